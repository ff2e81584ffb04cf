"""Depth-1 ConvBlock conv microbench on the GPU box (conv_tile.hip), every launch replayed from a hipGraph (no host launch
time in the numbers): conv1 (GroupNorm+SiLU prologue, statistics epilogue), conv2 (+ residual), data gradient at
[B, 32, 65536], next to a plain 2-read-1-write streaming kernel (adp_add) of the same tensors.  `warm` re-runs on one buffer
set (input in the Infinity Cache, as behind its producer in the step), `cold` rotates four sets (400 MB: past the cache).
TILE_CFGS = comma list of ADP_TILE_CFG values (stage gap in 10 ns ticks), TILE_BATCHES = comma list of batch sizes, TILE_C /
TILE_L = channels / length (8 / 262144 = the depth-0 convs on conv_direct8).
usage: python tools/tile_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402


def graph_time(fns, reps=5):
    """us per call of the captured sequence `fns` (list of callables)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fns:
            f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / len(fns) * 1e3


def main():
    dev = torch.device("cuda:0")
    C, L, G = int(os.environ.get("TILE_C", "32")), int(os.environ.get("TILE_L", "65536")), 8  # (TILE_C=8 TILE_L=262144: depth 0, conv_direct8)
    for B in [int(v) for v in os.environ.get("TILE_BATCHES", "4").split(",")]:
        sets = []
        for _ in range(4):
            x = torch.randn(B, C, L, device=dev)
            sets.append(dict(x=x, res=torch.randn(B, C, L, device=dev), out=torch.empty(B, C, L, device=dev),
                             stats=ops.gn_stats(x, G)))
        w = torch.randn(C, C, 3, device=dev) * 0.1
        bias, gamma, beta = torch.randn(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
        A = 4 * B * C * L

        def conv1(s):
            return lambda: ops.conv1d(s["x"], w, bias, pad=1, prologue=1, pro_stats=s["stats"], pro_gamma=gamma,
                                      pro_beta=beta, groups=G, out=s["out"], gn=ops.GnPart() if C % 32 == 0 else None)

        def conv2(s):
            return lambda: ops.conv1d(s["x"], w, bias, pad=1, prologue=1, pro_stats=s["stats"], pro_gamma=gamma,
                                      pro_beta=beta, groups=G, out=s["out"], res=s["res"])

        def dgrad(s):
            return lambda: ops.conv1d(s["x"], w, None, pad=1, transposed=True, out=s["out"])

        def wgrad(s):
            dw, db = torch.empty(C, C, 3, device=dev), torch.empty(C, device=dev)
            return lambda: ops.conv1d_wgrad(s["x"], s["res"], 3, pad=1, prologue=1, pro_stats=s["stats"], pro_gamma=gamma,
                                            pro_beta=beta, groups=G, dw=dw, dbias=db)

        def add(s):
            return lambda: ops.add(s["x"], s["res"], out=s["out"])
        for cfg in os.environ.get("TILE_CFGS", "120").split(","):
            os.environ["ADP_TILE_CFG"] = cfg
            row = []
            for temp, nset in (("warm", 1), ("cold", 4)):
                for name, mk, nb in (("conv1", conv1, 2), ("conv2", conv2, 3), ("dgrad", dgrad, 2), ("wgrad(+reduce)", wgrad, 4), ("add", add, 3)):
                    t = graph_time([mk(sets[i % nset]) for i in range(20)])
                    if nb == 4:  # per CALL = two launches (partials + second stage); bytes: x + dy
                        nb = 2
                    row.append(f"{name} {t:5.1f} us {nb * A / t / 1e6:5.2f} TB/s")
                row.append("|")
            print(f"B{B} cfg {cfg:>4} warm: " + "  ".join(row), flush=True)
    os.environ.pop("ADP_TILE_CFG", None)


if __name__ == "__main__":
    main()
