"""A/B microbench of the bf16-split conv family (conv_bs, ADP_CONV_BS=1) against the exact-f32 MFMA conv_mm on the deep
ResnetItem shapes (GPU box):  python tools/bs_bench.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402


def timeit(fn, n=20):
    """us per call, n calls replayed from one hipGraph (eager event timing is host-bound below ~12 us per launch)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    for C, L in ((256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128)):
        x = torch.randn(B, C, L, device=dev)
        w = torch.randn(C, C, 3, device=dev) * 0.05
        bias = torch.randn(C, device=dev)
        res = torch.randn(B, C, L, device=dev)
        fl = 2 * B * C * C * 3 * L
        row = f"C{C:5d} L{L:5d}:"
        outs = {}
        for mode in ("0", "1"):
            os.environ["ADP_CONV_BS"] = mode
            outs[mode] = (ops.conv1d(x, w, bias, pad=1, res=res), ops.conv1d(x, w, None, pad=1, transposed=True))
            t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res))
            t_t = timeit(lambda: ops.conv1d(x, w, None, pad=1, transposed=True))
            row += f"  [{'bf16 split' if mode == '1' else 'f32 mfma  '}] fwd {t_f:6.1f} us {fl / t_f / 1e6:6.1f} TF  dgrad {t_t:6.1f} us {fl / t_t / 1e6:6.1f} TF"
        ref64 = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding=1) + res.double()
        e = [((outs[m][0].double() - ref64).abs().max() / ref64.abs().max()).item() for m in ("0", "1")]
        d = ((outs["1"][1] - outs["0"][1]).abs().max() / outs["0"][1].abs().max()).item()
        print(row + f"  | err vs fp64: f32 {e[0]:.1e} split {e[1]:.1e}; dgrad split vs f32 {d:.1e}", flush=True)


if __name__ == "__main__":
    main()
