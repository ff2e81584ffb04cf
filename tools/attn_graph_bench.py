"""Attention core per call inside a hipGraph (GPU box): 20 back-to-back calls of one shape (warm instruction cache), and the same
call placed between `--cold` other kernels of the step's kind (a [B,512,n] GroupNorm pass and a conv), which is how an attention item
meets it inside a replayed step.  usage: python tools/attn_graph_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from tools.tile_bench import graph_time  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H, D = 8, 64
    shapes = [(1, 4096, 64), (1, 1024, 64), (1, 256, 64), (1, 128, 64), (1, 1024, 1024), (1, 256, 256), (1, 128, 128)]
    only = os.environ.get("ATTN_SHAPES")
    if only:
        shapes = [tuple(int(v) for v in s.split("x")) for s in only.split(",")]
    for B, n, m in shapes:
        q = torch.randn(B, H * D, n, device=dev)
        kv = torch.randn(B, 2 * H * D, m, device=dev)
        o, lse = ops.attn_fwd(q, kv, H, D)
        do = torch.randn_like(o)
        tf = graph_time([lambda: ops.attn_fwd(q, kv, H, D)] * 20)
        tb = graph_time([lambda: ops.attn_bwd(q, kv, o, do, lse, H, D)] * 20)
        # the same calls with other kernels in between (their code evicts the attention kernels' from the instruction caches)
        xs = torch.randn(B, 512, max(n, 256), device=dev)
        w = torch.randn(512, 512, 3, device=dev) * 0.02
        gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)

        def filler():
            st, a = ops.gn_stats_act(xs, 8, gam, bet)
            ops.conv1d(a, w, None, pad=1)
        t0 = graph_time([filler] * 10)
        tfc = graph_time([lambda: (filler(), ops.attn_fwd(q, kv, H, D))] * 10) - t0
        tbc = graph_time([lambda: (filler(), ops.attn_bwd(q, kv, o, do, lse, H, D))] * 10) - t0
        fl = 4 * B * H * n * m * D
        print(f"B{B} n{n} m{m}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF)  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF) | "
              f"between other kernels: fwd {tfc:7.1f}  bwd {tbc:7.1f}", flush=True)


if __name__ == "__main__":
    main()
