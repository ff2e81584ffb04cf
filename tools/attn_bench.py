"""Attention core microbench (GPU box): times adp_attn_fwd / adp_attn_bwd per shape with HIP events via torch.
usage: python tools/attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    H, D = 8, 64
    shapes = [(1, 1024, 1024), (1, 512, 512), (1, 256, 256), (1, 128, 128), (4, 1024, 1024), (4, 256, 256),
              (1, 4096, 64), (1, 1024, 64), (1, 128, 64), (8, 4096, 64)]
    for B, n, m in shapes:
        q = torch.randn(B, H * D, n, device=dev)
        kv = torch.randn(B, 2 * H * D, m, device=dev)
        o, lse = ops.attn_fwd(q, kv, H, D)
        do = torch.randn_like(o)
        tf = timeit(lambda: ops.attn_fwd(q, kv, H, D))
        tb = timeit(lambda: ops.attn_bwd(q, kv, o, do, lse, H, D))
        fl = 4 * B * H * n * m * D
        print(f"B{B} n{n} m{m}: fwd {tf:8.1f} us {fl / tf / 1e6:6.1f} TF | bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:6.1f} TF(2.5x)",
              flush=True)


if __name__ == "__main__":
    main()
