"""Conv kernel-family microbench (GPU box): times ops.conv1d on the bench's ResnetItem shapes.  The family is picked by
the dispatcher; run once per ADP_CONV_WAVE setting (0 = staged conv_mm only, 2 = wave-tile kernel wherever eligible).
usage: ADP_CONV_WAVE=2 python tools/conv_bench.py [batch]"""
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
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    print(f"ADP_CONV_WAVE={os.environ.get('ADP_CONV_WAVE', '(default)')} batch {B}")
    for C, L in ((64, 16384), (128, 4096), (256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128)):
        x = torch.randn(B, C, L, device=dev)
        w = torch.randn(C, C, 3, device=dev) * 0.05
        bias = torch.randn(C, device=dev)
        res = torch.randn(B, C, L, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        stats = ops.gn_stats(x, 8)
        fl = 2 * B * C * C * 3 * L
        t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res))
        t_p = timeit(lambda: ops.conv1d(x, w, bias, pad=1, prologue=1, pro_stats=stats, pro_gamma=gamma, pro_beta=beta,
                                        groups=8))
        t_t = timeit(lambda: ops.conv1d(x, w, None, pad=1, transposed=True))
        dw, db = torch.empty_like(w), torch.empty_like(bias)
        t_w = timeit(lambda: ops.conv1d_wgrad(x, res, 3, pad=1, dw=dw, dbias=db))
        t_wp = timeit(lambda: ops.conv1d_wgrad(x, res, 3, pad=1, prologue=1, pro_stats=stats, pro_gamma=gamma,
                                               pro_beta=beta, groups=8, dw=dw, dbias=db))
        print(f"C{C:5d} L{L:6d}: fwd {t_f:7.1f} us {fl / t_f / 1e6:6.1f} TF | gn+silu fwd {t_p:7.1f} us {fl / t_p / 1e6:6.1f} TF"
              f" | dgrad {t_t:7.1f} us {fl / t_t / 1e6:6.1f} TF | wgrad {t_w:7.1f} us {fl / t_w / 1e6:6.1f} TF"
              f" | gn+silu wgrad {t_wp:7.1f} us {fl / t_wp / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
