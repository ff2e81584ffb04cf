"""Kernel work: isolated launches of the stride-1 kernel-3 convs per block width (ADP_MM_NSP = 1 / 2 / 4: 64 / 128 / 256
positions per conv_mm block), forward with the GroupNorm+SiLU prologue + residual, plain forward + residual, data gradient.
usage: python tools/nsp_micro.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from conv_family_ab import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
for C, L in ((64, 16384), (128, 4096), (256, 2048), (512, 1024), (512, 512), (1024, 256)):
    x = torch.randn(B, C, L, device=dev)
    dy = torch.randn(B, C, L, device=dev)
    w = torch.randn(C, C, 3, device=dev) * 0.05
    bias = torch.randn(C, device=dev)
    res = torch.randn(B, C, L, device=dev)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    stats = ops.gn_stats(x, 8)
    fl = 2 * B * C * C * 3 * L
    for nsp in ("1", "2", "4"):
        os.environ["ADP_MM_NSP"] = nsp
        gn = ops.GnPart()
        t_p = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res, prologue=1, pro_stats=stats, pro_gamma=gamma, pro_beta=beta,
                                        groups=8, gn=ops.GnPart()))
        t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res))
        t_t = timeit(lambda: ops.conv1d(dy, w, None, pad=1, transposed=True))
        print(f"B{B} C{C:5d} L{L:6d} nsp {nsp}: fwd+pro {t_p:6.1f} us {fl / t_p / 1e6:6.1f} TF | fwd {t_f:6.1f} us {fl / t_f / 1e6:6.1f} TF"
              f" | dgrad {t_t:6.1f} us {fl / t_t / 1e6:6.1f} TF", flush=True)
