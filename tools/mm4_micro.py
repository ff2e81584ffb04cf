"""Kernel work: isolated launches (hipGraph of 20) of the wide kernel-3 convs, conv_mm's Winograd F(2,3) blocks against
conv_mm4's F(4,3) block (ADP_CONV_WINO4 = 0 / 1): plain forward + residual + GroupNorm partials, data gradient.
usage: python tools/mm4_micro.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from conv_family_ab import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
for C, L in ((128, 4096), (256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128)):
    x = torch.randn(B, C, L, device=dev)
    dy = torch.randn(B, C, L, device=dev)
    w = torch.randn(C, C, 3, device=dev) * 0.05
    bias = torch.randn(C, device=dev)
    res = torch.randn(B, C, L, device=dev)
    fl = 2 * B * C * C * 3 * L
    ref = None
    for w4 in ("0", "1"):
        os.environ["ADP_CONV_WINO4"] = w4
        y = ops.conv1d(x, w, bias, pad=1, res=res, gn=ops.GnPart())
        if ref is None:
            ref = y
        err = (y - ref).abs().max().item() / ref.abs().max().item()
        t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res, gn=ops.GnPart()))
        t_t = timeit(lambda: ops.conv1d(dy, w, None, pad=1, transposed=True))
        print(f"B{B} C{C:5d} L{L:6d} wino4 {w4}: fwd {t_f:6.1f} us {fl / t_f / 1e6:6.1f} TF | dgrad {t_t:6.1f} us {fl / t_t / 1e6:6.1f} TF"
              f" | vs F(2,3) {err:.1e}", flush=True)
os.environ.pop("ADP_CONV_WINO4", None)
