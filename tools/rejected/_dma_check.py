"""Child process of test_kernels.py::test_conv_mm_lds_dma_loaders: ADP_MM_DMA=2 (latched at the first dispatch of a
process) makes every plain stride-1 conv_mm launch use the LDS-DMA loaders (global_load_lds, three LDS buffers,
counted vmcnt), whatever its channel count.  argv[1] = emul | hip."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from audio_diffusion_pytorch_amd import _C, ops  # noqa: E402

dev = conftest._use_emulator() if sys.argv[1] == "emul" else conftest._use_hip()
g = torch.Generator().manual_seed(0)
worst = 0.0


def check(name, got, ref, tol=1e-4):
    global worst
    e = conftest.rel_err(got, ref)
    worst = max(worst, e)
    if e > tol:
        print("MISMATCH", name, e)


big = dev.type == "cuda"
cases = [(2, 64, 64, 200, 3), (1, 128, 32, 64, 3), (1, 32, 96, 300, 3), (2, 256, 64, 128, 1), (1, 512, 64, 72, 3)]
if big:
    cases += [(4, 1024, 1024, 256, 3), (4, 512, 512, 1024, 3), (1, 1024, 1024, 128, 3)]
for B, R, M, L, KT in cases:
    pad = (KT - 1) // 2
    x = torch.randn(B, R, L, generator=g)
    w = torch.randn(M, R, KT, generator=g) * 0.05
    b = torch.randn(M, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    res, sc = torch.randn(B, M, L, generator=g), torch.randn(B * M, generator=g)
    pre_ref = F.conv1d(x, w, b, padding=pad)
    pre = torch.empty(B, M, L).to(dev)
    for rep in range(3 if big else 1):   # repeated launches: a race between the DMA and the readers shows up as flicker
        out = ops.conv1d(xd, wd, b.to(dev), pad=pad, e_scale=sc.to(dev), res=res.to(dev), out_pre=pre)
        check(f"fwd {B,R,M,L,KT} rep{rep}", out, pre_ref * sc.view(B, M, 1) + res)
        check(f"pre {B,R,M,L,KT} rep{rep}", pre, pre_ref)
    dy = torch.randn(B, R, L, generator=g)
    wt = torch.randn(R, M, KT, generator=g) * 0.05
    dref = F.conv_transpose1d(dy, wt, None, padding=pad)
    for rep in range(3 if big else 1):
        check(f"dgrad {B,R,M,L,KT} rep{rep}", ops.conv1d(dy.to(dev), wt.to(dev), None, pad=pad, transposed=True), dref)
print("WORST", worst)
sys.exit(0 if worst < 1e-4 else 1)
