"""Child process of test_kernels.py::test_conv_wave_family: ADP_CONV_WAVE=2 (latched at the first dispatch of a
process) routes every eligible shape to the barrier-free wave-tile conv kernel (csrc/conv_wave.hip).
argv[1] = emul | hip."""
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


def check(name, got, ref):
    global worst
    e = conftest.rel_err(got, ref)
    worst = max(worst, e)
    if e > 1e-4:
        print("MISMATCH", name, e)


# B, R, M, L, KT  -- R < 128: four position tiles per workgroup; otherwise the four waves split K
for B, R, M, L, KT in ((2, 64, 64, 200, 3), (1, 128, 128, 100, 3), (1, 256, 64, 64, 1), (2, 32, 128, 72, 1),
                       (1, 96, 192, 130, 3)):
    pad = (KT - 1) // 2
    x = torch.randn(B, R, L, generator=g) * 1.3 + 0.2
    w = torch.randn(M, R, KT, generator=g) * 0.1
    b = torch.randn(M, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    KT, 1, 1, pad, 1, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", d) == 64064, "case must dispatch to the wave-tile kernel"
    # forward: bias + SkipModulate-style epilogue (e_scale, residual, pre-merge copy)
    res, sc = torch.randn(B, M, L, generator=g), torch.randn(B * M, generator=g)
    pre_ref = F.conv1d(x, w, b, padding=pad)
    pre = torch.empty(B, M, L).to(dev)
    out = ops.conv1d(xd, wd, b.to(dev), pad=pad, e_scale=sc.to(dev), res=res.to(dev), out_pre=pre)
    check(f"fwd {B,R,M,L,KT}", out, pre_ref * sc.view(B, M, 1) + res)
    check(f"pre {B,R,M,L,KT}", pre, pre_ref)
    # data gradient (transposed weight view): weight [R_fwd = M here .. ] read as [R][M][KT]
    dy = torch.randn(B, R, L, generator=g)
    wt = torch.randn(R, M, KT, generator=g) * 0.1
    check(f"dgrad {B,R,M,L,KT}", ops.conv1d(dy.to(dev), wt.to(dev), None, pad=pad, transposed=True),
          F.conv_transpose1d(dy, wt, None, padding=pad))
    # GroupNorm + SiLU prologue with a residual
    G = 8
    gamma, beta = torch.randn(R, generator=g) * 0.5 + 1, torch.randn(R, generator=g) * 0.1
    act = F.silu(F.group_norm(x, G, gamma, beta, eps=1e-5))
    stats = ops.gn_stats(xd, G)
    out = ops.conv1d(xd, wd, b.to(dev), pad=pad, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                     pro_beta=beta.to(dev), groups=G, res=res.to(dev))
    check(f"pro {B,R,M,L,KT}", out, F.conv1d(act, w, b, padding=pad) + res)
print("WORST", worst)
sys.exit(0 if worst < 1e-4 else 1)
