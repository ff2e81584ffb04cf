"""Per-kernel parity: every C-ABI entry point against the stock torch.nn.functional CPU op it replaces.
Each test runs twice (fixture `dev`): on the SIMT-emulated build of the kernel sources (CPU suite) and,
with -m gpu, on the real gfx950 library.  Tolerance: 1e-3 rel (north_star), fp32; in practice ~1e-6.
"""
import math

import os

import pytest
import torch
import torch.nn.functional as F

from audio_diffusion_pytorch_amd import _C, ops
from conftest import rel_err

TOL = 1e-4  # tighter than the 1e-3 contract on purpose: these are single ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def ref_gn_silu(x, G, gamma, beta):
    return F.silu(F.group_norm(x, G, gamma, beta, eps=1e-5))


# ------------------------------------------------------------------ conv forward family
CONV_CASES = [
    # B, R, M, L, KT, stride, pad, up  -- shapes cover: narrow (M<=32), 64-tile, 128-tile, ragged edges
    (2, 8, 8, 300, 3, 1, 1, 1),
    (1, 2, 8, 256, 1, 1, 0, 1),
    (2, 32, 32, 260, 3, 1, 1, 1),
    (1, 48, 80, 200, 3, 1, 1, 1),
    (1, 8, 32, 256, 4, 4, 0, 1),
    (2, 32, 64, 128, 4, 4, 0, 1),
    (1, 16, 32, 128, 2, 2, 0, 1),
    (1, 40, 24, 66, 3, 1, 1, 2),
    (1, 32, 8, 50, 3, 1, 1, 4),
    (3, 136, 130, 140, 3, 1, 1, 1),
]


@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", CONV_CASES)
def test_conv1d_plain(dev, B, R, M, L, KT, stride, pad, up):
    x, w, b = rnd(B, R, L, seed=1), rnd(M, R, KT, seed=2, scale=0.2), rnd(M, seed=3)
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    ref = F.conv1d(xr, w, b, stride=stride, padding=pad)
    out = ops.conv1d(x.to(dev), w.to(dev), b.to(dev), stride=stride, pad=pad, up=up)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("B,R,M,L,KT,tr", [(1, 512, 64, 64, 3, False), (1, 256, 96, 128, 3, True),
                                           (2, 1024, 32, 60, 1, False), (1, 512, 128, 32, 2, False)])
def test_conv1d_cross_workgroup_split_k(dev, B, R, M, L, KT, tr):
    """Small grids (batch-1 deep layers) split the channel reduction over several workgroups: partial tiles in the
    caller's scratch (adp_conv1d_ws_bytes) + a fixed-order reduce kernel that owns the epilogue (bias, e_scale,
    residual, out_pre)."""
    from ctypes import byref
    from audio_diffusion_pytorch_amd import _C
    stride = 2 if KT == 2 else 1
    x = rnd(B, R, L, seed=1)
    w = rnd(R, M, KT, seed=2, scale=0.05) if tr else rnd(M, R, KT, seed=2, scale=0.05)
    N = L // stride
    b, res, sc = rnd(M, seed=3), rnd(B, M, N, seed=4), rnd(B * M, seed=5)
    if tr:
        ref = F.conv_transpose1d(x, w, None, padding=1)
    else:
        ref = F.conv1d(x, w, None, stride=stride, padding=1 if KT == 3 else 0)
    pre_ref = ref + b[None, :, None]
    ref = pre_ref * sc.view(B, M, 1) + res
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M,
                    N, KT, stride, 1, 1 if KT == 3 else 0, 1, int(tr), 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_ws_bytes", byref(d)) > 0, "this shape is meant to take the split-K path"
    pre = torch.empty(B, M, N).to(dev)
    out = ops.conv1d(xd, wd, b.to(dev), stride=stride, pad=1 if KT == 3 else 0, transposed=tr, e_scale=sc.to(dev),
                     res=res.to(dev), out_pre=pre)
    assert rel_err(out, ref) < TOL and rel_err(pre, pre_ref) < TOL


@pytest.mark.parametrize("B,R,M,L,tr", [(2, 256, 64, 200, False), (1, 512, 64, 64, False), (2, 256, 96, 132, True),
                                        (1, 1024, 32, 64, True), (3, 288, 32, 4, False), (1, 256, 32, 260, True),
                                        (4, 256, 128, 512, False)])
def test_conv1d_winograd_family(dev, B, R, M, L, tr, monkeypatch):
    """Winograd F(2,3) form of the wide kernel-3 convs on the exact-f32 matrix cores (conv_mm's WN variant: transforms
    in the MMA waves' registers).  Forward and transposed-weight view, ragged last tile, a length shorter than one
    tile, the cross-workgroup K split, 32- and 64-row blocks, the full epilogue and the GroupNorm partial statistics.
    fp32 throughout: the bound against an fp64 reference is 1e-5, next to the direct form (ADP_CONV_WINO=0)."""
    x = rnd(B, R, L, seed=1)
    w = rnd(R, M, 3, seed=2, scale=0.05) if tr else rnd(M, R, 3, seed=2, scale=0.05)
    b, res, sc = rnd(M, seed=3), rnd(B, M, L, seed=4), rnd(B * M, seed=5)
    x64, w64 = x.double(), w.double()
    ref = F.conv_transpose1d(x64, w64, None, padding=1) if tr else F.conv1d(x64, w64, None, padding=1)
    pre_ref = ref + b.double()[None, :, None]
    ref = pre_ref * sc.double().view(B, M, 1) + res.double()
    args = (x.to(dev), w.to(dev), b.to(dev))
    kw = dict(pad=1, transposed=tr, e_scale=sc.to(dev), res=res.to(dev))
    err = lambda a, r: ((a.cpu().double() - r).abs().max() / r.abs().max()).item()  # noqa: E731
    from ctypes import byref
    from audio_diffusion_pytorch_amd import _C
    d = _C.ConvDesc(_C.ptr(args[0]), None, _C.ptr(args[1]), None, None, None, None, None, None, _C.ptr(args[0]), None, B, R, R,
                    L, M, L, 3, 1, 1, 1, 1, int(tr), 0, 1, 0, 1, 0)
    monkeypatch.setenv("ADP_CONV_WINO", "1")
    assert _C.query("adp_conv1d_tile", byref(d)) // 10000000 == 4, "this shape is meant to take the Winograd variant"
    pre = torch.empty(B, M, L).to(dev)
    gn = ops.GnPart()
    out = ops.conv1d(*args, out_pre=pre, gn=gn, **kw)
    monkeypatch.setenv("ADP_CONV_WINO", "0")
    out_direct = ops.conv1d(*args, **kw)
    e_w, e_d = err(out, ref), err(out_direct, ref)
    assert e_w < 1e-5 and err(pre, pre_ref) < 1e-5, (e_w, e_d)
    assert e_w < 8 * e_d + 1e-6, (e_w, e_d)
    if M % 32 == 0 and gn.part is not None:
        st = ops.gn_finalize(gn.part, 8)
        ref_st = ops.gn_stats(out, 8)
        assert gn.part[..., 2].sum(dim=2).eq(4 * L).all()
        assert rel_err(st[..., 1], ref_st[..., 1]) < 1e-5
        assert (st[..., 0] - ref_st[..., 0]).abs().max() < 1e-5


@pytest.mark.parametrize("C,L,tr", [(64, 200, False), (96, 132, True), (128, 64, False)])
def test_conv1d_winograd_with_groupnorm_prologue(dev, C, L, tr, monkeypatch):
    """The Winograd variant under the GroupNorm+SiLU loader prologue (the mid-depth ConvBlocks), channel counts below
    the default threshold switched in with ADP_WINO_MIN_R."""
    monkeypatch.setenv("ADP_CONV_WINO", "1")
    monkeypatch.setenv("ADP_WINO_MIN_R", "32")
    B, G = 2, 8
    x = rnd(B, C, L, seed=1) * 1.7 + 0.3
    w = rnd(C, C, 3, seed=2, scale=0.2)
    b, gamma, beta, res = rnd(C, seed=3), rnd(C, seed=4) * 0.5 + 1, rnd(C, seed=5) * 0.1, rnd(B, C, L, seed=6)
    a = ref_gn_silu(x, G, gamma, beta)
    ref = (F.conv_transpose1d(a, w, None, padding=1) + b[None, :, None] if tr else F.conv1d(a, w, b, padding=1)) + res
    xd = x.to(dev)
    stats = ops.gn_stats(xd, G)
    out = ops.conv1d(xd, w.to(dev), b.to(dev), pad=1, transposed=tr, prologue=1, pro_stats=stats,
                     pro_gamma=gamma.to(dev), pro_beta=beta.to(dev), groups=G, res=res.to(dev))
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("nsp", [2, 4])
@pytest.mark.parametrize("B,C,M,L,tr,pro", [(2, 64, 64, 300, False, True), (1, 64, 128, 520, True, False),
                                            (2, 128, 64, 256, False, False), (1, 96, 64, 68, True, True)])
def test_conv1d_winograd_wide_blocks(dev, nsp, B, C, M, L, tr, pro, monkeypatch):
    """Wide-N blocks of the Winograd variant (NSP = 2 / 4: 64 rows x 128 / 256 positions, K groups traded for position
    tiles): ragged last block (whole 64-position tiles of a block beyond the row's end), GroupNorm+SiLU prologue,
    transposed weight view, full epilogue, GroupNorm partial statistics -- against fp64 and, bit for bit where the K
    order is the same (NSP = 4 sums a row's channels in one wave: not the same order), against the 64-position block."""
    monkeypatch.setenv("ADP_CONV_WINO", "1")
    monkeypatch.setenv("ADP_MM_MIN_BLOCKS", "1")
    G = 8
    x = rnd(B, C, L, seed=1) * 1.3 + 0.2
    w = rnd(C, M, 3, seed=2, scale=0.1) if tr else rnd(M, C, 3, seed=2, scale=0.1)
    b, res, sc = rnd(M, seed=3), rnd(B, M, L, seed=4), rnd(B * M, seed=5)
    gamma, beta = rnd(C, seed=6) * 0.5 + 1, rnd(C, seed=7) * 0.1
    a64 = (ref_gn_silu(x.double(), G, gamma.double(), beta.double()) if pro else x.double())
    ref = F.conv_transpose1d(a64, w.double(), None, padding=1) if tr else F.conv1d(a64, w.double(), None, padding=1)
    pre_ref = ref + b.double()[None, :, None]
    ref = pre_ref * sc.double().view(B, M, 1) + res.double()
    xd = x.to(dev)
    kw = dict(pad=1, transposed=tr, e_scale=sc.to(dev), res=res.to(dev))
    if pro:
        kw.update(prologue=1, pro_stats=ops.gn_stats(xd, G), pro_gamma=gamma.to(dev), pro_beta=beta.to(dev), groups=G)
    from ctypes import byref
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(w.to(dev)), None, None, None, None, None, None, _C.ptr(xd), None, B, C, C,
                    L, M, L, 3, 1, 1, 1, 1, int(tr), 0, 1, 0, 1, 0)
    outs = {}
    for n in (1, nsp):
        monkeypatch.setenv("ADP_MM_NSP", str(n))
        tile = _C.query("adp_conv1d_tile", byref(d))
        assert tile // 10000000 == 4 and tile % 1000 == 64 * n and (tile // 1000) % 1000 == 64, tile
        pre = torch.empty(B, M, L).to(dev)
        gn = ops.GnPart()
        out = ops.conv1d(xd, w.to(dev), b.to(dev), out_pre=pre, gn=gn, **kw)
        outs[n] = (out, pre, gn)
    err = lambda a, r: ((a.cpu().double() - r).abs().max() / r.abs().max()).item()  # noqa: E731
    out, pre, gn = outs[nsp]
    assert err(out, ref) < 1e-5 and err(pre, pre_ref) < 1e-5
    assert gn.part is not None and gn.part[..., 2].sum(dim=2).eq(4 * L).all()
    st, ref_st = ops.gn_finalize(gn.part, 8), ops.gn_stats(out, 8)
    assert rel_err(st[..., 1], ref_st[..., 1]) < 1e-5 and (st[..., 0] - ref_st[..., 0]).abs().max() < 1e-5
    assert err(out, outs[1][0].cpu().double()) < 2e-6


@pytest.mark.parametrize("nsp", [2, 4])
@pytest.mark.parametrize("B,R,M,L,KT,tr,up", [(2, 64, 128, 300, 1, True, 1), (1, 96, 64, 516, 1, False, 1),
                                               (1, 64, 64, 130, 3, False, 2), (2, 128, 64, 36, 3, False, 4)])
def test_conv1d_wide_blocks_1x1_and_upsample(dev, nsp, B, R, M, L, KT, tr, up, monkeypatch):
    """The wide-N blocks under the 1x1 convs (direct form, both weight views) and the nearest-upsample Winograd convs."""
    monkeypatch.setenv("ADP_CONV_WINO", "1")
    monkeypatch.setenv("ADP_MM_MIN_BLOCKS", "1")
    pad = (KT - 1) // 2
    x = rnd(B, R, L, seed=1)
    w = rnd(R, M, KT, seed=2, scale=0.1) if tr else rnd(M, R, KT, seed=2, scale=0.1)
    N = L * up
    b, res, sc = rnd(M, seed=3), rnd(B, M, N, seed=4), rnd(B * M, seed=5)
    xr = F.interpolate(x.double(), scale_factor=up, mode="nearest") if up > 1 else x.double()
    ref = F.conv_transpose1d(xr, w.double(), None, padding=pad) if tr else F.conv1d(xr, w.double(), None, padding=pad)
    ref = (ref + b.double()[None, :, None]) * sc.double().view(B, M, 1) + res.double()
    xd, wd = x.to(dev), w.to(dev)
    from ctypes import byref
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R,
                    L, M, N, KT, 1, 1, pad, up, int(tr), 0, 1, 0, 1, 0)
    monkeypatch.setenv("ADP_MM_NSP", str(nsp))
    tile = _C.query("adp_conv1d_tile", byref(d))
    want = min(nsp, 2) if KT == 1 else nsp  # the 1x1 convs stop at 128 positions per block
    assert tile % 1000 == 64 * want and (tile // 1000) % 1000 == 64, tile
    out = ops.conv1d(xd, wd, b.to(dev), pad=pad, up=up, transposed=tr, e_scale=sc.to(dev), res=res.to(dev))
    assert ((out.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


def test_conv1d_big_tile(dev):
    # enough workgroups to select the 128x128 tile on the dispatcher
    B, R, M, L = 6, 32, 128, 1024 if dev.type == "cuda" else 1024
    x, w, b = rnd(B, R, L, seed=1), rnd(M, R, 3, seed=2, scale=0.2), rnd(M, seed=3)
    if dev.type != "cuda":
        B = 6
    ref = F.conv1d(x, w, b, padding=1)
    out = ops.conv1d(x.to(dev), w.to(dev), b.to(dev), pad=1)
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("C,G,L,dil", [(8, 8, 300, 1), (32, 8, 130, 1), (64, 8, 96, 2)])
def test_convblock_gn_silu_conv_residual(dev, C, G, L, dil):
    """ConvBlock: GroupNorm -> SiLU -> Conv1d(k3) with the GN+SiLU applied in the loader, + residual epilogue."""
    B = 2
    x, w, b = rnd(B, C, L, seed=1) * 1.7 + 0.3, rnd(C, C, 3, seed=2, scale=0.2), rnd(C, seed=3)
    gamma, beta, res = rnd(C, seed=4) * 0.5 + 1, rnd(C, seed=5) * 0.1, rnd(B, C, L, seed=6)
    ref = F.conv1d(ref_gn_silu(x, G, gamma, beta), w, b, padding=dil, dilation=dil) + res
    xd = x.to(dev)
    stats = ops.gn_stats(xd, G)
    sref = torch.stack([x.view(B, G, -1).mean(-1), (x.view(B, G, -1).var(-1, unbiased=False) + 1e-5).rsqrt()], -1)
    assert rel_err(stats, sref) < 1e-5
    out = ops.conv1d(xd, w.to(dev), b.to(dev), pad=dil, dil=dil, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                     pro_beta=beta.to(dev), groups=G, res=res.to(dev))
    assert rel_err(out, ref) < TOL


def test_gn_stats_large_group(dev):
    B, C, L, G = 1, 8, 9000, 4
    x = rnd(B, C, L, seed=7) * 3 + 5
    stats = ops.gn_stats(x.to(dev), G)
    xv = x.view(B, G, -1).double()
    sref = torch.stack([xv.mean(-1), (xv.var(-1, unbiased=False) + 1e-5).rsqrt()], -1).float()
    assert rel_err(stats, sref) < 1e-5


def test_conv1d_ln_prologue(dev):
    """1x1 projection with the LayerNorm-over-channels prologue (attention to_q / to_kv)."""
    B, C, L, M = 2, 40, 70, 48
    x, w = rnd(B, C, L, seed=1) * 2 + 1, rnd(M, C, 1, seed=2, scale=0.2)
    gamma, beta = rnd(C, seed=3) * 0.3 + 1, rnd(C, seed=4) * 0.1
    xn = F.layer_norm(x.transpose(1, 2), (C,), gamma, beta, eps=1e-5).transpose(1, 2)
    ref = F.conv1d(xn, w)
    xd = x.to(dev)
    stats = ops.ln_stats(xd)
    out = ops.conv1d(xd, w.to(dev), None, prologue=2, pro_stats=stats, pro_gamma=gamma.to(dev), pro_beta=beta.to(dev))
    assert rel_err(out, ref) < TOL


def test_conv1d_concat_and_skipmod(dev):
    """x2 channel concat (AppendChannels / InjectChannels) and the SkipModulate epilogue out = skip + scale*conv."""
    B, R1, R2, M, L = 2, 5, 3, 6, 90
    x, x2, w, b = rnd(B, R1, L, seed=1), rnd(B, R2, L, seed=2), rnd(M, R1 + R2, 3, seed=3, scale=0.3), rnd(M, seed=4)
    skip, bank = rnd(B, M, L, seed=5), rnd(B, 17, seed=6)
    scale = bank[:, 4:4 + M]
    ref = skip + scale[:, :, None] * F.conv1d(torch.cat([x, x2], 1), w, b, padding=1)
    bank_d = bank.to(dev)
    out = ops.conv1d(x.to(dev), w.to(dev), b.to(dev), pad=1, x2=x2.to(dev), e_scale=bank_d.view(-1)[4:], e_bstride=17,
                     res=skip.to(dev))
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("R1,R2,M,L", [(8, 0, 8, 2304), (5, 3, 8, 1100), (2, 0, 6, 516)])
def test_conv_direct8_prologue_concat(dev, R1, R2, M, L):
    """conv_direct8 (kernel 3, <= 8 input channels) with the GroupNorm+SiLU prologue: the window's halo travels between lanes
    AFTER the activation, zero padding stays zero; x2 concat; lengths with several waves, a ragged last workgroup."""
    B, R, G = 2, R1 + R2, (R1 + R2) if (R1 + R2) < 8 else 8
    xa = rnd(B, R, L, seed=1) * 1.5 + 0.3
    w, b = rnd(M, R, 3, seed=3, scale=0.3), rnd(M, seed=4)
    gamma, beta = rnd(R, seed=5) * 0.3 + 1.0, rnd(R, seed=6) * 0.2
    ref = F.conv1d(ref_gn_silu(xa, G, gamma, beta), w, b, padding=1)
    xd = xa.to(dev)
    stats = ops.gn_stats(xd, G)
    x1 = xd[:, :R1].contiguous()
    x2 = xd[:, R1:].contiguous() if R2 else None
    out = ops.conv1d(x1, w.to(dev), b.to(dev), pad=1, x2=x2, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                     pro_beta=beta.to(dev), groups=G)
    assert rel_err(out, ref) < TOL


# ------------------------------------------------------------------ deep-layer GEMM conv (conv_mm.hip)
MM_CASES = [
    # B, R, M, L, KT, pad, dil -- every (tile, K-group) variant of the split-K-in-block kernel, multi-chunk K,
    # ragged last position tile, dilation, kernel 1
    (1, 64, 64, 128, 3, 1, 1),        # 32x64 tile, 4 K groups, 2 chunks
    (2, 96, 32, 72, 3, 1, 1),         # ragged N (72 = 64 + 8), 3 chunks
    (3, 32, 64, 64 * 64, 3, 1, 1),    # 192 tiles -> 64x64 tile, 4 K groups
    (1, 32, 64, 64 * 384, 3, 1, 1),   # 384 tiles -> 64x64 tile, 2 K groups
    (1, 32, 64, 64 * 768, 1, 0, 1),   # 768 tiles -> 64x64 tile, 1 K group, kernel 1
    (2, 64, 96, 100, 1, 0, 1),        # M % 64 != 0 -> 32-row tile, kernel 1, 2 chunks
    (1, 32, 32, 64, 3, 3, 3),         # dilation 3
    (1, 32, 64, 64 * 1024, 1, 0, 1),  # 1024 tiles -> 64x128 tile (T2), guarded single-chunk pipeline
    (2, 96, 64, 64 * 512, 1, 0, 1),   # T2, 3 chunks -> prefetch distance 2 with a remainder iteration
    (2, 32, 32, 128 * 256 + 36, 3, 1, 1),  # 32 -> 32 channels, length not a multiple of 64: stays on conv_mm
    (1, 128, 64, 64 * 200, 3, 1, 1),  # 64x64 tile (T0), 4 chunks -> unguarded 2-stage pipeline
    (1, 160, 32, 192, 3, 1, 1),       # 32x64 tile (T1), 5 chunks (odd) -> remainder iteration
]


@pytest.mark.parametrize("B,R,M,L,KT,pad,dil", MM_CASES)
def test_conv_mm_family(dev, B, R, M, L, KT, pad, dil):
    from audio_diffusion_pytorch_amd import _C
    if dev.type != "cuda" and B * R * M * L * KT > 80e6:
        pytest.skip("emulator: the large-grid cases run on the GPU only")
    G = 8
    x = (rnd(B, R, L, seed=1) * 1.3 + 0.2).requires_grad_()
    w, b = rnd(M, R, KT, seed=2, scale=0.2), rnd(M, seed=3)
    gamma, beta = rnd(R, seed=4) * 0.5 + 1, rnd(R, seed=5) * 0.1
    xd, wd = x.detach().to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    KT, 1, dil, pad, 1, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", d) >= 1000000, "case must dispatch to the conv_mm family"
    # plain forward + bias
    y = F.conv1d(x, w, b, padding=pad, dilation=dil)
    out = ops.conv1d(xd, wd, b.to(dev), pad=pad, dil=dil)
    assert out.shape == y.shape
    assert rel_err(out, y) < TOL
    # data gradient (transposed weight view)
    dy = rnd(*y.shape, seed=9)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dx = ops.conv1d(dy.to(dev), wd, None, pad=(KT - 1) * dil - pad, dil=dil, transposed=True)
    assert rel_err(dx, dx_ref) < TOL
    # GroupNorm + SiLU prologue, residual epilogue
    res = rnd(B, M, L, seed=6)
    ref = F.conv1d(ref_gn_silu(x.detach(), G, gamma, beta), w, b, padding=pad, dilation=dil) + res
    stats = ops.gn_stats(xd, G)
    out = ops.conv1d(xd, wd, b.to(dev), pad=pad, dil=dil, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                     pro_beta=beta.to(dev), groups=G, res=res.to(dev))
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("nw", [1, 4, 16])
@pytest.mark.parametrize("B,L,pro,res,shift", [(1, 256, 0, 0, 0.0), (2, 1024, 1, 1, 0.0), (3, 192, 1, 0, 100.0),
                                               (1, 64 * 300, 1, 1, 0.0)])
def test_conv_tile32(dev, nw, B, L, pro, res, shift, monkeypatch):
    """conv_tile.hip: barrier-free wave-tile kernel (one wave = one 32 x 64 output tile through a wave-private LDS region,
    Winograd F(4,3) on v_mfma_f32_16x16x4_f32) for the 32 -> 32 channel kernel-3 ConvBlock convs: forward with / without GroupNorm+SiLU prologue and
    residual, data gradient, and the GroupNorm partial statistics of the output (shifted sums per row quad, Chan-combined
    over the 1-16 waves of a workgroup) -- `shift` puts the output mean at 100 sigma: the statistics must keep their digits
    (the one-pass sum-of-squares form this replaced lost them)."""
    from ctypes import byref
    if dev.type != "cuda" and L > 4096:
        pytest.skip("emulator: large case runs on the GPU only")
    monkeypatch.setenv("ADP_TILE_NW", str(nw))
    C, G = 32, 8
    x = rnd(B, C, L, seed=1) * 1.3 + 0.2
    w, b = rnd(C, C, 3, seed=2, scale=0.2), rnd(C, seed=3) + shift
    gamma, beta = rnd(C, seed=4) * 0.5 + 1, rnd(C, seed=5) * 0.1
    r = rnd(B, C, L, seed=6) if res else None
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, C, C, L, C, L,
                    3, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 32064, "case must dispatch to the wave-tile kernel"
    gn = ops.GnPart()
    if pro:
        ref = F.conv1d(ref_gn_silu(x, G, gamma, beta), w, b, padding=1)
        out = ops.conv1d(xd, wd, b.to(dev), pad=1, prologue=1, pro_stats=ops.gn_stats(xd, G), pro_gamma=gamma.to(dev),
                         pro_beta=beta.to(dev), groups=G, res=r.to(dev) if res else None, gn=gn)
    else:
        ref = F.conv1d(x, w, b, padding=1)
        out = ops.conv1d(xd, wd, b.to(dev), pad=1, res=r.to(dev) if res else None, gn=gn)
    if res:
        ref = ref + r
    assert rel_err(out, ref) < TOL
    assert gn.part is not None and gn.part[..., 2].sum(dim=2).eq(4 * L).all()
    st = ops.gn_finalize(gn.part, G)
    g64 = ref.double().view(B, G, -1)
    assert rel_err(st[..., 0], g64.mean(-1)) < 2e-6
    assert rel_err(st[..., 1], (g64.var(-1, unbiased=False) + 1e-5).rsqrt()) < 2e-5
    dx = ops.conv1d(xd, wd, None, pad=1, transposed=True)
    assert rel_err(dx, F.conv_transpose1d(x, w, None, padding=1)) < TOL


@pytest.mark.parametrize("rb", [1, 2])
@pytest.mark.parametrize("B,R,M,L,res,shift", [(1, 256, 32, 128, 0, 0.0), (2, 512, 64, 192, 1, 0.0), (1, 256, 96, 64, 1, 100.0)])
def test_conv_tilek(dev, rb, B, R, M, L, res, shift, monkeypatch):
    """conv_tilek.hip: the wave tile of conv_tile.hip for the deep layers -- eight waves of a workgroup split the input
    channels (16-channel chunks parked in wave-private LDS, U = G g built by the wave), partial tiles summed in LDS in wave
    order, 16- or 32-row tiles (rb): forward with bias / residual and the GroupNorm partial statistics of the output, data
    gradient (transposed weight view, taps flipped), non-square channel counts; against F.conv1d and the statistics of the
    fp64 result (`shift`: output mean at 100 sigma)."""
    from ctypes import byref
    monkeypatch.setenv("ADP_TILEK_MIN_R", "256")
    monkeypatch.setenv("ADP_TILEK_MIN_TILES", "1")
    monkeypatch.setenv("ADP_TILEK_RB", str(rb))
    G = 8
    x = rnd(B, R, L, seed=1) * 1.3 + 0.2
    w, b = rnd(M, R, 3, seed=2, scale=0.05), rnd(M, seed=3) + shift
    r = rnd(B, M, L, seed=6) if res else None
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    3, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 48000064, "case must dispatch to the deep-layer wave-tile kernel"
    assert _C.query("adp_conv1d_ws_bytes", byref(d)) == 0
    gn = ops.GnPart()
    ref = F.conv1d(x, w, b, padding=1)
    out = ops.conv1d(xd, wd, b.to(dev), pad=1, res=r.to(dev) if res else None, gn=gn)
    if res:
        ref = ref + r
    assert rel_err(out, ref) < TOL
    assert gn.part is not None and gn.part.shape == (B, M // 4, L // 64, 3) and gn.part[..., 2].eq(256).all()
    st = ops.gn_finalize(gn.part, G)
    g64 = ref.double().view(B, G, -1)
    assert rel_err(st[..., 0], g64.mean(-1)) < 2e-6
    assert rel_err(st[..., 1], (g64.var(-1, unbiased=False) + 1e-5).rsqrt()) < 2e-5
    # data gradient of the same conv: input = a [B, M, L] gradient, weight viewed transposed
    gy = rnd(B, M, L, seed=7)
    dx = ops.conv1d(gy.to(dev), wd, None, pad=1, transposed=True) if M >= 128 else None
    if dx is not None:
        assert rel_err(dx, F.conv_transpose1d(gy, w, None, padding=1)) < TOL
    # ... and a gradient-shaped call that does dispatch here whatever M is: R and M swapped
    w2 = rnd(R, M, 3, seed=8, scale=0.05)   # forward weight of a conv M -> R; its data gradient maps [B, R, L] -> [B, M, L]
    gy2 = rnd(B, R, L, seed=9)
    d2 = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                     3, 1, 1, 1, 1, 1, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d2)) == 48000064
    dx2 = ops.conv1d(gy2.to(dev), w2.to(dev), None, pad=1, transposed=True)
    assert rel_err(dx2, F.conv_transpose1d(gy2, w2, None, padding=1)) < TOL
    monkeypatch.setenv("ADP_CONV_TILEK", "0")  # the kernel family it replaces agrees
    out0 = ops.conv1d(xd, wd, b.to(dev), pad=1, res=r.to(dev) if res else None)
    assert rel_err(out, out0) < 1e-5


def test_conv_dispatch_of_the_readme_layers(emul):
    """Which kernel family adp_conv1d picks for the ResnetItem ConvBlock convs of the README U-Net (channels
    [8,32,64,128,256,512,512,1024,1024] at [B,2,2**18]) -- the table DESIGN.md section 4 describes, as a test: depth 1 on the
    32-channel wave tile, the wide layers on the F(4,3) block, and the deep layers whose tiles cannot fill the chip on
    conv_tilek (depths 5-7 at batch 1, depth 8 at batch 2-4) with NO split-K scratch, depth 8 at batch 1 / 8 on conv_mm."""
    from ctypes import byref
    a = torch.zeros(64)  # (aligned dummy storage: the queries never dereference)
    p = _C.ptr(a)

    def tile(B, C, L, tr=0):
        d = _C.ConvDesc(p, None, p, None, None, None, None, None, None, p, None, B, C, C, L, C, L, 3, 1, 1, 1, 1, tr, 0, 1, 0,
                        1, 0)
        return _C.query("adp_conv1d_tile", byref(d)), _C.query("adp_conv1d_ws_bytes", byref(d))
    TILE32, TILEK, MM4 = 32064, 48000064, 64032128
    deep = {5: (512, 1024), 6: (512, 512), 7: (1024, 256), 8: (1024, 128)}
    for tr in (0, 1):
        assert tile(4, 32, 65536, tr)[0] == TILE32 and tile(1, 32, 65536, tr)[0] == TILE32
        for B in (4, 8):
            for depth in (2, 3, 4, 5, 6, 7):
                C, L = {2: (64, 16384), 3: (128, 4096), 4: (256, 2048), **deep}[depth]
                assert tile(B, C, L, tr)[0] == MM4, (B, depth)
        for depth in (5, 6, 7):
            assert tile(1, *deep[depth], tr) == (TILEK, 0), depth
        for B in (2, 4):
            assert tile(B, *deep[8], tr) == (TILEK, 0), B
        for B in (1, 8):  # 64 / 512 tiles of 32 x 64: outside conv_tilek's window
            t, ws = tile(B, *deep[8], tr)
            assert t != TILEK and t != TILE32, B
        assert tile(1, *deep[8], tr)[1] > 0   # batch 1: conv_mm with its cross-workgroup K split (scratch for partial tiles)


def test_conv_tile32_trained_weight_dynamic_range(dev):
    """F(4,3)'s transform constants (1/24 ... 8) amplify fp32 rounding more than F(2,3) / the direct form: bound the wave-tile
    kernel's error against an fp64 reference at a TRAINED-weight-like dynamic range -- |w| log-uniform over 1e-3 ... 10 with
    random signs, inputs after GroupNorm+SiLU carrying a 30 sigma outlier per row -- for the forward (prologue + residual) and
    the data gradient.  The contract is 1e-3 relative to the tensor's max norm (north_star); the kernel has to keep a
    wide margin: 1e-5 in that norm (measured 7e-7 forward, 1.4e-6 data gradient; stock fp32 torch 2e-7), and 1e-4 ELEMENT-WISE
    against each output's own conditioning sum |w|.|a| (measured 3.8e-5 / 1.9e-5, stock fp32 3.6e-7: a Winograd tile spreads the
    rounding of its largest input -- the 30 sigma outlier x a weight of 10 -- over the tile's other outputs; still 10x inside
    the contract even when read element by element)."""
    B, C, L, G = 2, 32, 1024, 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, L, generator=g)
    x[:, :, 100] += 30.0                                      # one 30 sigma outlier column (survives SiLU as ~+30)
    x[:, 5, 700] -= 30.0
    mag = torch.exp(torch.empty(C, C, 3).uniform_(math.log(1e-3), math.log(10.0), generator=g))
    w = mag * torch.where(torch.rand(C, C, 3, generator=g) < 0.5, -1.0, 1.0)
    b = torch.randn(C, generator=g)
    gamma, beta = torch.randn(C, generator=g) * 0.5 + 1, torch.randn(C, generator=g) * 0.1
    r = torch.randn(B, C, L, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    a64 = F.silu(F.group_norm(x.double(), G, gamma.double(), beta.double(), eps=1e-5))
    ref = F.conv1d(a64, w.double(), b.double(), padding=1) + r.double()
    out = ops.conv1d(xd, wd, b.to(dev), pad=1, prologue=1, pro_stats=ops.gn_stats(xd, G), pro_gamma=gamma.to(dev),
                     pro_beta=beta.to(dev), groups=G, res=r.to(dev))
    assert rel_err(out, ref) < 1e-5
    # element-wise too, relative to each output's own conditioning sum |w| * |a| (what an fp32 dot product can promise)
    cond = F.conv1d(a64.abs(), w.double().abs(), None, padding=1) + r.double().abs() + b.double().abs()[None, :, None]
    assert ((out.double().cpu() - ref).abs() / cond).max().item() < 1e-4
    dref = F.conv_transpose1d(x.double(), w.double(), None, padding=1)
    dx = ops.conv1d(xd, wd, None, pad=1, transposed=True)
    assert rel_err(dx, dref) < 1e-5
    dcond = F.conv_transpose1d(x.double().abs(), w.double().abs(), None, padding=1)
    assert ((dx.double().cpu() - dref).abs() / dcond).max().item() < 1e-4


def _trained_like(shape, gen, lo=1e-3, hi=10.0):
    """|w| log-uniform over lo ... hi with random signs (the recipe of test_conv_tile32_trained_weight_dynamic_range)."""
    mag = torch.exp(torch.empty(*shape).uniform_(math.log(lo), math.log(hi), generator=gen))
    return mag * torch.where(torch.rand(*shape, generator=gen) < 0.5, -1.0, 1.0)


@pytest.mark.parametrize("family", ["mm4", "tilek"])
def test_deep_f43_convs_trained_weight_dynamic_range(dev, family, monkeypatch):
    """The F(4,3) kernels that carry the step's flops -- conv_mm4 (depths 3-7) and conv_tilek (depth 8, batch-1 deep layers) --
    at the LONGEST reduction of the README net (R = 1024 input channels) with a trained-weight-like dynamic range: |w| log-uniform
    over 1e-3 ... 10, inputs with 30 sigma outliers, fp64 reference.  Same bounds as the 32-channel wave tile: 1e-5 of the output's
    max norm, 1e-4 element-wise against each output's conditioning sum |w|.|x| (+ |res| + |bias|); forward and data gradient."""
    from ctypes import byref
    B, R, M, L = 1, 1024, 32, 128
    if family == "mm4":
        monkeypatch.setenv("ADP_MM4_MIN_BLOCKS", "1")
        monkeypatch.setenv("ADP_MM4_LIGHT_MIN_BLOCKS", "1000000")
        monkeypatch.setenv("ADP_CONV_TILEK", "0")
        want = 64032128
    else:
        monkeypatch.setenv("ADP_TILEK_MIN_R", "256")
        monkeypatch.setenv("ADP_TILEK_MIN_TILES", "1")
        want = 48000064
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, R, L, generator=g)
    x[:, :, 37] += 30.0
    x[:, 11, 90] -= 30.0
    w = _trained_like((M, R, 3), g)
    b, r = torch.randn(M, generator=g), torch.randn(B, M, L, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    3, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == want, "the case must dispatch to the kernel under test"
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=1) + r.double()
    out = ops.conv1d(xd, wd, b.to(dev), pad=1, res=r.to(dev))
    assert rel_err(out, ref) < 1e-5
    cond = F.conv1d(x.double().abs(), w.double().abs(), None, padding=1) + r.double().abs() + b.double().abs()[None, :, None]
    assert ((out.double().cpu() - ref).abs() / cond).max().item() < 1e-4
    # data gradient of a conv 32 -> 1024 read through the transposed weight view: [B, 1024, L] -> [B, 32, L], the same reduction
    wt = _trained_like((R, M, 3), g)
    d2 = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                     3, 1, 1, 1, 1, 1, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d2)) == want
    dref = F.conv_transpose1d(x.double(), wt.double(), None, padding=1)
    dx = ops.conv1d(xd, wt.to(dev), None, pad=1, transposed=True)
    assert rel_err(dx, dref) < 1e-5
    dcond = F.conv_transpose1d(x.double().abs(), wt.double().abs(), None, padding=1)
    assert ((dx.double().cpu() - dref).abs() / dcond).max().item() < 1e-4


def test_wgrad_f43_trained_gradient_dynamic_range(dev, monkeypatch):
    """The W4 (F(4,3)) weight gradient at the shape with the LONGEST position reduction it serves in the README net -- [4, 128, 4096]:
    16384 positions per (m, r, tap) through A e (constants up to 8) and B^T d (up to 5) -- with a heavy-tailed upstream gradient
    (|dy| log-uniform over 1e-3 ... 10) and 30 sigma outliers in the input; fp64 reference; 1e-5 of the max norm and 1e-4
    element-wise against sum |dy| |x|.  (The emulator runs a shorter row: the arithmetic per chunk is the same.)"""
    B, C, L = (4, 128, 4096) if dev.type == "cuda" else (1, 64, 512)
    _wgrad_family_env(monkeypatch, "4")
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, L, generator=g)
    x[:, :, 100] += 30.0
    x[:, 3, 300] -= 30.0
    dy = _trained_like((B, C, L), g)
    w = torch.zeros(C, C, 3, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(F.conv1d(x.double(), w, None, padding=1), w, dy.double())
    wa = torch.zeros(C, C, 3, dtype=torch.float64, requires_grad=True)
    cond, = torch.autograd.grad(F.conv1d(x.double().abs(), wa, None, padding=1), wa, dy.double().abs())
    dw, _ = ops.conv1d_wgrad(x.to(dev), dy.to(dev), 3, pad=1)
    assert rel_err(dw, ref) < 1e-5
    assert ((dw.double().cpu() - ref).abs() / cond).max().item() < 1e-4


@pytest.mark.parametrize("B,R,M,L,tr", [(2, 128, 64, 256, False), (1, 160, 160, 132, False), (2, 128, 128, 128, True),
                                        (1, 256, 32, 260, True), (3, 128, 96, 4, False), (1, 128, 64, 1024, False),
                                        (1, 192, 64, 384, True)])
@pytest.mark.parametrize("bkt", ["32", "64", "light", "npg3", "vpre64", "vprelight"])
def test_conv_mm4_winograd_f43(dev, B, R, M, L, tr, bkt, monkeypatch):
    """conv_mm4.hip: the Winograd F(4,3) block of the wide kernel-3 convs (MMA waves split the six planes and the chunk's
    channels; 32 rows x 128 positions per block): forward with bias / e_scale / residual / out_pre and the GroupNorm partial
    statistics of the output (row-pair entries per 128-position tile), data gradient through the transposed weight view;
    ragged last tiles (L = 132, 260), a tile shorter than one quad row (L = 4), several tiles per row (L = 1024); against fp64."""
    from ctypes import byref
    monkeypatch.setenv("ADP_MM4_MIN_BLOCKS", "1")
    monkeypatch.setenv("ADP_WINO4_MIN_R", "128")
    if bkt.startswith("vpre"):  # the loader waves stage the transformed inputs (full 128-position tiles only)
        if L % 128 or (bkt == "vpre64" and R % 64):
            pytest.skip("pre-transformed inputs: whole 128-position tiles")
        monkeypatch.setenv("ADP_MM4_VPRE", "1")
        bkt = "light" if bkt == "vprelight" else "64"
    else:
        monkeypatch.setenv("ADP_MM4_VPRE", "0")
    if bkt == "light":  # the 8-wave block with two K groups (short-K layers, two blocks per CU): one GroupNorm entry per tile
        monkeypatch.setenv("ADP_MM4_LIGHT_MIN_BLOCKS", "1")
    elif bkt == "npg3":  # the 16-wave block: three plane groups of two planes (ADP_MM4_NPG=3, 64-channel chunks)
        if R % 64:
            pytest.skip("64-channel chunks")
        monkeypatch.setenv("ADP_MM4_LIGHT_MIN_BLOCKS", "1000000")
        monkeypatch.setenv("ADP_MM4_NPG", "3")
    else:
        monkeypatch.setenv("ADP_MM4_LIGHT_MIN_BLOCKS", "1000000")
        monkeypatch.setenv("ADP_MM4_BKT", bkt)
    G = 8
    x = rnd(B, R, L, seed=1)
    w = rnd(R, M, 3, seed=2, scale=0.05) if tr else rnd(M, R, 3, seed=2, scale=0.05)
    b, res, sc = rnd(M, seed=3), rnd(B, M, L, seed=4), rnd(B * M, seed=5)
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    3, 1, 1, 1, 1, int(tr), 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 64032128, "case must dispatch to the F(4,3) block"
    if tr:
        ref = F.conv_transpose1d(x.double(), w.double(), None, padding=1)
    else:
        ref = F.conv1d(x.double(), w.double(), None, padding=1)
    pre_ref = ref + b.double()[None, :, None]
    ref = pre_ref * sc.double().view(B, M, 1) + res.double()
    pre = torch.empty(B, M, L).to(dev)
    gn = ops.GnPart()
    out = ops.conv1d(xd, wd, b.to(dev), pad=1, transposed=tr, e_scale=sc.to(dev), res=res.to(dev), out_pre=pre, gn=gn)
    assert rel_err(out, ref) < 1e-5 and rel_err(pre, pre_ref) < 1e-5
    assert gn.part is not None and gn.part.shape[2] == (1 if bkt == "light" else 2) * ((L + 127) // 128)
    assert gn.part[..., 2].sum(dim=2).eq(4 * L).all()
    if M % (4 * G) == 0:
        st = ops.gn_finalize(gn.part, G)
        g64 = ref.view(B, G, -1)
        assert rel_err(st[..., 0], g64.mean(-1)) < 2e-5
        assert rel_err(st[..., 1], (g64.var(-1, unbiased=False) + 1e-5).rsqrt()) < 2e-5
    # plain call (no epilogue operands) and the switch back to conv_mm's F(2,3) variant agree
    plain = ops.conv1d(xd, wd, None, pad=1, transposed=tr)
    monkeypatch.setenv("ADP_CONV_WINO4", "0")
    assert _C.query("adp_conv1d_tile", byref(d)) != 64032128
    assert rel_err(plain, ops.conv1d(xd, wd, None, pad=1, transposed=tr)) < 1e-5


@pytest.mark.parametrize("B,R,M,L", [(1, 512, 512, 128), (1, 1024, 512, 256), (2, 256, 48, 64), (1, 512, 1024, 192)])
@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("nkw", ["8", "16"])  # K slices = waves per workgroup
def test_conv_tilek1_projection(dev, B, R, M, L, tr, nkw, monkeypatch):
    """conv_tilek1.hip: the attention items' 1x1 projections at batch 1 (components.py:92-93) -- eight waves of a workgroup split
    the input channels of one 16 x 64 output tile, no cross-workgroup K split and no reduce launch: forward with bias / residual
    and the GroupNorm partial statistics of the output, data gradient through the transposed weight view; against fp64, and
    against conv_mm's split-K path on the same operands."""
    from ctypes import byref
    if nkw == "16" and R % 512:
        pytest.skip("sixteen slices take chunk pairs of 512 channels")
    monkeypatch.setenv("ADP_TILEK1_MIN_TILES", "1")
    monkeypatch.setenv("ADP_TILEK1_NKW", nkw)
    G = 4
    x = rnd(B, R, L, seed=1) * 1.3 + 0.2
    w = rnd(R, M, 1, seed=2, scale=0.05) if tr else rnd(M, R, 1, seed=2, scale=0.05)
    b, r = rnd(M, seed=3), rnd(B, M, L, seed=4)
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    1, 1, 1, 0, 1, int(tr), 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 47000064, "case must dispatch to the 1x1 wave-tile kernel"
    assert _C.query("adp_conv1d_ws_bytes", byref(d)) == 0
    if tr:
        ref = F.conv_transpose1d(x.double(), w.double(), None)
        out = ops.conv1d(xd, wd, None, transposed=True)
        assert rel_err(out, ref) < 1e-5
    else:
        ref = F.conv1d(x.double(), w.double(), b.double()) + r.double()
        gn = ops.GnPart()
        out = ops.conv1d(xd, wd, b.to(dev), res=r.to(dev), gn=gn)
        assert rel_err(out, ref) < 1e-5
        assert gn.part is not None and gn.part.shape == (B, M // 4, L // 64, 3) and gn.part[..., 2].eq(256).all()
        if M % (4 * G) == 0:
            st = ops.gn_finalize(gn.part, G)
            g64 = ref.view(B, G, -1)
            assert rel_err(st[..., 0], g64.mean(-1)) < 2e-5
            assert rel_err(st[..., 1], (g64.var(-1, unbiased=False) + 1e-5).rsqrt()) < 2e-5
    sh = None
    if tr and nkw == "8":  # pixel-shuffle store (the DownsampleItem data gradient: out[b][m / sp][n * sp + m % sp] + residual)
        for sp in (2, 4):
            rs = rnd(B, M // sp, L * sp, seed=7)
            sh = ops.conv1d(xd, wd, None, transposed=True, store=1, sp=sp, res=rs.to(dev))
            want = ref.view(B, M // sp, sp, L).permute(0, 1, 3, 2).reshape(B, M // sp, L * sp) + rs.double()
            assert rel_err(sh, want) < 1e-5
    monkeypatch.setenv("ADP_CONV_TILEK1", "0")
    assert _C.query("adp_conv1d_tile", byref(d)) != 47000064
    other = ops.conv1d(xd, wd, None, transposed=True) if tr else ops.conv1d(xd, wd, b.to(dev), res=r.to(dev))
    assert rel_err(out, other) < 1e-5
    if sh is not None:
        assert rel_err(sh, ops.conv1d(xd, wd, None, transposed=True, store=1, sp=4, res=rs.to(dev))) < 1e-5


@pytest.mark.parametrize("B,C,L", [(2, 128, 256), (1, 160, 132), (2, 192, 1024), (1, 256, 260), (2, 256, 128), (1, 512, 64)])
@pytest.mark.parametrize("bkt", ["64", "light", "tilek1", "tilek2", "mm"])
def test_conv_mm4_leaves_first_stage_of_groupnorm_backward(dev, B, C, L, bkt, monkeypatch):
    """The data gradient of a ConvBlock conv is the output gradient of SiLU(GroupNorm(x)) (components.py:89): conv_mm4's epilogue
    leaves (sum ds * xhat, sum ds) per row and 128-position tile (adp_conv_desc.gnb_ab; conv_tilek: per 64-position tile, 16- / 32-row
    tiles), adp_gn_silu_bwd_apply_ab finishes from them -- against autograd through the fp64 composite, and against the two-launch
    GroupNorm backward on the same da."""
    from ctypes import byref
    tilek = bkt.startswith("tilek")
    if tilek:
        if C % 256 or L % 64:
            pytest.skip("conv_tilek takes chunk pairs of 8 K slices and whole 64-position tiles")
        monkeypatch.setenv("ADP_TILEK_MIN_R", "256")
        monkeypatch.setenv("ADP_TILEK_MIN_TILES", "1")
        monkeypatch.setenv("ADP_TILEK_RB", bkt[-1])
    else:
        monkeypatch.setenv("ADP_CONV_TILEK", "0")
    if bkt == "mm":  # conv_mm's F(2,3) blocks; the small cases take its cross-workgroup K split (sums left by the reduce kernel)
        monkeypatch.setenv("ADP_CONV_WINO4", "0")
    monkeypatch.setenv("ADP_MM4_MIN_BLOCKS", "1")
    monkeypatch.setenv("ADP_WINO4_MIN_R", "64")
    monkeypatch.setenv("ADP_MM4_LIGHT_MIN_BLOCKS", "1" if bkt == "light" else "1000000")
    G = 8
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.4).double().requires_grad_()
    gamma = (rnd(C, seed=2) * 0.5 + 1).double().requires_grad_()
    beta = (rnd(C, seed=3) * 0.2).double().requires_grad_()
    w = rnd(C, C, 3, seed=4, scale=0.05)  # forward weight [M = C, R = C, 3]
    gy, dres = rnd(B, C, L, seed=5), rnd(B, C, L, seed=6)
    y = F.conv1d(ref_gn_silu(x, G, gamma, beta), w.double(), None, padding=1)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (x, gamma, beta), gy.double())
    xd, wd, gyd = x.detach().float().to(dev), w.to(dev), gy.to(dev)
    gd, bd = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    st = ops.gn_stats(xd, G)
    d = _C.ConvDesc(_C.ptr(gyd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(gyd), None, B, C, C, L, C, L,
                    3, 1, 1, 1, 1, 1, 0, 1, 0, 1, 0)
    tile = _C.query("adp_conv1d_tile", byref(d))
    if bkt == "mm":
        assert tile not in (48000064, 64032128) and tile >= 40000000, "case must dispatch to conv_mm's Winograd blocks"
    else:
        assert tile == (48000064 if tilek else 64032128), "case must dispatch to the named kernel"
    gb = ops.GnBwdPart(xd, st, gd, bd, G)
    dact = ops.conv1d(gyd, wd, None, pad=1, transposed=True, gnb=gb)
    split = bkt == "mm" and _C.query("adp_conv1d_ws_bytes", byref(d)) > 0
    if bkt == "mm" and (B, C, L) in ((2, 256, 128), (1, 512, 64)):
        assert split, "the small cases are meant to cover the split-K reduce kernel"
    want_e = (L + 1023) // 1024 if split else ((L + 63) // 64 if (tilek or bkt == "mm") else (L + 127) // 128)
    assert gb.ab is not None and tuple(gb.ab.shape) == (B, C, want_e, 2)
    dx, dg, db = ops.gn_silu_bwd(xd, dact, st, gd, bd, G, dres=dres.to(dev), ab=gb.ab)
    assert rel_err(dx, dx_ref + dres.double()) < 2e-5 and rel_err(dg, dg_ref) < 2e-5 and rel_err(db, db_ref) < 2e-5
    dx2, dg2, db2 = ops.gn_silu_bwd(xd, dact, st, gd, bd, G, dres=dres.to(dev))  # the two-launch form on the same da
    assert rel_err(dx, dx2) < 2e-6 and rel_err(dg, dg2) < 2e-6 and rel_err(db, db2) < 2e-6


@pytest.mark.parametrize("B,L,nw", [(2, 256, 1), (1, 1024, 4), (2, 2048, 16)])
def test_conv_tile32_leaves_first_stage_of_groupnorm_backward(dev, B, L, nw, monkeypatch):
    """The same contract for the 32-channel wave-tile kernel of depth 1 (conv_tile.hip): one entry per channel and workgroup
    (1 / 4 / 16 waves = 64 / 256 / 1024 positions), summed over the workgroup's waves in wave order."""
    from ctypes import byref
    monkeypatch.setenv("ADP_TILE_NW", str(nw))
    C, G = 32, 8
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.4).double().requires_grad_()
    gamma = (rnd(C, seed=2) * 0.5 + 1).double().requires_grad_()
    beta = (rnd(C, seed=3) * 0.2).double().requires_grad_()
    w = rnd(C, C, 3, seed=4, scale=0.1)
    gy, dres = rnd(B, C, L, seed=5), rnd(B, C, L, seed=6)
    y = F.conv1d(ref_gn_silu(x, G, gamma, beta), w.double(), None, padding=1)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (x, gamma, beta), gy.double())
    xd, wd, gyd = x.detach().float().to(dev), w.to(dev), gy.to(dev)
    gd, bd = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    st = ops.gn_stats(xd, G)
    d = _C.ConvDesc(_C.ptr(gyd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(gyd), None, B, C, C, L, C, L,
                    3, 1, 1, 1, 1, 1, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 32064, "case must dispatch to the 32-channel wave-tile kernel"
    gb = ops.GnBwdPart(xd, st, gd, bd, G)
    dact = ops.conv1d(gyd, wd, None, pad=1, transposed=True, gnb=gb)
    assert gb.ab is not None and tuple(gb.ab.shape) == (B, C, L // 64 // nw, 2)
    assert rel_err(dact, F.conv_transpose1d(gy.double(), w.double(), None, padding=1)) < 1e-5
    dx, dg, db = ops.gn_silu_bwd(xd, dact, st, gd, bd, G, dres=dres.to(dev), ab=gb.ab)
    assert rel_err(dx, dx_ref + dres.double()) < 2e-5 and rel_err(dg, dg_ref) < 2e-5 and rel_err(db, db_ref) < 2e-5
    dx2, dg2, db2 = ops.gn_silu_bwd(xd, dact, st, gd, bd, G, dres=dres.to(dev))
    assert rel_err(dx, dx2) < 2e-6 and rel_err(dg, dg2) < 2e-6 and rel_err(db, db2) < 2e-6


@pytest.mark.parametrize("B,R,M,L,tr,ksmax", [(1, 1024, 32, 128, False, 2), (2, 2048, 32, 132, True, 4), (1, 1024, 64, 8, False, 2)])
def test_conv_mm4_cross_workgroup_split_k(dev, B, R, M, L, tr, ksmax, monkeypatch):
    """conv_mm4 with fewer tiles than the chip has CUs (depth 8 at batch 4: 128 tiles of 32 x 128): the channel reduction is cut
    into 2 / 4 slices run by separate workgroups, the raw output-transformed partial tiles go to the caller's scratch and
    conv_splitk_reduce[_gn] sums them in a fixed order and owns the epilogue + the GroupNorm partials."""
    from ctypes import byref
    monkeypatch.setenv("ADP_WINO4_MIN_R", "128")
    monkeypatch.setenv("ADP_MM4_KS_MAX", str(ksmax))
    blocks = (M // 32) * ((L + 127) // 128) * B
    monkeypatch.setenv("ADP_MM4_MIN_BLOCKS", str(blocks * ksmax))
    x = rnd(B, R, L, seed=1)
    w = rnd(R, M, 3, seed=2, scale=0.05) if tr else rnd(M, R, 3, seed=2, scale=0.05)
    b, res, sc = rnd(M, seed=3), rnd(B, M, L, seed=4), rnd(B * M, seed=5)
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L,
                    3, 1, 1, 1, 1, int(tr), 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 64032128
    assert _C.query("adp_conv1d_ws_bytes", byref(d)) == ksmax * B * M * L * 4, "this shape is meant to take the split-K path"
    ref = F.conv_transpose1d(x.double(), w.double(), None, padding=1) if tr else F.conv1d(x.double(), w.double(), None, padding=1)
    pre_ref = ref + b.double()[None, :, None]
    ref = pre_ref * sc.double().view(B, M, 1) + res.double()
    pre = torch.empty(B, M, L).to(dev)
    gn = ops.GnPart()
    out = ops.conv1d(xd, wd, b.to(dev), pad=1, transposed=tr, e_scale=sc.to(dev), res=res.to(dev), out_pre=pre, gn=gn)
    assert rel_err(out, ref) < 1e-5 and rel_err(pre, pre_ref) < 1e-5
    assert gn.part is not None and gn.part[..., 2].sum(dim=2).eq(4 * L).all()
    st = ops.gn_finalize(gn.part, 8)
    g64 = ref.view(B, 8, -1)
    assert rel_err(st[..., 0], g64.mean(-1)) < 2e-5 and rel_err(st[..., 1], (g64.var(-1, unbiased=False) + 1e-5).rsqrt()) < 2e-5


@pytest.mark.parametrize("bkt", ["32", "64"])
@pytest.mark.parametrize("B,R,M,L,tr,big", [(2, 256, 64, 200, False, False), (1, 320, 96, 132, True, False),
                                             (2, 256, 128, 260, True, True), (1, 512, 64, 64, False, True)])
def test_conv_mm_1x1_wide_chunks(dev, B, R, M, L, tr, big, bkt, monkeypatch):
    """1x1 convs on conv_mm with 64-channel staged chunks (16 instead of 8 MFMAs per wave and barrier: the DownsampleItem data
    gradients over the space-to-depth view, the attention projections) against the 32-channel chunks and torch: 32- and 64-row
    tiles (`big`: with the wide-N block), ragged last tile, a channel count that is a multiple of 64 but not of 128."""
    monkeypatch.setenv("ADP_MM_K1_BKT", bkt)
    if big:
        monkeypatch.setenv("ADP_MM_MIN_BLOCKS", "1")
    x = rnd(B, R, L, seed=1)
    w = rnd(R, M, 1, seed=2, scale=0.05) if tr else rnd(M, R, 1, seed=2, scale=0.05)
    b, res = rnd(M, seed=3), rnd(B, M, L, seed=4)
    ref = (F.conv_transpose1d(x, w, None) + b[None, :, None] if tr else F.conv1d(x, w, b)) + res
    out = ops.conv1d(x.to(dev), w.to(dev), b.to(dev), transposed=tr, res=res.to(dev))
    assert rel_err(out, ref) < TOL


MM_RESAMPLE_CASES = [
    # B, R, M, Lin, KT, stride, pad, up -- DownsampleItem (kernel = stride) and UpsampleItem (nearest + k3) on conv_mm
    (2, 32, 64, 256, 2, 2, 0, 1),
    (1, 64, 32, 136, 2, 2, 0, 1),     # ragged tile, 2 chunks
    (2, 32, 64, 512, 4, 4, 0, 1),     # 16-channel chunks
    (1, 64, 128, 256, 4, 4, 0, 1),
    (2, 32, 32, 66, 3, 1, 1, 2),      # Lin * up = 132: ragged
    (1, 64, 64, 40, 3, 1, 1, 4),
    (3, 96, 32, 64, 3, 1, 1, 2),
]


@pytest.mark.parametrize("wino", ["0", "1"])
@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", MM_RESAMPLE_CASES)
def test_conv_mm_resample(dev, B, R, M, L, KT, stride, pad, up, wino, monkeypatch):
    from audio_diffusion_pytorch_amd import _C
    if wino == "1" and KT != 3:
        pytest.skip("Winograd F(2,3) is the kernel-3 form")
    monkeypatch.setenv("ADP_CONV_WINO", wino)
    monkeypatch.setenv("ADP_WINO_MIN_R", "32")
    x, w, b = rnd(B, R, L, seed=1), rnd(M, R, KT, seed=2, scale=0.2), rnd(M, seed=3)
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    ref = F.conv1d(xr, w, b, stride=stride, padding=pad)
    xd, wd = x.to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M,
                    ref.shape[-1], KT, stride, 1, pad, up, 0, 0, 1, 0, 1, 0)
    tile = _C.query("adp_conv1d_tile", d)
    assert tile >= 1000000 and (tile >= 40000000) == (wino == "1"), "case must dispatch to the conv_mm family"
    # with the SkipModulate epilogue of the up path: out = skip + scale[b, m] * (conv + bias), pre-merge value kept
    skip, scale = rnd(*ref.shape, seed=5), rnd(B, M, seed=6)
    pre = torch.empty(ref.shape, device=dev)
    out = ops.conv1d(xd, wd, b.to(dev), stride=stride, pad=pad, up=up, e_scale=scale.to(dev).view(-1), e_bstride=M,
                     res=skip.to(dev), out_pre=pre)
    assert rel_err(pre, ref) < TOL
    assert rel_err(out, skip + scale[:, :, None] * ref) < TOL


@pytest.mark.parametrize("wino", ["0", "1", "wide"])
@pytest.mark.parametrize("B,Rf,Mf,L,up", [(2, 64, 96, 72, 2), (1, 32, 64, 40, 4), (2, 64, 32, 64, 4)])
def test_conv_mm_upsample_dgrad_pooled_store(dev, B, Rf, Mf, L, up, wino, monkeypatch):
    """Data gradient of UpsampleItem (nearest x up, then k3 conv) on conv_mm: the transposed-weight conv over dy with
    the pooled store (sum of the `up` replicas of each source position) + the residual epilogue, direct and Winograd
    ("wide": the Winograd variant's 256-position blocks)."""
    monkeypatch.setenv("ADP_CONV_WINO", "0" if wino == "0" else "1")
    monkeypatch.setenv("ADP_WINO_MIN_R", "32")
    if wino == "wide":
        monkeypatch.setenv("ADP_MM_MIN_BLOCKS", "1")
    x = rnd(B, Rf, L, seed=1).requires_grad_()
    w = rnd(Mf, Rf, 3, seed=2, scale=0.2)
    y = F.conv1d(F.interpolate(x, scale_factor=up, mode="nearest"), w, None, padding=1)
    dy, res = rnd(*y.shape, seed=9), rnd(B, Rf, L, seed=10)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dx = ops.conv1d(dy.to(dev), w.to(dev), None, pad=1, transposed=True, store=2, sp=up, res=res.to(dev))
    assert rel_err(dx, dx_ref + res) < TOL


@pytest.mark.parametrize("light", ["0", "1"])
@pytest.mark.parametrize("B,R,M,L,up", [(2, 128, 64, 66, 2), (1, 160, 32, 40, 4), (1, 256, 96, 260, 2), (2, 128, 32, 4, 4)])
def test_conv_mm4_upsample_and_pooled_dgrad(dev, B, R, M, L, up, light, monkeypatch):
    """conv_mm4 for the UpsampleItem convs (nearest x up folded into the loader: the LDS tile holds virtual positions) and for their
    data gradients (transposed weight view over dy, pooled store = sums of `up` adjacent outputs of a lane's quad, + residual);
    both block shapes; ragged tiles; the gather bit for bit against the same kernel on the materialised upsampled tensor."""
    from ctypes import byref
    monkeypatch.setenv("ADP_MM4_MIN_BLOCKS", "1")
    monkeypatch.setenv("ADP_MM4_LIGHT_MIN_BLOCKS", "1" if light == "1" else "1000000")
    x = rnd(B, R, L, seed=1).requires_grad_()
    w = rnd(M, R, 3, seed=2, scale=0.05)
    b, sc = rnd(M, seed=3), rnd(B * M, seed=5)
    xu = F.interpolate(x, scale_factor=up, mode="nearest")
    y = F.conv1d(xu, w, None, padding=1)
    res = rnd(*y.shape, seed=4)
    xd, wd = x.detach().to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M, L * up,
                    3, 1, 1, 1, up, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", byref(d)) == 64032128, "the upsample conv must dispatch to the F(4,3) block"
    pre = torch.empty_like(y).to(dev)
    gn = ops.GnPart()
    out = ops.conv1d(xd, wd, b.to(dev), pad=1, up=up, e_scale=sc.to(dev), res=res.to(dev), out_pre=pre, gn=gn)
    pre_ref = y.detach() + b[None, :, None]
    assert rel_err(pre, pre_ref) < 1e-5 and rel_err(out, pre_ref * sc.view(B, M, 1) + res) < 1e-5
    assert gn.part is not None and gn.part[..., 2].sum(dim=2).eq(4 * L * up).all()
    # data gradient: dy [B, M, L*up] -> dx [B, R, L] through the pooled store
    dy, rx = rnd(*y.shape, seed=9), rnd(B, R, L, seed=10)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    if M >= 128:  # (the gradient's input channels are the conv's output channels: the F(4,3) block takes it from 128)
        dd = _C.ConvDesc(_C.ptr(pre), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, M, M, L * up, R,
                         L * up, 3, 1, 1, 1, 1, 1, 0, 1, 2, up, 0)
        assert _C.query("adp_conv1d_tile", byref(dd)) == 64032128
    dx = ops.conv1d(dy.to(dev), wd, None, pad=1, transposed=True, store=2, sp=up, res=rx.to(dev))
    assert rel_err(dx, dx_ref + rx) < 1e-5
    # integer index math, bit-exact: F(4,3)'s constants (1/6, 1/24) do not keep integer inputs exact, so the one-hot-tap identity
    # test of the F(2,3) / direct kernels does not carry over -- instead the SAME kernel is fed the materialised upsampled tensor:
    # identical arithmetic on identical LDS tiles, so the outputs are equal bit for bit iff the loader's gather (source index
    # floor(u / up), zero padding at both ends) filled the tile with exactly the values the materialised row holds
    plain = ops.conv1d(xu.detach().contiguous().to(dev), wd, b.to(dev), pad=1)
    assert torch.equal(ops.conv1d(xd, wd, b.to(dev), pad=1, up=up).cpu(), plain.cpu())


@pytest.mark.parametrize("B,R,M,L,up", [(2, 128, 128, 66, 2), (1, 160, 128, 40, 4)])
def test_conv_mm4_pooled_dgrad_of_wide_upsample_conv(dev, B, R, M, L, up, monkeypatch):
    """The pooled-store data gradient on conv_mm4 proper (>= 128 gradient channels), against autograd."""
    from ctypes import byref
    monkeypatch.setenv("ADP_MM4_MIN_BLOCKS", "1")
    x = rnd(B, R, L, seed=1).requires_grad_()
    w = rnd(M, R, 3, seed=2, scale=0.05)
    y = F.conv1d(F.interpolate(x, scale_factor=up, mode="nearest"), w, None, padding=1)
    dy, rx = rnd(*y.shape, seed=9), rnd(B, R, L, seed=10)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dyd, wd = dy.to(dev), w.to(dev)
    dd = _C.ConvDesc(_C.ptr(dyd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(dyd), None, B, M, M, L * up, R,
                     L * up, 3, 1, 1, 1, 1, 1, 0, 1, 2, up, 0)
    assert _C.query("adp_conv1d_tile", byref(dd)) == 64032128, "the pooled-store gradient must dispatch to the F(4,3) block"
    dx = ops.conv1d(dyd, wd, None, pad=1, transposed=True, store=2, sp=up, res=rx.to(dev))
    assert rel_err(dx, dx_ref + rx) < 1e-5
    assert rel_err(ops.conv1d(dyd, wd, None, pad=1, transposed=True, store=2, sp=up), dx_ref) < 1e-5


# ------------------------------------------------------------------ integer index math: bit-exact (north_star)
@pytest.mark.parametrize("wino", ["0", "1"])
@pytest.mark.parametrize("B,C,L,up", [(2, 32, 96, 2), (1, 64, 40, 4), (2, 8, 50, 4), (1, 2, 64, 2), (1, 32, 30, 3)])
def test_nearest_upsample_gather_is_bit_exact(dev, B, C, L, up, wino, monkeypatch):
    """UpsampleItem = nearest upsample (source index floor(dst / f)) then a k3 conv, with the gather folded into the conv
    loaders.  With a one-hot centre tap the conv is the identity, so the output must EQUAL the nearest-upsampled input
    -- integer-valued inputs keep the evaluation orders this test PINS exact: the direct form (wino = 0) and Winograd F(2,3)
    (wino = 1 with ADP_WINO_MIN_R = 32: conv_mm's WN variant, whose constants 1/2 are dyadic), on every kernel family that serves
    these <= 64-channel shapes (MFMA, direct VALU, generic), so torch.equal tests the index math alone.  The F(4,3) kernels
    (conv_mm4 from 128 channels, conv_tile32 at 32 -> 32 'same' convs without upsample) are NOT pinned here: their constants
    1/6 and 1/24 are not dyadic, so integer inputs do not make them exact -- their fused gather is compared bit-for-bit against
    the materialised upsample through the SAME arithmetic in test_conv_mm4_upsample_and_pooled_dgrad."""
    monkeypatch.setenv("ADP_CONV_WINO", wino)
    monkeypatch.setenv("ADP_WINO_MIN_R", "32")
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-64, 64, (B, C, L), generator=g).float()
    w = torch.zeros(C, C, 3)
    w[torch.arange(C), torch.arange(C), 1] = 1.0
    out = ops.conv1d(x.to(dev), w.to(dev), None, pad=1, up=up)
    ref = x.repeat_interleave(up, dim=2)  # dst l reads source l // up
    assert torch.equal(out.cpu(), ref)
    # shifted taps: output l = upsampled[l - 1] / upsampled[l + 1] with the zero padding at the ends
    for tap, sh in ((0, 1), (2, -1)):
        w = torch.zeros(C, C, 3)
        w[torch.arange(C), torch.arange(C), tap] = 1.0
        out = ops.conv1d(x.to(dev), w.to(dev), None, pad=1, up=up).cpu()
        exp = torch.zeros_like(ref)
        if sh == 1:
            exp[..., 1:] = ref[..., :-1]
        else:
            exp[..., :-1] = ref[..., 1:]
        assert torch.equal(out, exp), tap


@pytest.mark.parametrize("B,R,f,L", [(2, 32, 2, 128), (1, 32, 4, 256), (2, 8, 4, 64), (1, 2, 4, 1024), (1, 16, 2, 36),
                                     (2, 4, 3, 30)])
def test_strided_window_is_bit_exact(dev, B, R, f, L):
    """DownsampleItem = Conv1d(kernel = stride = f): output l reads the window x[l*f .. l*f + f - 1].  One-hot weights
    that route input (r, k) to output channel r*f + k turn the conv into the space-to-depth permutation, which must be
    reproduced exactly (torch.equal) by every strided kernel variant and by adp_unshuffle's path for f = 3."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, R, L, generator=g)
    M = R * f
    w = torch.zeros(M, R, f)
    for r in range(R):
        for k in range(f):
            w[r * f + k, r, k] = 1.0
    ref = x.view(B, R, L // f, f).permute(0, 1, 3, 2).reshape(B, M, L // f)
    if f in (2, 4):
        out = ops.conv1d(x.to(dev), w.to(dev), None, stride=f)
    else:
        out = ops.conv1d(ops.unshuffle(x.to(dev), f), w.view(M, R * f, 1).to(dev), None)
    assert torch.equal(out.cpu(), ref)


DIRECT_CASES = [
    # B, R, M, Lin, KT, stride, pad, up -- narrow layers on the VALU direct kernel (conv_direct.hip)
    (2, 8, 8, 2052, 3, 1, 1, 1),      # two workgroups of 1024 positions + ragged tail
    (1, 2, 8, 1024, 1, 1, 0, 1),
    (2, 8, 2, 512, 3, 1, 1, 1),
    (1, 8, 32, 1024, 4, 4, 0, 1),
    (2, 4, 16, 64, 2, 2, 0, 1),
    (1, 32, 8, 60, 3, 1, 1, 4),
    (2, 16, 8, 34, 3, 1, 1, 2),
    (1, 8, 40, 64, 3, 1, 1, 1),       # 5 output-channel groups
    (2, 5, 12, 772, 3, 1, 1, 1),      # 8-row kernel with 5 rows (clamped row loads), ragged channel group, threads beyond N
    (1, 2, 8, 260, 3, 1, 1, 1),       # 2-row instantiation
    (1, 8, 8, 4, 3, 1, 1, 1),         # one quad: both neighbours are padding
    (2, 32, 8, 1100, 3, 1, 1, 4),     # x4 upsample conv as four phase convs (conv_up4): several waves, ragged last workgroup
    (1, 12, 5, 66, 3, 1, 1, 4),       # 16-row instantiation with 12 rows, 5 output channels
    (1, 8, 8, 2, 3, 1, 1, 4),         # one input pair
]


@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", DIRECT_CASES)
def test_conv_direct_family(dev, B, R, M, L, KT, stride, pad, up):
    from audio_diffusion_pytorch_amd import _C
    x = rnd(B, R, L, seed=1).requires_grad_()
    w, b = rnd(M, R, KT, seed=2, scale=0.2), rnd(M, seed=3)
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    ref = F.conv1d(xr, w, b, stride=stride, padding=pad)
    xd, wd = x.detach().to(dev), w.to(dev)
    d = _C.ConvDesc(_C.ptr(xd), None, _C.ptr(wd), None, None, None, None, None, None, _C.ptr(xd), None, B, R, R, L, M,
                    ref.shape[-1], KT, stride, 1, pad, up, 0, 0, 1, 0, 1, 0)
    assert _C.query("adp_conv1d_tile", d) == 8999, "case must dispatch to the direct kernel"
    skip, scale = rnd(*ref.shape, seed=5), rnd(B, M, seed=6)
    pre = torch.empty(ref.shape, device=dev)
    out = ops.conv1d(xd, wd, b.to(dev), stride=stride, pad=pad, up=up, e_scale=scale.to(dev).view(-1), e_bstride=M,
                     res=skip.to(dev), out_pre=pre)
    assert rel_err(pre, ref.detach()) < TOL
    assert rel_err(out, skip + scale[:, :, None] * ref.detach()) < TOL
    if stride == 1:
        dy = rnd(*ref.shape, seed=9)
        (dx_ref,) = torch.autograd.grad(ref, x, dy)
        dx = ops.conv1d(dy.to(dev), wd, None, pad=(KT - 1) - pad, transposed=True, store=2 if up > 1 else 0, sp=up)
        assert rel_err(dx, dx_ref) < TOL


def test_conv_direct_concat_gn(dev):
    """x2 channel concat (AppendChannelsPlugin) and the GroupNorm+SiLU prologue on the direct kernel."""
    B, R1, R2, M, L, G = 2, 2, 2, 8, 256, 2
    x, x2 = rnd(B, R1, L, seed=1), rnd(B, R2, L, seed=2)
    w, b = rnd(M, R1 + R2, 1, seed=3, scale=0.3), rnd(M, seed=4)
    ref = F.conv1d(torch.cat([x, x2], 1), w, b)
    out = ops.conv1d(x.to(dev), w.to(dev), b.to(dev), x2=x2.to(dev))
    assert rel_err(out, ref) < TOL
    C = 8
    xx, ww = rnd(B, C, L, seed=5) * 1.4 + 0.3, rnd(C, C, 3, seed=6, scale=0.2)
    gamma, beta, res = rnd(C, seed=7) * 0.5 + 1, rnd(C, seed=8) * 0.1, rnd(B, C, L, seed=9)
    ref = F.conv1d(ref_gn_silu(xx, G, gamma, beta), ww, b, padding=1) + res
    xd = xx.to(dev)
    stats = ops.gn_stats(xd, G)
    out = ops.conv1d(xd, ww.to(dev), b.to(dev), pad=1, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                     pro_beta=beta.to(dev), groups=G, res=res.to(dev))
    assert rel_err(out, ref) < TOL


# ------------------------------------------------------------------ conv data gradients
@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", CONV_CASES)
def test_conv1d_dgrad(dev, B, R, M, L, KT, stride, pad, up):
    x = rnd(B, R, L, seed=1).requires_grad_()
    w = rnd(M, R, KT, seed=2, scale=0.2)
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    y = F.conv1d(xr, w, None, stride=stride, padding=pad)
    dy = rnd(*y.shape, seed=9)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dyd, wd = dy.to(dev), w.to(dev)
    if stride == 1:
        # transposed-weight conv over dy; the nearest-upsample gradient is the pooled store
        dx = ops.conv1d(dyd, wd, None, pad=(KT - 1) - pad, transposed=True, store=2 if up > 1 else 0, sp=up)
    else:
        # kernel == stride: a [C_in*f, C_out] GEMM whose rows are scattered back (pixel-shuffle store)
        assert KT == stride and pad == 0
        w2 = wd.permute(0, 1, 2).reshape(M, R * KT, 1)  # view [M][R*KT][1] == transposed layout [r=M][m=R*KT][1]
        dx = ops.conv1d(dyd, w2, None, transposed=True, store=1, sp=stride)
        if dx.shape[-1] < L:  # input tail not covered by any window
            dx = F.pad(dx, (0, L - dx.shape[-1]))
    assert dx.shape == dx_ref.shape
    assert rel_err(dx, dx_ref) < TOL


# ------------------------------------------------------------------ conv weight gradients
@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", CONV_CASES)
def test_conv1d_wgrad(dev, B, R, M, L, KT, stride, pad, up):
    x = rnd(B, R, L, seed=1)
    w = rnd(M, R, KT, seed=2, scale=0.2).requires_grad_()
    b = rnd(M, seed=3).requires_grad_()
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    y = F.conv1d(xr, w, b, stride=stride, padding=pad)
    dy = rnd(*y.shape, seed=9)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    dw, db = ops.conv1d_wgrad(x.to(dev), dy.to(dev), KT, stride=stride, pad=pad, up=up)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL


@pytest.mark.parametrize("B,C,L,G", [(2, 16, 700, 8), (3, 64, 200, 8), (5, 72, 90, 4)])
def test_conv1d_wgrad_prologue_accumulate(dev, B, C, L, G):
    x = rnd(B, C, L, seed=1) * 1.3 + 0.2
    gamma, beta = rnd(C, seed=4) * 0.5 + 1, rnd(C, seed=5) * 0.1
    w = rnd(C, C, 3, seed=2, scale=0.2).requires_grad_()
    y = F.conv1d(ref_gn_silu(x, G, gamma, beta), w, None, padding=1)
    dy = rnd(*y.shape, seed=9)
    (dw_ref,) = torch.autograd.grad(y, w, dy)
    xd = x.to(dev)
    stats = ops.gn_stats(xd, G)
    dw0 = rnd(C, C, 3, seed=11)
    dw, _ = ops.conv1d_wgrad(xd, dy.to(dev), 3, pad=1, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                             pro_beta=beta.to(dev), groups=G, dw=dw0.clone().to(dev), want_bias=False,
                             accumulate=True)
    assert rel_err(dw, dw_ref + dw0) < TOL


WGMM_CASES = [
    # B, R, M, L, KT -- 64x64 and 32x32 tiles, cross-workgroup position split and direct write, ragged chunk,
    # batch boundary inside a split, kernel 1
    (2, 64, 64, 256, 3),      # 1 tile, 8 chunks -> split across workgroups
    (3, 32, 96, 100, 3),      # 32x32 tiles, ragged last chunk (100 = 64 + 36)
    (2, 128, 64, 72, 3),      # 2 tiles of 64x64
    (1, 1024, 1024, 64, 3),   # 256 tiles -> no split, direct write (+ accumulate)
    (5, 64, 128, 64, 1),      # kernel 1, one chunk per batch element
]


def _wgrad_family_env(monkeypatch, wino):
    """wino: "0" direct form, "1" Winograd F(2,3) (WN), "4" Winograd F(4,3) (W4) -- each switched in for every channel count."""
    monkeypatch.setenv("ADP_CONV_WINO", "0" if wino == "0" else "1")
    monkeypatch.setenv("ADP_WINO_WGRAD_MIN_R", "32")
    monkeypatch.setenv("ADP_WGRAD_WINO4", "1" if wino == "4" else "0")
    monkeypatch.setenv("ADP_WINO4_WGRAD_MIN_R", "32")


@pytest.mark.parametrize("wino", ["0", "1", "4"])
@pytest.mark.parametrize("B,R,M,L,KT", WGMM_CASES)
def test_wgrad_mm_family(dev, B, R, M, L, KT, wino, monkeypatch):
    """wino = 1: the kernel-3 cases on the Winograd F(2,3) weight-gradient variant (WN; four rank-1 updates per
    output pair instead of six); wino = 4: the F(4,3) variant (W4; six rank-1 updates per output quad instead of twelve)."""
    if dev.type != "cuda" and R * M > 65536:
        pytest.skip("emulating 256 16-wave workgroups takes minutes; covered on the GPU")
    if wino != "0" and KT != 3:
        pytest.skip("the Winograd forms are kernel-3 forms")
    _wgrad_family_env(monkeypatch, wino)
    G = 8
    pad = (KT - 1) // 2
    x = rnd(B, R, L, seed=1) * 1.3 + 0.2
    gamma, beta = rnd(R, seed=4) * 0.5 + 1, rnd(R, seed=5) * 0.1
    w = rnd(M, R, KT, seed=2, scale=0.2).requires_grad_()
    b = rnd(M, seed=3).requires_grad_()
    dy = rnd(B, M, L, seed=9)
    xd, dyd = x.to(dev), dy.to(dev)
    # plain input, with bias gradient
    y = F.conv1d(x, w, b, padding=pad)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    dw, db = ops.conv1d_wgrad(xd, dyd, KT, pad=pad)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL
    # GroupNorm+SiLU recomputed in the loader, accumulate into existing gradients
    y = F.conv1d(ref_gn_silu(x, G, gamma, beta), w, b, padding=pad)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    stats = ops.gn_stats(xd, G)
    dw0, db0 = rnd(M, R, KT, seed=11), rnd(M, seed=12)
    dw, db = ops.conv1d_wgrad(xd, dyd, KT, pad=pad, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev),
                              pro_beta=beta.to(dev), groups=G, dw=dw0.clone().to(dev), dbias=db0.clone().to(dev),
                              accumulate=True)
    assert rel_err(dw, dw_ref + dw0) < TOL
    assert rel_err(db, db_ref + db0) < TOL


@pytest.mark.parametrize("wino", ["0", "1", "4"])
@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", MM_RESAMPLE_CASES + [(2, 64, 64, 128, 2, 2, 0, 1),
                                                                         (2, 64, 128, 20, 3, 1, 1, 4)])
def test_wgrad_mm_resample(dev, B, R, M, L, KT, stride, pad, up, wino, monkeypatch):
    """Weight gradients of DownsampleItem (kernel = stride) and UpsampleItem (nearest + k3) on wgrad_mm (direct form, Winograd
    F(2,3), Winograd F(4,3): the LDS tile holds the virtual upsampled positions, so the quads are those of the upsampled row)."""
    if wino != "0" and KT != 3:
        pytest.skip("the Winograd forms are kernel-3 forms")
    _wgrad_family_env(monkeypatch, wino)
    x = rnd(B, R, L, seed=1)
    w = rnd(M, R, KT, seed=2, scale=0.2).requires_grad_()
    b = rnd(M, seed=3).requires_grad_()
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    y = F.conv1d(xr, w, b, stride=stride, padding=pad)
    if y.shape[-1] % 4 != 0:
        pytest.skip("wgrad_mm needs N % 4 == 0 (falls back to the generic kernel, covered above)")
    dy = rnd(*y.shape, seed=9)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    dw, db = ops.conv1d_wgrad(x.to(dev), dy.to(dev), KT, stride=stride, pad=pad, up=up)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL


@pytest.mark.parametrize("B,C,L", [(4, 128, 1024), (2, 64, 4096), (4, 32, 65536)])
def test_wgrad_winograd_f43_accuracy(dev, B, C, L, monkeypatch):
    """The F(4,3) weight gradient sums B * L positions through transform constants up to 8 (A e) and 5 (B^T d): bound its fp32
    error against fp64 next to the F(2,3) and direct forms on long rows (measured on the GPU: see DESIGN.md section 4)."""
    if dev.type != "cuda" and B * C * L > 300000:
        pytest.skip("emulator: the long-row case runs on the GPU only")
    x, dy = rnd(B, C, L, seed=1), rnd(B, C, L, seed=2)
    w = torch.zeros(C, C, 3, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(F.conv1d(x.double(), w, None, padding=1), w, dy.double())
    errs = {}
    for wino in ("0", "1", "4"):
        _wgrad_family_env(monkeypatch, wino)
        dw, _ = ops.conv1d_wgrad(x.to(dev), dy.to(dev), 3, pad=1)
        errs[wino] = rel_err(dw, ref)
    assert errs["4"] < 1e-5 and errs["1"] < 1e-5 and errs["0"] < 1e-5, errs


@pytest.mark.parametrize("R1,R2,M,L", [(8, 0, 8, 2304), (5, 3, 8, 1100), (2, 0, 6, 516), (8, 0, 8, 36)])
def test_wgrad_direct8_prologue_concat(dev, R1, R2, M, L):
    """wgrad_direct8 (kernel 3, <= 8 x 8 channels: lane = (input row, quad), halo through DPP row shifts of the ACTIVATED values)
    with the GroupNorm+SiLU prologue, x2 concat, several spans per wave, a partly filled last span, accumulate."""
    B, R, G = 3, R1 + R2, (R1 + R2) if (R1 + R2) < 8 else 8
    xa = rnd(B, R, L, seed=1) * 1.5 + 0.3
    w = rnd(M, R, 3, seed=3, scale=0.3).requires_grad_()
    b = rnd(M, seed=4).requires_grad_()
    gamma, beta = rnd(R, seed=5) * 0.3 + 1.0, rnd(R, seed=6) * 0.2
    y = F.conv1d(ref_gn_silu(xa, G, gamma, beta), w, b, padding=1)
    dy = rnd(*y.shape, seed=9)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    xd = xa.to(dev)
    stats = ops.gn_stats(xd, G)
    x1 = xd[:, :R1].contiguous()
    x2 = xd[:, R1:].contiguous() if R2 else None
    kw = dict(pad=1, x2=x2, prologue=1, pro_stats=stats, pro_gamma=gamma.to(dev), pro_beta=beta.to(dev), groups=G)
    dw, db = ops.conv1d_wgrad(x1, dy.to(dev), 3, **kw)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL
    base = rnd(M, R, 3, seed=11).to(dev)
    dw2, _ = ops.conv1d_wgrad(x1, dy.to(dev), 3, dw=base.clone(), want_bias=False, accumulate=True, **kw)
    assert rel_err(dw2, base.cpu() + dw_ref) < TOL


@pytest.mark.parametrize("B,R,M,L,KT,stride,pad,up", DIRECT_CASES[:7] + [(3, 8, 8, 1300, 3, 1, 1, 1), (2, 5, 7, 772, 3, 1, 1, 1),
                                                                       (1, 2, 8, 260, 3, 1, 1, 1), (1, 8, 8, 4, 3, 1, 1, 1),
                                                                       (2, 32, 8, 1100, 3, 1, 1, 4), (1, 12, 5, 66, 3, 1, 1, 4),
                                                                       (1, 8, 8, 2, 3, 1, 1, 4), (3, 32, 8, 4096, 3, 1, 1, 4)])
def test_wgrad_direct_family(dev, B, R, M, L, KT, stride, pad, up):
    """Weight gradients of the narrow layers on the VALU streaming kernel (wgrad_direct.hip), incl. x2 concat."""
    x = rnd(B, R, L, seed=1)
    w = rnd(M, R, KT, seed=2, scale=0.2).requires_grad_()
    b = rnd(M, seed=3).requires_grad_()
    xr = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    y = F.conv1d(xr, w, b, stride=stride, padding=pad)
    dy = rnd(*y.shape, seed=9)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    dw, db = ops.conv1d_wgrad(x.to(dev), dy.to(dev), KT, stride=stride, pad=pad, up=up)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL
    if R >= 2 and up == 1:
        r1 = R // 2
        dw2, _ = ops.conv1d_wgrad(x[:, :r1].contiguous().to(dev), dy.to(dev), KT, stride=stride, pad=pad,
                                  x2=x[:, r1:].contiguous().to(dev))
        assert rel_err(dw2, dw_ref) < TOL


@pytest.mark.parametrize("whole,split", [(False, "1"), (True, "0"), (True, "1")])
@pytest.mark.parametrize("C,L,n,bias,acc", [(32, 4096, 3, True, False), (64, 1024, 10, True, True), (32, 2048, 2, False, False)])
def test_wgrad_parked_second_stage(dev, C, L, n, bias, acc, whole, split, monkeypatch):
    """Parked weight gradients (ops.WgradPark).  whole = False: adp_wgrad_desc.accumulate bit 1 + adp_wgrad_reduce_batch -- n
    same-shape split weight gradients leave their partial slices in their own scratch and are summed by one launch per 8.
    whole = True: the calls themselves wait and run as one launch per 8 (adp_conv1d_wgrad_batch: blockIdx.x = item * splits +
    split) + one batched second stage.  Same result as the plain calls (bitwise: same arithmetic, same order) -- except that a
    batched launch, whose items fill the chip together, splits each item's positions n times less (ADP_WGRAD_BATCH_SPLIT=0 keeps
    the lone split): then the partial sums group differently and the results agree to rounding."""
    B = 2
    park = ops.WgradPark()
    if not whole:
        park.BATCH_BYTES = 0
    outs, refs = [], []
    for i in range(n):
        x, dy = rnd(B, C, L, seed=10 + i).to(dev), rnd(B, C, L, seed=50 + i).to(dev)
        base, bb = rnd(C, C, 3, seed=90 + i).to(dev), rnd(C, seed=130 + i).to(dev)
        kw = dict(pad=1, want_bias=bias, accumulate=acc)
        dw_ref, db_ref = ops.conv1d_wgrad(x, dy, 3, dw=base.clone() if acc else None, dbias=bb.clone() if acc and bias else None, **kw)
        dw, db = ops.conv1d_wgrad(x, dy, 3, dw=base.clone() if acc else None, dbias=bb.clone() if acc and bias else None,
                                  park=park, **kw)
        outs.append((dw, db))
        refs.append((dw_ref, db_ref))
    assert len(park.calls if whole else park.items) == n, "these shapes must take the split matrix-core path"
    monkeypatch.setenv("ADP_WGRAD_BATCH_SPLIT", split)
    park.flush()
    assert not park.items and not park.calls
    for (dw, db), (dw_ref, db_ref) in zip(outs, refs):
        if split == "0" or not whole:  # the items keep the lone launch's position split: same arithmetic, same order
            assert torch.equal(dw.cpu(), dw_ref.cpu())
            if bias:
                assert torch.equal(db.cpu(), db_ref.cpu())
        else:  # a batched launch splits every item's positions n times less (round 5): another summation order
            assert rel_err(dw, dw_ref) < 1e-5
            if bias:
                assert rel_err(db, db_ref) < 1e-5


# ------------------------------------------------------------------ GroupNorm+SiLU backward
@pytest.mark.parametrize("B,C,L,G", [(2, 8, 3000, 8), (2, 32, 130, 8), (1, 64, 40, 8), (2, 1024, 128, 8)])
def test_gn_silu_bwd(dev, B, C, L, G):
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.4).requires_grad_()
    gamma = (rnd(C, seed=2) * 0.5 + 1).requires_grad_()
    beta = (rnd(C, seed=3) * 0.2).requires_grad_()
    y = ref_gn_silu(x, G, gamma, beta)
    dact, dres = rnd(B, C, L, seed=4), rnd(B, C, L, seed=5)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (x, gamma, beta), dact)
    xd = x.detach().to(dev)
    stats = ops.gn_stats(xd, G)
    dx, dg, db = ops.gn_silu_bwd(xd, dact.to(dev), stats, gamma.detach().to(dev), beta.detach().to(dev), G,
                                 dres=dres.to(dev))
    assert rel_err(dx, dx_ref + dres) < TOL
    assert rel_err(dg, dg_ref) < TOL
    assert rel_err(db, db_ref) < TOL


# ------------------------------------------------------------------ Modulation / LayerNorm over channels
@pytest.mark.parametrize("B,C,L", [(2, 8, 300), (2, 32, 70), (1, 130, 64), (2, 100, 50), (2, 300, 40), (1, 1024, 24),
                                   (4, 512, 784), (4, 512, 400), (2, 1024, 128)])  # 16 / 8 / 4-position tiles
def test_modulation_fwd_bwd(dev, B, C, L):
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.4).requires_grad_()
    NT = 2 * C + 7
    bank = (rnd(B, NT, seed=2) * 0.5).requires_grad_()
    off = 3
    scale, shift = bank[:, off:off + C], bank[:, off + C:off + 2 * C]
    xn = F.layer_norm(x.transpose(1, 2), (C,), eps=1e-5)
    y = (xn * (1 + scale[:, None, :]) + shift[:, None, :]).transpose(1, 2)
    dy = rnd(B, C, L, seed=3)
    dx_ref, dbank_ref = torch.autograd.grad(y, (x, bank), dy)
    xd, bank_d = x.detach().to(dev), bank.detach().to(dev)
    yd, stats = ops.modulation_fwd(xd, bank_d.view(-1)[off:], NT)
    assert rel_err(yd, y) < TOL
    dbank = torch.zeros(B, NT, device=dev)
    dx = ops.modulation_bwd(xd, dy.to(dev), bank_d.view(-1)[off:], NT, stats, dbank.view(-1)[off:], NT)
    assert rel_err(dx, dx_ref) < TOL
    assert rel_err(dbank, dbank_ref) < TOL


@pytest.mark.parametrize("B,C,L,dual", [(2, 8, 300, False), (2, 32, 72, True), (1, 130, 64, False), (2, 300, 40, True),
                                        (1, 1024, 24, False), (4, 512, 784, True), (2, 1024, 128, False),
                                        (2, 100, 50, True)])   # (L % 4 != 0: the two launches it stands for)
def test_modulation_ln_fwd(dev, B, C, L, dual):
    """adp_modulation_ln_fwd (norm.hip, chained mode of chan_lnv_fwd): ModulationItem + the LayerNorm(s) of the attention item
    behind it in one launch = adp_modulation_fwd followed by adp_ln_affine_fwd on its output (every output, both statistics),
    and both against torch."""
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.4)
    NT = 2 * C + 7
    bank = rnd(B, NT, seed=2) * 0.5
    off = 3
    scale, shift = bank[:, off:off + C], bank[:, off + C:off + 2 * C]
    g1, b1, g2, b2 = (rnd(C, seed=10 + i) * 0.5 + (1.0 if i % 2 == 0 else 0.0) for i in range(4))
    y_ref = (F.layer_norm(x.transpose(1, 2), (C,), eps=ops.MODULATION_LN_EPS) * (1 + scale[:, None, :])
             + shift[:, None, :])
    xn_ref = F.layer_norm(y_ref, (C,), g1, b1, eps=ops.ATTENTION_LN_EPS).transpose(1, 2)
    cn_ref = F.layer_norm(y_ref, (C,), g2, b2, eps=ops.ATTENTION_LN_EPS).transpose(1, 2)
    xd, bank_d = x.to(dev), bank.to(dev)
    gd = [t.to(dev) for t in (g1, b1, g2, b2)]
    y, st, xn, cn, lst = ops.modulation_ln_fwd(xd, bank_d.view(-1)[off:], NT, gd[0], gd[1], gd[2] if dual else None,
                                               gd[3] if dual else None)
    y0, st0 = ops.modulation_fwd(xd, bank_d.view(-1)[off:], NT)
    xn0, cn0, lst0 = ops.ln_affine_fwd(y0, gd[0], gd[1], gd[2] if dual else None, gd[3] if dual else None)
    assert torch.equal(y, y0) and torch.equal(st, st0)
    assert rel_err(y, y_ref.transpose(1, 2)) < TOL and rel_err(xn, xn_ref) < TOL
    assert rel_err(xn, xn0) < 1e-5 and rel_err(lst, lst0) < 1e-5
    assert (cn is None) == (not dual)
    if dual:
        assert rel_err(cn, cn_ref) < TOL and rel_err(cn, cn0) < 1e-5


@pytest.mark.parametrize("B,C,L,res", [(2, 8, 300, True), (2, 32, 72, False), (1, 130, 64, True), (2, 300, 40, True),
                                       (1, 1024, 24, True), (4, 512, 784, True), (2, 1024, 128, False),
                                       (2, 100, 50, True)])   # (L % 4 != 0: the two launches it stands for)
def test_modulation_ln_bwd(dev, B, C, L, res):
    """adp_modulation_ln_bwd_partial (norm.hip, chan_lnv_bwd_chain_kernel): d(xn) -> d(y) -> d(x) through the attention item's
    LayerNorm and the ModulationItem in front of it in one pass, against torch autograd and against the two separate calls
    (adp_ln_bwd, adp_modulation_bwd_partial): dx, the LayerNorm's [dgamma | dbeta], the Modulation's scale / shift gradient."""
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.4).requires_grad_()
    NT = 2 * C + 7
    bank = (rnd(B, NT, seed=2) * 0.5).requires_grad_()
    off = 3
    scale, shift = bank[:, off:off + C], bank[:, off + C:off + 2 * C]
    g1 = (rnd(C, seed=10) * 0.5 + 1.0).requires_grad_()
    b1 = (rnd(C, seed=11) * 0.5).requires_grad_()
    y = F.layer_norm(x.transpose(1, 2), (C,), eps=ops.MODULATION_LN_EPS) * (1 + scale[:, None, :]) + shift[:, None, :]
    xn = F.layer_norm(y, (C,), g1, b1, eps=ops.ATTENTION_LN_EPS).transpose(1, 2)
    dxn, dres = rnd(B, C, L, seed=3), rnd(B, C, L, seed=4)
    out = (xn * dxn).sum() + ((y.transpose(1, 2) * dres).sum() if res else 0.0)
    dx_ref, dbank_ref, dg_ref, db_ref = torch.autograd.grad(out, (x, bank, g1, b1))
    xd, bank_d, g1d, b1d = x.detach().to(dev), bank.detach().to(dev), g1.detach().to(dev), b1.detach().to(dev)
    ssv = bank_d.view(-1)[off:]
    yd, st, xnd, _, lst = ops.modulation_ln_fwd(xd, ssv, NT, g1d, b1d)
    dxn_d, dres_d = dxn.to(dev), (dres.to(dev) if res else None)
    # one pass
    sums, dbank = ops.ModulationSums(), torch.zeros(B, NT, device=dev)
    dgb = torch.full((2 * C,), float("nan"), device=dev)
    dx = sums.partial_ln(off, xd, ssv, NT, st, dbank.view(-1)[off:], NT, yd, dxn_d, g1d, lst, dres_d, dgb)
    sums.flush()
    assert rel_err(dx, dx_ref) < TOL and rel_err(dbank, dbank_ref) < TOL
    assert rel_err(dgb[:C], dg_ref) < TOL and rel_err(dgb[C:], db_ref) < TOL
    # the two calls it replaces
    dy0, dgb0 = ops.ln_bwd(yd, dxn_d, lst, g1d, dres=dres_d)
    sums0, dbank0 = ops.ModulationSums(), torch.zeros(B, NT, device=dev)
    dx0 = sums0.partial(off, xd, dy0, ssv, NT, st, dbank0.view(-1)[off:], NT)
    sums0.flush()
    assert rel_err(dx, dx0) < 2e-5 and rel_err(dbank, dbank0) < 2e-5 and rel_err(dgb, dgb0) < 2e-5


@pytest.mark.parametrize("B,C,L,n", [(2, 64, 200, 3), (1, 32, 9000, 2), (2, 1024, 12, 9)])
def test_modulation_bwd_parked_sums(dev, B, C, L, n):
    """adp_modulation_bwd_partial + adp_modulation_bwd_reduce: n Modulation backwards of one shape, second stages summed by
    one launch per eight (both reduce forms: tile-major workgroups, and a wave per output for narrow + long), have the bits
    of n separate adp_modulation_bwd calls; a flush by bank-row range takes only the items inside it."""
    NT = n * 2 * C + 5
    bank = (rnd(B, NT, seed=2) * 0.5).to(dev)
    xs = [(rnd(B, C, L, seed=10 + i) * 1.5 + 0.4).to(dev) for i in range(n)]
    dys = [rnd(B, C, L, seed=30 + i).to(dev) for i in range(n)]
    offs = [3 + i * 2 * C for i in range(n)]
    stats = [ops.modulation_fwd(xs[i], bank.view(-1)[offs[i]:], NT)[1] for i in range(n)]
    d0 = torch.zeros(B, NT, device=dev)
    dx0 = [ops.modulation_bwd(xs[i], dys[i], bank.view(-1)[offs[i]:], NT, stats[i], d0.view(-1)[offs[i]:], NT) for i in range(n)]
    d1 = torch.zeros(B, NT, device=dev)
    sums = ops.ModulationSums()
    dx1 = [sums.partial(offs[i], xs[i], dys[i], bank.view(-1)[offs[i]:], NT, stats[i], d1.view(-1)[offs[i]:], NT)
           for i in range(n)]
    sums.flush(offs[1], NT)          # everything but the first item
    assert len(sums.items) == 1 and d1[:, offs[0]:offs[0] + 2 * C].eq(0).all()
    sums.flush()
    assert not sums.items
    assert torch.equal(d0, d1) and all(torch.equal(a, b) for a, b in zip(dx0, dx1))


def test_ln_bwd(dev):
    B, C, L = 2, 40, 70
    x = (rnd(B, C, L, seed=1) * 2 + 1).requires_grad_()
    gamma = (rnd(C, seed=3) * 0.3 + 1).requires_grad_()
    beta = (rnd(C, seed=4) * 0.1).requires_grad_()
    xn = F.layer_norm(x.transpose(1, 2), (C,), gamma, beta, eps=1e-5).transpose(1, 2)
    dxn, dres = rnd(B, C, L, seed=5), rnd(B, C, L, seed=6)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(xn, (x, gamma, beta), dxn)
    xd = x.detach().to(dev)
    stats = ops.ln_stats(xd)
    dx, dgb = ops.ln_bwd(xd, dxn.to(dev), stats, gamma.detach().to(dev), dres=dres.to(dev))
    assert rel_err(dx, dx_ref + dres) < TOL
    assert rel_err(dgb[:C], dg_ref) < TOL
    assert rel_err(dgb[C:], db_ref) < TOL


def test_skipmod_bwd(dev):
    B, C, L, NT = 2, 6, 2500, 11
    g, x, bank = rnd(B, C, L, seed=1), rnd(B, C, L, seed=2), rnd(B, NT, seed=3)
    scale = bank[:, 2:2 + C]
    dx_ref = scale[:, :, None] * g
    ds_ref = (g * x).sum(-1)
    bank_d = bank.to(dev)
    dbank = torch.zeros(B, NT, device=dev)
    dx = ops.skipmod_bwd(g.to(dev), x.to(dev), bank_d.view(-1)[2:], NT, dbank.view(-1)[2:], NT)
    assert rel_err(dx, dx_ref) < TOL
    assert rel_err(dbank[:, 2:2 + C], ds_ref) < TOL
    assert dbank[:, :2].abs().max().item() == 0 and dbank[:, 2 + C:].abs().max().item() == 0


# ------------------------------------------------------------------ conditioning path (small-batch Linear)
@pytest.mark.parametrize("B,K,N,act,post", [(1, 257, 64, 0, 2), (4, 1024, 37, 1, 0), (8, 1500, 20, 2, 0), (37, 96, 50, 1, 0),
                                            (3, 1024, 4100, 1, 0),   # >= 4096 rows: eight rows per wave, ragged last wave
                                            (2, 1300, 4096, 0, 2),   # the same with two K chunks (+ GELU at the end)
                                            (16, 64, 9, 1, 2),
                                            (2, 512, 4101, 0, 2), (4, 256, 33, 2, 0)])  # streaming form: 2 / 1 pieces of 256
def test_linear_fwd_bwd(dev, B, K, N, act, post):
    x = rnd(B, K, seed=1).requires_grad_()
    w = rnd(N, K, seed=2, scale=K ** -0.5).requires_grad_()
    b = rnd(N, seed=3).requires_grad_()
    a = {0: lambda t: t, 1: F.silu, 2: F.gelu}[act]
    xa = a(x)
    y = F.linear(xa, w, b)
    yp = F.gelu(y) if post == 2 else y
    out = ops.linear_fwd(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev), act, post)
    assert rel_err(out, yp) < TOL
    dy = rnd(B, N, seed=4)
    dxa_ref, dw_ref, db_ref = torch.autograd.grad(y, (xa, w, b), dy)
    dxa = ops.linear_bwd_data(dy.to(dev), w.detach().to(dev))
    assert rel_err(dxa, dxa_ref) < TOL
    dw, db = ops.linear_bwd_weight(dy.to(dev), x.detach().to(dev), act)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL
    # activation backward helper
    if act:
        (dx_ref,) = torch.autograd.grad(xa, x, dxa_ref)
        dx = ops.act_bwd(x.detach().to(dev), dxa, act)
        assert rel_err(dx, dx_ref) < TOL


def test_time_fourier(dev):
    B, H = 3, 128
    t = torch.tensor([0.0, 0.31, 1.0])
    w = rnd(H, seed=1).requires_grad_()
    f = t[:, None] * w[None, :] * 2 * math.pi
    four = torch.cat([t[:, None], f.sin(), f.cos()], -1)
    out = ops.time_fourier_fwd(t.to(dev), w.detach().to(dev))
    assert rel_err(out, four) < 1e-5
    dfour = rnd(B, 2 * H + 1, seed=2)
    (dw_ref,) = torch.autograd.grad(four, w, dfour)
    dw = ops.time_fourier_bwd(t.to(dev), w.detach().to(dev), dfour.to(dev))
    assert rel_err(dw, dw_ref) < TOL


# ------------------------------------------------------------------ v-objective math
def test_v_noise_mse_step(dev):
    B, C, L = 3, 2, 1000
    x, n = rnd(B, C, L, seed=1), rnd(B, C, L, seed=2)
    sig = torch.tensor([0.0, 0.37, 1.0])
    ang = sig.view(-1, 1, 1) * math.pi / 2
    a, b = torch.cos(ang), torch.sin(ang)
    xn, vt = ops.v_noise(x.to(dev), n.to(dev), sig.to(dev))
    assert rel_err(xn, a * x + b * n) < 1e-6
    assert rel_err(vt, a * n - b * x) < 1e-6
    vp = rnd(B, C, L, seed=3).requires_grad_()
    loss_ref = F.mse_loss(vp, vt.cpu())
    loss = ops.mse_fwd(vp.detach().to(dev), vt)
    assert abs(loss.item() - loss_ref.item()) < 1e-6 * abs(loss_ref.item())
    (dv_ref,) = torch.autograd.grad(loss_ref, vp)
    gl = torch.tensor(0.7)
    dv = ops.mse_bwd(vp.detach().to(dev), vt, gl.to(dev))
    assert rel_err(dv, dv_ref * 0.7) < 1e-6
    ab4 = torch.tensor([0.3, 0.9, 0.5, 0.8])
    xo = ops.v_step(x.to(dev), n.to(dev), ab4.to(dev))
    ref = 0.5 * (0.3 * x - 0.9 * n) + 0.8 * (0.9 * x + 0.3 * n)
    assert rel_err(xo, ref) < 1e-6


# ------------------------------------------------------------------ attention core
@pytest.mark.parametrize("B,H,D,n,m", [(2, 2, 64, 160, 160), (1, 3, 32, 70, 45), (1, 8, 64, 128, 64),
                                       (1, 2, 64, 512, 64),    # few key tiles: the dk/dv pass splits the queries
                                       (2, 1, 16, 203, 37),    # ragged both ways, query split with a short last slice
                                       (1, 2, 64, 96, 256),
                                       (2, 2, 64, 70, 45), (1, 2, 64, 64, 20)])  # few keys: ragged second key block / one block
# backward: both passes in one launch (own delta) / two launches, delta through ws; "fk": the four-wave kernels where 64 channels
# meet at most 64 keys (the other two modes switch them off: the one-wave forms still serve those shapes at large batch * length)
@pytest.mark.parametrize("mode", ["fk", "merged", "split"])
def test_attention_fwd_bwd(dev, B, H, D, n, m, mode, monkeypatch):
    if mode == "fk" and not (D == 64 and m <= 64):
        pytest.skip("not a few-keys shape")
    monkeypatch.setenv("ADP_ATTN_MERGE", "0" if mode == "split" else "1")
    monkeypatch.setenv("ADP_ATTN_FEWKEYS", "1" if mode == "fk" else "0")
    mid = H * D
    q = rnd(B, mid, n, seed=1).requires_grad_()
    kv = rnd(B, 2 * mid, m, seed=2).requires_grad_()
    qh = q.view(B, H, D, n).transpose(2, 3)
    kh = kv[:, :mid].reshape(B, H, D, m).transpose(2, 3)
    vh = kv[:, mid:].reshape(B, H, D, m).transpose(2, 3)
    att = torch.softmax(torch.einsum("bhnd,bhmd->bhnm", qh, kh) * D ** -0.5, dim=-1)
    o_ref = torch.einsum("bhnm,bhmd->bhnd", att, vh).transpose(2, 3).reshape(B, mid, n)
    o, lse = ops.attn_fwd(q.detach().to(dev), kv.detach().to(dev), H, D)
    assert rel_err(o, o_ref) < TOL
    lse_ref = torch.logsumexp(torch.einsum("bhnd,bhmd->bhnm", qh, kh) * D ** -0.5, dim=-1)
    assert rel_err(lse.view(B, H, n), lse_ref) < 1e-5
    do = rnd(B, mid, n, seed=3)
    dq_ref, dkv_ref = torch.autograd.grad(o_ref, (q, kv), do)
    dq, dkv = ops.attn_bwd(q.detach().to(dev), kv.detach().to(dev), o, do.to(dev), lse, H, D)
    assert rel_err(dq, dq_ref) < TOL
    assert rel_err(dkv, dkv_ref) < TOL


@pytest.mark.parametrize("I,M2,E", [(3, 16, 12), (2, 40, 70), (5, 64, 33)])
def test_ctx_fold_fwd_bwd(dev, I, M2, E):
    """adp_ctx_fold_fwd / _bwd (ctx_bank.hip): w_all = W_i diag(gamma_i), bias_all = W_i beta_i per item of the bank and the
    transpose of that map, written at given offsets of a flat gradient buffer (ragged E: not a multiple of the wave / of 32)."""
    W = [rnd(M2, E, seed=10 + i).to(dev) for i in range(I)]
    ga = [rnd(E, seed=20 + i).to(dev) for i in range(I)]
    be = [rnd(E, seed=30 + i).to(dev) for i in range(I)]
    tab = torch.tensor([[t.data_ptr() for t in W], [t.data_ptr() for t in ga], [t.data_ptr() for t in be]],
                       dtype=torch.int64).to(dev)
    w_all, b_all = ops.ctx_fold_fwd(tab, I, M2, E)
    for i in range(I):
        assert rel_err(w_all[i * M2:(i + 1) * M2], W[i] * ga[i][None, :]) < TOL
        assert rel_err(b_all[i * M2:(i + 1) * M2], W[i] @ be[i]) < TOL
    dw_all, db_all = rnd(I * M2, E, seed=40).to(dev), rnd(I * M2, seed=41).to(dev)
    per = M2 * E + 2 * E + 5  # (a gap between the items' regions)
    flat = torch.full((I * per + 3,), 7.0, device=dev)
    dw_off = torch.tensor([3 + i * per + 2 * E + 5 for i in range(I)], dtype=torch.int64).to(dev)
    dgb_off = torch.tensor([3 + i * per for i in range(I)], dtype=torch.int64).to(dev)
    ops.ctx_fold_bwd(tab, dw_all, db_all, I, M2, E, flat, dw_off, dgb_off)
    for i in range(I):
        dW, db = dw_all[i * M2:(i + 1) * M2], db_all[i * M2:(i + 1) * M2]
        a = 3 + i * per
        assert rel_err(flat[a + 2 * E + 5:a + per].view(M2, E), dW * ga[i][None, :] + db[:, None] * be[i][None, :]) < TOL
        assert rel_err(flat[a:a + E], (dW * W[i]).sum(0)) < TOL
        assert rel_err(flat[a + E:a + 2 * E], db @ W[i]) < TOL
        assert bool((flat[a + 2 * E:a + 2 * E + 5] == 7.0).all())
    assert bool((flat[:3] == 7.0).all())


@pytest.mark.parametrize("B,R,M,L,KT,stride,up", [
    (2, 32, 64, 200, 3, 1, 1),     # conv_mm, ragged last tile
    (1, 64, 32, 128, 1, 1, 1),     # conv_mm 1x1
    (2, 32, 32, 512, 3, 1, 1),     # conv_tile32 (wave tiles)
    (3, 32, 32, 1280, 3, 1, 1),    # conv_tile32, batch that does not divide the CU count
    (1, 32, 64, 256, 4, 4, 1),     # DownsampleItem (kernel = stride = 4)
    (1, 64, 32, 96, 3, 1, 2),      # UpsampleItem loader with the SkipModulate epilogue
    (1, 512, 64, 64, 3, 1, 1),     # small grid: cross-workgroup K split, statistics from the reduce kernel
    (2, 1024, 32, 60, 1, 1, 1),    # K split, kernel 1, ragged rows
])
def test_groupnorm_statistics_from_conv_epilogue(dev, B, R, M, L, KT, stride, up):
    """adp_conv_desc.gn_part: the conv epilogue's per-row (mean, M2, count) slices + adp_gn_finalize reproduce
    adp_gn_stats of the output tensor (so the consumer GroupNorm needs no statistics pass), also with the residual /
    e_scale epilogue terms folded in; and adp_gn_act matches the fused statistics+activation path."""
    G = 8
    pad = 1 if KT == 3 else 0
    x, w, b = rnd(B, R, L, seed=1), rnd(M, R, KT, seed=2, scale=0.3), rnd(M, seed=3)
    N = ops.conv_out_len(L, KT, stride, 1, pad, up)
    res, sc = rnd(B, M, N, seed=4) + 0.5, rnd(B * M, seed=5)
    gn = ops.GnPart()
    out = ops.conv1d(x.to(dev), w.to(dev), b.to(dev), stride=stride, pad=pad, up=up, e_scale=sc.to(dev) if up > 1 else None,
                     res=res.to(dev), gn=gn)
    assert gn.part is not None and gn.covers(out), "this shape is meant to produce epilogue statistics"
    assert gn.part.shape[:2] == (B, M // 4) and torch.isfinite(gn.part).all()
    assert gn.part[..., 2].sum(dim=2).eq(4 * N).all()    # the slices of every row quad cover each element exactly once
    st = ops.gn_finalize(gn.part, G)
    ref = ops.gn_stats(out, G)
    assert rel_err(st[..., 0], ref[..., 0]) < 1e-4 or (st[..., 0] - ref[..., 0]).abs().max() < 1e-5
    assert rel_err(st[..., 1], ref[..., 1]) < 1e-5
    gamma, beta = rnd(M, seed=6) * 0.5 + 1, rnd(M, seed=7) * 0.1
    act = ops.gn_act(out, st, G, gamma.to(dev), beta.to(dev))
    assert rel_err(act, ref_gn_silu(out.cpu(), G, gamma, beta)) < TOL
    st2, act2 = ops.gn_finalize_act(out, gn.part, G, gamma.to(dev), beta.to(dev))
    assert torch.equal(act2.cpu(), act.cpu()) or rel_err(act2, act) < 1e-6
    assert rel_err(st2, st) < 1e-6


@pytest.mark.parametrize("B,C,L,G", [(2, 16, 300, 8), (1, 64, 1030, 8), (2, 512, 24, 8)])
def test_gn_stats_act(dev, B, C, L, G):
    """adp_gn_stats_act: statistics + materialised SiLU(GroupNorm(x)) (the wide-layer path)."""
    x = rnd(B, C, L, seed=1) * 1.7 + 0.3
    gamma, beta = rnd(C, seed=2) * 0.5 + 1, rnd(C, seed=3) * 0.1
    stats, act = ops.gn_stats_act(x.to(dev), G, gamma.to(dev), beta.to(dev))
    assert rel_err(act, ref_gn_silu(x, G, gamma, beta)) < TOL
    ref_stats = ops.gn_stats(x.to(dev), G)
    assert rel_err(stats, ref_stats) < 1e-6


def test_calibration_probes(dev):
    """csrc/probe.hip (bench.py's `calibration` object): the streaming copy copies, the MFMA probe reports the flops of its
    launch and leaves the value its four accumulator chains add up to, the empty launch succeeds; argument errors are codes."""
    x = rnd(4096, seed=1).to(dev)
    y = torch.zeros_like(x)
    _C.call("adp_probe_copy", _C.ptr(x), _C.ptr(y), x.numel(), _C.stream())
    assert torch.equal(y.cpu(), x.cpu())
    out = torch.zeros(512 * 256).to(dev)
    iters = 2
    flops = _C.call_value("adp_probe_mfma", iters, _C.ptr(out), out.numel(), _C.stream())
    assert flops == 512 * 4 * iters * 4 * 4096
    # A[i][k] = lane * 1e-3 with k = lane >> 5, B likewise 2e-3: every accumulator element (i, j) sums over k = 0, 1
    lane = torch.arange(64, dtype=torch.float64)
    a, b = (lane * 1e-3).view(2, 32), (lane * 2e-3).view(2, 32)   # [k][i], [k][j]
    tile = torch.einsum("ki,kj->ij", a, b) * iters                  # one accumulator tile after `iters` MFMAs
    assert abs(out.cpu().double().sum().item() / (512 * 4 * 4) - tile.sum().item()) < 1e-3 * tile.sum().item()
    _C.call("adp_probe_launch", 3, _C.stream())
    # the latency probe walks the cycle the host laid out: 7 lines of 16 ints, 0 -> 3 -> 5 -> 1 -> 6 -> 2 -> 4 -> 0
    order = [0, 3, 5, 1, 6, 2, 4]
    chain = torch.zeros(7 * 16, dtype=torch.int32)
    for a, b in zip(order, order[1:] + order[:1]):
        chain[a * 16] = b * 16
    o = torch.zeros(1, dtype=torch.int32).to(dev)
    _C.call("adp_probe_chase", _C.ptr(chain.to(dev), torch.int32), 10, _C.ptr(o, torch.int32), _C.stream())
    assert int(o.cpu()) == order[10 % 7] * 16
    assert _C.lib().adp_probe_copy(None, None, 4, 0) == -5 and _C.lib().adp_probe_launch(0, 0) == -1
    assert _C.lib().adp_probe_mfma(1, _C.ptr(out), 16, 0) == -1
