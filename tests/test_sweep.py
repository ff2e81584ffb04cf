"""Seeded random shape sweeps across the dispatcher's family boundaries (shape sweeps per kernel against
torch.nn.functional).  The hand-picked cases of test_kernels.py pin each kernel family; these draw channel counts,
lengths, batch sizes and epilogue / prologue options at random so that eligibility edges (channels % 32, ragged tiles,
split thresholds, alignment of odd lengths) are crossed by shapes nobody chose.  Cases are generated at import time from
fixed seeds, so a failure names a reproducible parameter tuple."""
import random

import pytest
import torch
import torch.nn.functional as F

from audio_diffusion_pytorch_amd import ops
from conftest import rel_err

TOL = 1e-4
CH = [2, 3, 8, 12, 16, 24, 32, 40, 64, 72, 96, 128, 136, 160]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _conv_cases(n, seed):
    r = random.Random(seed)
    out = []
    while len(out) < n:
        kind = r.choice(["k3", "k3", "k3", "k1", "down2", "down4", "up2", "up4", "k3dil"])
        B, R, M = r.randint(1, 3), r.choice(CH), r.choice(CH)
        L = r.choice([16, 20, 33, 64, 100, 128, 130, 200, 257, 384])
        KT, stride, up, dil = 3, 1, 1, 1
        if kind == "k1":
            KT = 1
        elif kind in ("down2", "down4"):
            KT = stride = int(kind[-1])
            L = max(L // stride * stride, stride * 4)
        elif kind in ("up2", "up4"):
            up = int(kind[-1])
            L = min(L, 130)
        elif kind == "k3dil":
            dil = r.choice([2, 3])
        pad = dil * (KT - 1) // 2 if stride == 1 else 0
        tr = kind in ("k3", "k1") and r.random() < 0.35
        pro = kind in ("k3", "k1", "k3dil") and not tr and R % 4 == 0 and r.random() < 0.4
        if R * M * L * B > 6_000_000:  # keeps the emulated run in seconds
            continue
        opts = tuple(r.random() < 0.5 for _ in range(4))  # bias, e_scale, res, out_pre
        out.append((B, R, M, L, KT, stride, pad, dil, up, tr, pro, opts))
    return out


@pytest.mark.parametrize("case", _conv_cases(36, 1234), ids=lambda c: "-".join(str(int(v)) if not isinstance(v, tuple) else
                                                                               "".join(str(int(b)) for b in v) for v in c))
def test_conv1d_forward_sweep(dev, case):
    B, R, M, L, KT, stride, pad, dil, up, tr, pro, (has_b, has_s, has_r, has_p) = case
    x = rnd(B, R, L, seed=1) * 1.2 + 0.1
    w = rnd(R, M, KT, seed=2, scale=0.2) if tr else rnd(M, R, KT, seed=2, scale=0.2)
    xin, kw = x, {}
    if pro:
        G = 4 if R % 8 else 8
        gamma, beta = rnd(R, seed=4) * 0.5 + 1, rnd(R, seed=5) * 0.1
        xin = F.silu(F.group_norm(x, G, gamma, beta, eps=1e-5))
        xd = x.to(dev)
        kw = dict(prologue=1, pro_stats=ops.gn_stats(xd, G), pro_gamma=gamma.to(dev), pro_beta=beta.to(dev), groups=G)
    xr = F.interpolate(xin, scale_factor=up, mode="nearest") if up > 1 else xin
    if tr:
        ref = F.conv_transpose1d(xr, w, None, padding=pad, dilation=dil)
    else:
        ref = F.conv1d(xr, w, None, stride=stride, padding=pad, dilation=dil)
    N = ref.shape[-1]
    b = rnd(M, seed=3) if has_b else None
    sc = rnd(B * M, seed=6) if has_s else None
    res = rnd(B, M, N, seed=7) if has_r else None
    pre_ref = ref + (b[None, :, None] if has_b else 0)
    ref = pre_ref * (sc.view(B, M, 1) if has_s else 1) + (res if has_r else 0)
    pre = torch.empty(B, M, N).to(dev) if has_p else None
    mv = lambda t: None if t is None else t.to(dev)  # noqa: E731
    out = ops.conv1d(x.to(dev), w.to(dev), mv(b), stride=stride, dil=dil, pad=pad, up=up, transposed=tr, e_scale=mv(sc),
                     res=mv(res), out_pre=pre, **kw)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL
    if has_p:
        assert rel_err(pre, pre_ref) < TOL


def _wgrad_cases(n, seed):
    r = random.Random(seed)
    out = []
    while len(out) < n:
        kind = r.choice(["k3", "k3", "k1", "down2", "down4", "up2", "up4"])
        B, R, M = r.randint(1, 4), r.choice(CH), r.choice(CH)
        L = r.choice([16, 36, 64, 100, 128, 132, 260, 400])
        KT, stride, up = 3, 1, 1
        if kind == "k1":
            KT = 1
        elif kind in ("down2", "down4"):
            KT = stride = int(kind[-1])
            L = max(L // stride * stride, stride * 4)
        elif kind in ("up2", "up4"):
            up = int(kind[-1])
            L = min(L, 132)
        pad = (KT - 1) // 2 if stride == 1 else 0
        pro = kind in ("k3", "k1") and R % 4 == 0 and r.random() < 0.4
        if R * M * L * B > 5_000_000:
            continue
        out.append((B, R, M, L, KT, stride, pad, up, pro, r.random() < 0.3))
    return out


@pytest.mark.parametrize("case", _wgrad_cases(28, 4321), ids=lambda c: "-".join(str(int(v)) for v in c))
def test_conv1d_wgrad_sweep(dev, case):
    B, R, M, L, KT, stride, pad, up, pro, accumulate = case
    x = rnd(B, R, L, seed=1) * 1.2 + 0.1
    w = rnd(M, R, KT, seed=2, scale=0.2).requires_grad_()
    b = rnd(M, seed=3).requires_grad_()
    xin, kw = x, {}
    if pro:
        G = 4 if R % 8 else 8
        gamma, beta = rnd(R, seed=4) * 0.5 + 1, rnd(R, seed=5) * 0.1
        xin = F.silu(F.group_norm(x, G, gamma, beta, eps=1e-5))
        kw = dict(prologue=1, pro_stats=ops.gn_stats(x.to(dev), G), pro_gamma=gamma.to(dev), pro_beta=beta.to(dev), groups=G)
    xr = F.interpolate(xin, scale_factor=up, mode="nearest") if up > 1 else xin
    y = F.conv1d(xr, w, b, stride=stride, padding=pad)
    dy = rnd(*y.shape, seed=9)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    if accumulate:
        dw0, db0 = rnd(M, R, KT, seed=11), rnd(M, seed=12)
        kw.update(dw=dw0.clone().to(dev), dbias=db0.clone().to(dev), accumulate=True)
        dw_ref, db_ref = dw_ref + dw0, db_ref + db0
    dw, db = ops.conv1d_wgrad(x.to(dev), dy.to(dev), KT, stride=stride, pad=pad, up=up, **kw)
    assert rel_err(dw, dw_ref) < TOL
    assert rel_err(db, db_ref) < TOL


def _norm_cases(n, seed):
    r = random.Random(seed)
    return [(r.randint(1, 3), r.choice([8, 16, 24, 32, 64, 72, 128]), r.choice([17, 64, 100, 256, 513, 1024]), r.choice([4, 8]))
            for _ in range(n)]


@pytest.mark.parametrize("B,C,L,G", _norm_cases(10, 99))
def test_groupnorm_silu_fwd_bwd_sweep(dev, B, C, L, G):
    """GroupNorm statistics + SiLU(GroupNorm) backward (gn_stats / gn_bwd) against autograd on random shapes."""
    x = (rnd(B, C, L, seed=1) * 1.5 + 0.3).requires_grad_()
    gamma, beta = (rnd(C, seed=2) * 0.5 + 1).requires_grad_(), (rnd(C, seed=3) * 0.1).requires_grad_()
    y = F.silu(F.group_norm(x, G, gamma, beta, eps=1e-5))
    dy = rnd(B, C, L, seed=4)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (x, gamma, beta), dy)
    xd = x.detach().to(dev)
    stats = ops.gn_stats(xd, G)
    dx, dg, db = ops.gn_silu_bwd(xd, dy.to(dev), stats, gamma.detach().to(dev), beta.detach().to(dev), G)
    assert rel_err(dx, dx_ref) < TOL
    assert rel_err(dg, dg_ref) < TOL
    assert rel_err(db, db_ref) < TOL


@pytest.mark.parametrize("seed", range(8))
def test_attention_sweep(dev, seed):
    r = random.Random(700 + seed)
    B, H, D = r.randint(1, 2), r.randint(1, 4), r.choice([16, 32, 64])
    n, m = r.choice([20, 64, 97, 160, 256]), r.choice([7, 33, 64, 130, 200])
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
    do = rnd(B, mid, n, seed=3)
    dq_ref, dkv_ref = torch.autograd.grad(o_ref, (q, kv), do)
    dq, dkv = ops.attn_bwd(q.detach().to(dev), kv.detach().to(dev), o, do.to(dev), lse, H, D)
    assert rel_err(dq, dq_ref) < TOL
    assert rel_err(dkv, dkv_ref) < TOL
