"""Host-side operator layer: one Python function per C-ABI entry point of include/adp.h.

These only marshal tensors (device pointers, shapes, the current HIP stream) into the hand-written
gfx950 kernels; they hold no arithmetic of their own and have no fallback.  Outputs are allocated by
the caller or with torch.empty on the input's device (PyTorch = device memory + streams only).
"""
import contextlib
import ctypes
import os
from ctypes import byref
from typing import Optional

import torch
from torch import Tensor

from . import _C
from ._C import ConvDesc, WgradDesc, ptr

GN_EPS = 1e-5
# LayerNorm epsilons of the two item families, separately switchable (oracle/a_unet_restatement.py [switch] block: a_unet's
# Modulation may carry eps=1e-6 on its own LayerNorm; unverifiable offline, both 1e-5 until tools/pin_a_unet.py says otherwise)
MODULATION_LN_EPS = 1e-5
ATTENTION_LN_EPS = 1e-5


def _ws(nbytes: int, like: Tensor) -> Tensor:
    return torch.empty(max(1, (nbytes + 3) // 4), dtype=torch.float32, device=like.device)


def conv_out_len(Lin: int, KT: int, stride: int, dil: int, pad: int, up: int) -> int:
    return (Lin * up + 2 * pad - dil * (KT - 1) - 1) // stride + 1


def conv1d(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, stride: int = 1, dil: int = 1, pad: int = 0,
           up: int = 1, transposed: bool = False, x2: Optional[Tensor] = None, prologue: int = 0,
           pro_stats: Optional[Tensor] = None, pro_gamma: Optional[Tensor] = None, pro_beta: Optional[Tensor] = None,
           groups: int = 1, e_scale: Optional[Tensor] = None, e_bstride: int = 0, res: Optional[Tensor] = None,
           store: int = 0, sp: int = 1, N: Optional[int] = None, out: Optional[Tensor] = None,
           out_pre: Optional[Tensor] = None, gn: Optional["GnPart"] = None, gnb: Optional["GnBwdPart"] = None) -> Tensor:
    """Fused implicit-GEMM conv (adp_conv1d).  w: [M, R, KT] (or [R, M, KT] when transposed).
    `gn`: a GnPart to fill with the GroupNorm partial statistics of the output (left empty when the dispatched kernel
    family cannot produce them; the consumer then runs adp_gn_stats).
    `gnb`: a GnBwdPart naming the SiLU(GroupNorm(x)) whose output gradient this (data-gradient) launch produces: the epilogue
    leaves the first stage of that backward in gnb.ab when the dispatched kernel can (else gnb.ab stays None and
    gn_silu_bwd runs its own pass)."""
    B, R1, Lin = x.shape
    R = R1 + (x2.shape[1] if x2 is not None else 0)
    if transposed:
        Rw, M, KT = w.shape
    else:
        M, Rw, KT = w.shape
    assert Rw == R, f"weight expects {Rw} input channels, got {R}"
    if N is None:
        N = conv_out_len(Lin, KT, stride, dil, pad, up)
    if store == 0:
        oshape = (B, M, N)
    elif store == 1:
        oshape = (B, M // sp, N * sp)
    else:
        oshape = (B, M, N // sp)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    else:
        assert tuple(out.shape) == oshape, (out.shape, oshape)
    d = ConvDesc(ptr(x), ptr(x2), ptr(w), ptr(bias), ptr(pro_stats), ptr(pro_gamma), ptr(pro_beta), ptr(e_scale),
                 ptr(res), ptr(out), ptr(out_pre), B, R, R1, Lin, M, N, KT, stride, dil, pad, up, int(transposed), prologue, groups,
                 store, sp, e_bstride, None, None)
    need = _C.query("adp_conv1d_ws_bytes", byref(d))
    if need > 0:  # split-K partial tiles (small grids: the deep layers at batch 1)
        ws = _ws(need, x)
        d.ws = ptr(ws)
    if gn is not None:
        E = _C.query("adp_conv1d_gn_entries", byref(d))
        if E > 0:
            gn.part = torch.empty((B, M // 4, E, 3), dtype=torch.float32, device=x.device)
            gn.of = out
            d.gn_part = ptr(gn.part)
    if gnb is not None and GNB_EPILOGUE:
        d.gnb_groups = gnb.groups
        d.gnb_x, d.gnb_stats, d.gnb_gamma, d.gnb_beta = ptr(gnb.x), ptr(gnb.stats), ptr(gnb.gamma), ptr(gnb.beta)
        E = _C.query("adp_conv1d_gnb_entries", byref(d))  # (some kernels look at the operands' alignment)
        if E > 0:
            gnb.ab = torch.empty((B, M, E, 2), dtype=torch.float32, device=x.device)
            d.gnb_ab = ptr(gnb.ab)
    if _C.PROFILE is not None:  # algorithmic work of this launch: A_in + A_out (+A_res) + weights
        _C.tag(flops=2 * B * M * N * R * KT,
               bytes=4 * (B * R * Lin + out.numel() + w.numel() + (res.numel() if res is not None else 0)),
               shape=f"B{B} R{R} M{M} N{N} KT{KT} s{stride} up{up} tr{int(transposed)} pro{prologue}")
    _C.call("adp_conv1d", byref(d), _C.stream())
    if _C.REPLAY is not None:
        keep = (x, x2, w, bias, pro_stats, pro_gamma, pro_beta, e_scale, res, out, out_pre, gn.part if gn is not None else None,
                (gnb.x, gnb.stats, gnb.gamma, gnb.beta, gnb.ab) if gnb is not None else None, ws if need > 0 else None)
        _C.REPLAY.append((f"B{B} R{R} M{M} N{N} KT{KT} s{stride} up{up} tr{int(transposed)} pro{prologue}",
                          4 * (B * R * Lin + out.numel() + w.numel() + (res.numel() if res is not None else 0)),
                          lambda d=d, keep=keep: _C.call("adp_conv1d", byref(d), _C.stream())))
    return out


class WgradPark:
    """Parked second stages of split weight gradients (adp_wgrad_desc.accumulate bit 1): the partial slices stay in their own
    scratch until flush() sums every parked gradient of one shape in ONE launch (adp_wgrad_reduce_batch) -- the ConvBlock
    convs of a U-Net depth share their shape, so a block side's four to eight second stages become one."""

    # whole weight-gradient CALLS are collected too (`calls`): small problems, where the inputs of a block side's gradients stay
    # in the Infinity Cache until the side is done, run as one launch per shape (adp_conv1d_wgrad_batch)
    # x + dy of one call; above it only the second stage is parked.  (Round 4: 24 MB, 40 / 80 within noise.  Round 5, with the
    # batch-aware position split of adp_conv1d_wgrad_batch -- an item of a batched launch takes 1/n of the lone split: 24 -> 40 ->
    # 80 MB 12.77 -> 12.72 -> 12.68 ms per step; 80 MB covers every layer of the README configuration)
    BATCH_BYTES = 80 << 20

    # flush() on a SIDE stream (ADP_WGRAD_SIDE=1): nothing downstream of a weight gradient is on the backward's critical path --
    # the data-gradient chain goes on while the batched launches run beside it (a branch of the captured graph), join() is where
    # somebody reads the gradients (the data-parallel hook, the end of the backward).  What the launches read stays referenced
    # until join(): freed on the main stream it could be handed to a later main-stream kernel while the side stream still reads it.
    SIDE = os.environ.get("ADP_WGRAD_SIDE", "0") != "0"
    _side_streams = {}

    def __init__(self):
        self.items = []  # (key = (partials, cnt, M, accumulate, has_bias), ws, dw, dbias)
        self.calls = []  # (shape key, WgradDesc, tensors kept alive)
        self.held = []   # what side-stream launches still read
        self.forked = None
        if "ADP_WGRAD_BATCH_MB" in os.environ:  # (A/B: 0 = only second stages are parked)
            self.BATCH_BYTES = int(os.environ["ADP_WGRAD_BATCH_MB"]) << 20

    def _fork(self, t: Tensor):
        if not (self.SIDE and t.is_cuda):
            return contextlib.nullcontext()
        side = self.forked
        if side is None:
            side = self._side_streams.get(t.device)
            if side is None:
                side = self._side_streams[t.device] = torch.cuda.Stream(device=t.device)
            self.forked = side
        side.wait_stream(torch.cuda.current_stream(t.device))  # (the inputs were produced on the main stream)
        return torch.cuda.stream(side)

    def join(self) -> None:
        """The main stream waits for the side-stream launches: gradients are final for whatever is enqueued next."""
        if self.forked is not None:
            torch.cuda.current_stream(self.forked.device).wait_stream(self.forked)
            self.forked = None
            self.held.clear()

    def add(self, key, ws: Tensor, dw: Tensor, dbias: Optional[Tensor]) -> None:
        self.items.append((key, ws, dw, dbias))

    def add_call(self, key, d: WgradDesc, keep) -> None:
        self.calls.append((key, d, keep))

    def flush(self) -> None:
        if not self.calls and not self.items:
            return
        with self._fork(self.calls[0][2][0] if self.calls else self.items[0][1]):
            self._flush()

    def _flush(self) -> None:
        if self.calls:
            groups = {}
            for key, d, keep in self.calls:
                groups.setdefault(key, []).append(d)
            calls, self.calls = self.calls, []
            for key, ds in groups.items():
                arr = (WgradDesc * len(ds))(*ds)
                if _C.PROFILE is not None:
                    B, R, _, Lin, M, N, KT = key[:7]
                    _C.tag(flops=2 * B * M * N * R * KT * len(ds), bytes=4 * len(ds) * (B * R * Lin + B * M * N + M * R * KT),
                           shape=f"n{len(ds)} B{B} R{R} M{M} N{N} KT{KT}")
                _C.call("adp_conv1d_wgrad_batch", arr, len(ds), _C.stream())
            if self.forked is not None:
                self.held.append(calls)
            del calls
        if not self.items:
            return
        groups = {}
        for key, ws, dw, dbias in self.items:
            groups.setdefault(key, []).append((ws, dw, dbias))
        if self.forked is not None:
            self.held.append(self.items)
        self.items = []
        for (partials, cnt, M, acc, has_bias), g in groups.items():
            n = len(g)
            arr = ctypes.c_void_p * n
            wsa, dwa = arr(*[ptr(t[0]) for t in g]), arr(*[ptr(t[1]) for t in g])
            dba = arr(*[ptr(t[2]) for t in g]) if has_bias else None
            _C.tag(bytes=4 * n * (partials + 1) * (cnt + M), shape=f"n{n} P{partials} cnt{cnt}")
            _C.call("adp_wgrad_reduce_batch", wsa, dwa, dba, n, partials, cnt, M, acc, _C.stream())


def conv1d_wgrad(x: Tensor, dy: Tensor, KT: int, *, stride: int = 1, dil: int = 1, pad: int = 0, up: int = 1,
                 x2: Optional[Tensor] = None, prologue: int = 0, pro_stats: Optional[Tensor] = None,
                 pro_gamma: Optional[Tensor] = None, pro_beta: Optional[Tensor] = None, groups: int = 1,
                 dw: Optional[Tensor] = None, dbias: Optional[Tensor] = None, want_bias: bool = True,
                 accumulate: bool = False, park: Optional[WgradPark] = None):
    B, R1, Lin = x.shape
    R = R1 + (x2.shape[1] if x2 is not None else 0)
    _, M, N = dy.shape
    if dw is None:
        dw = torch.empty((M, R, KT), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((M,), dtype=torch.float32, device=x.device)
    d = WgradDesc(ptr(x), ptr(x2), ptr(dy), ptr(pro_stats), ptr(pro_gamma), ptr(pro_beta), ptr(dw), ptr(dbias), None,
                  B, R, R1, Lin, M, N, KT, stride, dil, pad, up, prologue, groups, int(accumulate))
    ws = _ws(_C.query("adp_conv1d_wgrad_ws_bytes", byref(d)), x)
    d.ws = ptr(ws)
    if park is not None and x2 is None and 4 * (x.numel() + dy.numel()) <= park.BATCH_BYTES:
        # the whole call waits for park.flush(): one launch per shape with the block side's other gradients
        # (tagged there: a tag set here would ride on whatever profiled launch comes next)
        key = (B, R, R1, Lin, M, N, KT, stride, dil, pad, up, prologue, groups, int(accumulate), pro_stats is not None,
               pro_gamma is not None, pro_beta is not None, dbias is not None)
        park.add_call(key, d, (x, dy, pro_stats, pro_gamma, pro_beta, dw, dbias, ws))
        return dw, dbias
    if _C.PROFILE is not None:  # A_x + A_dy + weight-gradient write
        _C.tag(flops=2 * B * M * N * R * KT, bytes=4 * (B * R * Lin + dy.numel() + dw.numel()),
               shape=f"B{B} R{R} M{M} N{N} KT{KT} s{stride} up{up} pro{prologue}")
    if park is not None:
        partials = _C.query("adp_conv1d_wgrad_partials", byref(d))
        if partials > 1:  # the second stage waits for park.flush(); ws belongs to the parked item until then
            d.accumulate = int(accumulate) | 2
            park.add((partials, M * R * KT, M, int(accumulate), dbias is not None), ws, dw, dbias)
    _C.call("adp_conv1d_wgrad", byref(d), _C.stream())
    return dw, dbias


GNB_EPILOGUE = os.environ.get("ADP_GNB_EPILOGUE", "1") != "0"  # (A/B switch: 0 = the GroupNorm backward always runs its own first stage)


class GnBwdPart:
    """Names the SiLU(GroupNorm(x)) behind a data-gradient conv (x, its statistics and affine parameters); after the conv,
    `ab` [B, C, E, 2] holds the first stage of that GroupNorm backward if the conv's epilogue produced it (adp_conv_desc.gnb_ab)."""
    __slots__ = ("x", "stats", "gamma", "beta", "groups", "ab")

    def __init__(self, x: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, groups: int):
        self.x, self.stats, self.gamma, self.beta, self.groups = x, stats, gamma, beta, groups
        self.ab: Optional[Tensor] = None


class GnPart:
    """GroupNorm partial statistics [B, C/4, E, 3] = (mean, M2, count) per slice of a 4-channel row quad, written by
    the kernel that PRODUCED tensor `of` (conv epilogue), so that the consuming GroupNorm needs no pass of its own."""
    __slots__ = ("part", "of")

    def __init__(self):
        self.part: Optional[Tensor] = None
        self.of: Optional[Tensor] = None

    def covers(self, x: Tensor) -> bool:
        return self.part is not None and self.of is x


def gn_finalize(part: Tensor, groups: int, eps: float = GN_EPS) -> Tensor:
    """stats [B, G, 2] from producer-side partials (adp_gn_finalize)."""
    B, CQ, E, _ = part.shape
    C = CQ * 4
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=part.device)
    _C.tag(bytes=4 * part.numel(), shape=f"B{B} C{C} E{E}")
    _C.call("adp_gn_finalize", ptr(part), B, C, E, groups, eps, ptr(stats), _C.stream())
    return stats


def gn_finalize_act(x: Tensor, part: Tensor, groups: int, gamma: Tensor, beta: Tensor, eps: float = GN_EPS):
    """(stats, SiLU(GroupNorm(x))) from producer-side partials in one launch (adp_gn_finalize_act)."""
    B, C, L = x.shape
    E = part.shape[2]
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    act = torch.empty_like(x)
    _C.tag(bytes=8 * x.numel(), shape=f"B{B} C{C} L{L} E{E}")
    _C.call("adp_gn_finalize_act", ptr(x), ptr(part), B, C, L, E, groups, eps, ptr(gamma), ptr(beta), ptr(stats),
            ptr(act), _C.stream())
    return stats, act


def gn_act(x: Tensor, stats: Tensor, groups: int, gamma: Tensor, beta: Tensor) -> Tensor:
    """SiLU(GroupNorm(x)) materialised from finished statistics (adp_gn_act)."""
    B, C, L = x.shape
    act = torch.empty_like(x)
    _C.tag(bytes=8 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_gn_act", ptr(x), ptr(stats), ptr(gamma), ptr(beta), B, C, L, groups, ptr(act), _C.stream())
    return act


def gn_stats(x: Tensor, groups: int, eps: float = GN_EPS, out: Optional[Tensor] = None) -> Tensor:
    B, C, L = x.shape
    stats = out if out is not None else torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    ws = _ws(_C.query("adp_gn_stats_ws_bytes", B, C, L, groups), x)
    _C.tag(bytes=4 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_gn_stats", ptr(x), B, C, L, groups, eps, ptr(stats), ptr(ws), _C.stream())
    return stats


def gn_stats_act(x: Tensor, groups: int, gamma: Tensor, beta: Tensor, eps: float = GN_EPS):
    """(stats [B, G, 2], act = SiLU(GroupNorm(x)) materialised) in two launches (adp_gn_stats_act)."""
    B, C, L = x.shape
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    act = torch.empty_like(x)
    ws = _ws(_C.query("adp_gn_stats_ws_bytes", B, C, L, groups), x)
    _C.tag(bytes=12 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_gn_stats_act", ptr(x), B, C, L, groups, eps, ptr(gamma), ptr(beta), ptr(stats), ptr(act), ptr(ws),
            _C.stream())
    return stats, act


def gn_silu_bwd(x: Tensor, dact: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, groups: int,
                dres: Optional[Tensor] = None, dx: Optional[Tensor] = None, dgamma: Optional[Tensor] = None,
                dbeta: Optional[Tensor] = None, accumulate: bool = False, ab: Optional[Tensor] = None):
    """Backward of SiLU(GroupNorm(x)): returns (dx [+ dres], dgamma, dbeta).  `ab` [B, C, E, 2]: the first stage, already
    left by the epilogue of the conv that produced dact (GnBwdPart.ab): one launch instead of two."""
    B, C, L = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty_like(gamma)
    if dbeta is None:
        dbeta = torch.empty_like(beta)
    s = _C.stream()
    NS = _C.query("adp_row_nsplit", B * C, L)
    if ab is not None:
        _C.tag(bytes=(12 + (4 if dres is not None else 0)) * x.numel(), shape=f"B{B} C{C} L{L}")
        _C.call("adp_gn_silu_bwd_apply_ab", ptr(x), ptr(dact), ptr(stats), ptr(gamma), ptr(beta), ptr(ab), ptr(dres), B, C,
                L, groups, NS, ab.shape[2], ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), s)
        return dx, dgamma, dbeta
    ab = torch.empty((B, C, NS, 2), dtype=torch.float32, device=x.device)
    _C.tag(bytes=8 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_gn_silu_bwd_reduce", ptr(x), ptr(dact), ptr(stats), ptr(gamma), ptr(beta), B, C, L, groups, NS,
            ptr(ab), s)
    _C.tag(bytes=(12 + (4 if dres is not None else 0)) * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_gn_silu_bwd_apply", ptr(x), ptr(dact), ptr(stats), ptr(gamma), ptr(beta), ptr(ab), ptr(dres), B, C,
            L, groups, NS, ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), s)
    return dx, dgamma, dbeta


def modulation_fwd(x: Tensor, ss: Tensor, ss_bstride: int, eps: float = MODULATION_LN_EPS, y: Optional[Tensor] = None,
                   stats: Optional[Tensor] = None):
    """ss: 1-D view whose element [b*ss_bstride + c] is scale and [b*ss_bstride + C + c] is shift."""
    B, C, L = x.shape
    if y is None:
        y = torch.empty_like(x)
    if stats is None:
        stats = torch.empty((B, L, 2), dtype=torch.float32, device=x.device)
    _C.tag(bytes=8 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_modulation_fwd", ptr(x), ptr(ss), ss_bstride, B, C, L, eps, ptr(y), ptr(stats), _C.stream())
    return y, stats


def modulation_ln_fwd(x: Tensor, ss: Tensor, ss_bstride: int, gamma: Tensor, beta: Tensor, gamma2: Optional[Tensor] = None,
                      beta2: Optional[Tensor] = None, eps: float = MODULATION_LN_EPS, eps_ln: float = ATTENTION_LN_EPS):
    """ModulationItem + the LayerNorm(s) of the attention item behind it in one launch (adp_modulation_ln_fwd):
    (y, stats) as modulation_fwd, (xn, xn2 or None, ln_stats) as ln_affine_fwd(y, gamma, beta, gamma2, beta2)."""
    B, C, L = x.shape
    y, xn = torch.empty_like(x), torch.empty_like(x)
    xn2 = torch.empty_like(x) if gamma2 is not None else None
    stats = torch.empty((B, L, 2), dtype=torch.float32, device=x.device)
    ln_stats = torch.empty((B, L, 2), dtype=torch.float32, device=x.device)
    _C.tag(bytes=4 * x.numel() * (4 if xn2 is not None else 3), shape=f"B{B} C{C} L{L}")
    _C.call("adp_modulation_ln_fwd", ptr(x), ptr(ss), ss_bstride, B, C, L, eps, ptr(y), ptr(stats), eps_ln, ptr(gamma),
            ptr(beta), ptr(xn), ptr(gamma2), ptr(beta2), ptr(xn2), ptr(ln_stats), _C.stream())
    return y, stats, xn, xn2, ln_stats


def modulation_bwd(x: Tensor, dy: Tensor, ss: Tensor, ss_bstride: int, stats: Tensor, dss: Tensor, dss_bstride: int,
                   dx: Optional[Tensor] = None) -> Tensor:
    B, C, L = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    ws = _ws(_C.query("adp_chan_ln_bwd_ws_bytes", B, C, L), x)
    _C.tag(bytes=12 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_modulation_bwd", ptr(x), ptr(dy), ptr(ss), ss_bstride, ptr(stats), B, C, L, ptr(dx), ptr(dss),
            dss_bstride, ptr(ws), _C.stream())
    return dx


class ModulationSums:
    """Parked second stages of Modulation backwards (adp_modulation_bwd_partial / _reduce): the per-tile channel sums of
    each backward stay in its workspace until `flush`, which sums every parked item whose scale / shift gradient slice
    lies in [lo, hi) of the conditioning-bank row -- the items of one U-Net depth, which share their shape -- in one
    launch per eight."""

    def __init__(self):
        self.items = []  # (offset into the bank row, ws, dss view, B, C, NT, dss_bstride)

    def partial(self, off: int, x: Tensor, dy: Tensor, ss: Tensor, ss_bstride: int, stats: Tensor, dss: Tensor,
                dss_bstride: int) -> Tensor:
        B, C, L = x.shape
        dx = torch.empty_like(x)
        ws = _ws(_C.query("adp_chan_ln_bwd_ws_bytes", B, C, L), x)
        _C.tag(bytes=12 * x.numel(), shape=f"B{B} C{C} L{L}")
        NT = _C.call_value("adp_modulation_bwd_partial", ptr(x), ptr(dy), ptr(ss), ss_bstride, ptr(stats), B, C, L, ptr(dx),
                           ptr(ws), _C.stream())
        self.items.append((off, ws, dss, B, C, NT, dss_bstride))
        return dx

    def partial_ln(self, off: int, x: Tensor, ss: Tensor, ss_bstride: int, stats: Tensor, dss: Tensor, dss_bstride: int,
                   y: Tensor, dxn: Tensor, gamma: Tensor, ln_stats: Tensor, dres: Optional[Tensor], dgb: Tensor) -> Tensor:
        """`partial` for a ModulationItem whose output y went through an attention item's LayerNorm (gamma, ln_stats): the
        LayerNorm's backward of d(xn) (+ dres) and this item's backward in one pass (adp_modulation_ln_bwd_partial);
        dgb = the LayerNorm's [dgamma | dbeta] destination."""
        B, C, L = x.shape
        dx = torch.empty_like(x)
        nbytes = _C.query("adp_chan_ln_bwd_ws_bytes", B, C, L)
        ws, ws_ln = _ws(nbytes, x), _ws(nbytes, x)
        _C.tag(bytes=16 * x.numel(), shape=f"B{B} C{C} L{L}")
        NT = _C.call_value("adp_modulation_ln_bwd_partial", ptr(x), ptr(ss), ss_bstride, ptr(stats), ptr(y), ptr(dxn),
                           ptr(gamma), ptr(ln_stats), ptr(dres), B, C, L, 0, ptr(dx), ptr(ws), ptr(dgb), ptr(ws_ln),
                           _C.stream())
        self.items.append((off, ws, dss, B, C, NT, dss_bstride))
        return dx

    def flush(self, lo: Optional[int] = None, hi: Optional[int] = None) -> None:
        take = [it for it in self.items if lo is None or lo <= it[0] < hi]
        if not take:
            return
        self.items = [it for it in self.items if not (lo is None or lo <= it[0] < hi)]
        groups = {}
        for it in take:
            groups.setdefault(it[3:], []).append(it)
        for (B, C, NT, bstride), its in groups.items():
            n = len(its)
            wsp = (ctypes.c_void_p * n)(*[ptr(it[1]) for it in its])
            dsp = (ctypes.c_void_p * n)(*[ptr(it[2]) for it in its])
            _C.call("adp_modulation_bwd_reduce", wsp, dsp, n, B, C, NT, bstride, _C.stream())


def ln_stats(x: Tensor, eps: float = ATTENTION_LN_EPS) -> Tensor:
    B, C, L = x.shape
    stats = torch.empty((B, L, 2), dtype=torch.float32, device=x.device)
    _C.call("adp_ln_stats", ptr(x), B, C, L, eps, ptr(stats), _C.stream())
    return stats


def ln_affine_fwd(x: Tensor, gamma: Tensor, beta: Tensor, gamma2: Optional[Tensor] = None,
                  beta2: Optional[Tensor] = None, eps: float = ATTENTION_LN_EPS):
    """(y, y2 or None, stats): y = LayerNorm_C(x) * gamma + beta, y2 likewise with (gamma2, beta2), stats [B, L, 2]."""
    B, C, L = x.shape
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if gamma2 is not None else None
    stats = torch.empty((B, L, 2), dtype=torch.float32, device=x.device)
    _C.tag(bytes=4 * x.numel() * (3 if y2 is not None else 2), shape=f"B{B} C{C} L{L}")
    _C.call("adp_ln_affine_fwd", ptr(x), B, C, L, eps, ptr(gamma), ptr(beta), ptr(y), ptr(gamma2), ptr(beta2), ptr(y2),
            ptr(stats), _C.stream())
    return y, y2, stats


def ln_bwd(x: Tensor, dxn: Tensor, stats: Tensor, gamma: Tensor, dres: Optional[Tensor] = None,
           dgb: Optional[Tensor] = None, accumulate: bool = False):
    """Backward of LayerNorm-over-channels with affine: returns (dx [+ dres], [dgamma | dbeta])."""
    B, C, L = x.shape
    dx = torch.empty_like(x)
    if dgb is None:
        dgb = torch.empty((2 * C,), dtype=torch.float32, device=x.device)
    ws = _ws(_C.query("adp_chan_ln_bwd_ws_bytes", B, C, L), x)
    _C.call("adp_ln_bwd", ptr(x), ptr(dxn), ptr(stats), ptr(gamma), ptr(dres), B, C, L, int(accumulate), ptr(dx),
            ptr(dgb), ptr(ws), _C.stream())
    return dx, dgb


LIN_ROWS = 16  # activation rows one adp_linear_* launch holds in LDS (LIN_BMAX in csrc/linear.hip)


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], act: int = 0, post: int = 0,
               y: Optional[Tensor] = None) -> Tensor:
    B, K = x.shape
    N = w.shape[0]
    if y is None:
        y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _C.tag(bytes=4 * (w.numel() + x.numel() + y.numel()), shape=f"B{B} K{K} N{N}")
    for b0 in range(0, B, LIN_ROWS):  # batches beyond 16 rows: the weight matrix is streamed once per 16 rows
        nb = min(LIN_ROWS, B - b0)
        _C.call("adp_linear_fwd", ptr(x[b0:b0 + nb]), ptr(w), ptr(bias), nb, K, N, act, post, ptr(y[b0:b0 + nb]), N,
                _C.stream())
    return y


def linear_bwd_data(dy: Tensor, w: Tensor, dxa: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    B, N = dy.shape
    K = w.shape[1]
    if dxa is None:
        dxa = torch.empty((B, K), dtype=torch.float32, device=dy.device)
    ws = _ws(_C.query("adp_linear_bwd_data_ws_bytes", min(B, LIN_ROWS), K, N), dy)
    _C.tag(bytes=4 * (w.numel() + B * N + dxa.numel()), shape=f"B{B} K{K} N{N}")
    for b0 in range(0, B, LIN_ROWS):
        nb = min(LIN_ROWS, B - b0)
        _C.call("adp_linear_bwd_data", ptr(dy[b0:b0 + nb]), N, ptr(w), nb, K, N, int(accumulate),
                ptr(dxa[b0:b0 + nb]), ptr(ws), _C.stream())
    return dxa


def linear_bwd_weight(dy: Tensor, x: Tensor, act: int = 0, dw: Optional[Tensor] = None,
                      dbias: Optional[Tensor] = None, want_bias: bool = True, accumulate: bool = False,
                      rows: Optional[int] = None, dy_bstride: Optional[int] = None):
    """dy [B, N]; or, with `rows`, a 1-D view whose element [b*dy_bstride + n] is dy[b, n] for n < rows (a column
    slice of a wider table, e.g. one depth's rows of the conditioning bank)."""
    if rows is None:
        B, N = dy.shape
        dy_bstride = N
        dy = dy.reshape(-1)
    else:
        B, N = x.shape[0], rows
    K = x.shape[1]
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    if dbias is None and want_bias:
        dbias = torch.empty((N,), dtype=torch.float32, device=dy.device)
    _C.tag(bytes=4 * (dw.numel() + B * N + x.numel()), shape=f"B{B} K{K} N{N}")
    for b0 in range(0, B, LIN_ROWS):  # the second and later groups of 16 rows accumulate into dw / dbias
        nb = min(LIN_ROWS, B - b0)
        _C.call("adp_linear_bwd_weight", ptr(dy[b0 * dy_bstride:]), dy_bstride, ptr(x[b0:b0 + nb]), nb, K, N, act,
                int(accumulate or b0 > 0), ptr(dw), ptr(dbias), _C.stream())
    return dw, dbias


def time_fourier_fwd(t: Tensor, w: Tensor) -> Tensor:
    B, H = t.shape[0], w.shape[0]
    four = torch.empty((B, 2 * H + 1), dtype=torch.float32, device=t.device)
    _C.call("adp_time_fourier_fwd", ptr(t), ptr(w), B, H, ptr(four), _C.stream())
    return four


def time_fourier_bwd(t: Tensor, w: Tensor, dfour: Tensor, dw: Optional[Tensor] = None,
                     accumulate: bool = False) -> Tensor:
    B, H = t.shape[0], w.shape[0]
    if dw is None:
        dw = torch.empty_like(w)
    _C.call("adp_time_fourier_bwd", ptr(t), ptr(w), ptr(dfour), B, H, int(accumulate), ptr(dw), _C.stream())
    return dw


def act_fwd(x: Tensor, act: int) -> Tensor:
    y = torch.empty_like(x)
    _C.call("adp_act_fwd", ptr(x), x.numel(), act, ptr(y), _C.stream())
    return y


def act_bwd(x: Tensor, dy: Tensor, act: int, dx: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    if dx is None:
        dx = torch.empty_like(x)
    _C.call("adp_act_bwd", ptr(x), ptr(dy), x.numel(), act, int(accumulate), ptr(dx), _C.stream())
    return dx


def skipmod_bwd(g: Tensor, x: Tensor, scale: Tensor, scale_bstride: int, dscale: Tensor, dscale_bstride: int,
                dx: Optional[Tensor] = None) -> Tensor:
    B, C, L = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    ws = _ws(_C.query("adp_skipmod_bwd_ws_bytes", B, C, L), x)
    _C.tag(bytes=12 * x.numel(), shape=f"B{B} C{C} L{L}")
    _C.call("adp_skipmod_bwd", ptr(g), ptr(x), ptr(scale), scale_bstride, B, C, L, ptr(dx), ptr(dscale),
            dscale_bstride, ptr(ws), _C.stream())
    return dx


def v_noise(x: Tensor, noise: Tensor, sigma: Tensor):
    B = x.shape[0]
    per = x.numel() // B
    x_noisy, v_target = torch.empty_like(x), torch.empty_like(x)
    _C.call("adp_v_noise", ptr(x), ptr(noise), ptr(sigma), B, per, ptr(x_noisy), ptr(v_target), _C.stream())
    return x_noisy, v_target


def mse_fwd(v_pred: Tensor, v_target: Tensor) -> Tensor:
    n = v_pred.numel()
    loss = torch.empty((), dtype=torch.float32, device=v_pred.device)
    ws = _ws(_C.query("adp_mse_ws_bytes", n), v_pred)
    _C.call("adp_mse_fwd", ptr(v_pred), ptr(v_target), n, ptr(loss), ptr(ws), _C.stream())
    return loss


def mse_bwd(v_pred: Tensor, v_target: Tensor, gloss: Optional[Tensor]) -> Tensor:
    dv = torch.empty_like(v_pred)
    _C.call("adp_mse_bwd", ptr(v_pred), ptr(v_target), ptr(gloss), v_pred.numel(), ptr(dv), _C.stream())
    return dv


def v_step(x: Tensor, v: Tensor, ab4: Tensor, out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = torch.empty_like(x)
    _C.call("adp_v_step", ptr(x), ptr(v), ptr(ab4), x.numel(), ptr(out), _C.stream())
    return out


def v_inpaint_step(x: Tensor, v: Tensor, source: Tensor, noise: Tensor, mask_u8: Tensor, ab4: Tensor,
                   out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = torch.empty_like(x)
    _C.call("adp_v_inpaint_step", ptr(x), ptr(v), ptr(source), ptr(noise), ptr(mask_u8, torch.uint8), ptr(ab4),
            x.numel(), ptr(out), _C.stream())
    return out


def cfg_mix(y2: Tensor, scale: float) -> Tensor:
    """y2 [2B, ...] -> y2[B:] + (y2[:B] - y2[B:]) * scale."""
    half = y2.numel() // 2
    out = torch.empty((y2.shape[0] // 2,) + tuple(y2.shape[1:]), dtype=torch.float32, device=y2.device)
    _C.call("adp_cfg_mix", ptr(y2), half, float(scale), ptr(out), _C.stream())
    return out


def select_rows(a: Tensor, b: Tensor, pick_u8: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[r] = pick[r] ? a[r] : b[r] for the leading dimension r."""
    rows = a.shape[0]
    per = a.numel() // rows
    if out is None:
        out = torch.empty_like(a)
    _C.call("adp_select_rows", ptr(a), ptr(b), ptr(pick_u8, torch.uint8), rows, per, ptr(out), _C.stream())
    return out


def resample(x: Tensor, kern: Tensor, fi: int, fo: int, width: int, out_len: int) -> Tensor:
    """x [B, C, length], kern [fo, J] -> [B, C, out_len] (adp_resample)."""
    B, C, length = x.shape
    J = kern.shape[-1]
    out = torch.empty((B, C, out_len), dtype=torch.float32, device=x.device)
    _C.tag(bytes=4 * (x.numel() + out.numel()), shape=f"rows{B * C} len{length} {fi}->{fo}")
    _C.call("adp_resample", ptr(x), ptr(kern), B * C, length, fi, fo, J, width, out_len, ptr(out), _C.stream())
    return out


def add(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = torch.empty_like(a)
    _C.call("adp_add", ptr(a), ptr(b), a.numel(), ptr(out), _C.stream())
    return out


def axpby(a: float, x: Tensor, b: float = 0.0, y: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """out = a * x + b * y (y optional)."""
    if out is None:
        out = torch.empty_like(x)
    _C.call("adp_axpby", float(a), ptr(x), float(b), ptr(y), x.numel(), ptr(out), _C.stream())
    return out


def concat_channels(x: Tensor, x2: Tensor) -> Tensor:
    """torch.cat([x, x2], dim=1) of [B, C, L] tensors as two strided row copies (adp_copy2d)."""
    B, C1, L = x.shape
    C2 = x2.shape[1]
    assert x2.shape[0] == B and x2.shape[2] == L, "concat_channels: batch / length mismatch"
    out = torch.empty((B, C1 + C2, L), dtype=torch.float32, device=x.device)
    flat = out.view(-1)
    s = _C.stream()
    _C.call("adp_copy2d", ptr(x), C1 * L, ptr(flat), (C1 + C2) * L, B, C1 * L, s)
    _C.call("adp_copy2d", ptr(x2), C2 * L, ptr(flat[C1 * L:]), (C1 + C2) * L, B, C2 * L, s)
    return out


def split_channels(g: Tensor, C1: int):
    """Inverse of concat_channels: contiguous (g[:, :C1], g[:, C1:])."""
    B, C, L = g.shape
    a = torch.empty((B, C1, L), dtype=torch.float32, device=g.device)
    b = torch.empty((B, C - C1, L), dtype=torch.float32, device=g.device)
    flat = g.view(-1)
    s = _C.stream()
    _C.call("adp_copy2d", ptr(flat), C * L, ptr(a), C1 * L, B, C1 * L, s)
    _C.call("adp_copy2d", ptr(flat[C1 * L:]), C * L, ptr(b), (C - C1) * L, B, (C - C1) * L, s)
    return a, b


def unshuffle(x: Tensor, f: int) -> Tensor:
    """[B, C, L] -> [B, C*f, L/f] with out[b, c*f + k, l] = x[b, c, l*f + k] (adp_unshuffle)."""
    B, C, L = x.shape
    assert L % f == 0, "length must be divisible by the downsample factor"
    out = torch.empty((B, C * f, L // f), dtype=torch.float32, device=x.device)
    _C.call("adp_unshuffle", ptr(x), B * C, L, f, ptr(out), _C.stream())
    return out


def pool_sum(x: Tensor, f: int, res: Optional[Tensor] = None) -> Tensor:
    """[B, C, L*f] -> [B, C, L]: sums f adjacent positions (+ res) (adp_pool_sum)."""
    B, C, Lf = x.shape
    out = torch.empty((B, C, Lf // f), dtype=torch.float32, device=x.device)
    _C.call("adp_pool_sum", ptr(x), B * C, Lf // f, f, ptr(res), ptr(out), _C.stream())
    return out


def attn_fwd(q: Tensor, kv: Tensor, heads: int, head_features: int):
    """q [B, H*D, n]; kv [B, 2*H*D, m] (k = first half of the channels, v = second half) -> o [B, H*D, n], lse."""
    B, mid, n = q.shape
    m = kv.shape[2]
    H, D = heads, head_features
    assert mid == H * D and kv.shape[1] == 2 * mid
    o = torch.empty_like(q)
    lse = torch.empty((B, H, n), dtype=torch.float32, device=q.device)
    kvf = kv.view(-1)
    need = _C.query("adp_attn_fwd_ws_bytes", B, H, D, n, m)
    ws = _ws(need, q) if need > 0 else None  # key-split partials (small grids)
    _C.tag(flops=4 * B * H * n * m * D, bytes=4 * (2 * q.numel() + kv.numel()), shape=f"B{B} H{H} D{D} n{n} m{m}")
    _C.call("adp_attn_fwd", ptr(q), ptr(kvf), ptr(kvf[mid * m:]), B, H, D, n, m, mid * n, 2 * mid * m, ptr(o),
            ptr(lse), ptr(ws), _C.stream())
    return o, lse


def ctx_fold_fwd(tab: Tensor, I: int, M2: int, E: int):
    """Folded context bank of I CrossAttentionItems (adp_ctx_fold_fwd): tab = device int64 [3, I] pointers of (to_kv weight
    [M2, E], norm_context weight [E], norm_context bias [E]) -> (w_all [I*M2, E] = W_i diag(gamma_i), bias_all [I*M2] = W_i beta_i)."""
    w_all = torch.empty((I * M2, E), dtype=torch.float32, device=tab.device)
    bias_all = torch.empty((I * M2,), dtype=torch.float32, device=tab.device)
    _C.tag(bytes=8 * w_all.numel(), shape=f"I{I} M{M2} E{E}")
    _C.call("adp_ctx_fold_fwd", ptr(tab[0], torch.int64), ptr(tab[1], torch.int64), ptr(tab[2], torch.int64), I, M2, E,
            ptr(w_all), ptr(bias_all), _C.stream())
    return w_all, bias_all


def ctx_fold_bwd(tab: Tensor, dw_all: Tensor, dbias_all: Tensor, I: int, M2: int, E: int, flat: Tensor, dw_off: Tensor,
                 dgb_off: Tensor) -> None:
    """Gradients of the folded bank back to the items' parameters, written into the flat gradient buffer (adp_ctx_fold_bwd)."""
    _C.tag(bytes=12 * dw_all.numel(), shape=f"I{I} M{M2} E{E}")
    _C.call("adp_ctx_fold_bwd", ptr(tab[0], torch.int64), ptr(tab[1], torch.int64), ptr(tab[2], torch.int64), ptr(dw_all),
            ptr(dbias_all), I, M2, E, ptr(flat), ptr(dw_off, torch.int64), ptr(dgb_off, torch.int64), _C.stream())


def copy_rows(src: Tensor, src_stride: int, dst: Tensor, dst_stride: int, rows: int, cols: int) -> None:
    """dst[r * dst_stride + c] = src[r * src_stride + c] over flat fp32 views (adp_copy2d)."""
    _C.call("adp_copy2d", ptr(src), src_stride, ptr(dst), dst_stride, rows, cols, _C.stream())


def attn_bwd(q: Tensor, kv: Tensor, o: Tensor, dout: Tensor, lse: Tensor, heads: int, head_features: int,
             dkv: Optional[Tensor] = None):
    """Returns (dq [B, H*D, n], dkv [B, 2*H*D, m])."""
    B, mid, n = q.shape
    m = kv.shape[2]
    H, D = heads, head_features
    dq = torch.empty_like(q)
    if dkv is None:
        dkv = torch.empty_like(kv)
    ws = _ws(_C.query("adp_attn_bwd_ws_bytes", B, H, D, n, m), q)
    kvf, dkvf = kv.view(-1), dkv.view(-1)
    # QK^T is recomputed in both passes: dV, dP, dK in the key/value pass and dP, dQ in the query pass (7 contractions)
    _C.tag(flops=14 * B * H * n * m * D, bytes=4 * (4 * q.numel() + 2 * kv.numel()), shape=f"B{B} H{H} D{D} n{n} m{m}")
    _C.call("adp_attn_bwd", ptr(q), ptr(kvf), ptr(kvf[mid * m:]), ptr(o), ptr(dout), ptr(lse), B, H, D, n, m,
            mid * n, 2 * mid * m, ptr(dq), ptr(dkvf), ptr(dkvf[mid * m:]), ptr(ws), _C.stream())
    return dq, dkv
