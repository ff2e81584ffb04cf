"""AttentionItem / CrossAttentionItem on the gfx950 kernels (host orchestration only).

Semantics (a_unet Attention as used at /root/reference/audio_diffusion_pytorch/components.py:92-93; SURVEY.md 8a
row a15, oracle/a_unet_restatement.py:Attention):
    x [B,C,n] -> x + to_out(softmax(q k^T D^-0.5) v),  q = to_q(LN(x)),  k,v = chunk(to_kv(LN_ctx(ctx)))
    ctx = x (self attention, separate LayerNorm parameters) or the embedding [B,m,E] (cross attention).

MI355X structure: activations never leave the channel-major [B,C,n] layout (the reference packs to [B,n,C] and
back).  The LayerNorms are one channel-axis kernel (both normalisations of a self-attention item in one pass), a
Linear over channels is a 1x1 conv on the MFMA conv kernel; the attention core is the flash-style kernel in csrc/attention.hip, which consumes the
projections exactly as the convs produce them (k and v are the two channel halves of one to_kv output) and
never materialises the [n, m] score matrix; the residual add is the out-projection's epilogue.
"""
import os
from typing import Optional

import torch
from torch import Tensor

from . import ops


class CtxBank:
    """The context side of every CrossAttentionItem of one forward pass as ONE weight bank (csrc/ctx_bank.hip): all items read
    the SAME embedding, so xhat = LayerNorm-without-affine(context) is computed once and kv_i = (W_i diag(gamma_i)) xhat +
    W_i beta_i for all items is one 1x1 conv over the folded bank; the backward is one weight-gradient and one data-gradient
    launch over the bank + the un-fold.  Replaces, per item, LayerNorm + projection (+ split-K reduce) in the forward and
    weight gradient + reduce, data gradient + reduce, LayerNorm backward + reduce and the embedding-gradient add in the
    backward -- about ten launches on a [B, E, 64] tensor each (BASELINE config 4: 32 items).
    Data-parallel safe: the items' context-side gradients (norm_context, to_kv) live in their OWN trailing region of the flat
    gradient buffer (UNetV0Net._param_offsets), `backward` runs as soon as the shallowest cross-attention block is done and
    the hook then gets that region as one bucket (unet._UNetFn.backward)."""

    @staticmethod
    def prepare(run, context: Tensor):
        net = run.net
        items = [p for d in range(len(net.blocks)) for mods in (net.blocks[d].items_down, net.blocks[d].items_up)
                 for t, p in zip(net.item_types[d], mods) if t == "cross_attention"]
        if len(items) < 2:
            return None
        # the folded weights and their gradient live for the whole step (2 x I * 2HD * E floats: ~200 MB in config 4): a cap
        M2_, E_ = items[0].to_kv.weight.shape
        if 8 * len(items) * M2_ * E_ > (int(os.environ.get("ADP_CTX_BANK_MAX_MB", "4096")) << 20):
            return None  # (per-item path: nothing is kept beyond an item's own backward)
        dev = context.device
        ptrs = tuple(t.data_ptr() for p in items for t in (p.to_kv.weight, p.norm_context.weight, p.norm_context.bias))
        cache = getattr(net, "_ctx_tables", None)
        if cache is None or cache["key"] != (dev, ptrs):
            if context.is_cuda and torch.cuda.is_current_stream_capturing():
                return None  # (the pointer tables are uploaded outside a capture; this call takes the per-item path)
            offs = net._param_offsets()
            net._named_params()
            names = net._pname_cache
            tab = torch.tensor([[p.to_kv.weight.data_ptr() for p in items], [p.norm_context.weight.data_ptr() for p in items],
                                [p.norm_context.bias.data_ptr() for p in items]], dtype=torch.int64).to(dev)
            dw_off = torch.tensor([offs[names[id(p.to_kv.weight)]][0] for p in items], dtype=torch.int64).to(dev)
            dgb_off = torch.tensor([offs[names[id(p.norm_context.weight)]][0] for p in items], dtype=torch.int64).to(dev)
            E = items[0].to_kv.weight.shape[1]
            cache = net._ctx_tables = dict(key=(dev, ptrs), tab=tab, dw_off=dw_off, dgb_off=dgb_off,
                                           ones=torch.ones(E, device=dev), zeros=torch.zeros(E, device=dev))
        self = CtxBank()
        self.items, self.index, self.cache = items, {id(p): i for i, p in enumerate(items)}, cache
        self.I, self.M2, self.E = len(items), items[0].to_kv.weight.shape[0], items[0].to_kv.weight.shape[1]
        self.B, self.m = context.shape[0], context.shape[1]
        self.ctx = context.transpose(1, 2).contiguous()  # [B, E, m] channel-major (layout change only)
        self.xhat, _, self.st = ops.ln_affine_fwd(self.ctx, cache["ones"], cache["zeros"])
        self.w_all, self.bias_all = ops.ctx_fold_fwd(cache["tab"], self.I, self.M2, self.E)
        self.kv_all = ops.conv1d(self.xhat, self.w_all.view(self.I * self.M2, self.E, 1), self.bias_all)  # [B, I*M2, m]
        self.dkv_all = None
        return self

    def kv(self, p) -> Tensor:
        """k | v of item p, packed [B, 2HD, m]: a view of the bank's output at batch 1, a row copy otherwise."""
        i, n = self.index[id(p)], self.M2 * self.m
        if self.B == 1:
            return self.kv_all[:, i * self.M2:(i + 1) * self.M2, :]
        out = torch.empty((self.B, self.M2, self.m), dtype=torch.float32, device=self.kv_all.device)
        ops.copy_rows(self.kv_all.view(-1)[i * n:], self.I * n, out.view(-1), n, self.B, n)
        return out

    def dkv_slot(self, p):
        """Where item p's attention backward leaves d(k | v): its slice of dkv_all at batch 1 (written in place), else None."""
        if self.dkv_all is None:
            self.dkv_all = torch.empty_like(self.kv_all)
        i = self.index[id(p)]
        return self.dkv_all[:, i * self.M2:(i + 1) * self.M2, :] if self.B == 1 else None

    def put_dkv(self, p, dkv: Tensor) -> None:
        if self.B > 1:
            i, n = self.index[id(p)], self.M2 * self.m
            ops.copy_rows(dkv.view(-1), n, self.dkv_all.view(-1)[i * n:], self.I * n, self.B, n)

    def backward(self, run) -> None:
        """After every item's backward: the bank's weight gradient, the un-fold into the items' parameter gradients, the data
        gradient and the LayerNorm backward of the shared normalisation."""
        if self.dkv_all is None:
            return
        c, IM = self.cache, self.I * self.M2
        dw_all, db_all = ops.conv1d_wgrad(self.xhat, self.dkv_all, 1)
        ops.ctx_fold_bwd(c["tab"], dw_all.view(IM, self.E), db_all, self.I, self.M2, self.E, run.flat, c["dw_off"], c["dgb_off"])
        if run.want_emb_grad:
            dxhat = ops.conv1d(self.dkv_all, self.w_all.view(IM, self.E, 1), None, transposed=True)
            run.emb_grad, _ = ops.ln_bwd(self.ctx, dxhat, self.st, c["ones"])


class PendingLN:
    """Gradient handed from a cross-attention item's backward to the backward of the ModulationItem in front of it: the
    LayerNorm backward of d(xn) (+ the residual gradient) has not run yet."""

    def __init__(self, y, dxn, gamma, ln_stats, dres, dgb):
        self.y, self.dxn, self.gamma, self.ln_stats, self.dres, self.dgb = y, dxn, gamma, ln_stats, dres, dgb


def attention_item(run, p, x: Tensor, context: Optional[Tensor]) -> Tensor:
    net = run.net
    H, D = net.heads, net.head_features
    mid = H * D
    B, C, n = x.shape
    is_cross = context is not None
    wq = p.to_q.weight.view(mid, C, 1)
    bank = getattr(run, "ctx_bank", None) if is_cross else None
    # LayerNorm(s) of x that the ModulationItem in front of this item formed in its own launch (unet._Run.modulation)
    ready, run.ln_ready = getattr(run, "ln_ready", None), None
    if ready is not None and ready[0] is not x:
        ready = None
    if bank is not None:
        assert context.shape[0] == B, "embedding batch mismatch"
        if ready is not None:
            xn, st_x = ready[1], ready[3]
        else:
            xn, _, st_x = ops.ln_affine_fwd(x, p.norm.weight, p.norm.bias)
        cn = ctx = st_c = None
    elif is_cross:
        assert context.shape[0] == B, "embedding batch mismatch"
        # [B, E, m] channel-major (layout change only), made ONCE per forward: every cross-attention item of the
        # U-Net reads the same embedding (32 items in BASELINE config 4)
        cached = getattr(run, "_ctx_cm", None)
        if cached is None or cached[0] is not context:
            cached = run._ctx_cm = (context, context.transpose(1, 2).contiguous())
        ctx = cached[1]
        if ready is not None:
            xn, st_x = ready[1], ready[3]
        else:
            xn, _, st_x = ops.ln_affine_fwd(x, p.norm.weight, p.norm.bias)
        cn, _, st_c = ops.ln_affine_fwd(ctx, p.norm_context.weight, p.norm_context.bias)
    else:  # self attention: one pass over x yields both normalisations (same statistics, two affine maps)
        ctx = x
        if ready is not None and ready[2] is not None:
            xn, cn, st_x = ready[1], ready[2], ready[3]
        else:
            xn, cn, st_x = ops.ln_affine_fwd(x, p.norm.weight, p.norm.bias, p.norm_context.weight, p.norm_context.bias)
        st_c = st_x
    # the normalised tensors are materialised once (2-4 MB) so that the projections are plain MFMA 1x1 convs; the
    # LayerNorm-in-the-loader variant ran on the generic conv kernel at 1-4 TF (round-1 profile)
    q = ops.conv1d(xn, wq, None)
    if bank is not None:
        kv = bank.kv(p)
    else:
        Cc = ctx.shape[1]
        wkv = p.to_kv.weight.view(2 * mid, Cc, 1)
        kv = ops.conv1d(cn, wkv, None)
    o, lse = ops.attn_fwd(q, kv, H, D)
    wo = p.to_out.weight.view(C, mid, 1)
    y = ops.conv1d(o, wo, None, res=x)

    if run.need_grad:
        from . import unet as unet_mod
        defer_ln = ready is not None and is_cross and unet_mod.MOD_LN_BWD_FUSE

        def bwd(gy):
            ops.conv1d_wgrad(o, gy, 1, dw=run.g(p.to_out.weight).view(C, mid, 1), want_bias=False, park=run.wpark)
            do = ops.conv1d(gy, wo, None, transposed=True)
            dq, dkv = ops.attn_bwd(q, kv, o, do, lse, H, D, dkv=bank.dkv_slot(p) if bank is not None else None)
            # q path: weight gradient with the LayerNorm applied in the loader, then LayerNorm backward (+ residual)
            ops.conv1d_wgrad(xn, dq, 1, dw=run.g(p.to_q.weight).view(mid, C, 1), want_bias=False, park=run.wpark)
            dxn = ops.conv1d(dq, wq, None, transposed=True)
            # [dgamma | dbeta] land directly in the flat gradient buffer (weight and bias are adjacent parameters)
            if defer_ln:
                # cross attention behind a ModulationItem that formed `xn`: the LayerNorm's backward runs inside that item's
                # backward -- the next tape entry -- as one pass (ops.ModulationSums.partial_ln)
                dx = PendingLN(x, dxn, p.norm.weight, st_x, gy, run.gspan(p.norm.weight, 2 * C))
                net._ln_deferred = getattr(net, "_ln_deferred", 0) + 1  # (visible to tests: which path a step took)
            else:
                dx, _ = ops.ln_bwd(x, dxn, st_x, p.norm.weight, dres=gy, dgb=run.gspan(p.norm.weight, 2 * C))
            if bank is not None:  # k/v path: the bank's backward runs once, after the last item (CtxBank.backward)
                bank.put_dkv(p, dkv)
                return dx
            # k/v path
            ops.conv1d_wgrad(cn, dkv, 1, dw=run.g(p.to_kv.weight).view(2 * mid, Cc, 1), want_bias=False, park=run.wpark)
            dcn = ops.conv1d(dkv, wkv, None, transposed=True)
            gbc = run.gspan(p.norm_context.weight, 2 * Cc)
            if is_cross:
                dctx, _ = ops.ln_bwd(ctx, dcn, st_c, p.norm_context.weight, dgb=gbc)
                if run.want_emb_grad:  # kept channel-major [B, E, m]; the items' contributions meet in adp_add
                    run.emb_grad = dctx if run.emb_grad is None else ops.add(run.emb_grad, dctx, out=run.emb_grad)
            else:
                dx, _ = ops.ln_bwd(x, dcn, st_x, p.norm_context.weight, dres=dx, dgb=gbc)
            return dx
        run.tape.append((bwd, None))
    return y
