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
from typing import Optional

import torch
from torch import Tensor

from . import ops


def attention_item(run, p, x: Tensor, context: Optional[Tensor]) -> Tensor:
    net = run.net
    H, D = net.heads, net.head_features
    mid = H * D
    B, C, n = x.shape
    is_cross = context is not None
    wq = p.to_q.weight.view(mid, C, 1)
    if is_cross:
        assert context.shape[0] == B, "embedding batch mismatch"
        # [B, E, m] channel-major (layout change only), made ONCE per forward: every cross-attention item of the
        # U-Net reads the same embedding (32 items in BASELINE config 4)
        cached = getattr(run, "_ctx_cm", None)
        if cached is None or cached[0] is not context:
            cached = run._ctx_cm = (context, context.transpose(1, 2).contiguous())
        ctx = cached[1]
        xn, _, st_x = ops.ln_affine_fwd(x, p.norm.weight, p.norm.bias)
        cn, _, st_c = ops.ln_affine_fwd(ctx, p.norm_context.weight, p.norm_context.bias)
    else:  # self attention: one pass over x yields both normalisations (same statistics, two affine maps)
        ctx = x
        xn, cn, st_x = ops.ln_affine_fwd(x, p.norm.weight, p.norm.bias, p.norm_context.weight, p.norm_context.bias)
        st_c = st_x
    Cc = ctx.shape[1]
    wkv = p.to_kv.weight.view(2 * mid, Cc, 1)
    # the normalised tensors are materialised once (2-4 MB) so that the projections are plain MFMA 1x1 convs; the
    # LayerNorm-in-the-loader variant ran on the generic conv kernel at 1-4 TF (round-1 profile)
    q = ops.conv1d(xn, wq, None)
    kv = ops.conv1d(cn, wkv, None)
    o, lse = ops.attn_fwd(q, kv, H, D)
    wo = p.to_out.weight.view(C, mid, 1)
    y = ops.conv1d(o, wo, None, res=x)

    if run.need_grad:
        def bwd(gy):
            ops.conv1d_wgrad(o, gy, 1, dw=run.g(p.to_out.weight).view(C, mid, 1), want_bias=False)
            do = ops.conv1d(gy, wo, None, transposed=True)
            dq, dkv = ops.attn_bwd(q, kv, o, do, lse, H, D)
            # q path: weight gradient with the LayerNorm applied in the loader, then LayerNorm backward (+ residual)
            ops.conv1d_wgrad(xn, dq, 1, dw=run.g(p.to_q.weight).view(mid, C, 1), want_bias=False)
            dxn = ops.conv1d(dq, wq, None, transposed=True)
            # [dgamma | dbeta] land directly in the flat gradient buffer (weight and bias are adjacent parameters)
            dx, _ = ops.ln_bwd(x, dxn, st_x, p.norm.weight, dres=gy, dgb=run.gspan(p.norm.weight, 2 * C))
            # k/v path
            ops.conv1d_wgrad(cn, dkv, 1, dw=run.g(p.to_kv.weight).view(2 * mid, Cc, 1), want_bias=False)
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
