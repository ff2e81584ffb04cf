"""
ORACLE (test infrastructure, NOT product code) -- CPU restatement of the U-Net hot path.

PARITY UNPINNED: the block arithmetic of the reference lives in the third-party package
`a-unet` (required unpinned at /root/reference/setup.py:20, imported at
/root/reference/audio_diffusion_pytorch/components.py:5-24; last known PyPI release 0.0.16).
That package is NOT under /root/reference, not installed and not installable offline, and
the reference repository holds no golden vectors for this path.  This file
restates a-unet's published algorithm (a_unet/blocks.py + a_unet/apex.py, as recalled) out
of stock `torch.nn` modules only, so per-primitive numerics are pinned by PyTorch itself
(F.group_norm, F.silu, F.conv1d, F.layer_norm, F.linear, nearest Upsample, softmax).  Which
blocks are built, in what order and with which hyper-parameters follows the reference's own
call site: /root/reference/audio_diffusion_pytorch/components.py:34-105.

Every recalled semantic choice is a single named switch below so it can be flipped in one
place should a real a_unet ever become available.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from math import pi
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

# ----------------------------------------------------------------------------------------
# [switch] constants -- recalled a_unet semantics
# ----------------------------------------------------------------------------------------
DOWNSAMPLE_WIDTH = 1            # Downsample: Conv1d(kernel=factor*width, stride=factor, pad=(k-f)//2)
UPSAMPLE_KERNEL_SIZE = 3        # UpsampleInterpolate: nearest upsample then Conv1d(k=3, padding=1)
RESNET_KERNEL_SIZE = 3
MODULATION_ONE_PLUS_SCALE = True  # Modulation: LN(x, no affine) * (1 + scale) + shift
TIME_EMBED_DIM = 256            # NumberEmbedder(dim=256): [t, sin(2 pi t w), cos(2 pi t w)], w in R^128
TIME_NUM_LAYERS = 2             # TimeConditioningPlugin(num_layers=2): Linear+GELU repeated
TIME_GELU_AFTER_EMBEDDER = True  # time_features = F.gelu(embedder(time)) before the MLP
ATTN_SEPARATE_CONTEXT_NORM = True  # Attention: norm(x) for q, norm_context(context) for k,v (also in self-attn)
SKIP_SCALES_BRANCH = True       # MergeModulate: skip + scale(features) * x_branch
SKIP_CAT_SCALE = 2 ** -0.5      # MergeCat (SkipCat, use_modulation=False): Conv1x1(cat[skip * scale, x_branch])
GN_EPS = 1e-5
# Two LayerNorm epsilons, separately switchable: a_unet's Modulation builds its own LayerNorm(elementwise_affine=False) and may
# carry a DiT-style eps=1e-6 there (a round-4 reviewer's recollection of a_unet 0.0.16 blocks.py; unverifiable offline, ~1e-6
# in effect), while the attention norms are plain nn.LayerNorm (default 1e-5).  Both stay 1e-5 until tools/pin_a_unet.py can
# compare with the real package; the product reads the same two names (audio_diffusion_pytorch_amd/unet.A_UNET_SEMANTICS).
MODULATION_LN_EPS = 1e-5
ATTENTION_LN_EPS = 1e-5

ITEM_RESNET = "resnet"
ITEM_MODULATION = "modulation"
ITEM_INJECT = "inject"
ITEM_ATTENTION = "attention"
ITEM_CROSS_ATTENTION = "cross_attention"


def item_list(items: int, use_modulation: bool, ctx_channels: int, att: int, cross: int) -> List[str]:
    """Item types of one depth, exactly as components.py:88-95 composes them."""
    return (
        [ITEM_RESNET]
        + [ITEM_MODULATION] * int(use_modulation)
        + [ITEM_INJECT] * int(ctx_channels > 0)
        + [ITEM_ATTENTION] * att
        + [ITEM_CROSS_ATTENTION] * cross
    ) * items


class ResnetBlock(nn.Module):
    """a_unet ResnetBlock: x + ConvBlock2(ConvBlock1(x)); ConvBlock = GroupNorm -> SiLU -> Conv1d(k=3,p=1).
    in == out channels in UNetV0's usage (ResnetItem passes channels for both) -> identity shortcut."""

    def __init__(self, channels: int, groups: int):
        super().__init__()
        k = RESNET_KERNEL_SIZE
        self.gn1 = nn.GroupNorm(groups, channels, eps=GN_EPS)
        self.conv1 = nn.Conv1d(channels, channels, k, padding=(k - 1) // 2)
        self.gn2 = nn.GroupNorm(groups, channels, eps=GN_EPS)
        self.conv2 = nn.Conv1d(channels, channels, k, padding=(k - 1) // 2)

    def forward(self, x: Tensor) -> Tensor:
        h = self.conv1(F.silu(self.gn1(x)))
        h = self.conv2(F.silu(self.gn2(h)))
        return x + h


class Modulation(nn.Module):
    """a_unet Packed(Modulation): over [B,n,C]: LayerNorm(C, no affine)(x) * (1+scale) + shift,
    [scale|shift] = Linear(MF -> 2C)(SiLU(features))."""

    def __init__(self, channels: int, modulation_features: int):
        super().__init__()
        self.channels = channels
        self.to_scale_shift = nn.Linear(modulation_features, 2 * channels)

    def forward(self, x: Tensor, features: Tensor) -> Tensor:
        ss = self.to_scale_shift(F.silu(features))  # [B, 2C]
        scale, shift = ss.chunk(2, dim=-1)
        xt = x.transpose(1, 2)  # [B, L, C]
        xn = F.layer_norm(xt, (self.channels,), eps=MODULATION_LN_EPS)
        if MODULATION_ONE_PLUS_SCALE:
            y = xn * (1 + scale[:, None, :]) + shift[:, None, :]
        else:
            y = xn * scale[:, None, :] + shift[:, None, :]
        return y.transpose(1, 2)


class Attention(nn.Module):
    """a_unet Packed(Attention) / Packed(CrossAttention) over [B,n,C]:
    x + to_out(softmax(q k^T * D^-0.5) v), q = to_q(norm(x)), k,v = chunk(to_kv(norm_context(ctx))),
    ctx = x (self) or embedding [B,m,E] (cross); all three Linear layers without bias."""

    def __init__(self, channels: int, head_features: int, num_heads: int, context_features: Optional[int] = None):
        super().__init__()
        self.channels = channels
        self.h, self.d = num_heads, head_features
        self.is_cross = context_features is not None
        cf = context_features if self.is_cross else channels
        mid = num_heads * head_features
        self.norm = nn.LayerNorm(channels, eps=ATTENTION_LN_EPS)
        self.norm_context = nn.LayerNorm(cf, eps=ATTENTION_LN_EPS)
        self.to_q = nn.Linear(channels, mid, bias=False)
        self.to_kv = nn.Linear(cf, 2 * mid, bias=False)
        self.to_out = nn.Linear(mid, channels, bias=False)

    def forward(self, x: Tensor, context: Optional[Tensor] = None) -> Tensor:
        xt = x.transpose(1, 2)  # [B, n, C]
        skip = xt
        ctx = context if self.is_cross else xt
        assert ctx is not None, "You must provide a context when using context_features"
        xn = self.norm(xt)
        cn = self.norm_context(ctx) if ATTN_SEPARATE_CONTEXT_NORM else (xn if not self.is_cross else self.norm_context(ctx))
        q = self.to_q(xn)
        k, v = self.to_kv(cn).chunk(2, dim=-1)
        B, n, _ = q.shape
        m = k.shape[1]
        q = q.view(B, n, self.h, self.d).transpose(1, 2)
        k = k.view(B, m, self.h, self.d).transpose(1, 2)
        v = v.view(B, m, self.h, self.d).transpose(1, 2)
        sim = torch.einsum("bhnd,bhmd->bhnm", q, k) * (self.d ** -0.5)
        attn = sim.softmax(dim=-1)
        out = torch.einsum("bhnm,bhmd->bhnd", attn, v)
        out = out.transpose(1, 2).reshape(B, n, self.h * self.d)
        return (skip + self.to_out(out)).transpose(1, 2)


class InjectChannels(nn.Module):
    """a_unet InjectChannelsItem: Conv1x1(cat[x, channels[depth]]) + x."""

    def __init__(self, channels: int, context_channels: int, depth: int):
        super().__init__()
        self.depth = depth
        self.conv = nn.Conv1d(channels + context_channels, channels, 1)

    def forward(self, x: Tensor, channels: Sequence[Optional[Tensor]]) -> Tensor:
        ctx = channels[self.depth]
        return self.conv(torch.cat([x, ctx], dim=1)) + x


class MergeModulate(nn.Module):
    """a_unet SkipModulate -> MergeModulate: skip + Linear(MF -> C)(SiLU(features))[:, :, None] * x."""

    def __init__(self, channels: int, modulation_features: int):
        super().__init__()
        self.to_scale = nn.Linear(modulation_features, channels)

    def forward(self, skip: Tensor, x: Tensor, features: Tensor) -> Tensor:
        scale = self.to_scale(F.silu(features))[:, :, None]
        if SKIP_SCALES_BRANCH:
            return skip + scale * x
        return scale * skip + x


class MergeCat(nn.Module):
    """a_unet SkipCat -> MergeCat (selected at components.py:99 when use_modulation is False):
    Conv1d(2C -> C, kernel 1)(cat[skip * 2^-1/2, x])."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv1d(2 * channels, channels, 1)

    def forward(self, skip: Tensor, x: Tensor, features=None) -> Tensor:
        return self.conv(torch.cat([skip * SKIP_CAT_SCALE, x], dim=1))


class Upsample(nn.Module):
    """a_unet UpsampleInterpolate: nn.Upsample(scale_factor=f, mode='nearest') then Conv1d(k=3, padding=1)."""

    def __init__(self, in_channels: int, out_channels: int, factor: int):
        super().__init__()
        self.factor = factor
        k = UPSAMPLE_KERNEL_SIZE
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k))
        self.bias = nn.Parameter(torch.empty(out_channels))
        ref = nn.Conv1d(in_channels, out_channels, k, padding=k // 2)
        with torch.no_grad():
            self.weight.copy_(ref.weight)
            self.bias.copy_(ref.bias)

    def forward(self, x: Tensor) -> Tensor:
        k = UPSAMPLE_KERNEL_SIZE
        if self.factor > 1:
            x = F.interpolate(x, scale_factor=self.factor, mode="nearest")
        return F.conv1d(x, self.weight, self.bias, padding=k // 2)


class Block(nn.Module):
    """One depth of a_unet XUNet.  Order (a_unet Block): skip_adapter | downsample, items, inner, items_up,
    upsample | skip merge.  `down`: Conv1d(in -> C, kernel=f*width, stride=f, padding=(f*width-f)//2)."""

    def __init__(self, depth: int, in_channels: int, out_channels: int, channels: int, factor: int,
                 items: List[str], context_channels: int, cfg: dict):
        super().__init__()
        width = DOWNSAMPLE_WIDTH if factor > 1 else 1
        self.down = nn.Conv1d(in_channels, channels, factor * width, stride=factor,
                              padding=(factor * width - factor) // 2)
        self.items_down = nn.ModuleList([self._make(t, depth, channels, context_channels, cfg) for t in items])
        self.items_up = nn.ModuleList([self._make(t, depth, channels, context_channels, cfg) for t in items])
        self.item_types = list(items)
        self.up = Upsample(channels, out_channels, factor)
        self.skip_adapter = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else None
        if cfg["use_modulation"]:
            self.skip = MergeModulate(out_channels, cfg["modulation_features"])
        else:
            self.skip = MergeCat(out_channels)

    @staticmethod
    def _make(t: str, depth: int, channels: int, context_channels: int, cfg: dict) -> nn.Module:
        if t == ITEM_RESNET:
            return ResnetBlock(channels, cfg["resnet_groups"])
        if t == ITEM_MODULATION:
            return Modulation(channels, cfg["modulation_features"])
        if t == ITEM_INJECT:
            return InjectChannels(channels, context_channels, depth)
        if t == ITEM_ATTENTION:
            return Attention(channels, cfg["attention_features"], cfg["attention_heads"])
        if t == ITEM_CROSS_ATTENTION:
            return Attention(channels, cfg["attention_features"], cfg["attention_heads"],
                             context_features=cfg["embedding_features"])
        raise ValueError(t)

    def run_items(self, mods, x, features, embedding, channels):
        for t, m in zip(self.item_types, mods):
            if t == ITEM_RESNET:
                x = m(x)
            elif t == ITEM_MODULATION:
                x = m(x, features)
            elif t == ITEM_INJECT:
                x = m(x, channels)
            elif t == ITEM_ATTENTION:
                x = m(x)
            elif t == ITEM_CROSS_ATTENTION:
                x = m(x, embedding)
        return x


class UNetV0Oracle(nn.Module):
    """CPU restatement of `UNetV0(...)` (components.py:34-105) = TimeConditioningPlugin(XUNet).
    forward(x [B,C,L], time [B], *, features=None, embedding=None, channels=None) -> [B,C_out,L]."""

    def __init__(self, dim: int = 1, in_channels: int = 2, channels: Sequence[int] = (), factors: Sequence[int] = (),
                 items: Sequence[int] = (), attentions: Optional[Sequence[int]] = None,
                 cross_attentions: Optional[Sequence[int]] = None, context_channels: Optional[Sequence[int]] = None,
                 attention_features: Optional[int] = None, attention_heads: Optional[int] = None,
                 embedding_features: Optional[int] = None, resnet_groups: int = 8, use_modulation: bool = True,
                 modulation_features: int = 1024, embedding_max_length: Optional[int] = None,
                 use_time_conditioning: bool = True, use_embedding_cfg: bool = False,
                 use_text_conditioning: bool = False, out_channels: Optional[int] = None):
        super().__init__()
        assert dim == 1
        n = len(channels)
        attentions = list(attentions) if attentions is not None else [0] * n
        cross_attentions = list(cross_attentions) if cross_attentions is not None else [0] * n
        context_channels = list(context_channels) if context_channels is not None else [0] * n
        xs = (channels, factors, items, attentions, cross_attentions, context_channels)
        assert all(len(x) == n for x in xs)
        assert not use_embedding_cfg and not use_text_conditioning, "CFG / T5 plugins are out of scope (SURVEY 8a-19)"
        assert use_modulation or not use_time_conditioning, "use_time_conditioning requires use_modulation=True"
        self.use_time = use_time_conditioning
        out_channels = out_channels if out_channels is not None else in_channels
        cfg = dict(resnet_groups=resnet_groups, modulation_features=modulation_features,
                   attention_features=attention_features, attention_heads=attention_heads,
                   embedding_features=embedding_features, use_modulation=use_modulation)
        mf = modulation_features
        # TimeConditioningPlugin: NumberEmbedder(features=MF, dim=256) + num_layers x (Linear + GELU)
        if use_time_conditioning:
            half = TIME_EMBED_DIM // 2
            self.time_weights = nn.Parameter(torch.randn(half))
            self.time_linear = nn.Linear(TIME_EMBED_DIM + 1, mf)
            self.time_mlp = nn.ModuleList([nn.Linear(mf, mf) for _ in range(TIME_NUM_LAYERS)])
        blocks = []
        for d in range(n):
            in_ch = in_channels if d == 0 else channels[d - 1]
            out_ch = out_channels if d == 0 else in_ch
            its = item_list(items[d], use_modulation, context_channels[d], attentions[d], cross_attentions[d])
            blocks.append(Block(d, in_ch, out_ch, channels[d], factors[d], its, context_channels[d], cfg))
        self.blocks = nn.ModuleList(blocks)

    def time_features(self, time: Tensor) -> Tensor:
        t = time.reshape(-1, 1).to(torch.float32)
        freqs = t * self.time_weights[None, :] * 2 * pi
        four = torch.cat([t, freqs.sin(), freqs.cos()], dim=-1)  # [B, 257]
        f = self.time_linear(four)
        if TIME_GELU_AFTER_EMBEDDER:
            f = F.gelu(f)
        for lin in self.time_mlp:
            f = F.gelu(lin(f))
        return f

    def run_block(self, d: int, x: Tensor, features, embedding, channels) -> Tensor:
        if d == len(self.blocks):
            return x
        blk: Block = self.blocks[d]
        skip = blk.skip_adapter(x) if blk.skip_adapter is not None else x
        x = blk.down(x)
        x = blk.run_items(blk.items_down, x, features, embedding, channels)
        x = self.run_block(d + 1, x, features, embedding, channels)
        x = blk.run_items(blk.items_up, x, features, embedding, channels)
        x = blk.up(x)
        return blk.skip(skip, x, features)

    def forward(self, x: Tensor, time: Optional[Tensor] = None, *, features: Optional[Tensor] = None,
                embedding: Optional[Tensor] = None, channels=None) -> Tensor:
        if self.use_time:
            assert time is not None, "TimeConditioningPlugin requires time in forward"
            tf = self.time_features(time)
            features = features + tf if features is not None else tf
        else:
            assert time is None, "the bare XUNet takes x only (keyword conditioning)"
        return self.run_block(0, x, features, embedding, channels)


def rand_bool(shape, proba: float) -> Tensor:
    """a_unet rand_bool [recalled]: Bernoulli(proba) draw as a bool tensor (constant for proba 0 / 1)."""
    if proba == 1:
        return torch.ones(shape, dtype=torch.bool)
    if proba == 0:
        return torch.zeros(shape, dtype=torch.bool)
    return torch.bernoulli(torch.full(shape, float(proba))).to(torch.bool)


class ClassifierFreeGuidanceOracle(nn.Module):
    """Restatement of a_unet ClassifierFreeGuidancePlugin around a UNetV0Oracle (components.py:66-69; SURVEY
    Appendix A [recalled]): a learned positional FixedEmbedding stands in for the masked embedding; training masks
    whole batch elements with probability `embedding_mask_proba`; `embedding_scale != 1` evaluates the net twice and
    extrapolates  out_masked + (out - out_masked) * scale."""

    def __init__(self, net: nn.Module, embedding_max_length: int, embedding_features: int):
        super().__init__()
        self.net = net
        self.max_length = embedding_max_length
        self.fixed_embedding = nn.Embedding(embedding_max_length, embedding_features)

    def forward(self, x: Tensor, time: Optional[Tensor] = None, *, embedding: Optional[Tensor] = None,
                embedding_scale: float = 1.0, embedding_mask_proba: float = 0.0, batch_mask: Optional[Tensor] = None,
                **kwargs) -> Tensor:
        assert embedding is not None, "ClassiferFreeGuidancePlugin requires embedding"
        b, length = embedding.shape[0], embedding.shape[1]
        assert length <= self.max_length
        fixed = self.fixed_embedding(torch.arange(length))[None].expand(b, -1, -1)
        if embedding_mask_proba > 0.0:
            if batch_mask is None:  # `batch_mask` lets a test inject the draw
                batch_mask = rand_bool((b, 1, 1), embedding_mask_proba)
            embedding = torch.where(batch_mask.reshape(b, 1, 1), fixed, embedding)
        if embedding_scale != 1.0:
            out = self.net(x, time, embedding=embedding, **kwargs)
            out_masked = self.net(x, time, embedding=fixed, **kwargs)
            return out_masked + (out - out_masked) * embedding_scale
        return self.net(x, time, embedding=embedding, **kwargs)


def AppendChannelsOracle(net_t, channels: int):
    """Restatement of AppendChannelsPlugin (components.py:162-180)."""

    def Net(in_channels: int, out_channels: Optional[int] = None, **kwargs) -> nn.Module:
        out_channels = out_channels if out_channels is not None else in_channels
        net = net_t(in_channels=in_channels + channels, out_channels=out_channels, **kwargs)

        class _M(nn.Module):
            def __init__(self):
                super().__init__()
                self.net = net

            def forward(self, x, *args, append_channels, **kw):
                return self.net(torch.cat([x, append_channels], dim=1), *args, **kw)

        return _M()

    return Net
