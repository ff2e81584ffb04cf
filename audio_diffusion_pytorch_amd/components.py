"""`UNetV0` factory and the `AppendChannelsPlugin`, API-compatible with
/root/reference/audio_diffusion_pytorch/components.py:34-105 and :162-180.

The reference composes a_unet module templates; here the same constructor arguments configure ONE
kernel-backed module (unet.UNetV0Net); use_embedding_cfg wraps it in the classifier-free-guidance module below
(the guided and the masked evaluation run as ONE batched U-Net call).  Out of scope:
use_text_conditioning (T5 download), use_modulation=False (SkipCat), LTPlugin, MelSpectrogram -- each raises a
clear error instead of silently degrading.
"""
from typing import Callable, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .unet import UNetV0Net
from .utils import default, exists


def rand_bool(shape, proba: float, device=None) -> Tensor:
    """Bernoulli(proba) draw as a bool tensor on torch's generator (constant for proba 0 / 1), as a_unet's helper."""
    if proba == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    if proba == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.bernoulli(torch.full(shape, float(proba), device=device)).to(torch.bool)


class _SelectRows(torch.autograd.Function):
    """out[b] = pick[b] ? a[b] : other[b] (adp_select_rows) with its gradient routed the same way."""

    @staticmethod
    def forward(ctx, a: Tensor, other: Tensor, pick_u8: Tensor) -> Tensor:
        ctx.save_for_backward(pick_u8)
        return ops.select_rows(a.contiguous(), other.contiguous(), pick_u8)

    @staticmethod
    def backward(ctx, g: Tensor):
        (pick_u8,) = ctx.saved_tensors
        g = g.contiguous()
        zero = torch.zeros_like(g)
        return ops.select_rows(g, zero, pick_u8), ops.select_rows(zero, g, pick_u8), None


class ClassifierFreeGuidanceNet(nn.Module):
    """UNetV0(use_embedding_cfg=True, embedding_max_length=N): a_unet's ClassifierFreeGuidancePlugin
    (/root/reference/audio_diffusion_pytorch/components.py:66-69; forward kwargs `embedding_scale`,
    `embedding_mask_proba`, README.md:62-76).  MI355X-first differences: with embedding_scale != 1 the
    guided and the masked evaluation are ONE U-Net call on a doubled batch [x | x] with embeddings
    [embedding | fixed], and the extrapolation out_masked + (out - out_masked) * scale is one kernel (adp_cfg_mix)
    -- every weight is streamed once per sampling step instead of twice."""

    def __init__(self, net: UNetV0Net, embedding_max_length: int, embedding_features: int):
        super().__init__()
        self.net = net
        self.max_length = embedding_max_length
        self.fixed_embedding = nn.Embedding(embedding_max_length, embedding_features)

    def forward(self, x: Tensor, time: Optional[Tensor] = None, *, embedding: Optional[Tensor] = None,
                embedding_scale: float = 1.0, embedding_mask_proba: float = 0.0, batch_mask: Optional[Tensor] = None,
                features: Optional[Tensor] = None, channels=None, **kwargs) -> Tensor:
        assert exists(embedding), "ClassiferFreeGuidancePlugin requires embedding"
        b, length = embedding.shape[0], embedding.shape[1]
        assert length <= self.max_length, "embedding longer than embedding_max_length"
        fixed = self.fixed_embedding.weight[:length][None].expand(b, -1, -1)  # positional table, one row set per sample
        if embedding_mask_proba > 0.0:
            if batch_mask is None:  # `batch_mask` lets a harness inject the draw
                batch_mask = rand_bool((b, 1, 1), embedding_mask_proba, device=embedding.device)
            pick = batch_mask.reshape(b).to(device=embedding.device, dtype=torch.uint8).contiguous()
            embedding = _SelectRows.apply(fixed, embedding, pick)
        if embedding_scale != 1.0:
            two = lambda t: None if t is None else torch.cat([t, t], dim=0)  # noqa: E731  (memory movement only)
            emb2 = torch.cat([embedding, fixed], dim=0)
            ch2 = None if channels is None else [two(c) for c in channels]
            y2 = self.net(two(x), two(time), embedding=emb2, features=two(features), channels=ch2, **kwargs)
            return ops.cfg_mix(y2.contiguous(), float(embedding_scale))
        return self.net(x, time, embedding=embedding, features=features, channels=channels, **kwargs)


def UNetV0(
    dim: int,
    in_channels: int,
    channels: Sequence[int],
    factors: Sequence[int],
    items: Sequence[int],
    attentions: Optional[Sequence[int]] = None,
    cross_attentions: Optional[Sequence[int]] = None,
    context_channels: Optional[Sequence[int]] = None,
    attention_features: Optional[int] = None,
    attention_heads: Optional[int] = None,
    embedding_features: Optional[int] = None,
    resnet_groups: int = 8,
    use_modulation: bool = True,
    modulation_features: int = 1024,
    embedding_max_length: Optional[int] = None,
    use_time_conditioning: bool = True,
    use_embedding_cfg: bool = False,
    use_text_conditioning: bool = False,
    out_channels: Optional[int] = None,
) -> nn.Module:
    num_layers = len(channels)
    attentions = default(attentions, [0] * num_layers)
    cross_attentions = default(cross_attentions, [0] * num_layers)
    context_channels = default(context_channels, [0] * num_layers)
    xs = (channels, factors, items, attentions, cross_attentions, context_channels)
    assert all(len(x) == num_layers for x in xs)

    if use_embedding_cfg:
        assert exists(embedding_max_length), "use_embedding_cfg requires embedding_max_length"
        assert exists(embedding_features), "use_embedding_cfg requires embedding_features"
    if use_text_conditioning:
        raise NotImplementedError("TextConditioningPlugin needs the t5-base download; pass `embedding=` directly")
    assert use_time_conditioning, "UNetV0 on MI355X is built with TimeConditioningPlugin (the reference default)"
    assert use_modulation, "use_time_conditioning requires use_modulation=True"

    net = UNetV0Net(
        dim=dim, in_channels=in_channels, channels=channels, factors=factors, items=items, attentions=attentions,
        cross_attentions=cross_attentions, context_channels=context_channels,
        attention_features=attention_features, attention_heads=attention_heads,
        embedding_features=embedding_features, resnet_groups=resnet_groups,
        modulation_features=modulation_features, out_channels=out_channels)
    if use_embedding_cfg:
        return ClassifierFreeGuidanceNet(net, embedding_max_length, embedding_features)
    return net


class _AppendChannelsNet(nn.Module):
    """Module returned by AppendChannelsPlugin(net_t, channels)(...): the concat is not materialised --
    the depth-0 convs read the two tensors through two base pointers (adp_conv_desc.x2)."""

    def __init__(self, net: nn.Module):
        super().__init__()
        self.net = net

    def forward(self, x: Tensor, *args, append_channels: Tensor, **kwargs) -> Tensor:
        return self.net(x, *args, x_append=append_channels, **kwargs)


def AppendChannelsPlugin(net_t: Callable, channels: int):
    def Net(in_channels: int, out_channels: Optional[int] = None, **kwargs) -> nn.Module:
        out_channels = default(out_channels, in_channels)
        net = net_t(in_channels=in_channels + channels, out_channels=out_channels, **kwargs)
        assert isinstance(net, (UNetV0Net, ClassifierFreeGuidanceNet)), "AppendChannelsPlugin wraps UNetV0 on this backend"
        return _AppendChannelsNet(net)

    return Net
