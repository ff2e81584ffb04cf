"""`UNetV0` factory and the `AppendChannelsPlugin`, API-compatible with
/root/reference/audio_diffusion_pytorch/components.py:34-105 and :162-180.

The reference composes a_unet module templates; here the same constructor arguments configure ONE
kernel-backed module (unet.UNetV0Net); use_embedding_cfg wraps it in the classifier-free-guidance module below
(the guided and the masked evaluation run as ONE batched U-Net call); use_text_conditioning wraps it in the
text-conditioning module (the embedder is the caller's module; the t5-base default needs local weights);
use_modulation=False selects SkipCat; use_time_conditioning=False leaves `features` to the caller.  Out of scope
(SURVEY.md section 2): LTPlugin, MelSpectrogram.
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
            if kwargs.get("x_append") is not None:  # AppendChannelsPlugin's second input pointer follows the batch
                kwargs = dict(kwargs, x_append=two(kwargs["x_append"]))
            y2 = self.net(two(x), two(time), embedding=emb2, features=two(features), channels=ch2, **kwargs)
            return ops.cfg_mix(y2.contiguous(), float(embedding_scale))
        return self.net(x, time, embedding=embedding, features=features, channels=channels, **kwargs)


class T5Embedder(nn.Module):
    """a_unet's T5Embedder as TextConditioningPlugin instantiates it (t5-base encoder, max_length tokens, frozen;
    components.py:70-72).  The encoder is the `transformers` model on stock PyTorch: text encoding is outside the
    denoising hot path.  Needs the weights in the local HuggingFace cache (no download here)."""

    def __init__(self, model: str = "t5-base", max_length: int = 64):
        super().__init__()
        from transformers import AutoTokenizer, T5EncoderModel
        try:
            self.tokenizer = AutoTokenizer.from_pretrained(model, local_files_only=True)
            self.transformer = T5EncoderModel.from_pretrained(model, local_files_only=True)
        except Exception as e:  # no network on the build / GPU boxes
            raise NotImplementedError(
                f"TextConditioningPlugin's default T5Embedder needs the '{model}' weights in the local HuggingFace "
                "cache; pass UNetV0(..., use_text_conditioning=True, text_embedder=<module: List[str] -> [B, m, E]>) "
                "or feed `embedding=` directly") from e
        self.max_length = max_length

    @torch.no_grad()
    def forward(self, texts) -> Tensor:
        enc = self.tokenizer(texts, truncation=True, max_length=self.max_length, padding="max_length",
                             return_tensors="pt")
        device = next(self.transformer.parameters()).device
        self.transformer.eval()
        return self.transformer(input_ids=enc["input_ids"].to(device),
                                attention_mask=enc["attention_mask"].to(device))["last_hidden_state"]


class TextConditioningNet(nn.Module):
    """a_unet TextConditioningPlugin (components.py:70-72): forward kwarg `text` (list of strings) is embedded and
    passed on as `embedding` (concatenated in front of a caller-supplied `embedding` along the token axis)."""

    def __init__(self, net: nn.Module, embedder: Optional[nn.Module] = None):
        super().__init__()
        self.net = net
        self.embedder = embedder if exists(embedder) else T5Embedder()

    def _embed(self, x: Tensor, text, embedding: Optional[Tensor]) -> Tensor:
        assert exists(text), "TextConditioningPlugin requires `text` in forward"
        text_embedding = self.embedder(text).to(device=x.device, dtype=torch.float32)
        if exists(embedding):
            text_embedding = torch.cat([text_embedding, embedding], dim=1)  # token axis; memory movement only
        return text_embedding

    def forward(self, x: Tensor, *args, text=None, embedding: Optional[Tensor] = None,
                resolved_text_embedding: Optional[Tensor] = None, **kwargs) -> Tensor:
        if resolved_text_embedding is not None:  # a sampling loop embedded the text once (prepare_sampling_kwargs)
            return self.net(x, *args, embedding=resolved_text_embedding, **kwargs)
        return self.net(x, *args, embedding=self._embed(x, text, embedding), **kwargs)

    def prepare_sampling_kwargs(self, x: Tensor, kwargs: dict) -> dict:
        """Called once by VSampler / VInpainter before their step loop: the text is embedded ONCE per sampling run
        (the reference's plugin re-encodes it on each of the N steps) and reaches the steps as a plain device tensor,
        so the step has no host-side tokenizer / H2D copy in it and can be captured in a hipGraph."""
        if not exists(kwargs.get("text")):
            return kwargs
        rest = {k: v for k, v in kwargs.items() if k not in ("text", "embedding")}
        rest["resolved_text_embedding"] = self._embed(x, kwargs["text"], kwargs.get("embedding")).contiguous()
        return rest


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
    text_embedder: Optional[nn.Module] = None,
) -> nn.Module:
    """Same arguments as the reference factory (components.py:34-54) plus `text_embedder` (the module
    TextConditioningPlugin would otherwise build from t5-base).  Plugin nesting order as at components.py:66-76:
    Time(Text(CFG(XUNet))) -- time conditioning lives inside the kernel-backed net."""
    num_layers = len(channels)
    attentions = default(attentions, [0] * num_layers)
    cross_attentions = default(cross_attentions, [0] * num_layers)
    context_channels = default(context_channels, [0] * num_layers)
    xs = (channels, factors, items, attentions, cross_attentions, context_channels)
    assert all(len(x) == num_layers for x in xs)

    if use_embedding_cfg:
        msg = "use_embedding_cfg requires embedding_max_length"
        assert exists(embedding_max_length), msg
        assert exists(embedding_features), "use_embedding_cfg requires embedding_features"
    if use_time_conditioning:
        assert use_modulation, "use_time_conditioning requires use_modulation=True"

    net: nn.Module = UNetV0Net(
        dim=dim, in_channels=in_channels, channels=channels, factors=factors, items=items, attentions=attentions,
        cross_attentions=cross_attentions, context_channels=context_channels,
        attention_features=attention_features, attention_heads=attention_heads,
        embedding_features=embedding_features, resnet_groups=resnet_groups,
        modulation_features=modulation_features, out_channels=out_channels, use_modulation=use_modulation,
        use_time_conditioning=use_time_conditioning)
    if use_embedding_cfg:
        net = ClassifierFreeGuidanceNet(net, embedding_max_length, embedding_features)
    if use_text_conditioning:
        net = TextConditioningNet(net, text_embedder)
    return net


def load_reference_state_dict(net: nn.Module, sd) -> dict:
    """EXPERIMENTAL like UNetV0Net.load_a_unet_state_dict, which it extends to whatever `UNetV0(...)` /
    `AppendChannelsPlugin(UNetV0, n)(...)` returned: a checkpoint of the reference's net incl. its plugins.  The reference
    nests Time(Text(CFG(XUNet))) (components.py:66-76) and every plugin is a `Module([own modules..., net], fn)`, so its
    floating-point tensors come as [time embedder + MLP | T5 encoder (use_text_conditioning) | fixed_embedding
    (use_embedding_cfg) | XUNet blocks]: the time tensors and the block tensors go through the positional loader, the
    fixed-embedding table (identified by its `[embedding_max_length, embedding_features]` shape right in front of the
    blocks) into ClassifierFreeGuidanceNet.fixed_embedding, and the frozen pretrained T5 tensors are NOT loaded (they are
    t5-base's own weights; the embedder here is the caller's module or the local t5-base).  Returns {checkpoint key: what it
    was loaded as / "skipped: ..."}."""
    cfg_net, has_text, inner = None, False, net
    while not isinstance(inner, UNetV0Net):
        if isinstance(inner, ClassifierFreeGuidanceNet):
            cfg_net = inner
        elif isinstance(inner, TextConditioningNet):
            has_text = True
        elif not isinstance(inner, _AppendChannelsNet):
            raise TypeError(f"load_reference_state_dict: {type(inner).__name__} does not wrap a UNetV0")
        inner = inner.net
    theirs = [(k, v) for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point]
    n_unet = len(inner.a_unet_key_order())
    n_time = sum(1 for k in inner.a_unet_key_order() if k.startswith("time_"))
    n_extra = len(theirs) - n_unet
    if n_extra < (1 if cfg_net is not None else 0) or (n_extra > (1 if cfg_net is not None else 0) and not has_text):
        raise ValueError(f"checkpoint holds {len(theirs)} floating-point tensors; this net expects {n_unet} for the U-Net"
                         + (" + the fixed-embedding table" if cfg_net is not None else "")
                         + (" + the text encoder's" if has_text else ""))
    extras = theirs[n_time:n_time + n_extra]
    out = {}
    if cfg_net is not None:
        k, v = extras[-1]
        if tuple(v.shape) != tuple(cfg_net.fixed_embedding.weight.shape):
            raise ValueError(f"checkpoint entry {k!r} has shape {tuple(v.shape)} where the fixed-embedding table "
                             f"{tuple(cfg_net.fixed_embedding.weight.shape)} is expected (right in front of the blocks)")
        with torch.no_grad():
            cfg_net.fixed_embedding.weight.copy_(v)
        out[k] = "fixed_embedding.weight"
        extras = extras[:-1]
    for k, _ in extras:
        out[k] = "skipped: frozen pretrained text-encoder tensor (TextConditioningPlugin's T5)"
    core = dict(theirs[:n_time] + theirs[n_time + n_extra:])
    out.update(inner.load_a_unet_state_dict(core))
    return out


class _ConcatChannels(torch.autograd.Function):
    """cat([x, extra], dim=1) on adp_copy2d, with the split as its gradient."""

    @staticmethod
    def forward(ctx, x: Tensor, extra: Tensor) -> Tensor:
        ctx.c1 = x.shape[1]
        return ops.concat_channels(x.contiguous(), extra.contiguous())

    @staticmethod
    def backward(ctx, g: Tensor):
        ga, gb = ops.split_channels(g.contiguous(), ctx.c1)
        return ga, gb


def _accepts_x_append(net: nn.Module) -> bool:
    """True when the wrapped net ends in a UNetV0Net, whose depth-0 convs can read a second input pointer."""
    while isinstance(net, (ClassifierFreeGuidanceNet, TextConditioningNet)):
        net = net.net
    return isinstance(net, UNetV0Net)


class _AppendChannelsNet(nn.Module):
    """Module returned by AppendChannelsPlugin(net_t, channels)(...).  Around a UNetV0 the concat is not
    materialised -- the depth-0 convs read the two tensors through two base pointers (adp_conv_desc.x2); around any
    other net it is one strided-copy kernel pair (adp_copy2d) and the net sees the [B, C + channels, L] tensor."""

    def __init__(self, net: nn.Module):
        super().__init__()
        self.net = net
        self.two_pointer = _accepts_x_append(net)

    def prepare_sampling_kwargs(self, x: Tensor, kwargs: dict) -> dict:
        inner = getattr(self.net, "prepare_sampling_kwargs", None)
        return inner(x, kwargs) if inner is not None else kwargs

    def forward(self, x: Tensor, *args, append_channels: Tensor, **kwargs) -> Tensor:
        if self.two_pointer:
            return self.net(x, *args, x_append=append_channels, **kwargs)
        return self.net(_ConcatChannels.apply(x, append_channels), *args, **kwargs)


def AppendChannelsPlugin(net_t: Callable, channels: int):
    def Net(in_channels: int, out_channels: Optional[int] = None, **kwargs) -> nn.Module:
        out_channels = default(out_channels, in_channels)
        return _AppendChannelsNet(net_t(in_channels=in_channels + channels, out_channels=out_channels, **kwargs))

    return Net
