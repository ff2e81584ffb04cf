"""Model wrappers of the denoising hot path: `DiffusionModel`, `DiffusionUpsampler`, `DiffusionAE`.

Public contract = /root/reference/audio_diffusion_pytorch/models.py:22-45 (DiffusionModel), :48-131 (DiffusionAE and its
encoder / adapter interfaces) and :134-165 (DiffusionUpsampler): same constructor arguments, `forward` -> loss,
`sample` / `decode` -> tensor under no_grad, `diffusion_*` / `sampler_*` kwarg prefixes routed to the diffusion and
the sampler, everything else to `net_t`, ONE net shared by both.

What is organised around the gfx950 kernels rather than around torch ops:
  * the starting noise of `sample` / `decode` is drawn on the host like the reference's (utils.py:123-125), so seeded
    scripts reproduce its samples; one 8 MB copy per sampling run;
  * `reupsample` is two launches of the polyphase resampler (adp_resample) on the caller's tensor -- the kernel
    never writes its input, so no defensive clone;
  * the appended / injected conditioning tensors are consumed through a second input pointer of the depth-0 /
    inject convs (components.AppendChannelsPlugin, unet inject items): no concatenated copy exists.
Vocoder / AR wrappers are outside the hot-path scope.
"""
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
from torch import Generator, Tensor

from .components import AppendChannelsPlugin
from .diffusion import VDiffusion, VSampler
from .utils import closest_power_2, downsample, upsample


def _split_prefixed(kwargs: Dict[str, Any], *prefixes: str):
    """({prefix: {stripped name: value}}, the rest) -- the reference's chained utils.groupby calls in one pass."""
    routed: Dict[str, Dict[str, Any]] = {p: {} for p in prefixes}
    rest: Dict[str, Any] = {}
    for name, value in kwargs.items():
        for p in prefixes:
            if name.startswith(p):
                routed[p][name[len(p):]] = value
                break
        else:
            rest[name] = value
    return routed, rest


def _start_noise(shape: Sequence[int], like: Tensor, generator: Optional[Generator]) -> Tensor:
    """N(0, 1) starting point of DiffusionUpsampler.sample.  Drawn exactly where the reference draws it (utils.randn_like,
    utils.py:123-125 <- models.py:163): on the HOST -- from `generator`, or from torch's global CPU generator when none is
    given -- then moved to `like`'s device in one copy, so that `torch.manual_seed(s); model.sample(...)` scripts produce the
    reference's starting noise on any backend.  (A generator that lives on the device is honoured in place.)
    DiffusionAE.decode does NOT come through here: the reference draws that noise on the latent's device (models.py:120-125)."""
    if generator is not None and generator.device.type != "cpu":
        return torch.randn(tuple(shape), generator=generator, dtype=like.dtype, device=generator.device).to(like.device)
    return torch.randn(tuple(shape), generator=generator, dtype=like.dtype).to(like.device)


class DiffusionModel(nn.Module):
    """net + diffusion (training objective) + sampler over the same net (models.py:22-45)."""

    def __init__(self, net_t: Callable, diffusion_t: Callable = VDiffusion, sampler_t: Callable = VSampler,
                 loss_fn: Callable = torch.nn.functional.mse_loss, dim: int = 1, **kwargs):
        super().__init__()
        routed, net_kwargs = _split_prefixed(kwargs, "diffusion_", "sampler_")
        self.net = net_t(dim=dim, **net_kwargs)
        self.diffusion = diffusion_t(net=self.net, loss_fn=loss_fn, **routed["diffusion_"])
        self.sampler = sampler_t(net=self.net, **routed["sampler_"])

    def forward(self, *args, **kwargs) -> Tensor:
        """Training loss of one batch (VDiffusion: fused noising kernel -> U-Net -> fused MSE)."""
        return self.diffusion(*args, **kwargs)

    @torch.no_grad()
    def sample(self, *args, **kwargs) -> Tensor:
        """Runs the sampler from the given noise (VSampler: hipGraph-replayed steps)."""
        return self.sampler(*args, **kwargs)


class EncoderBase(nn.Module, ABC):
    """DiffusionAE encoder interface (models.py:48-55): subclasses set `out_channels` and `downsample_factor`."""

    @abstractmethod
    def __init__(self):
        super().__init__()
        self.out_channels = None
        self.downsample_factor = None


class AdapterBase(nn.Module, ABC):
    """DiffusionAE adapter interface (models.py:58-67): maps the waveform to / from the domain the U-Net works in."""

    @abstractmethod
    def encode(self, x: Tensor) -> Tensor:
        pass

    @abstractmethod
    def decode(self, x: Tensor) -> Tensor:
        pass


class DiffusionAE(DiffusionModel):
    """Diffusion auto-encoder (models.py:70-131).  `encoder` is the caller's torch module; its latent reaches the
    U-Net as `channels[inject_depth]`, where the InjectChannelsItem's 1x1 conv reads [x | latent] through two input
    pointers, and the latent's gradient comes back from that conv's HIP data-gradient kernel into the encoder."""

    def __init__(self, in_channels: int, channels: Sequence[int], encoder: EncoderBase, inject_depth: int,
                 latent_factor: Optional[int] = None, adapter: Optional[AdapterBase] = None, **kwargs):
        per_depth_context = [encoder.out_channels if d == inject_depth else 0 for d in range(len(channels))]
        super().__init__(in_channels=in_channels, channels=channels, context_channels=per_depth_context, **kwargs)
        self.in_channels = in_channels
        self.inject_depth = inject_depth
        self.encoder = encoder
        self.latent_factor = latent_factor if latent_factor is not None else encoder.downsample_factor
        self.adapter = None if adapter is None else adapter.requires_grad_(False)

    def _context(self, latent: Tensor):
        """`channels` list for the U-Net: placeholders down to the injection depth, then the latent."""
        return [None] * self.inject_depth + [latent]

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    def forward(self, x: Tensor, with_info: bool = False, **kwargs) -> Union[Tensor, Tuple[Tensor, Any]]:
        latent, info = self.encode(x, with_info=True)
        target = x if self.adapter is None else self.adapter.encode(x)
        loss = super().forward(target, channels=self._context(latent), **kwargs)
        return (loss, info) if with_info else loss

    @torch.no_grad()
    def decode(self, latent: Tensor, generator: Optional[Generator] = None, **kwargs) -> Tensor:
        length = closest_power_2(latent.shape[2] * self.latent_factor)
        # drawn like the reference's decode (models.py:117-125): on the latent's device, from `generator` or that device's
        # global generator -- a seeded decode() script sees the reference's starting noise
        noise = torch.randn((latent.shape[0], self.in_channels, length), device=latent.device, dtype=latent.dtype,
                            generator=generator)
        out = super().sample(noise, channels=self._context(latent), **kwargs)
        return out if self.adapter is None else self.adapter.decode(out)


class DiffusionUpsampler(DiffusionModel):
    """Diffusion upsampler (models.py:134-165): the U-Net denoises the full-rate signal conditioned on the
    low-rate one re-upsampled to full rate, appended as extra input channels."""

    def __init__(self, in_channels: int, upsample_factor: int, net_t: Callable, **kwargs):
        self.upsample_factor = upsample_factor
        super().__init__(net_t=AppendChannelsPlugin(net_t, channels=in_channels), in_channels=in_channels, **kwargs)

    def reupsample(self, x: Tensor) -> Tensor:
        """down x f then up x f with the windowed-sinc polyphase kernel (utils.py:82-117 semantics); `x` is only read."""
        f = self.upsample_factor
        return upsample(downsample(x, factor=f), factor=f)

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        return super().forward(x, *args, append_channels=self.reupsample(x), **kwargs)

    @torch.no_grad()
    def sample(self, downsampled: Tensor, generator: Optional[Generator] = None, **kwargs) -> Tensor:
        conditioning = upsample(downsampled, factor=self.upsample_factor)
        noise = _start_noise(conditioning.shape, conditioning, generator)
        return super().sample(noise, append_channels=conditioning, **kwargs)
