"""`DiffusionModel` and `DiffusionUpsampler`, API-compatible with
/root/reference/audio_diffusion_pytorch/models.py:22-45 and :134-165 (kwargs-prefix routing via
utils.groupby; the same `net` object shared by diffusion and sampler), plus `DiffusionAE` (models.py:48-131; the
latent is injected at `inject_depth` through InjectChannelsItem, whose concat is read through two base pointers by
the 1x1 conv).  Vocoder / AR wrappers are out of the hot-path scope."""
from abc import ABC, abstractmethod
from typing import Any, Callable, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
from torch import Generator, Tensor

from .components import AppendChannelsPlugin
from .diffusion import VDiffusion, VSampler
from .utils import closest_power_2, default, downsample, exists, groupby, randn_like, upsample


class DiffusionModel(nn.Module):
    def __init__(self, net_t: Callable, diffusion_t: Callable = VDiffusion, sampler_t: Callable = VSampler,
                 loss_fn: Callable = torch.nn.functional.mse_loss, dim: int = 1, **kwargs):
        super().__init__()
        diffusion_kwargs, kwargs = groupby("diffusion_", kwargs)
        sampler_kwargs, kwargs = groupby("sampler_", kwargs)

        self.net = net_t(dim=dim, **kwargs)
        self.diffusion = diffusion_t(net=self.net, loss_fn=loss_fn, **diffusion_kwargs)
        self.sampler = sampler_t(net=self.net, **sampler_kwargs)

    def forward(self, *args, **kwargs) -> Tensor:
        return self.diffusion(*args, **kwargs)

    @torch.no_grad()
    def sample(self, *args, **kwargs) -> Tensor:
        return self.sampler(*args, **kwargs)


class EncoderBase(nn.Module, ABC):
    """Abstract class for DiffusionAE encoder (models.py:48-55): sets `out_channels` and `downsample_factor`."""

    @abstractmethod
    def __init__(self):
        super().__init__()
        self.out_channels = None
        self.downsample_factor = None


class AdapterBase(nn.Module, ABC):
    """Abstract class for DiffusionAE adapter (models.py:58-67)."""

    @abstractmethod
    def encode(self, x: Tensor) -> Tensor:
        pass

    @abstractmethod
    def decode(self, x: Tensor) -> Tensor:
        pass


class DiffusionAE(DiffusionModel):
    """Diffusion Auto Encoder (models.py:70-131).  The user-supplied `encoder` is an ordinary torch module; its
    latent enters the U-Net at `inject_depth` as `channels[inject_depth]` and receives its gradient from the U-Net
    backward (the encoder trains through the HIP kernels' data gradient of the inject conv)."""

    def __init__(self, in_channels: int, channels: Sequence[int], encoder: EncoderBase, inject_depth: int,
                 latent_factor: Optional[int] = None, adapter: Optional[AdapterBase] = None, **kwargs):
        context_channels = [0] * len(channels)
        context_channels[inject_depth] = encoder.out_channels
        super().__init__(in_channels=in_channels, channels=channels, context_channels=context_channels, **kwargs)
        self.in_channels = in_channels
        self.encoder = encoder
        self.inject_depth = inject_depth
        self.latent_factor = default(latent_factor, self.encoder.downsample_factor)
        self.adapter = adapter.requires_grad_(False) if exists(adapter) else None

    def forward(self, x: Tensor, with_info: bool = False, **kwargs) -> Union[Tensor, Tuple[Tensor, Any]]:
        latent, info = self.encode(x, with_info=True)
        channels = [None] * self.inject_depth + [latent]
        x = self.adapter.encode(x) if exists(self.adapter) else x
        loss = super().forward(x, channels=channels, **kwargs)
        return (loss, info) if with_info else loss

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    @torch.no_grad()
    def decode(self, latent: Tensor, generator: Optional[Generator] = None, **kwargs) -> Tensor:
        b = latent.shape[0]
        noise_length = closest_power_2(latent.shape[2] * self.latent_factor)
        noise = torch.randn((b, self.in_channels, noise_length), device=latent.device, dtype=latent.dtype,
                            generator=generator)
        channels = [None] * self.inject_depth + [latent]
        out = super().sample(noise, channels=channels, **kwargs)
        return self.adapter.decode(out) if exists(self.adapter) else out


class DiffusionUpsampler(DiffusionModel):
    def __init__(self, in_channels: int, upsample_factor: int, net_t: Callable, **kwargs):
        self.upsample_factor = upsample_factor
        super().__init__(net_t=AppendChannelsPlugin(net_t, channels=in_channels), in_channels=in_channels, **kwargs)

    def reupsample(self, x: Tensor) -> Tensor:
        x = x.clone()
        x = downsample(x, factor=self.upsample_factor)
        x = upsample(x, factor=self.upsample_factor)
        return x

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        reupsampled = self.reupsample(x)
        return super().forward(x, *args, append_channels=reupsampled, **kwargs)

    @torch.no_grad()
    def sample(self, downsampled: Tensor, generator: Optional[Generator] = None, **kwargs) -> Tensor:
        reupsampled = upsample(downsampled, factor=self.upsample_factor)
        noise = randn_like(reupsampled, generator=generator)
        return super().sample(noise, append_channels=reupsampled, **kwargs)
