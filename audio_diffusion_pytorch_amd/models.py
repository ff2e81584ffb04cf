"""`DiffusionModel` and `DiffusionUpsampler`, API-compatible with
/root/reference/audio_diffusion_pytorch/models.py:22-45 and :134-165 (kwargs-prefix routing via
utils.groupby; the same `net` object shared by diffusion and sampler).  DiffusionAE / Vocoder / AR wrappers are
out of the hot-path scope."""
from typing import Callable, Optional

import torch
import torch.nn as nn
from torch import Generator, Tensor

from .components import AppendChannelsPlugin
from .diffusion import VDiffusion, VSampler
from .utils import downsample, groupby, randn_like, upsample


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
