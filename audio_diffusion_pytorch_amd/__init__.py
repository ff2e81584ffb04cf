"""MI355X-native (gfx950, hand-written HIP) implementation of the audio-diffusion-pytorch denoising hot path.
Exports mirror /root/reference/audio_diffusion_pytorch/__init__.py:1-20 for the in-scope symbols."""
from .components import AppendChannelsPlugin, ClassifierFreeGuidanceNet, UNetV0
from .diffusion import (
    Diffusion,
    Distribution,
    LinearSchedule,
    Sampler,
    Schedule,
    UniformDistribution,
    VDiffusion,
    VInpainter,
    VSampler,
)
from .models import AdapterBase, DiffusionAE, DiffusionModel, DiffusionUpsampler, EncoderBase
from .unet import UNetV0Net

XUNet = UNetV0Net

__all__ = [
    "AppendChannelsPlugin", "UNetV0", "XUNet", "UNetV0Net", "Diffusion", "Distribution", "LinearSchedule", "Sampler",
    "Schedule", "UniformDistribution", "VDiffusion", "VInpainter", "VSampler", "DiffusionModel", "DiffusionUpsampler",
    "DiffusionAE", "EncoderBase", "AdapterBase", "ClassifierFreeGuidanceNet",
]
