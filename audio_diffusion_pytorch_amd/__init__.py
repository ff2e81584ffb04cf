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


def _out_of_scope(name: str, why: str):
    """Reference exports outside the denoising hot path (SURVEY section 8 / DESIGN section 7): importable CLASSES (the
    reference exports classes: isinstance / issubclass / subclassing keep working), whose construction says what they
    are instead of failing with an ImportError at the call site."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{name} is not part of the MI355X-native hot path ({why}); see DESIGN.md section 7")
    return type(name, (), {"__init__": __init__, "__doc__": f"Out of scope: {why}."})


DiffusionVocoder = _out_of_scope("DiffusionVocoder", "needs torchaudio's STFT / mel filterbank")
MelSpectrogram = _out_of_scope("MelSpectrogram", "needs torchaudio's STFT / mel filterbank")
DiffusionAR = _out_of_scope("DiffusionAR", "autoregressive ARVDiffusion / ARVSampler use a different net signature")


def LTPlugin(*args, **kwargs):
    """Reference: a factory (components.py:113-159), so a plain callable here too."""
    raise NotImplementedError("LTPlugin is not part of the MI355X-native hot path (learned-transform front end, not on "
                              "the UNetV0 denoising path); see DESIGN.md section 7")


__all__ = [
    "AppendChannelsPlugin", "UNetV0", "XUNet", "UNetV0Net", "Diffusion", "Distribution", "LinearSchedule", "Sampler",
    "Schedule", "UniformDistribution", "VDiffusion", "VInpainter", "VSampler", "DiffusionModel", "DiffusionUpsampler",
    "DiffusionAE", "EncoderBase", "AdapterBase", "ClassifierFreeGuidanceNet", "DiffusionVocoder", "MelSpectrogram",
    "DiffusionAR", "LTPlugin",
]
