"""
ORACLE (test infrastructure, NOT product code) -- CPU restatement of the v-objective math.

Follows /root/reference/audio_diffusion_pytorch/diffusion.py:
  * get_alpha_beta         :77-80   angle = sigma * pi / 2 ; alpha = cos, beta = sin (fp32)
  * VDiffusion.forward     :82-95   x_noisy = a x + b n ; v_target = a n - b x ; loss(net(x_noisy, sigma), v_target)
  * LinearSchedule.forward :147-148 linspace(start, end, num_steps)
  * VSampler.forward       :172-190 per step: x_pred = a_i x - b_i v ; n_pred = b_i x + a_i v ;
                                    x = a_{i+1} x_pred + b_{i+1} n_pred
  * VInpainter.forward     :321-354 per (step i, resample r): rotate to level j = i + (r == last); re-noise the
                                    source to level j (torch.randn_like(source)); x = s_noisy*mask + x*~mask
and /root/reference/audio_diffusion_pytorch/utils.py:82-117 (windowed-sinc `resample`).

PINNED: tests/test_oracle.py checks this restatement against the live reference module
(when /root/reference is present) and against the committed fixtures in tests/golden/
generated from the live reference by tests/golden/make_golden.py.
"""
from math import ceil, pi
from typing import Callable, Optional

import torch
import torch.nn.functional as F
from torch import Tensor


def alpha_beta(sigmas: Tensor):
    angle = sigmas * pi / 2
    return torch.cos(angle), torch.sin(angle)


def v_noise(x: Tensor, noise: Tensor, sigmas: Tensor):
    """Returns (x_noisy, v_target) for per-sample sigmas [B]."""
    s = sigmas.view(-1, *([1] * (x.ndim - 1)))
    a, b = alpha_beta(s)
    return a * x + b * noise, a * noise - b * x


def v_loss(net: Callable, x: Tensor, noise: Tensor, sigmas: Tensor, loss_fn=F.mse_loss, **kw) -> Tensor:
    x_noisy, v_target = v_noise(x, noise, sigmas)
    return loss_fn(net(x_noisy, sigmas, **kw), v_target)


def linear_schedule(num_steps: int, start: float = 1.0, end: float = 0.0, device=None) -> Tensor:
    return torch.linspace(start, end, num_steps, device=device)


@torch.no_grad()
def v_sample(net: Callable, x_noisy: Tensor, num_steps: int, sigmas: Optional[Tensor] = None, **kw) -> Tensor:
    b = x_noisy.shape[0]
    if sigmas is None:
        sigmas = linear_schedule(num_steps + 1, device=x_noisy.device)
    sig = sigmas[:, None].expand(num_steps + 1, b)
    a, bt = alpha_beta(sig.reshape(num_steps + 1, b, *([1] * (x_noisy.ndim - 1))))
    for i in range(num_steps):
        v = net(x_noisy, sig[i], **kw)
        x_pred = a[i] * x_noisy - bt[i] * v
        n_pred = bt[i] * x_noisy + a[i] * v
        x_noisy = a[i + 1] * x_pred + bt[i + 1] * n_pred
    return x_noisy


@torch.no_grad()
def v_inpaint(net: Callable, source: Tensor, mask: Tensor, num_steps: int, num_resamples: int,
              x_noisy: Optional[Tensor] = None, **kw) -> Tensor:
    """VInpainter.forward (diffusion.py:321-354) with the same torch.randn_like call order."""
    x_noisy = x_noisy if x_noisy is not None else torch.randn_like(source)
    b = x_noisy.shape[0]
    sigmas = linear_schedule(num_steps + 1, device=x_noisy.device)
    sig = sigmas[:, None].expand(num_steps + 1, b)
    a, bt = alpha_beta(sig.reshape(num_steps + 1, b, *([1] * (x_noisy.ndim - 1))))
    for i in range(num_steps):
        for r in range(num_resamples):
            v = net(x_noisy, sig[i], **kw)
            x_pred = a[i] * x_noisy - bt[i] * v
            n_pred = bt[i] * x_noisy + a[i] * v
            j = int(r == num_resamples - 1)
            x_noisy = a[i + j] * x_pred + bt[i + j] * n_pred
            s_noisy = a[i + j] * source + bt[i + j] * torch.randn_like(source)
            x_noisy = s_noisy * mask + x_noisy * ~mask
    return x_noisy


def resample(waveforms: Tensor, factor_in: int, factor_out: int, rolloff: float = 0.99,
             lowpass_filter_width: int = 6) -> Tensor:
    """Windowed-sinc polyphase resampler, one conv1d (utils.py:82-109)."""
    b, c, length = waveforms.shape
    target = int(factor_out * length / factor_in)
    kw = dict(device=waveforms.device, dtype=waveforms.dtype)
    base = min(factor_in, factor_out) * rolloff
    width = ceil(lowpass_filter_width * factor_in / base)
    idx = torch.arange(-width, width + factor_in, **kw)[None, None] / factor_in
    t = torch.arange(0, -factor_out, step=-1, **kw)[:, None, None] / factor_out + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width) * pi
    window = torch.cos(t / lowpass_filter_width / 2) ** 2
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t)
    kernels = kernels * (window * (base / factor_in))
    w = F.pad(waveforms.reshape(b * c, length), (width, width + factor_in))
    out = F.conv1d(w[:, None], kernels, stride=factor_in)  # [(b c), k, l]
    out = out.reshape(b, c, factor_out, -1).permute(0, 1, 3, 2).reshape(b, c, -1)
    return out[..., :target]


def downsample(w: Tensor, factor: int) -> Tensor:
    return resample(w, factor_in=factor, factor_out=1)


def upsample(w: Tensor, factor: int) -> Tensor:
    return resample(w, factor_in=1, factor_out=factor)
