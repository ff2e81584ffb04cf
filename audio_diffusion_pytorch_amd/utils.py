"""Host helpers with the reference's semantics (/root/reference/audio_diffusion_pytorch/utils.py:15-70, :123-125).
Pure host-side glue: kwargs routing for DiffusionModel and small predicates.  The windowed-sinc
`resample` family (utils.py:82-117) runs on the polyphase HIP kernel adp_resample."""
from functools import reduce
from inspect import isfunction
from math import ceil, floor, log2, pi
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Generator, Tensor


def exists(val) -> bool:
    return val is not None


def iff(condition: bool, value):
    return value if condition else None


def is_sequence(obj) -> bool:
    return isinstance(obj, (list, tuple))


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def to_list(val) -> List:
    if isinstance(val, tuple):
        return list(val)
    if isinstance(val, list):
        return val
    return [val]


def prod(vals: Sequence[int]) -> int:
    return reduce(lambda x, y: x * y, vals)


def closest_power_2(x: float) -> int:
    e = log2(x)
    best = min((floor(e), ceil(e)), key=lambda z: abs(x - 2 ** z))
    return 2 ** int(best)


def group_dict_by_prefix(prefix: str, d: Dict) -> Tuple[Dict, Dict]:
    with_prefix, without = {}, {}
    for k, v in d.items():
        (with_prefix if k.startswith(prefix) else without)[k] = v
    return with_prefix, without


def groupby(prefix: str, d: Dict, keep_prefix: bool = False) -> Tuple[Dict, Dict]:
    """Splits kwargs into (those starting with `prefix`, prefix stripped unless keep_prefix; the rest)."""
    with_prefix, rest = group_dict_by_prefix(prefix, d)
    if keep_prefix:
        return with_prefix, rest
    return {k[len(prefix):]: v for k, v in with_prefix.items()}, rest


def prefix_dict(prefix: str, d: Dict) -> Dict:
    return {prefix + str(k): v for k, v in d.items()}


_KERNEL_BANKS: Dict = {}


def resample_kernels(factor_in: int, factor_out: int, rolloff: float, lowpass_filter_width: int):
    """The reference's [factor_out, J] windowed-sinc kernel bank and its `width` (utils.py:91-102), evaluated with
    the reference's own fp32 torch expressions on the CPU (the reference CPU path's bits), cached per configuration."""
    key = (factor_in, factor_out, rolloff, lowpass_filter_width)
    hit = _KERNEL_BANKS.get(key)
    if hit is not None:
        return hit
    kw = dict(device="cpu", dtype=torch.float32)
    base = min(factor_in, factor_out) * rolloff
    width = ceil(lowpass_filter_width * factor_in / base)
    idx = torch.arange(-width, width + factor_in, **kw)[None, None] / factor_in
    t = torch.arange(0, -factor_out, step=-1, **kw)[:, None, None] / factor_out + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width) * pi
    window = torch.cos(t / lowpass_filter_width / 2) ** 2
    scale = base / factor_in
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels *= window * scale
    hit = (kernels.reshape(factor_out, -1).contiguous(), width)
    _KERNEL_BANKS[key] = hit
    return hit


def resample(waveforms: Tensor, factor_in: int, factor_out: int, rolloff: float = 0.99,
             lowpass_filter_width: int = 6) -> Tensor:
    """Windowed-sinc resampler (utils.py:82-109 semantics) on the polyphase HIP kernel `adp_resample`: the padding,
    the phase interleave "(b c) k l -> b c (l k)" and the crop are index arithmetic inside the kernel."""
    from . import ops
    b, c, length = waveforms.shape
    target = int(factor_out * length / factor_in)
    kernels, width = resample_kernels(factor_in, factor_out, rolloff, lowpass_filter_width)
    x = waveforms.to(torch.float32).contiguous()
    dkey = (factor_in, factor_out, rolloff, lowpass_filter_width, str(x.device))
    bank = _KERNEL_BANKS.get(dkey)
    if bank is None:  # device copy of the coefficient bank, made once per device
        bank = kernels.to(x.device)
        _KERNEL_BANKS[dkey] = bank
    out = ops.resample(x, bank, factor_in, factor_out, width, target)
    return out.to(waveforms.dtype)


def downsample(waveforms: Tensor, factor: int, **kwargs) -> Tensor:
    return resample(waveforms, factor_in=factor, factor_out=1, **kwargs)


def upsample(waveforms: Tensor, factor: int, **kwargs) -> Tensor:
    return resample(waveforms, factor_in=1, factor_out=factor, **kwargs)


def randn_like(tensor: Tensor, *args, generator: Optional[Generator] = None, **kwargs) -> Tensor:
    """randn_like that supports a (CPU) generator, then moves to the tensor's device (utils.py:123-125)."""
    return torch.randn(tensor.shape, *args, generator=generator, **kwargs).to(tensor)
