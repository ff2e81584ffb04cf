"""v-objective diffusion on the gfx950 kernels: `VDiffusion` (training loss) and `VSampler`
(DDIM-style loop) and `VInpainter` (RePaint-style resampling loop), API-compatible with
/root/reference/audio_diffusion_pytorch/diffusion.py:15-30, :62-95, :133-190, :300-354.

Differences from the reference are structural, not numerical:
  * noising (x_noisy, v_target) is one fused kernel (2 reads, 2 writes) instead of ~6 elementwise ops;
  * the default MSE loss and its gradient are fused kernels; a user `loss_fn` still works (autograd);
  * the sampler keeps the sigma schedule on the host and the (alpha, beta) table on the device: the
    loop never reads device memory (the reference's tqdm f-string syncs every step, diffusion.py:188),
    so one step (U-Net forward + rotation kernel) is captured in a hipGraph and replayed.
"""
from collections import OrderedDict
from contextlib import nullcontext
from math import pi
from typing import Any, List, Optional, Tuple

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from tqdm import tqdm

from . import ops



def _on_device_of(t: Tensor):
    """Kernel launches go to the current HIP device's stream: make the tensor's device current for the call."""
    return torch.cuda.device(t.device) if t.is_cuda else nullcontext()


""" Distributions """


class Distribution:
    """Interface used by different distributions"""

    def __call__(self, num_samples: int, device: torch.device):
        raise NotImplementedError()


class UniformDistribution(Distribution):
    def __init__(self, vmin: float = 0.0, vmax: float = 1.0):
        super().__init__()
        self.vmin, self.vmax = vmin, vmax

    def __call__(self, num_samples: int, device: torch.device = torch.device("cpu")):
        return (self.vmax - self.vmin) * torch.rand(num_samples, device=device) + self.vmin


def extend_dim(x: Tensor, dim: int):
    return x.view(*x.shape + (1,) * (dim - x.ndim))


""" Diffusion """


class Diffusion(nn.Module):
    """Interface used by different diffusion methods"""


class _VNoise(torch.autograd.Function):
    """x_noisy = a x + b n ; v_target = a n - b x  (diffusion.py:90-92) as one kernel."""

    @staticmethod
    def forward(ctx, x, noise, sigmas):
        x_noisy, v_target = ops.v_noise(x.contiguous(), noise.contiguous(), sigmas.contiguous())
        ctx.mark_non_differentiable(x_noisy, v_target)
        return x_noisy, v_target

    @staticmethod
    def backward(ctx, *g):  # data and noise are not differentiated on the training path
        return None, None, None


class _MSE(torch.autograd.Function):
    """F.mse_loss(v_pred, v_target) (diffusion.py:95) with a fused backward."""

    @staticmethod
    def forward(ctx, v_pred, v_target):
        v_pred, v_target = v_pred.contiguous(), v_target.contiguous()
        ctx.save_for_backward(v_pred, v_target)
        return ops.mse_fwd(v_pred, v_target)

    @staticmethod
    def backward(ctx, gloss):
        v_pred, v_target = ctx.saved_tensors
        return ops.mse_bwd(v_pred, v_target, gloss.contiguous()), None


def fused_mse_loss(v_pred: Tensor, v_target: Tensor) -> Tensor:
    return _MSE.apply(v_pred, v_target)


class VDiffusion(Diffusion):
    def __init__(self, net: nn.Module, sigma_distribution: Distribution = UniformDistribution(),
                 loss_fn: Any = F.mse_loss, use_graph: bool = True):
        """`use_graph` (not in the reference): replay the training step from hipGraphs where that is safe (graphed.py) --
        the README loop `loss = model(x); loss.backward()` then costs two graph launches instead of ~700 kernel launches
        issued from Python; False (or ADP_TRAIN_GRAPH=0) = launch every kernel eagerly."""
        super().__init__()
        self.net = net
        self.sigma_distribution = sigma_distribution
        self.loss_fn = loss_fn
        self.use_graph = use_graph

    def get_alpha_beta(self, sigmas: Tensor) -> Tuple[Tensor, Tensor]:
        angle = sigmas * pi / 2
        return torch.cos(angle), torch.sin(angle)

    def forward(self, x: Tensor, noise: Optional[Tensor] = None, **kwargs) -> Tensor:
        """`noise` (optional, default torch.randn_like(x) as at diffusion.py:88) lets a harness inject the draw."""
        if self._graph_path_ok(x, noise):
            with _on_device_of(x):
                loss = self.train_graphs().run(x, noise, kwargs)
            if loss is not None:
                return loss
        return self._forward_eager(x, noise, **kwargs)

    def train_graphs(self):
        """The captured training steps of this module (kept OFF the module: hipGraphs can be neither deep-copied nor pickled,
        and an EMA copy.deepcopy(model) / torch.save(model) must keep working after a step has been captured)."""
        from .graphed import GRAPHS_OF, TrainStepGraphs
        graphs = GRAPHS_OF.get(self)
        if graphs is None:
            graphs = GRAPHS_OF[self] = TrainStepGraphs(self)
        return graphs

    def _graph_path_ok(self, x: Tensor, noise: Optional[Tensor]) -> bool:
        """Whether this call may be served by the replayed step (graphed.py lists the conditions)."""
        if not (self.use_graph and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()) or x.requires_grad:
            return False
        if noise is not None and (not noise.is_cuda or noise.requires_grad or noise.shape != x.shape):
            return False
        if type(self.sigma_distribution) is not UniformDistribution or os.environ.get("ADP_TRAIN_GRAPH", "1") == "0":
            return False
        if torch.cuda.is_current_stream_capturing():  # (a caller's own whole-step capture: bench.py, parallel.capture_step)
            return False
        hooked = self.__dict__.get("_hook_sites")
        if hooked is None:  # the U-Nets whose backward may carry a data-parallel hook (collectives: not captured implicitly)
            hooked = self.__dict__["_hook_sites"] = [m for m in self.net.modules() if hasattr(m, "_param_offsets")]
        return all(getattr(m, "_grad_ready_hook", None) is None for m in hooked)

    def _forward_eager(self, x: Tensor, noise: Optional[Tensor] = None, **kwargs) -> Tensor:
        batch_size, device = x.shape[0], x.device
        with _on_device_of(x):
            sigmas = self.sigma_distribution(num_samples=batch_size, device=device)
            if noise is None:
                noise = torch.randn_like(x)
            x_noisy, v_target = _VNoise.apply(x, noise, sigmas.to(torch.float32))
            v_pred = self.net(x_noisy, sigmas, **kwargs)
            if self.loss_fn is F.mse_loss:
                return fused_mse_loss(v_pred, v_target)
            return self.loss_fn(v_pred, v_target)


""" Schedules """


class Schedule(nn.Module):
    """Interface used by different sampling schedules"""

    def forward(self, num_steps: int, device: torch.device) -> Tensor:
        raise NotImplementedError()


class LinearSchedule(Schedule):
    def __init__(self, start: float = 1.0, end: float = 0.0):
        super().__init__()
        self.start, self.end = start, end

    def forward(self, num_steps: int, device: Any) -> Tensor:
        return torch.linspace(self.start, self.end, num_steps, device=device)


""" Samplers """


class Sampler(nn.Module):
    pass


def _kw_spec(value, tensors: List[Tensor]):
    """Hashable structure of a forward kwarg (names / shapes / dtypes / python scalars -- never object identity);
    the tensors it holds are appended to `tensors` in traversal order.  None when it cannot be made static."""
    if isinstance(value, Tensor):
        tensors.append(value)
        return ("T", tuple(value.shape), value.dtype, value.device)
    if value is None or isinstance(value, (bool, int, float)):
        return ("S", type(value).__name__, value)
    # (strings are NOT static: a net that takes text runs a host-side tokenizer + H2D copy per call, which is illegal
    # inside stream capture; our own TextConditioningNet resolves text to a tensor before the loop instead)
    if isinstance(value, (list, tuple)):
        items = tuple(_kw_spec(v, tensors) for v in value)
        return None if any(i is None for i in items) else ("L", type(value).__name__, items)
    return None


def _kw_rebuild(value, it):
    """The same structure with every tensor replaced by the next one from `it` (the cache entry's static copy)."""
    if isinstance(value, Tensor):
        return next(it)
    if isinstance(value, (list, tuple)):
        return type(value)(_kw_rebuild(v, it) for v in value)
    return value


class VSampler(Sampler):

    diffusion_types = [VDiffusion]
    GRAPH_CACHE_ENTRIES = 4  # captured steps kept (LRU); each owns its private activation pool

    def __init__(self, net: nn.Module, schedule: Schedule = LinearSchedule(), use_graph: bool = True):
        super().__init__()
        self.net = net
        self.schedule = schedule
        self.use_graph = use_graph
        self._graph_cache: "OrderedDict" = OrderedDict()

    def get_alpha_beta(self, sigmas: Tensor) -> Tuple[Tensor, Tensor]:
        angle = sigmas * pi / 2
        return torch.cos(angle), torch.sin(angle)

    def _tables(self, num_steps: int, b: int, device):
        """sigma table [N+1, B] and per-step (a_i, b_i, a_{i+1}, b_{i+1}) table [N, 4], both on the device.
        The schedule is a pure function of num_steps, so nothing here (or in the loop) syncs with the host."""
        sigmas = self.schedule(num_steps + 1, device=device).to(torch.float32)
        alphas, betas = self.get_alpha_beta(sigmas)
        ab = torch.stack([alphas[:-1], betas[:-1], alphas[1:], betas[1:]], dim=1).contiguous()
        return sigmas[:, None].expand(num_steps + 1, b).contiguous(), ab

    HOIST_MAX_BYTES = 512 << 20  # cap of the hoisted conditioning table (README net: 360 KB per step and batch element)

    def _conditioning_table(self, sig: Tensor, num_steps: int, b: int, kwargs) -> Optional[Tensor]:
        """[num_steps, B, bank_total] or None.  The time values of ALL steps are known before the loop starts, and what the
        U-Net derives from them (time MLP, then the conditioning bank: one row of 90 K scale / shift values per call in the
        README configuration, 184 MB of weights streamed per call) does not depend on x: one batched pass in front of the
        loop replaces eight launches per step.  Only for a bare UNetV0 (the plugin wrappers re-shape their arguments) without
        `features`; ADP_SAMPLER_HOIST=0 keeps the per-step conditioning (A/B)."""
        table_fn = getattr(self.net, "conditioning_table", None)
        if table_fn is None or kwargs.get("features") is not None or os.environ.get("ADP_SAMPLER_HOIST", "1") == "0":
            return None
        total = getattr(self.net, "bank_total", 0)
        if total <= 0 or 4 * num_steps * b * total > self.HOIST_MAX_BYTES:
            return None
        table = table_fn(sig[:num_steps])
        return None if table is None else table.view(num_steps, b, total)

    @torch.no_grad()
    def forward(self, x_noisy: Tensor, num_steps: int, show_progress: bool = False, **kwargs) -> Tensor:
        with _on_device_of(x_noisy):
            b = x_noisy.shape[0]
            sig, ab = self._tables(num_steps, b, x_noisy.device)
            x = x_noisy.contiguous().clone()
            prepare = getattr(self.net, "prepare_sampling_kwargs", None)
            if prepare is not None:  # e.g. text -> embedding tensor, once per sampling run
                kwargs = prepare(x, kwargs)
            cond = self._conditioning_table(sig, num_steps, b, kwargs)
            if self.use_graph and x.is_cuda and not show_progress:
                out = self._forward_graph(x, sig, ab, num_steps, kwargs, cond)
                if out is not None:
                    return out
            bar = tqdm(range(num_steps), disable=not show_progress)
            host_sigmas = torch.linspace(self.schedule.start, self.schedule.end, num_steps + 1).tolist() \
                if (show_progress and isinstance(self.schedule, LinearSchedule)) else None
            for i in bar:
                v = self.net(x, sig[i], **kwargs) if cond is None else self.net(x, sig[i], conditioning=cond[i], **kwargs)
                x = ops.v_step(x, v.contiguous(), ab[i])
                if host_sigmas is not None:
                    bar.set_description(f"Sampling (noise={host_sigmas[i + 1]:.2f})")
            return x

    def _forward_graph(self, x: Tensor, sig: Tensor, ab: Tensor, num_steps: int, kwargs,
                       cond: Optional[Tensor] = None) -> Optional[Tensor]:
        """One step = U-Net forward + rotation kernel, captured once per call STRUCTURE and replayed.  The cache key
        is (x shape, kwarg names, tensor shapes/dtypes, python scalar values); the entry owns static copies of every
        tensor kwarg (also those nested in `channels`) and the caller's tensors are copied into them before the
        replays, so fresh conditioning tensors per call reuse the graph and can never be read after they are freed.
        Per step only two tiny device-to-device copies (sigma row, alpha/beta row) precede the replay.  Returns None
        (eager fallback) for kwargs that cannot be made static."""
        names = sorted(kwargs)
        live: List[Tensor] = []
        specs = tuple((k, _kw_spec(kwargs[k], live)) for k in names)
        if any(sp is None for _, sp in specs) or any(not t.is_cuda for t in live):
            return None
        key = (tuple(x.shape), x.device, specs, cond is not None)
        entry = self._graph_cache.get(key)
        if entry is None:
            sx, ssig, sab = torch.empty_like(x), torch.empty_like(sig[0]), torch.empty_like(ab[0])
            scond = torch.empty_like(cond[0]) if cond is not None else None  # this step's rows of the hoisted conditioning
            statics = [torch.empty_like(t, memory_format=torch.contiguous_format) for t in live]
            sx.copy_(x)
            ssig.copy_(sig[0])
            sab.copy_(ab[0])
            for st, t in zip(statics, live):
                st.copy_(t)
            it = iter(statics)
            skw = {k: _kw_rebuild(kwargs[k], it) for k in names}
            if scond is not None:
                scond.copy_(cond[0])
                skw["conditioning"] = scond
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up outside capture
                v = self.net(sx, ssig, **skw)
                ops.v_step(sx, v.contiguous(), sab, out=torch.empty_like(sx))
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                v = self.net(sx, ssig, **skw)
                ops.v_step(sx, v.contiguous(), sab, out=sx)  # in place: each element is read then written
            entry = (graph, sx, ssig, sab, statics, scond)
            self._graph_cache[key] = entry
            while len(self._graph_cache) > self.GRAPH_CACHE_ENTRIES:
                self._graph_cache.popitem(last=False)  # least recently used graph + its buffers
        else:
            self._graph_cache.move_to_end(key)
        graph, sx, ssig, sab, statics, scond = entry
        sx.copy_(x)
        for st, t in zip(statics, live):
            st.copy_(t)
        for i in range(num_steps):
            if scond is not None:
                scond.copy_(cond[i])  # (the captured step no longer reads the sigma row)
            else:
                ssig.copy_(sig[i])
            sab.copy_(ab[i])
            graph.replay()
        return sx.clone()


""" Inpainters """


class Inpainter(nn.Module):
    pass


class VInpainter(Inpainter):
    """diffusion.py:306-354.  Per resample the reference runs ~10 elementwise ops; here the rotation to the next
    noise level, the re-noising of the source and the masked blend are ONE kernel (adp_v_inpaint_step).  The noise
    draws stay on torch's generator (`torch.randn_like(source)`, same call order as the reference) so seeding
    behaves identically; the (alpha, beta) table lives on the device and the loop never syncs with the host."""

    diffusion_types = [VDiffusion]

    def __init__(self, net: nn.Module, schedule: Schedule = LinearSchedule()):
        super().__init__()
        self.net = net
        self.schedule = schedule

    def get_alpha_beta(self, sigmas: Tensor) -> Tuple[Tensor, Tensor]:
        angle = sigmas * pi / 2
        return torch.cos(angle), torch.sin(angle)

    @torch.no_grad()
    def forward(self, source: Tensor, mask: Tensor, num_steps: int, num_resamples: int, show_progress: bool = False,
                x_noisy: Optional[Tensor] = None, **kwargs) -> Tensor:
        with _on_device_of(source):
            return self._run(source, mask, num_steps, num_resamples, show_progress, x_noisy, kwargs)

    def _run(self, source, mask, num_steps, num_resamples, show_progress, x_noisy, kwargs) -> Tensor:
        x = (x_noisy if x_noisy is not None else torch.randn_like(source)).contiguous()
        prepare = getattr(self.net, "prepare_sampling_kwargs", None)
        if prepare is not None:
            kwargs = prepare(x, kwargs)
        b = x.shape[0]
        sigmas = self.schedule(num_steps + 1, device=x.device).to(torch.float32)
        alphas, betas = self.get_alpha_beta(sigmas)
        sig = sigmas[:, None].expand(num_steps + 1, b).contiguous()
        # rows (a_i, b_i, a_j, b_j) for j = i (re-noise at the same level) and j = i + 1 (move on)
        ab_stay = torch.stack([alphas[:-1], betas[:-1], alphas[:-1], betas[:-1]], dim=1).contiguous()
        ab_next = torch.stack([alphas[:-1], betas[:-1], alphas[1:], betas[1:]], dim=1).contiguous()
        src = source.to(torch.float32).contiguous()
        mask_u8 = mask.expand_as(src).to(torch.uint8).contiguous()
        host_sigmas = self.schedule(num_steps + 1, device="cpu").tolist() if show_progress else None
        bar = tqdm(range(num_steps), disable=not show_progress)
        for i in bar:
            for r in range(num_resamples):
                v = self.net(x, sig[i], **kwargs)
                last = r == num_resamples - 1
                x = ops.v_inpaint_step(x, v.contiguous(), src, torch.randn_like(src), mask_u8,
                                       (ab_next if last else ab_stay)[i])
            if host_sigmas is not None:
                bar.set_description(f"Inpainting (noise={host_sigmas[i + 1]:.2f})")
        return x
