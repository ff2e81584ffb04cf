"""The README training loop at kernel speed: `loss = model(audio); loss.backward()` replayed from two hipGraphs.

The reference's users write the loop of /root/reference/README.md:36-39 (`DiffusionModel.forward` -> `VDiffusion.forward`,
models.py:40-41, diffusion.py:82-95) and nothing else.  Launched call by call, a step of this package is ~700 C-ABI launches
issued from Python -- 16-30 ms of host time for 6-11 ms of kernels.  A training step on fixed shapes is a static launch
sequence (one autograd node for the whole U-Net, no host reads), so `VDiffusion.forward` captures it ONCE per call structure:

  * graph F: sigma draw + noise draw (torch's Philox generator, registered with the graph: every replay draws fresh values,
    `torch.manual_seed` keeps working), the fused noising kernel, the U-Net forward, the loss;
  * graph B (same memory pool): the whole backward, every parameter gradient written into the flat gradient buffer.

`model(x)` then costs a copy of x into the graph's static input + one graph launch, and `loss.backward()` one graph launch:
the autograd node `_Replay` hands the flat buffer's per-parameter views to autograd, so AccumulateGrad, parameter hooks,
gradient accumulation, `loss / k` scaling (the incoming gradient is a static input of graph B) and optimizers behave as in the
eager step.  What is captured is exactly the eager path (`VDiffusion._forward_eager`), kernels and arithmetic unchanged.

Taken only where it is safe, else the eager step runs as before (never an error): CUDA tensors, grad mode on, the stock
`UniformDistribution`, inputs / conditioning tensors that do not require grad, keyword arguments that can be made static
(tensors, None, python scalars, lists of those), no data-parallel hook on the U-Net (its collectives are captured explicitly
by parallel.capture_step), not already inside a stream capture.  `VDiffusion(use_graph=False)` or ADP_TRAIN_GRAPH=0 switch it
off.  A capture that fails (e.g. a user loss_fn that syncs with the host) marks its call structure as eager-only.
"""
import os
import weakref
from collections import OrderedDict
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.distributed
import torch.nn as nn
from torch import Tensor


class _ProxyParameters:
    """For the duration of a capture every Parameter of the module is replaced, in the modules' `_parameters` dicts, by a fresh
    Parameter over the SAME storage.  Why: a loss tensor of an earlier step that the caller still holds keeps that step's autograd
    graph alive, and with it the parameters' AccumulateGrad nodes, created on the stream that step ran on.  Autograd re-uses a
    live node, and when a gradient produced on the capture stream reaches a node of another stream it makes THAT stream wait for
    an event of the capture -- which pulls the caller's stream into the capture and breaks it (a segmentation fault inside
    hipStreamEndCapture on ROCm 7; torch's "AccumulateGrad node's stream does not match" warning describes the mechanism).
    Fresh tensor objects have no node yet; the kernels see the same memory, so the captured graphs serve the real parameters."""

    def __init__(self, module: nn.Module):
        self.params = tracked_parameters(module)
        self.holders = module.__dict__["_adp_param_cache"][1]
        self.proxies = [nn.Parameter(p.data, requires_grad=p.requires_grad) for p in self.params]

    def __enter__(self):
        for (d, leaf), q in zip(self.holders, self.proxies):
            d[leaf] = q
        return self.proxies

    def __exit__(self, *exc):
        for (d, leaf), p in zip(self.holders, self.params):
            d[leaf] = p
        return False


def tracked_parameters(module: nn.Module) -> List[nn.Parameter]:
    """`list(module.parameters())` without the module-tree walk (1-2 ms for ~600 parameters) on every call: the list is cached
    on the module together with where each entry is registered and re-validated by identity per call (~30 us); replaced
    Parameter objects (load_state_dict(assign=True), to_empty, ...) or a changed parameter count rebuild it."""
    cache = module.__dict__.get("_adp_param_cache")
    if cache is not None:
        params, holders = cache
        for p, (d, leaf) in zip(params, holders):
            if d.get(leaf) is not p:
                cache = None
                break
    if cache is None:
        params, holders, seen = [], [], set()
        for mod in module.modules():
            for leaf, p in mod._parameters.items():
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
                    holders.append((mod._parameters, leaf))
        module.__dict__["_adp_param_cache"] = (params, holders)
    return params


class _Entry:
    """One captured call structure: static inputs, the two graphs, the static loss / incoming gradient / parameter gradients."""
    __slots__ = ("g_f", "g_b", "sx", "snoise", "statics", "sloss", "sgloss", "params", "grads", "sig", "step")


class _Replay(torch.autograd.Function):
    """Autograd node of a replayed step: forward = graph F (returns a copy of the static loss), backward = graph B."""

    @staticmethod
    def forward(ctx, entry: _Entry, *params):
        entry.g_f.replay()
        entry.step += 1
        ctx.entry, ctx.step = entry, entry.step
        return entry.sloss.clone()

    @staticmethod
    def backward(ctx, gloss):
        entry: _Entry = ctx.entry
        if ctx.step != entry.step:
            raise RuntimeError("graph-replayed training step: backward() of a loss whose forward is no longer the last one run "
                               "at this shape (the replayed step keeps ONE set of activations); call backward before the next "
                               "forward, or construct the diffusion with use_graph=False / set ADP_TRAIN_GRAPH=0")
        entry.sgloss.copy_(gloss.reshape(entry.sgloss.shape))
        # gradient accumulation: a .grad that is still LAST step's view of the static gradient buffer (adopted by AccumulateGrad,
        # not cleared since) would be overwritten by the replay -- move it to memory of its own first
        for p, g in zip(entry.params, entry.grads):
            pg = p.grad
            if pg is not None and g is not None and pg.data_ptr() == g.data_ptr():
                p.grad = pg.clone()
        entry.g_b.replay()
        # fresh tensor objects (nobody else holds them): AccumulateGrad adopts them without a copy, as in the eager step
        return (None,) + tuple(None if g is None else g.detach() for g in entry.grads)


class TrainStepGraphs:
    """Cache of captured training steps of one diffusion module, keyed on the call STRUCTURE (shapes, kwarg names, tensor
    shapes / dtypes, python scalar values -- never object identity), LRU-bounded: every entry owns its activations."""

    MAX_ENTRIES = 2

    def __init__(self, owner: nn.Module):
        self._owner = weakref.ref(owner)  # (the registry below is keyed weakly on the owner: no cycle through this object)
        self.cache: "OrderedDict[Any, _Entry]" = OrderedDict()
        self.eager_only = set()
        self.captures = 0  # (visible to tests / bench: how many times a step was captured)
        self.replays = 0

    def _signature(self, params):
        return tuple((p.data_ptr(), p.requires_grad) for p in params)

    def run(self, x: Tensor, noise: Optional[Tensor], kwargs: Dict[str, Any]) -> Optional[Tensor]:
        from .diffusion import _kw_spec
        names = sorted(kwargs)
        live: List[Tensor] = []
        specs = tuple((k, _kw_spec(kwargs[k], live)) for k in names)
        if any(sp is None for _, sp in specs) or any((not t.is_cuda) or t.requires_grad for t in live):
            return None
        key = (tuple(x.shape), x.dtype, x.device, noise is not None, specs)
        if key in self.eager_only:
            return None
        all_params = tracked_parameters(self._owner())
        sig = self._signature(all_params)
        entry = self.cache.get(key)
        if entry is not None and entry.sig != sig:  # parameters moved / replaced / (un)frozen: the graph holds stale pointers
            del self.cache[key]
            entry = None
        if entry is None:
            params = [p for p in all_params if p.requires_grad]
            if not params:
                return None
            try:
                with _ProxyParameters(self._owner()) as proxies:
                    entry = self._capture(x, noise, kwargs, names, live, [q for q in proxies if q.requires_grad])
                entry.params = params
                del proxies
            except Exception as e:  # capture is a launch-overhead optimisation only
                import warnings
                warnings.warn(f"graph capture of the training step failed ({type(e).__name__}: {e}); this call structure runs "
                              f"eagerly from now on", RuntimeWarning)
                self.eager_only.add(key)
                torch.cuda.synchronize(x.device)
                return None
            entry.sig = sig
            self.cache[key] = entry
            while len(self.cache) > self.MAX_ENTRIES:
                self.cache.popitem(last=False)
        else:
            self.cache.move_to_end(key)
        entry.sx.copy_(x)
        if noise is not None:
            entry.snoise.copy_(noise)
        for st, t in zip(entry.statics, live):
            st.copy_(t)
        self.replays += 1
        return _Replay.apply(entry, *entry.params)

    def _capture(self, x, noise, kwargs, names, live, params) -> _Entry:
        from .diffusion import _kw_rebuild
        eager = self._owner()._forward_eager  # (x, noise, **kwargs) -> loss: the un-captured step
        e = _Entry()
        e.sx = x.detach().contiguous().clone()
        e.snoise = noise.detach().contiguous().clone() if noise is not None else None
        e.statics = [t.detach().clone(memory_format=torch.contiguous_format) for t in live]
        it = iter(e.statics)
        skw = {k: _kw_rebuild(kwargs[k], it) for k in names}
        e.sgloss = torch.ones((), dtype=torch.float32, device=x.device)
        e.params = params
        dev = x.device
        # two eager steps on a side stream (allocator pools, lazily built tables); they must not be seen by the caller: no
        # .grad is touched (autograd.grad) and the generator is put back, so a seeded script draws what the eager loop draws
        rng = torch.cuda.get_rng_state(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                loss = eager(e.sx, e.snoise, **skw)
                if loss.dim() != 0:
                    raise ValueError("the loss is not a scalar")
                torch.autograd.grad(loss, params, grad_outputs=e.sgloss, allow_unused=True)
                del loss
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        torch.cuda.set_rng_state(rng, dev)
        # inside a process group (e.g. under torch's DistributedDataParallel): its watchdog thread queries the events of earlier
        # collectives -- let it retire them first, and do not let a query from that thread abort the capture
        mode = "global"
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            from .parallel import quiesce_watchdog
            quiesce_watchdog()
            mode = "thread_local"
        e.g_f, e.g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.g_f, capture_error_mode=mode):
            e.sloss = eager(e.sx, e.snoise, **skw)
        with torch.cuda.graph(e.g_b, pool=e.g_f.pool(), capture_error_mode=mode):
            e.grads = torch.autograd.grad(e.sloss, params, grad_outputs=e.sgloss, allow_unused=True)
        # Drop the captured step's autograd graph: it keeps the parameters' AccumulateGrad nodes alive, and those were created
        # on the CAPTURE stream -- every later backward would then run its 600 gradient accumulations on that stream behind an
        # event record / wait pair each ("AccumulateGrad node's stream does not match ...": ~5 ms of a replayed step).
        e.sloss = e.sloss.detach()
        e.step = 0
        self.captures += 1
        return e


# diffusion module -> its captured steps (weak keys: the graphs and their activation pools die with the module)
GRAPHS_OF: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
