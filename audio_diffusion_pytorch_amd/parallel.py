"""Single-node data-parallel training: one process per MI355X, gradients all-reduced over RCCL/xGMI.

The reference has no distributed code at all; this is new work mandated by
BASELINE.json's north_star.  Design for xGMI (point-to-point links, no switch):
  * the U-Net backward writes every parameter gradient into ONE flat fp32 buffer in parameter order, so
    buckets are contiguous slices: no gradient copies, no flatten/unflatten;
  * a bucket is all-reduced (async, RCCL's own stream) the moment its slice is final.  Blocks finish
    deepest-first and depths 7-8 hold ~82 % of the bytes, so most of the transfer overlaps the long, HBM-heavy
    shallow half of the backward pass;
  * averaging uses RCCL's AVG reduction (no extra pass); the host never synchronises -- the compute stream
    waits on the collectives' events only at the end of backward.
`torch.distributed` (backend "nccl" == RCCL on ROCm) is used for bootstrap and the collective call only.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def graph_safe_rccl_env() -> None:
    """Environment under which a training step WITH its RCCL all-reduces can be captured in a hipGraph (set before the
    process group exists).  ProcessGroupNCCL's watchdog thread polls the collectives' events; an event query from another
    thread while a stream capture is open is what killed the round-3 capture attempt (segmentation fault inside
    libtorch_hip).  With the watchdog's error handling and monitoring off the captured step replays correctly -- one-rank RCCL
    group on MI355X / ROCm 7 / torch 2.10, tools/dp_capture_probe.py: 15.5-16.5 ms eager -> 13.3 ms replayed at batch 4, 16.5 ->
    7.6 ms at batch 1, gradients bit-identical to the eager step's.  The price: a hung collective is no longer aborted by the
    watchdog (it hangs until the launcher's own timeout).  `setdefault`: a caller's explicit setting wins."""
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")
    os.environ.setdefault("TORCH_NCCL_DUMP_ON_TIMEOUT", "0")


def quiesce_watchdog(seconds: float = 0.3) -> None:
    """Called between the eager warm-up collectives and a stream capture: ProcessGroupNCCL's watchdog thread retires finished
    work objects on its next pass (every ~100 ms) by QUERYING their events, and an event query that lands inside an open capture
    can still abort the process on this stack even in thread_local mode (one `dp1` run in about ten died that way, with a c10::Error
    out of a libtorch_hip thread).  After a device synchronisation every pending work is complete; one pause later the watchdog's
    list is empty, and works issued under capture are never handed to it."""
    import time
    import torch.distributed as dist
    torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized():
        time.sleep(seconds)


def capture_step(step, warmup: int = 2):
    """Captures `step()` -- a whole training step through a DataParallel wrapper, collectives included -- in a hipGraph and
    returns (graph, replay).  `warmup` eager steps run first on a side stream (RCCL communicators, allocator pools and lazily
    built tables must exist before the capture).  The capture uses capture_error_mode="thread_local": ProcessGroupNCCL's
    watchdog thread may still be polling the events of the EAGER warm-up collectives while the capture is open, and in the
    default "global" mode an event query from ANY thread aborts the process (seen once in five runs of the `dp1` leg); works
    issued under capture are not handed to the watchdog at all.  Call with gradients set to None (the captured AccumulateGrad
    then adopts views of the flat gradient buffer, as in the eager step)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(side)
    quiesce_watchdog()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        step()
    return graph, graph.replay


def capture_step_deferred(step, dp: "DataParallel", warmup: int = 2):
    """Fallback of capture_step for a stack on which collectives cannot be captured: the step's KERNELS are captured (the
    data-parallel hook only notes which regions of the flat gradient buffer it would have sent), and `replay()` = graph replay +
    the noted bucket all-reduces issued eagerly behind it on RCCL's stream + the wait.  No overlap with the backward pass, but no
    per-kernel host launches either (an eager data-parallel step is host bound at ~2x the kernels' time).  Returns (graph, replay)."""
    dp._deferred = []
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
                dp.flush_deferred()
        torch.cuda.current_stream().wait_stream(side)
        quiesce_watchdog()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            step()
        regions, flat = list(dp._deferred), dp._deferred_flat
    except Exception:
        dp._deferred = None
        raise
    dp._deferred = []

    def replay():
        graph.replay()
        dp._deferred, dp._deferred_flat = list(regions), flat
        dp.flush_deferred()
    return graph, replay


def init_process_group_from_env(backend: Optional[str] = None, graph_safe: bool = False) -> int:
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  `graph_safe=True`
    (opt-in: only jobs that will call capture_step need it) applies the settings of graph_safe_rccl_env, so that the
    data-parallel step can be replayed from a hipGraph -- at the price of the RCCL watchdog (a hung collective then hangs
    until the launcher's timeout); a warning says so when the settings take effect.  Eager jobs keep the watchdog."""
    if dist.is_initialized():
        return dist.get_rank()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if graph_safe:
        import warnings
        warnings.warn("init_process_group_from_env(graph_safe=True): ProcessGroupNCCL's watchdog (async error handling, "
                      "monitoring, dump on timeout) is switched off so that steps with collectives can be captured in a "
                      "hipGraph; a failed or hung RCCL collective will not be aborted by torch", RuntimeWarning, stacklevel=2)
        graph_safe_rccl_env()
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return dist.get_rank()


class DataParallel(nn.Module):
    """Wraps a DiffusionModel (or any module containing a UNetV0 instance): broadcasts rank 0's parameters once, then
    averages gradients across ranks during backward.  forward(*a, **kw) -> module(*a, **kw).

    Two gradient families:
      * the U-Net's parameters -- one flat buffer written by the kernels, all-reduced in place in buckets as the
        backward finalises contiguous regions (hook called from unet._UNetFn.backward);
      * every other trainable parameter of the wrapped module (ClassifierFreeGuidance's fixed embedding table, a
        DiffusionAE encoder, ...) -- ordinary autograd gradients, gathered into ONE trailing bucket and all-reduced
        when the whole backward pass is over (together with a per-parameter "some rank produced a gradient" flag, so
        parameters unused on every rank keep grad = None).
    Gradient accumulation: the U-Net's gradients are averaged per backward pass (so accumulating averaged gradients over
    several passes is exact); the trailing bucket averages whatever `p.grad` HOLDS at the end of a pass, so for parameters
    outside the U-Net call zero_grad between passes (accumulating there would average an already averaged part again).
    The U-Net's collectives are waited for (stream wait, no host sync) at the end of the U-Net's backward node --
    before autograd's AccumulateGrad touches the flat buffer's views; the trailing bucket is issued and waited for
    in an autograd final callback, after the last node of the backward pass."""

    def __init__(self, module: nn.Module, min_bucket_bytes: int = 32 << 20, process_group=None,
                 force_collectives: bool = False):
        """`force_collectives`: issue the bucketed all-reduces even in a ONE-rank group (they return their input): the
        whole RCCL path -- AVG op, RCCL stream against the compute stream on the flat buffer, the trailing bucket -- can
        then be exercised on a single GPU (tests/test_parallel.py, -m gpu)."""
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._collect = self.world > 1 or (force_collectives and dist.is_initialized())
        self.min_bucket = min_bucket_bytes // 4
        self._works: List = []
        self._deferred: Optional[List] = None  # capture_step_deferred: regions noted instead of sent
        self._deferred_flat: Optional[torch.Tensor] = None
        self._pending: List = []  # contiguous (start, end) regions not yet sent
        self._final_queued = False
        self.unet = self._find_unet(module)
        own = {id(p) for p in self.unet.parameters()}
        self._extra = [p for p in module.parameters() if id(p) not in own and p.requires_grad]
        dev = next(module.parameters()).device
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            raise RuntimeError(f"DataParallel: the module lives on {dev} but the current device is "
                               f"cuda:{torch.cuda.current_device()}; call torch.cuda.set_device first")
        if self._collect:
            with torch.no_grad():
                for p in module.parameters():
                    dist.broadcast(p, src=0, group=process_group)
            self.unet._grad_ready_hook = self._on_ready
            self._avg = dist.get_backend(process_group) == "nccl"
            for p in self._extra:
                p.register_post_accumulate_grad_hook(self._on_extra_grad)

    @staticmethod
    def _find_unet(module: nn.Module):
        from .unet import UNetV0Net
        for m in module.modules():
            if isinstance(m, UNetV0Net):
                return m
        raise ValueError("DataParallel needs a module containing a UNetV0 network")

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def sample(self, *args, **kwargs):  # sampling is replica-local: no communication
        return self.module.sample(*args, **kwargs)

    # ---- end-of-backward synchronisation (one autograd final callback per backward pass)
    def _queue_final(self):
        if self._deferred is not None:  # (flush_deferred runs the trailing bucket behind the replay)
            return
        if not self._final_queued:
            self._final_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _on_extra_grad(self, p):
        self._queue_final()

    def _tail_bucket(self):
        """The trailing bucket, allocated ONCE: [gradient of every non-U-Net parameter | one has-grad flag each], with a
        view per parameter and a pinned host row for the flags (one asynchronous copy per step instead of a scalar write per
        parameter)."""
        if getattr(self, "_tail", None) is None:
            sizes = [p.numel() for p in self._extra]
            ref = self._extra[0]
            total = sum(sizes)
            self._tail = torch.zeros(total + len(sizes), dtype=torch.float32, device=ref.device)
            self._tail_views, off = [], 0
            for p, n in zip(self._extra, sizes):
                self._tail_views.append(self._tail[off:off + n].view(p.shape))
                off += n
            self._tail_flags = self._tail[total:]
            self._tail_host = torch.zeros(len(sizes), dtype=torch.float32)
            self._tail_event = None
            if ref.is_cuda:
                self._tail_host = self._tail_host.pin_memory()
                self._tail_event = torch.cuda.Event()  # guards the pinned row against the next step's host write
        return self._tail

    def _finalize(self):
        self._final_queued = False
        if self._extra:
            buf = self._tail_bucket()
            have = [p.grad is not None for p in self._extra]
            if not all(have):
                buf.zero_()  # (slices of absent gradients must not carry last step's values into the sum)
            dst = [v for v, h in zip(self._tail_views, have) if h]
            if dst:
                torch._foreach_copy_(dst, [p.grad for p, h in zip(self._extra, have) if h])
            # (under a stream capture the copy becomes a node of the graph and the event is left alone: a captured event
            #  cannot be waited for from the host)
            guard = self._tail_event is not None and not torch.cuda.is_current_stream_capturing()
            if guard:
                self._tail_event.synchronize()  # the previous step's asynchronous copy has read the pinned row
            self._tail_host.copy_(torch.tensor(have, dtype=torch.float32))
            self._tail_flags.copy_(self._tail_host, non_blocking=True)
            if guard:
                self._tail_event.record()
            self._works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))
        self._wait_all()
        if self._extra:
            # Which parameters got a gradient on SOME rank?  A parameter this rank produced a gradient for certainly
            # did, so the summed flags only have to be read back (one tiny device-to-host copy = a host sync) when a
            # local gradient is missing -- a step in which every rank used every parameter stays sync-free.
            flags = [1.0] * len(have) if all(have) else self._tail_flags.tolist()
            buf.mul_(1.0 / self.world)
            tgt, src = [], []
            for p, v, used in zip(self._extra, self._tail_views, flags):
                if used > 0:
                    if p.grad is None:
                        p.grad = v.clone()
                    else:
                        tgt.append(p.grad)
                        src.append(v)
            if tgt:
                torch._foreach_copy_(tgt, src)

    def measure_overlap(self, step, timer, reps: int = 3) -> dict:
        """How much of the gradient all-reduce hides under the backward pass.  Three timings of `step()` (a full
        forward + backward through this wrapper; `timer(fn, n)` -> seconds per call incl. device sync):
          step_ms                      the data-parallel step as it runs
          step_without_allreduce_ms    the same step with the collectives skipped (a no-op hook: the same kernels, no sends)
          allreduce_alone_ms           the step's bucket sequence all-reduced back to back with no compute beside it
        hidden_frac = 1 - (step - step_without) / allreduce_alone: 1.0 = fully overlapped, 0.0 = fully exposed."""
        if not self._collect:
            raise RuntimeError("measure_overlap needs a process group of more than one rank (no collectives at world size 1)")
        t_step = timer(step, reps)
        sent: List = []
        hook, orig_send = self.unet._grad_ready_hook, self._send
        try:
            self._send = lambda flat, a, b: (sent.append((a, b)), orig_send(flat, a, b))[1]
            step()
        finally:
            self._send = orig_send
        if not sent:
            raise RuntimeError("measure_overlap: step() sent no gradient bucket -- it has to run a backward pass through the "
                               "wrapped U-Net")
        # a no-op hook, NOT None: a detached hook also switches the backward to the deferred one-launch conditioning-bank
        # gradient (unet._UNetFn.backward: defer_bank), i.e. to different kernels than the step being measured
        self.unet._grad_ready_hook = lambda flat, a, b: None
        try:
            t_plain = timer(step, reps)
        finally:
            self.unet._grad_ready_hook = hook
        total = max(b for _, b in sent)
        flat = torch.zeros(total, dtype=torch.float32, device=next(self.unet.parameters()).device)

        def comm_only():
            for a, b in sent:
                self._send(flat, a, b)
            self._wait_all()
        t_comm = timer(comm_only, reps)
        hidden = 1.0 - (t_step - t_plain) / t_comm if t_comm > 0 else None
        return {"step_ms": round(t_step * 1e3, 3), "step_without_allreduce_ms": round(t_plain * 1e3, 3),
                "allreduce_alone_ms": round(t_comm * 1e3, 3), "buckets_mb": [round((b - a) * 4 / 2 ** 20, 1) for a, b in sent],
                "hidden_frac": None if hidden is None else round(max(0.0, min(1.0, hidden)), 4)}

    def _wait_all(self):
        for w, scaled in self._works:
            w.wait()
            if scaled is not None:
                scaled.mul_(1.0 / self.world)
        self._works = []

    # ---- called from the U-Net backward as regions of the flat gradient become final
    def flush_deferred(self) -> None:
        """Sends the regions noted in deferred mode (capture_step_deferred) and waits for them (stream wait, no host sync)."""
        regions, flat = self._deferred or [], self._deferred_flat
        self._deferred = None
        try:
            for a, b in regions:
                self._send(flat, a, b)
            if self._extra:
                self._finalize()
            else:
                self._wait_all()
        finally:
            self._deferred = []

    def _send(self, flat: torch.Tensor, a: int, b: int):
        if self._deferred is not None:
            self._deferred.append((a, b))
            self._deferred_flat = flat
            return
        buf = flat[a:b]
        if self._avg:
            self._works.append((dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:  # gloo (CPU tests): SUM then scale
            self._works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf))

    def _on_ready(self, flat: torch.Tensor, start, end):
        if start is None:  # end of the U-Net backward: flush, then make the compute stream wait for the collectives
            for reg in self._pending:
                self._send(flat, *reg)
            self._pending = []
            self._wait_all()
            if self._extra:
                self._queue_final()
            return
        # Regions arrive as several interleaved address-descending streams (the blocks deepest-first, and each
        # depth's rows of the conditioning bank): merge a new region into the pending region it touches, and send a
        # pending region once it is big enough for the links.
        for i, (a, b) in enumerate(self._pending):
            if end == a:
                self._pending[i] = (start, b)
                break
            if start == b:
                self._pending[i] = (a, end)
                break
        else:
            self._pending.append((start, end))
        keep = []
        for a, b in self._pending:
            if b - a >= self.min_bucket:
                self._send(flat, a, b)
            else:
                keep.append((a, b))
        self._pending = keep
