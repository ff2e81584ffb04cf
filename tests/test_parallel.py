"""Data-parallel path: 2 processes over gloo on the CPU (kernels through the SIMT emulator).
Gradients after the bucketed all-reduce must equal the single-process gradients on the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    """A port the OS hands out now (a fixed pid-derived port can still sit in TIME_WAIT from a previous run)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


CFG = dict(in_channels=2, channels=[8, 16], factors=[2, 2], items=[1, 1], modulation_features=32)


CFG_GUIDED = dict(in_channels=2, channels=[8, 16], factors=[2, 2], items=[1, 1], modulation_features=32,
                  cross_attentions=[0, 1], attention_heads=2, attention_features=8, embedding_features=12,
                  use_embedding_cfg=True, embedding_max_length=5)


def _worker_cfg(rank, world, port, out_dir):
    """Same as _worker for UNetV0(use_embedding_cfg=True): the fixed-embedding table lives OUTSIDE the U-Net's flat
    gradient buffer and must be averaged through the trailing bucket."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import conftest
    conftest._use_emulator()
    import audio_diffusion_pytorch_amd as adp
    from audio_diffusion_pytorch_amd.parallel import DataParallel
    from test_unet import FixedSigmas
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    sig = [0.2, 0.7, 0.4, 0.9]
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig[2 * rank:2 * rank + 2]),
                               **CFG_GUIDED)
    dp = DataParallel(model, min_bucket_bytes=1024)
    assert len(dp._extra) == 1  # the fixed embedding table
    g = torch.Generator().manual_seed(7)
    x, noise = torch.randn(4, 2, 64, generator=g), torch.randn(4, 2, 64, generator=g)
    emb = torch.randn(4, 5, 12, generator=g)
    mask = torch.tensor([True, False, False, True])
    sl = slice(2 * rank, 2 * rank + 2)
    loss = dp(x[sl], noise=noise[sl], embedding=emb[sl], embedding_mask_proba=0.5, batch_mask=mask[sl])
    loss.backward()
    # the cross-attention context bank runs UNDER the data-parallel hook (its gradients are their own region of the flat buffer)
    assert dp.unet._ctx_bank_runs == 1 and dp.unet._ctx_bank_hooked_backwards == 1
    ca, cb = dp.unet.ctx_param_range()
    assert cb - ca == 2 * (2 * 12 + 2 * 2 * 8 * 12) and cb == sum(p.numel() for p in dp.unet.parameters())
    torch.save({"grads": {n: p.grad.clone() for n, p in model.named_parameters()},
                "params": {n: p.detach().clone() for n, p in model.named_parameters()}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_gloo_reduces_parameters_outside_the_unet(emul, tmp_path):
    import audio_diffusion_pytorch_amd as adp
    from test_unet import FixedSigmas
    port = _free_port()
    mp.spawn(_worker_cfg, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert "net.fixed_embedding.weight" in r0["grads"]
    for n in r0["grads"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n   # every parameter -- also outside the U-Net -- agrees
    sig = [0.2, 0.7, 0.4, 0.9]
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig), **CFG_GUIDED)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(r0["params"][n])
    g = torch.Generator().manual_seed(7)
    x, noise = torch.randn(4, 2, 64, generator=g), torch.randn(4, 2, 64, generator=g)
    emb = torch.randn(4, 5, 12, generator=g)
    mask = torch.tensor([True, False, False, True])
    model(x, noise=noise, embedding=emb, embedding_mask_proba=0.5, batch_mask=mask).backward()
    gmax = max(p.grad.abs().max().item() for p in model.parameters())
    for n, p in model.named_parameters():
        err = (r0["grads"][n] - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-3 * gmax)
        assert err < 1e-4, (n, err)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import conftest
    conftest._use_emulator()
    import audio_diffusion_pytorch_amd as adp
    from audio_diffusion_pytorch_amd.parallel import DataParallel
    from test_unet import FixedSigmas
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init per rank: the wrapper must broadcast rank 0's parameters
    sig = [0.2, 0.7, 0.4, 0.9]
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig[2 * rank:2 * rank + 2]),
                               **CFG)
    dp = DataParallel(model, min_bucket_bytes=1024)
    g = torch.Generator().manual_seed(7)
    x, noise = torch.randn(4, 2, 64, generator=g), torch.randn(4, 2, 64, generator=g)
    loss = dp(x[2 * rank:2 * rank + 2], noise=noise[2 * rank:2 * rank + 2])
    loss.backward()
    saved = {"grads": {n: p.grad.clone() for n, p in model.net.named_parameters()},
             "params": {n: p.detach().clone() for n, p in model.net.named_parameters()}}
    # the overlap measurement bench.py reports at N > 1: step / step without collectives / collectives alone
    import time

    def step():
        for p in model.parameters():
            p.grad = None
        dp(x[2 * rank:2 * rank + 2], noise=noise[2 * rank:2 * rank + 2]).backward()

    def timer(fn, n):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n
    saved["overlap"] = dp.measure_overlap(step, timer, reps=1)
    step()  # the wrapper still reduces after the measurement (hook re-attached)
    saved["grads_after"] = {n: p.grad.clone() for n, p in model.net.named_parameters()}
    # deferred mode (parallel.capture_step_deferred without the graph): the hook only notes its regions, flush_deferred sends them
    dp._deferred = []
    step()
    assert len(dp._deferred) >= 1 and not dp._works
    dp.flush_deferred()
    dp._deferred = None
    saved["grads_deferred"] = {n: p.grad.clone() for n, p in model.net.named_parameters()}
    torch.save(saved, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_gloo_equals_single_process(emul, tmp_path):
    import audio_diffusion_pytorch_amd as adp
    from test_unet import FixedSigmas
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    ov = r0["overlap"]
    assert {"step_ms", "step_without_allreduce_ms", "allreduce_alone_ms", "buckets_mb", "hidden_frac"} <= set(ov)
    assert len(ov["buckets_mb"]) >= 1 and 0.0 <= ov["hidden_frac"] <= 1.0
    for n in r0["grads"]:
        assert torch.equal(r0["grads_after"][n], r0["grads"][n]), n     # measuring leaves the reduction intact
        assert torch.equal(r0["grads_deferred"][n], r0["grads"][n]), n  # regions noted during backward, sent behind it
        assert torch.equal(r1["grads_deferred"][n], r0["grads"][n]), n
        assert torch.equal(r0["params"][n], r1["params"][n]), n          # broadcast happened
        assert torch.allclose(r0["grads"][n], r1["grads"][n], atol=0, rtol=0), n  # same averaged gradient
    # single process on the concatenated batch with rank 0's parameters
    sig = [0.2, 0.7, 0.4, 0.9]
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig), **CFG)
    with torch.no_grad():
        for n, p in model.net.named_parameters():
            p.copy_(r0["params"][n])
    g = torch.Generator().manual_seed(7)
    x, noise = torch.randn(4, 2, 64, generator=g), torch.randn(4, 2, 64, generator=g)
    model(x, noise=noise).backward()
    gmax = max(p.grad.abs().max().item() for p in model.net.parameters())
    for n, p in model.net.named_parameters():
        err = (r0["grads"][n] - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-3 * gmax)
        assert err < 1e-4, (n, err)


def _run_rccl_workers(world):
    import subprocess
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_rccl_worker.py"), str(r), str(world), port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
        assert "DataParallel over RCCL ok" in o


@pytest.mark.gpu
def test_dp_rccl_one_rank_group():
    """parallel.DataParallel on the REAL RCCL backend with a one-rank group (all a 1-GPU box offers) and
    force_collectives=True: bucketed ReduceOp.AVG all-reduces from inside the U-Net backward on RCCL's stream, the wait at the
    end of the backward node, the pre-allocated trailing bucket of the parameter outside the U-Net -- gradients must equal
    the unwrapped model's."""
    _run_rccl_workers(1)


@pytest.mark.gpu
def test_dp_rccl_two_ranks():
    """The same over two GPUs (when the box has them): each rank takes half of the batch, gradients = single-process
    gradients of the whole batch."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_rccl_workers(2)
