"""Worker of tests/test_parallel.py::test_dp_rccl_* (one process per rank, real RCCL backend on the GPU box):
gradients through parallel.DataParallel must equal the plain single-process gradients of the concatenated batch -- incl. a
parameter OUTSIDE the U-Net's flat buffer (ClassifierFreeGuidance's fixed embedding: the trailing bucket).
usage: python tests/_dp_rccl_worker.py <rank> <world> <port>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    import audio_diffusion_pytorch_amd as adp
    from audio_diffusion_pytorch_amd.parallel import DataParallel, graph_safe_rccl_env
    from test_parallel import CFG_GUIDED
    from test_unet import FixedSigmas
    graph_safe_rccl_env()  # (the watchdog stays away from the stream capture below)
    dist.init_process_group("nccl", rank=rank, world_size=world)

    class FixedSigmas(FixedSigmas):  # noqa: F811  (device-resident: a captured step cannot copy from pageable host memory)
        def __call__(self, num_samples, device=torch.device("cpu")):
            if getattr(self, "_dev", None) is None or self._dev.device != torch.device(device):
                self._dev = self.vals.to(device)
            return self._dev[:num_samples]
    per = 2
    sig = [0.2, 0.7, 0.4, 0.9][:per * world]
    g = torch.Generator().manual_seed(7)
    n = per * world
    x, noise = torch.randn(n, 2, 256, generator=g), torch.randn(n, 2, 256, generator=g)
    emb = torch.randn(n, 5, 12, generator=g)
    mask = torch.tensor([True, False, False, True][:n])
    # reference: the whole batch on this GPU, unwrapped
    torch.manual_seed(100)
    ref = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig), **CFG_GUIDED).to(dev)
    ref(x.to(dev), noise=noise.to(dev), embedding=emb.to(dev), embedding_mask_proba=0.5, batch_mask=mask.to(dev)).backward()
    # data parallel: this rank's slice (rank 0's parameters are broadcast, so every rank starts from seed 100's)
    torch.manual_seed(100 + rank)
    sl = slice(per * rank, per * rank + per)
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig[sl]), **CFG_GUIDED).to(dev)
    dp = DataParallel(model, min_bucket_bytes=1024, force_collectives=True)
    assert len(dp._extra) == 1 and dp._collect
    for step in range(2):  # second step: the pre-allocated trailing bucket is reused
        for p in model.parameters():
            p.grad = None
        dp(x[sl].to(dev), noise=noise[sl].to(dev), embedding=emb[sl].to(dev), embedding_mask_proba=0.5,
           batch_mask=mask[sl].to(dev)).backward()
    torch.cuda.synchronize()
    # config 4's optimisation runs where config 4 runs: the context bank is taken under the data-parallel hook
    assert dp.unet._ctx_bank_runs == 2 and dp.unet._ctx_bank_hooked_backwards == 2
    worst = 0.0
    for (name, p), q in zip(model.named_parameters(), ref.parameters()):
        assert (p.grad is None) == (q.grad is None), name
        if p.grad is not None:
            den = max(q.grad.abs().max().item(), 1e-6)
            worst = max(worst, (p.grad - q.grad).abs().max().item() / den)
    assert worst < 1e-4, worst
    # the same step CAPTURED in a hipGraph with its RCCL all-reduces (how bench.py runs at N > 1) and replayed twice
    eager = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    args = (x[sl].to(dev),)
    kw = dict(noise=noise[sl].to(dev), embedding=emb[sl].to(dev), embedding_mask_proba=0.5, batch_mask=mask[sl].to(dev))
    from audio_diffusion_pytorch_amd.parallel import capture_step

    def one_step():
        for p in model.parameters():
            p.grad = None
        dp(*args, **kw).backward()
    g, replay = capture_step(one_step, warmup=1)
    for _ in range(2):
        replay()
    torch.cuda.synchronize()
    for (name, p), e in zip(model.named_parameters(), eager):
        assert (p.grad is None) == (e is None), name
        if e is not None:
            assert torch.equal(p.grad, e), f"replayed data-parallel step differs from the eager one at {name}"
    # the fallback form: kernels captured, the bucket all-reduces issued eagerly behind every replay (capture_step_deferred)
    from audio_diffusion_pytorch_amd.parallel import capture_step_deferred
    g2, replay2 = capture_step_deferred(one_step, dp, warmup=1)
    for _ in range(2):
        replay2()
    torch.cuda.synchronize()
    for (name, p), e in zip(model.named_parameters(), eager):
        if e is not None:
            assert torch.equal(p.grad, e), f"deferred-collective replay differs from the eager step at {name}"
    dp._deferred = None  # back to sending from inside backward
    print(f"rank {rank}/{world}: DataParallel over RCCL ok, worst relative gradient difference {worst:.2e}; the step replays "
          f"from a hipGraph with its collectives, bit-identical", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
