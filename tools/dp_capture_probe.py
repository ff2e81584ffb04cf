"""Can the data-parallel step (kernels + RCCL all-reduces issued from inside backward) be captured in a hipGraph when the
ProcessGroupNCCL watchdog is kept away from the capture?  One-rank group, DataParallel(force_collectives=True).
usage: python tools/dp_capture_probe.py <mode>   mode = thread_local | relaxed | global   (run each in its own process)"""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29573")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the watchdog thread queries the collectives' events; an event query from another thread while a capture is open is what
# killed the round-3 attempt
os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")
os.environ.setdefault("TORCH_NCCL_DUMP_ON_TIMEOUT", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audio_diffusion_pytorch_amd import parallel  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"  # ("global" can abort: the watchdog polls the warm-up collectives)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
import audio_diffusion_pytorch_amd as adp  # noqa: E402


class Fixed(adp.Distribution):  # (a fixed noise level per sample: the replayed and the eager step then see the same problem)
    def __call__(self, num_samples, device=torch.device("cpu")):
        return torch.linspace(0.2, 0.8, num_samples, device=device)


torch.manual_seed(0)
model = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS, items=bench.ITEMS,
                           diffusion_sigma_distribution=Fixed()).to(dev)
x = torch.randn(B, 2, bench.LENGTH, device=dev)
noise = torch.randn(B, 2, bench.LENGTH, device=dev)
dp = parallel.DataParallel(model, force_collectives=True)


def zero():
    for p in model.parameters():
        p.grad = None


def step():
    zero()
    dp(x, noise=noise).backward()


t_eager = bench._time(step, 10, warmup=3)
print(f"[{mode}] eager data-parallel step: {t_eager * 1e3:.3f} ms", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
zero()
g = torch.cuda.CUDAGraph()
print(f"[{mode}] capturing ...", flush=True)
with torch.cuda.graph(g, capture_error_mode=mode):
    loss = dp(x, noise=noise)
    loss.backward()
print(f"[{mode}] captured; replaying ...", flush=True)
g.replay()
torch.cuda.synchronize()
ref = [p.grad.clone() for p in model.parameters()]
t = bench._time(g.replay, 20, warmup=3)
print(f"[{mode}] data-parallel step CAPTURED with its RCCL all-reduces and replayed: {t * 1e3:.3f} ms per step (batch {B}), "
      f"loss {loss.item():.6f}", flush=True)
dp.unet._grad_ready_hook = None
zero()
model(x, noise=noise).backward()
torch.cuda.synchronize()
worst = max((p.grad - r).abs().max().item() for p, r in zip(model.parameters(), ref))
print(f"[{mode}] max |grad(replayed DP step) - grad(plain eager step)| = {worst:.3e}", flush=True)
dist.destroy_process_group()
