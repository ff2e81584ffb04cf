"""Exercises parallel.DataParallel over the real RCCL backend with a ONE-rank process group (all a 1-GPU box
offers): init, parameter broadcast, hook-driven bucketed all-reduce (ReduceOp.AVG, async, RCCL stream) from inside the
U-Net backward, final wait -- and checks the gradients equal the plain single-process step."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audio_diffusion_pytorch_amd import parallel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
x = torch.randn(2, 2, bench.LENGTH, device=dev)
torch.manual_seed(7)
loss = model(x)
loss.backward()
ref = [p.grad.clone() for p in model.parameters()]
for p in model.parameters():
    p.grad = None
dp = parallel.DataParallel(model)
dp.world = 2  # pretend: forces the hook path; with one rank AVG returns the same values
dp.unet._grad_ready_hook = dp._on_ready
dp._avg = True
sent = []
orig = dp._send
dp._send = lambda flat, a, b: (sent.append(b - a), orig(flat, a, b))[1]
torch.manual_seed(7)
loss2 = dp(x)
loss2.backward()
torch.cuda.synchronize()
worst = max((p.grad - r).abs().max().item() for p, r in zip(model.parameters(), ref))
print(f"rccl 1-rank DataParallel ok: loss {loss.item():.6f} / {loss2.item():.6f}, buckets {len(sent)} "
      f"(MB: {[round(s * 4 / 2**20) for s in sent]}), max |grad diff| {worst:.3e}")
assert worst == 0.0 and abs(loss.item() - loss2.item()) == 0.0
dist.destroy_process_group()
