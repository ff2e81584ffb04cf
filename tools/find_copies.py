"""Which host-side torch ops put device copies (`__amd_rocclr_copyBuffer`, memset / fill kernels) into a training step?
usage: python tools/find_copies.py [batch]"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

os.environ["ADP_TRAIN_GRAPH"] = "0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
model = bench.build_model(dev)
x = torch.randn(B, 2, bench.LENGTH, device=dev)


def step():
    for p in model.parameters():
        p.grad = None
    model(x).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ops = Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::to",
                  "aten::_to_copy", "aten::cat", "aten::add_", "aten::mul_", "aten::mul", "aten::add"):
        st = [s for s in (e.stack or []) if "audio_diffusion_pytorch_amd" in s or "bench.py" in s]
        ops[(e.name, str(e.input_shapes)[:60], st[0][-90:] if st else "?")] += 1
for (name, shp, where), c in ops.most_common(40):
    print(f"{c:4d}  {name:18s} {shp:60s} {where}")
kern = Counter(e.name[:70] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
for k, c in kern.most_common(60):
    if any(t in k for t in ("copy", "Memcpy", "Memset", "fill", "elementwise")):
        print(f"GPU {c:4d}  {k}")
