"""Which host-side ops issue device copies / fills in one training step (GPU box; kernel-work tool)."""
import os, sys
from collections import Counter
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
m = bench.build_model(dev)
x = torch.randn(1, 2, bench.LENGTH, device=dev)
for _ in range(2):
    for p in m.parameters(): p.grad = None
    m(x).backward()
for p in m.parameters(): p.grad = None
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m(x).backward()
    torch.cuda.synchronize()
c = Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::sum"):
        st = [s for s in (e.stack or []) if "audio_diffusion" in s or "bench" in s][:2]
        shp = ""
        c[(e.name, tuple(st))] += 1
for k, v in c.most_common(25):
    print(v, k)
print([ (e.key, e.count) for e in prof.key_averages() if "copy" in e.key.lower() or "Memcpy" in e.key or "fill" in e.key.lower()][:20])
