"""A/B of two checkouts of the whole package (Python + library), one process each: copy the old package to tools/ab/old_pkg/audio_diffusion_pytorch_amd (built), then  python tools/pkg_ab.py <old|new>"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "tools", "ab", "old_pkg") if which == "old" else ROOT)
sys.path.insert(1, ROOT)
import audio_diffusion_pytorch_amd as adp
import bench
assert ("old_pkg" in adp.__file__) == (which == "old"), adp.__file__
dev = torch.device("cuda:0")
for B in (4, 1):
    torch.manual_seed(0)
    m = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS, items=bench.ITEMS).to(dev)
    x = torch.randn(B, 2, bench.LENGTH, device=dev)
    def zero():
        for p in m.parameters(): p.grad = None
    def step():
        zero(); m(x).backward()
    best = min(bench._time(bench._graphed(step, zero), 20) for _ in range(3))
    print(f"{which} batch {B}: {best * 1e3:.3f} ms", flush=True)
    del m
