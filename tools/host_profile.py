"""Where the HOST time of an eager training step goes (the data-parallel path at N > 1 is eager): cProfile of 5 eager steps at
batch 4, launches asynchronous (no sync inside the profiled region).  usage: python tools/host_profile.py [batch]"""
import cProfile
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
model = bench.build_model(dev)
x = torch.randn(B, 2, bench.LENGTH, device=dev)


def step():
    for p in model.parameters():
        p.grad = None
    model(x).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f"batch {B}: host time per eager step {t_host * 1e3:.2f} ms (launch loop only), {t_all * 1e3:.2f} ms incl. the device drain")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
