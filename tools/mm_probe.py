"""Runs one conv / wgrad shape repeatedly (for rocprofv3 --pmc passes on a single kernel).
usage: python tools/mm_probe.py <fwd|dgrad|wgrad|wgrad0> B C L [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_pytorch_amd import ops, _C  # noqa: E402

if os.environ.get("ADP_LIB"):  # probe builds of the kernel library (tools/probe/)
    _C._testing_use_library(os.environ["ADP_LIB"], allow_cpu=False)

kind, B, C, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
x = torch.randn(B, C, L, device=dev)
w = torch.randn(C, C, 3, device=dev) * 0.05
b = torch.randn(C, device=dev)
g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
stats = ops.gn_stats(x, 8)
out = torch.empty_like(x)
dy = torch.randn_like(x)
dw, db = torch.empty_like(w), torch.empty_like(b)
big = torch.empty(64 * 1024 * 1024, device=dev)  # 256 MB: flushes L2 / Infinity Cache between iterations


def run():
    if kind == "fwd":
        ops.conv1d(x, w, b, pad=1, prologue=1, pro_stats=stats, pro_gamma=g, pro_beta=be, groups=8, res=x, out=out)
    elif kind == "dgrad":
        ops.conv1d(dy, w, None, pad=1, transposed=True, out=out)
    elif kind == "wgrad0":  # weight gradient of a materialised (already activated) input
        ops.conv1d_wgrad(x, dy, 3, pad=1, dw=dw, dbias=db)
    else:
        ops.conv1d_wgrad(x, dy, 3, pad=1, prologue=1, pro_stats=stats, pro_gamma=g, pro_beta=be, groups=8, dw=dw, dbias=db)


for _ in range(3):
    run()
torch.cuda.synchronize()
for flush in (False, True):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(iters):
        if flush:
            big.zero_()
        s.record()
        run()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    fl = 2 * 3 * C * C * L * B
    print(f"{kind} B{B} C{C} L{L} flush={flush}: {tot / iters * 1e3:.1f} us  {fl / (tot / iters) / 1e9:.1f} TF")
