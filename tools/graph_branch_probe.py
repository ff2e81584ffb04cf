"""Do two independent branches of ONE hipGraph run concurrently on this ROCm?  (The captured data-parallel step forks RCCL's
all-reduces onto a side stream inside the graph; if the runtime serialises graph branches, nothing can hide under backward.)
Branch A: register-only MFMA kernels (matrix pipes, no memory); branch B: 256 MB streaming copies (HBM, few VALU).  Alone they
take t_A and t_B; perfectly concurrent ~max(t_A, t_B), serialised t_A + t_B.  Same pair on two eager streams for reference,
with the side stream at normal and at high priority.
usage: python tools/graph_branch_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_pytorch_amd import _C  # noqa: E402
from audio_diffusion_pytorch_amd._C import ptr  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = 64 << 20
src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
buf = torch.empty(512 * 256, device=dev)
K = 8
ITERS = int(os.environ.get("PROBE_MFMA_ITERS", "3000"))


def A():
    for _ in range(K):
        _C.call_value("adp_probe_mfma", ITERS, ptr(buf), buf.numel(), _C.stream())


def Bc():
    for _ in range(K):
        _C.call("adp_probe_copy", ptr(src), ptr(dst), n, _C.stream())


def ev_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def graph_of(body):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g.replay


def forked(prio):
    s2 = torch.cuda.Stream(priority=prio)

    def body():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            Bc()
        A()
        cur.wait_stream(s2)
    return body


def serial():
    A()
    Bc()


tA, tB = ev_ms(graph_of(A)), ev_ms(graph_of(Bc))
print(f"[branch] graph of {K} MFMA kernels alone: {tA:.3f} ms; graph of {K} 256 MB copies alone: {tB:.3f} ms "
      f"({8 * n * K / tB / 1e6:.0f} GB/s)")
print(f"[branch] ONE graph, serial chain A then B: {ev_ms(graph_of(serial)):.3f} ms (sum {tA + tB:.3f})")
for prio, label in ((0, "normal"), (-1, "high")):
    t = ev_ms(graph_of(forked(prio)))
    print(f"[branch] ONE graph, B forked onto a {label}-priority side stream: {t:.3f} ms  "
          f"(max {max(tA, tB):.3f}, sum {tA + tB:.3f}; overlap = {(tA + tB - t) / min(tA, tB):.2f} of the shorter branch)")
    te = ev_ms(forked(prio))
    print(f"[branch] eager, two streams ({label} priority side stream): {te:.3f} ms  (overlap {(tA + tB - te) / min(tA, tB):.2f})")
