"""Does an EXTERNAL event recorded inside a captured hipGraph release a side stream while the rest of the graph still runs?
(parallel.DataParallel.capture_step relies on it: all-reduce bucket k starts when its event node fires, mid-backward.)
usage: python tools/ext_event_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_pytorch_amd import _C  # noqa: E402
from audio_diffusion_pytorch_amd._C import ptr  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    buf = torch.empty(512 * 256, device=dev)
    a = torch.zeros(1 << 20, device=dev)

    def long_kernel():
        _C.call_value("adp_probe_mfma", 40000, ptr(buf), buf.numel(), _C.stream())

    long_kernel()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    long_kernel()
    torch.cuda.synchronize()
    print(f"long kernel alone: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    # raw HIP: torch refuses external events on ROCm, the runtime may not
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    try:
        evp = ctypes.c_void_p()
        print("hipEventCreateWithFlags(disable timing):", hip.hipEventCreateWithFlags(ctypes.byref(evp), 2))
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            a.add_(1)
            rc = hip.hipEventRecordWithFlags(evp, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1)
            long_kernel()
        print("hipEventRecordWithFlags(external) under capture:", rc)
        for rep in range(3):
            torch.cuda.synchronize()
            e_start, e_side, e_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e_start.record()
            g.replay()
            rcw = hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), evp, 0)
            e_side.record(side)
            e_end.record()
            torch.cuda.synchronize()
            print(f"raw external rep {rep} (wait rc {rcw}): side stream released at {e_start.elapsed_time(e_side):.3f} ms, graph done at "
                  f"{e_start.elapsed_time(e_end):.3f} ms")
    except Exception as e:
        print(f"raw HIP external event: {type(e).__name__}: {e}")
    for external in (False,):
        try:
            ev = torch.cuda.Event(external=external)
            side = torch.cuda.Stream()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                a.add_(1)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                a.add_(1)
                ev.record()
                long_kernel()
            for rep in range(3):
                torch.cuda.synchronize()
                e_start, e_side, e_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e_start.record()
                g.replay()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    e_side.record(side)
                e_end.record()
                torch.cuda.synchronize()
                print(f"external={external} rep {rep}: side stream released at {e_start.elapsed_time(e_side):.3f} ms, graph done at "
                      f"{e_start.elapsed_time(e_end):.3f} ms")
        except Exception as e:
            print(f"external={external}: {type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
