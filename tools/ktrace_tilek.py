"""In-kernel timeline of conv_tilek (tools/ktrace.py's measurement build): per 16-channel chunk of one wave -- wait for the chunk's
global loads, park + U transform (to the wave's LDS sync), issue of the next loads, the four K steps, closing sync.
  python tools/ktrace.py build            # here
  python tools/ktrace_tilek.py [B C L]    # on the GPU box"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import _C, ops  # noqa: E402
from tools.ktrace import LIB  # noqa: E402


def main():
    _C._testing_use_library(LIB.replace(".so", os.environ.get("KTRACE_TAG", "") + ".so"), allow_cpu=False)
    lib = _C.lib()
    B, C, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 1024, 256)
    dev = torch.device("cuda:0")
    x = torch.randn(B, C, L, device=dev)
    ws = [torch.randn(C, C, 3, device=dev) * 0.05 for _ in range(4)]
    res = torch.randn(B, C, L, device=dev)
    buf = torch.zeros(64 * 16 * 64, dtype=torch.int64, device=dev)
    for tr in (False, True):
        for w in ws[:3]:
            ops.conv1d(x, w, None, pad=1, res=res, transposed=tr)
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.adp_ktrace_set_tilek(ctypes.c_void_p(buf.data_ptr())) == 0
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.conv1d(x, ws[3], None, pad=1, res=res, transposed=tr)  # (a weight tensor not touched yet: cold, as in a step)
        b_.record()
        torch.cuda.synchronize()
        lib.adp_ktrace_set_tilek(ctypes.c_void_p(0))
        t = buf.view(64, 16, 64).cpu()
        print(f"conv_tilek {'data gradient' if tr else 'forward'} [{B},{C},{L}] traced launch {a.elapsed_time(b_) * 1e3:.1f} us (event pair)")
        blocks = [b for b in range(64) if t[b, 0, 0] > 0]
        names = ("wait loads", "park+U+sync", "issue next loads", "4 K steps", "closing sync")
        for b in blocks[:2]:
            for w in (0, 5):
                row = t[b, w]
                t0 = int(t[b, :8, 0].min())
                print(f"  block {b} wave {w}: entry +{(int(row[0]) - t0)} cycles; loop end +{int(row[60]) - t0}; after the partial-tile "
                      f"barrier +{int(row[61]) - t0}; end +{int(row[63]) - t0}")
                for c in range(8):
                    if row[6 + 6 * c] == 0:
                        break
                    m = [int(row[k + 6 * c]) for k in range(1, 7)]
                    gap = m[0] - (int(row[6 * c]) if c else int(row[0]))
                    print(f"     chunk {c}: before +{gap:5d} | " + " | ".join(f"{n} {m[i + 1] - m[i]:5d}" for i, n in enumerate(names)))
        # aggregate
        import statistics as st
        ph = [[] for _ in names]
        life = []
        for b in blocks:
            for w in range(8):
                row = t[b, w]
                for c in range(8):
                    if row[6 + 6 * c] == 0:
                        break
                    for i in range(5):
                        ph[i].append(int(row[2 + i + 6 * c]) - int(row[1 + i + 6 * c]))
                life.append((int(row[63]) - int(row[0]), int(row[1]) - int(row[0]), int(row[60]) - int(row[1]), int(row[63]) - int(row[60])))
        print("  median cycles per chunk and wave: " + ", ".join(f"{n} {st.median(v):.0f}" for n, v in zip(names, ph) if v))
        print(f"  wave lifetime median {st.median(v[0] for v in life):.0f} cycles = prologue {st.median(v[1] for v in life):.0f} + K loop "
              f"{st.median(v[2] for v in life):.0f} + exchange/epilogue {st.median(v[3] for v in life):.0f}")


if __name__ == "__main__":
    main()
