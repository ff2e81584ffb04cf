"""In-kernel timelines of the two F(4,3) matrix kernels (conv_mm4 at [4,1024,256], the batched W4 weight gradient of the same
layer): where a workgroup's time goes -- ramp, per chunk (barrier wait vs MFMA issue, loaders: wait for their global loads vs
LDS stores), K-group exchange, epilogue.  Builds a SEPARATE measurement library (-DADP_KTRACE: lane 0 of every wave drops
s_memtime into LDS at the marks, the first 64 workgroups dump them at the end); the product library has no marks.
  python tools/ktrace.py build [tag] [-DFLAG ...] [--notrace]   # here (hipcc cross-compiles): tools/ab/libadp_ktrace<tag>.so
  [KTRACE_TAG=<tag>] python tools/ktrace.py run [C L B]         # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "ab", "libadp_ktrace.so")


def build(tag="", extra=(), trace=True):
    """tools/ab/libadp_ktrace<tag>.so with the timeline marks (trace) and any extra -D flags (A/B builds of compile-time knobs)."""
    from audio_diffusion_pytorch_amd import build as b
    odir = os.path.join(ROOT, "tools", "ab", "kt_obj" + tag)
    os.makedirs(odir, exist_ok=True)
    lib = LIB.replace(".so", tag + ".so")
    objs, procs = [], []
    for s in b.sources():
        o = os.path.join(odir, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-DADP_KTRACE"] if trace else []) + \
            list(extra) + ["-c", "-x", "hip", s, "-o", o, "-I", os.path.join(ROOT, "include"), "-I", b.CSRC, "-Wno-unused-value"]
        procs.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


def report(name, t, nwaves, nmma, mhz=2400.0):
    """t: [blocks, 16, 64] int64 cycle marks."""
    import torch
    us = lambda c: c / mhz  # noqa: E731
    blocks = [b for b in range(t.shape[0]) if t[b, 0, 0] > 0]
    print(f"== {name}: {len(blocks)} traced workgroups, {nmma} MMA + {nwaves - nmma} loader waves; cycles at an assumed {mhz:.0f} MHz")
    for b in blocks[:3]:
        t0 = int(t[b, :nwaves, 0][t[b, :nwaves, 0] > 0].min())
        for w, label in ((0, "MMA wave 0"), (nmma, "loader wave 0")):
            row = t[b, w]
            nch = max(c for c in range(16) if row[3 + 3 * c] > 0) + 1 if any(row[3 + 3 * c] > 0 for c in range(16)) else 0
            wait = [int(row[2 + 3 * c] - row[1 + 3 * c]) for c in range(nch)]
            work = [int(row[3 + 3 * c] - row[2 + 3 * c]) for c in range(nch)]
            end = int(row[63]) if row[63] > 0 else int(row[60])
            print(f"  block {b} {label}: entry +{us(int(row[0]) - t0):.2f} us; first mark +{us(int(row[1]) - t0):.2f}; loop end "
                  f"+{us(int(row[60]) - t0):.2f}; after exchange +{us(int(row[62]) - t0) if row[62] > 0 else 0:.2f}; end +{us(end - t0):.2f} us")
            if w == 0:
                print(f"     per chunk barrier wait (cycles): {wait}")
                print(f"     per chunk MFMA issue  (cycles): {work}")
            else:
                print(f"     per chunk wait for global loads (cycles): {wait}")
                print(f"     per chunk LDS stores + next loads issue (cycles): {work}")
                gap = [int(row[1 + 3 * (c + 1)] - row[3 + 3 * c]) for c in range(nch - 1)]
                print(f"     per chunk time in the barrier (cycles): {gap}")
    # who closes the barrier: arrival of every wave at barrier B_c relative to the barrier's release (MMA wave 0's first mark
    # after it), averaged over chunks 4..12 -- the last arriver is the wave closest to 0
    for b in blocks[:2]:
        arr = []
        for w in range(nwaves):
            row = t[b, w]
            slot = (lambda c: 1 + 3 * c) if w < nmma else (lambda c: 3 + 3 * c)
            ds = [int(t[b, 0, 2 + 3 * c]) - int(row[slot(c)]) for c in range(4, 13) if row[slot(c)] > 0 and t[b, 0, 2 + 3 * c] > 0]
            arr.append(round(sum(ds) / max(1, len(ds))))
        print(f"  block {b}: cycles between a wave's arrival at the chunk barrier and its release, waves 0..{nmma - 1} MMA then loaders: {arr}")
    # when did the traced workgroups start / end (one resident generation, or several?)
    t00 = min(int(t[b, :nwaves, 0][t[b, :nwaves, 0] > 0].min()) for b in blocks)
    starts = [round(us(int(t[b, :nwaves, 0][t[b, :nwaves, 0] > 0].min()) - t00), 1) for b in blocks]
    ends = [round(us(max(int(v) for v in t[b, :nwaves, 60:64].flatten() if v > 0) - t00), 1) for b in blocks]
    print(f"  start of every traced workgroup (us after the first): {starts}")
    print(f"  end of every traced workgroup (us after the first start): {ends}")
    # aggregate over all traced blocks
    tot_wait = tot_work = n = 0
    span = []
    for b in blocks:
        row = t[b, 0]
        for c in range(16):
            if row[3 + 3 * c] > 0 and row[2 + 3 * c] > 0:
                tot_wait += int(row[2 + 3 * c] - row[1 + 3 * c])
                tot_work += int(row[3 + 3 * c] - row[2 + 3 * c])
                n += 1
        t0 = int(t[b, :nwaves, 0][t[b, :nwaves, 0] > 0].min())
        ends = [int(t[b, w, 63]) for w in range(nmma) if t[b, w, 63] > 0]
        if ends:
            span.append((max(ends) - t0, int(row[1]) - t0, int(row[60]) - int(row[1]), max(ends) - int(row[60])))
    if n:
        print(f"  MMA wave 0 over {len(blocks)} blocks: barrier wait {tot_wait / n:.0f} cycles per chunk, issue {tot_work / n:.0f} cycles per chunk "
              f"({tot_work / (tot_wait + tot_work):.1%} of the loop)")
    if span:
        import statistics as st
        print(f"  workgroup lifetime median {us(st.median(s[0] for s in span)):.2f} us = ramp {us(st.median(s[1] for s in span)):.2f} + K loop "
              f"{us(st.median(s[2] for s in span)):.2f} + exchange/epilogue {us(st.median(s[3] for s in span)):.2f}")


def run():
    import ctypes
    import torch
    from audio_diffusion_pytorch_amd import _C, ops
    _C._testing_use_library(LIB.replace(".so", os.environ.get("KTRACE_TAG", "") + ".so"), allow_cpu=False)
    lib = _C.lib()
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    dev = torch.device("cuda:0")
    x = torch.randn(B, C, L, device=dev)
    dy = torch.randn(B, C, L, device=dev)
    w = torch.randn(C, C, 3, device=dev) * 0.05
    bias = torch.randn(C, device=dev)
    res = torch.randn(B, C, L, device=dev)
    buf = torch.zeros(64 * 16 * 64, dtype=torch.int64, device=dev)
    null = ctypes.c_void_p(0)
    # ---- conv_mm4 forward
    for _ in range(3):
        ops.conv1d(x, w, bias, pad=1, res=res, gn=ops.GnPart())
    torch.cuda.synchronize()
    assert lib.adp_ktrace_set_mm4(ctypes.c_void_p(buf.data_ptr())) == 0
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv1d(x, w, bias, pad=1, res=res, gn=ops.GnPart())
    b_.record()
    torch.cuda.synchronize()
    lib.adp_ktrace_set_mm4(null)
    print(f"conv_mm4 forward [{B},{C},{L}] traced launch: {a.elapsed_time(b_) * 1e3:.1f} us (event pair)")
    report(f"conv_mm4 forward [{B},{C},{L}]", buf.view(64, 16, 64).cpu(), 12, 8)
    # ---- W4 weight gradient, batched like the step does (8 items at depth 7)
    buf.zero_()
    n = 8 if C >= 1024 else 4
    dws = [torch.empty(C, C, 3, device=dev) for _ in range(n)]
    dbs = [torch.empty(C, device=dev) for _ in range(n)]

    def batch():
        park = ops.WgradPark()
        for i in range(n):
            ops.conv1d_wgrad(x, dy, 3, pad=1, dw=dws[i], dbias=dbs[i], park=park)
        park.flush()
    for _ in range(2):
        batch()
    torch.cuda.synchronize()
    assert lib.adp_ktrace_set_wgrad(ctypes.c_void_p(buf.data_ptr())) == 0
    a.record()
    batch()
    b_.record()
    torch.cuda.synchronize()
    lib.adp_ktrace_set_wgrad(null)
    print(f"wgrad batch of {n} [{B},{C},{L}] traced: {a.elapsed_time(b_) * 1e3:.1f} us (event pair, incl. second stages)")
    report(f"wgrad_mm W4 [{B},{C},{L}] x {n}", buf.view(64, 16, 64).cpu(), 12, 8)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":  # build [tag] [-Dflags ...] [--notrace]
        rest = sys.argv[2:]
        tag = rest[0] if rest and not rest[0].startswith("-") else ""
        build(tag, [a for a in rest if a.startswith("-D")], trace="--notrace" not in rest)
    else:
        run()
