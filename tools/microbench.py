"""Per-kernel timing of the hot-path kernels at the BASELINE UNetV0 shapes (run on the GPU box).
Prints one line per (op, depth): ms, achieved TFLOP/s and algorithmic GB/s (SURVEY.md section 8d byte model)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_pytorch_amd import ops  # noqa: E402

CH = [8, 32, 64, 128, 256, 512, 512, 1024, 1024]
FAC = [1, 4, 4, 4, 2, 2, 2, 2, 2]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--length", type=int, default=2 ** 18)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L = a.batch, a.length
    rows = []
    Ld = L
    cin = 2
    for d, (C, f) in enumerate(zip(CH, FAC)):
        Lin = Ld
        Ld = Ld // f
        x = torch.randn(B, C, Ld, device=dev)
        w = torch.randn(C, C, 3, device=dev) * 0.05
        b = torch.randn(C, device=dev)
        g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        A = B * C * Ld * 4
        stats = ops.gn_stats(x, 8)
        out = torch.empty_like(x)
        dy = torch.randn_like(x)

        def rec(name, ms, flops, bytes_):
            rows.append(dict(op=name, depth=d, C=C, L=Ld, ms=round(ms, 4), tflops=round(flops / ms / 1e9, 2),
                             gbps=round(bytes_ / ms / 1e6, 1)))
            print(json.dumps(rows[-1]), flush=True)

        fl = 2 * 3 * C * C * Ld * B
        rec("gn_stats", timeit(lambda: ops.gn_stats(x, 8, out=stats)), 0, A)
        rec("convblock_fwd", timeit(lambda: ops.conv1d(x, w, b, pad=1, prologue=1, pro_stats=stats, pro_gamma=g,
                                                      pro_beta=be, groups=8, res=x, out=out)), fl, 3 * A + 12 * C * C)
        rec("conv_dgrad", timeit(lambda: ops.conv1d(dy, w, None, pad=1, transposed=True, out=out)), fl,
            2 * A + 12 * C * C)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        rec("conv_wgrad", timeit(lambda: ops.conv1d_wgrad(x, dy, 3, pad=1, prologue=1, pro_stats=stats, pro_gamma=g,
                                                          pro_beta=be, groups=8, dw=dw, dbias=db)), fl,
            2 * A + 12 * C * C)
        dx = torch.empty_like(x)
        rec("gn_silu_bwd", timeit(lambda: ops.gn_silu_bwd(x, dy, stats, g, be, 8, dres=dy, dx=dx)), 0, 6 * A)
        NT = 2 * C
        bank = torch.randn(B, NT, device=dev) * 0.1
        y, st = ops.modulation_fwd(x, bank.view(-1), NT)
        rec("modulation_fwd", timeit(lambda: ops.modulation_fwd(x, bank.view(-1), NT, y=y, stats=st)), 0, 2 * A)
        dbank = torch.zeros_like(bank)
        rec("modulation_bwd", timeit(lambda: ops.modulation_bwd(x, dy, bank.view(-1), NT, st, dbank.view(-1), NT,
                                                                dx=dx)), 0, 3 * A)
        # down / up convs of this depth
        xin = torch.randn(B, cin, Lin, device=dev)
        wd = torch.randn(C, cin, f, device=dev) * 0.05
        od = torch.empty(B, C, Ld, device=dev)
        rec("down_fwd", timeit(lambda: ops.conv1d(xin, wd, b, stride=f, out=od)), 2 * cin * C * f * Ld * B,
            B * cin * Lin * 4 + A)
        wu = torch.randn(cin, C, 3, device=dev) * 0.05
        ou = torch.empty(B, cin, Lin, device=dev)
        bu = torch.randn(cin, device=dev)
        rec("up_fwd", timeit(lambda: ops.conv1d(x, wu, bu, pad=1, up=f, out=ou)), 2 * 3 * C * cin * Lin * B,
            B * cin * Lin * 4 + A)
        cin = C
        del x, out, dy, dx, y, xin, od, ou
    if a.out:
        with open(a.out, "w") as fo:
            json.dump(rows, fo, indent=1)


if __name__ == "__main__":
    main()
