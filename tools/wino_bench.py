"""A/B microbench of the Winograd F(2,3) conv family (conv_wino, ADP_CONV_WINO=1) against the exact-f32 MFMA conv_mm on the deep
ResnetItem shapes (GPU box):  python tools/bs_bench.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402


def timeit(fn, n=20):
    """us per call, n calls replayed from one hipGraph (eager event timing is host-bound below ~12 us per launch)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    for C, L in ((256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128)):
        x = torch.randn(B, C, L, device=dev)
        w = torch.randn(C, C, 3, device=dev) * 0.05
        bias = torch.randn(C, device=dev)
        res = torch.randn(B, C, L, device=dev)
        fl = 2 * B * C * C * 3 * L
        row = f"C{C:5d} L{L:5d}:"
        outs = {}
        for mode in ("0", "1"):
            os.environ["ADP_CONV_WINO"] = mode
            outs[mode] = (ops.conv1d(x, w, bias, pad=1, res=res), ops.conv1d(x, w, None, pad=1, transposed=True))
            t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res))
            t_t = timeit(lambda: ops.conv1d(x, w, None, pad=1, transposed=True))
            row += f"  [{'winograd  ' if mode == '1' else 'f32 mfma  '}] fwd {t_f:6.1f} us {fl / t_f / 1e6:6.1f} TF  dgrad {t_t:6.1f} us {fl / t_t / 1e6:6.1f} TF"
        ref64 = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding=1) + res.double()
        e = [((outs[m][0].double() - ref64).abs().max() / ref64.abs().max()).item() for m in ("0", "1")]
        d = ((outs["1"][1] - outs["0"][1]).abs().max() / outs["0"][1].abs().max()).item()
        print(row + f"  | err vs fp64: f32 {e[0]:.1e} wino {e[1]:.1e}; dgrad wino vs f32 {d:.1e}", flush=True)




def cold():
    """same A/B with COLD weights: 40 distinct weight tensors (> the 256 MB Infinity Cache) cycled through, as in the
    model where every conv has its own weights"""
    B, dev = 4, torch.device("cuda:0")
    for C, L in ((1024, 256), (512, 1024)):
        nw = max(2, int(600e6 / (C * C * 12)))
        ws = [torch.randn(C, C, 3, device=dev) * 0.05 for _ in range(nw)]
        x = torch.randn(B, C, L, device=dev)
        xs = [torch.randn(B, C, L, device=dev) for _ in range(nw)]  # cold activations too (fresh tensor per conv)
        row = f"C{C} L{L} ({nw} weight tensors):"
        for mode in ("0", "1"):
            os.environ["ADP_CONV_WINO"] = mode

            def f():
                for w in ws:
                    ops.conv1d(x, w, None, pad=1)

            def fg():
                for w in ws:
                    ops.conv1d(x, w, None, pad=1, gn=ops.GnPart())

            def fc():
                for w, xc in zip(ws, xs):
                    ops.conv1d(xc, w, None, pad=1)

            def fw():  # written by the kernel right before, as in the model
                for w, xc in zip(ws, xs):
                    xc.mul_(1.0)
                    ops.conv1d(xc, w, None, pad=1)

            def fw0():
                for xc in xs:
                    xc.mul_(1.0)

            def t():
                for w in ws:
                    ops.conv1d(x, w, None, pad=1, transposed=True)
            row += (f"  [wino {mode}] fwd {timeit(f, 2) / nw:6.1f} us  cold x {timeit(fc, 2) / nw:6.1f} us  x just written "
                    f"{(timeit(fw, 2) - timeit(fw0, 2)) / nw:6.1f} us  dgrad {timeit(t, 2) / nw:6.1f} us")
        print(row, flush=True)


if __name__ == "__main__":
    cold() if len(sys.argv) > 1 and sys.argv[1] == "cold" else main()
