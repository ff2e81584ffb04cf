"""Deep-layer convs per launch inside a hipGraph (20 launches, two buffer sets): conv_tilek against the conv_mm / conv_mm4 path with its
cross-workgroup K split (reduce launch included), GPU box.  usage: python tools/tilek_micro.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from tools.tile_bench import graph_time  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for B, C, L in ((1, 512, 1024), (1, 512, 512), (1, 1024, 256), (1, 1024, 128), (4, 1024, 128)):
        xs = [torch.randn(B, C, L, device=dev) for _ in range(2)]
        rs = [torch.randn(B, C, L, device=dev) for _ in range(2)]
        # COLD weights, as inside a step: every launch of a graph takes another weight tensor, ~320 MB of them in total (the
        # Infinity Cache holds 256 MB); TILEK_WARM=1 = one tensor for all launches (L2 / MALL-resident after the first)
        nw = 1 if os.environ.get("TILEK_WARM") == "1" else max(2, (320 << 20) // (C * C * 12))
        ws = [torch.randn(C, C, 3, device=dev) * 0.02 for _ in range(nw)]
        w = ws[0]
        b = torch.randn(C, device=dev)
        row = []
        ref = None
        for name, env in (("mm", {"ADP_CONV_TILEK": "0"}),
                          ("tilek rb1 pf2", {"ADP_CONV_TILEK": "1", "ADP_TILEK_RB": "1", "ADP_TILEK_PF": "2"}),
                          ("tilek rb1 pf4", {"ADP_CONV_TILEK": "1", "ADP_TILEK_RB": "1", "ADP_TILEK_PF": "4"}),
                          ("tilek rb2", {"ADP_CONV_TILEK": "1", "ADP_TILEK_RB": "2"})):
            for k in ("ADP_CONV_TILEK", "ADP_TILEK_RB", "ADP_TILEK_PF"):
                os.environ.pop(k, None)
            os.environ.update(env)
            out = ops.conv1d(xs[0], w, b, pad=1, res=rs[0])
            if ref is None:
                ref = out
            err = ((out - ref).norm() / ref.norm()).item()
            n = max(20, nw)
            t1 = graph_time([(lambda i: (lambda: ops.conv1d(xs[i & 1], ws[i % nw], b, pad=1, gn=ops.GnPart())))(i) for i in range(n)])
            t2 = graph_time([(lambda i: (lambda: ops.conv1d(xs[i & 1], ws[i % nw], b, pad=1, res=rs[i & 1])))(i) for i in range(n)])
            t3 = graph_time([(lambda i: (lambda: ops.conv1d(xs[i & 1], ws[i % nw], None, pad=1, transposed=True)))(i) for i in range(n)])
            row.append(f"{name}: conv1+gn {t1:5.1f} conv2+res {t2:5.1f} dgrad {t3:5.1f} (err {err:.1e})")
        fl = 2 * B * C * C * L * 3
        print(f"[{B},{C},{L}] {fl / 1e9:.2f} GF | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
