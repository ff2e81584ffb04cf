"""1x1 projections per launch inside a hipGraph (20 launches, cold weights): conv_tilek1 against conv_mm with its cross-workgroup K
split (reduce launch included).  usage: python tools/tilek1_micro.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from tools.tile_bench import graph_time  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    os.environ["ADP_TILEK1_MAX_TILES"] = "100000"
    os.environ["ADP_TILEK1_MIN_TILES"] = "1"
    os.environ["ADP_TILEK1_MM_BLOCKS"] = "100000"
    for B, R, M, L in ((1, 512, 512, 512), (1, 1024, 512, 256), (1, 512, 1024, 256), (1, 1024, 512, 128), (1, 512, 1024, 128),
                       (1, 1024, 1024, 128), (1, 512, 512, 1024), (1, 256, 512, 2048), (1, 512, 256, 2048), (4, 1024, 512, 128),
                       (4, 512, 512, 512), (1, 1024, 2048, 128), (1, 512, 1024, 512)):
        xs = [torch.randn(B, R, L, device=dev) for _ in range(2)]
        rs = [torch.randn(B, M, L, device=dev) for _ in range(2)]
        gs = [torch.randn(B, M, L, device=dev) for _ in range(2)]
        nw = max(2, (320 << 20) // (R * M * 4))
        nw = min(nw, 64)
        ws = [torch.randn(M, R, 1, device=dev) * 0.02 for _ in range(nw)]
        b = torch.randn(M, device=dev)
        row = []
        for name, env, nkw in (("mm", "0", "8"), ("tilek1 x8", "1", "8"), ("tilek1 x16", "1", "16")):
            os.environ["ADP_CONV_TILEK1"] = env
            os.environ["ADP_TILEK1_NKW"] = nkw
            n = max(20, nw)
            t1 = graph_time([(lambda i: (lambda: ops.conv1d(xs[i & 1], ws[i % nw], b, res=rs[i & 1])))(i) for i in range(n)])
            t2 = graph_time([(lambda i: (lambda: ops.conv1d(gs[i & 1], ws[i % nw], None, transposed=True)))(i) for i in range(n)])
            row.append(f"{name}: fwd+res {t1:5.1f} dgrad {t2:5.1f}")
        print(f"[{B},{R}->{M},{L}] " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
