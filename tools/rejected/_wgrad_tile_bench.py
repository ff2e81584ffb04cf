"""Depth-1 weight gradient microbench (GPU box): wgrad_tile.hip vs wgrad_mm (ADP_WGRAD_TILE=1/0), hipGraph of 20 launches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from tools.tile_bench import graph_time  # noqa: E402
dev = torch.device("cuda:0")
C, L, G = 32, 65536, 8
for B in (4,):
    S = []
    for _ in range(4):
        x = torch.randn(B, C, L, device=dev)
        S.append(dict(x=x, dy=torch.randn(B, C, L, device=dev), st=ops.gn_stats(x, G)))
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dw, db = torch.empty(C, C, 3, device=dev), torch.empty(C, device=dev)
    for mode, gap in (("0", "0"), ("1", "0"), ("1", "60"), ("1", "120"), ("1", "200")):
        os.environ["ADP_WGRAD_TILE"] = mode
        os.environ["ADP_WGRAD_TILE_GAP"] = gap
        row = []
        for nset in (1, 4):
            t1 = graph_time([(lambda s: (lambda: ops.conv1d_wgrad(s["x"], s["dy"], 3, pad=1, prologue=1, pro_stats=s["st"], pro_gamma=gam, pro_beta=bet, groups=G, dw=dw, dbias=db)))(S[i % nset]) for i in range(20)])
            t0 = graph_time([(lambda s: (lambda: ops.conv1d_wgrad(s["x"], s["dy"], 3, pad=1, dw=dw, dbias=db)))(S[i % nset]) for i in range(20)])
            row.append(f"pro1 {t1:5.1f} us  pro0 {t0:5.1f} us ({2 * 4 * B * C * L / t0 / 1e6:4.2f} TB/s)")
        print(f"B{B} ADP_WGRAD_TILE={mode} gap {gap:>3}  warm: {row[0]}   cold: {row[1]}", flush=True)
