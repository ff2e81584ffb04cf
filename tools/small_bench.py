"""Per-launch cost (us, hipGraph replay of 20 back-to-back launches, launch boundary included) of the small normalisation kernels at
the bench step's shapes (GPU box; kernel work).  usage: python tools/small_bench.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from tools.tile_bench import graph_time  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    G = 8
    print(f"batch {B}: us per launch (graph, 20 launches, 2 buffer sets alternating)")
    for C, L in ((32, 65536), (64, 16384), (128, 4096), (256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128)):
        S = []
        for _ in range(2):
            x = torch.randn(B, C, L, device=dev)
            S.append(dict(x=x, d=torch.randn(B, C, L, device=dev), r=torch.randn(B, C, L, device=dev), o=torch.empty(B, C, L, device=dev),
                          st=ops.gn_stats(x, G), ss=torch.randn(B, 3 * C, device=dev) * 0.1, dss=torch.empty(B, 3 * C, device=dev)))
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        res = {}
        res["gn_stats"] = graph_time([(lambda s: (lambda: ops.gn_stats(s["x"], G)))(S[i & 1]) for i in range(20)]) * 2 / 2
        res["gn_stats_act"] = graph_time([(lambda s: (lambda: ops.gn_stats_act(s["x"], G, gam, bet)))(S[i & 1]) for i in range(20)])
        res["gn_act"] = graph_time([(lambda s: (lambda: ops.gn_act(s["x"], s["st"], G, gam, bet)))(S[i & 1]) for i in range(20)])
        res["gn_bwd(2)"] = graph_time([(lambda s: (lambda: ops.gn_silu_bwd(s["x"], s["d"], s["st"], gam, bet, G, dres=s["r"], dx=s["o"], dgamma=dg, dbeta=db)))(S[i & 1]) for i in range(20)])
        res["mod_fwd"] = graph_time([(lambda s: (lambda: ops.modulation_fwd(s["x"], s["ss"].view(-1), 3 * C, y=s["o"])))(S[i & 1]) for i in range(20)])
        stats = ops.modulation_fwd(S[0]["x"], S[0]["ss"].view(-1), 3 * C)[1]
        res["mod_bwd"] = graph_time([(lambda s: (lambda: ops.modulation_bwd(s["x"], s["d"], s["ss"].view(-1), 3 * C, stats, s["dss"].view(-1), 3 * C, dx=s["o"])))(S[i & 1]) for i in range(20)])
        res["add"] = graph_time([(lambda s: (lambda: ops.add(s["x"], s["r"], out=s["o"])))(S[i & 1]) for i in range(20)])
        mb = 4 * B * C * L / 1e6
        print(f"[{B},{C},{L}] {mb:5.1f} MB: " + "  ".join(f"{k} {v:5.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
