"""UpsampleItem conv microbench on the GPU box (hipGraph replay, us per call): forward (nearest xF + kernel 3 + SkipModulate merge
epilogue), data gradient (pooled store), weight gradient (+ second stage) at [B, R, L] -> [B, M, F*L].
usage: python tools/up_bench.py [R M L F]   (default 32 8 65536 4 = the depth-1 -> depth-0 UpsampleItem of the bench)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from tools.tile_bench import graph_time  # noqa: E402


def main():
    R, M, L, Fu = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (32, 8, 65536, 4)
    B, dev = 4, torch.device("cuda:0")
    S = []
    for _ in range(2):
        S.append(dict(h=torch.randn(B, R, L, device=dev), skip=torch.randn(B, M, Fu * L, device=dev),
                      u=torch.empty(B, M, Fu * L, device=dev), y=torch.empty(B, M, Fu * L, device=dev),
                      du=torch.randn(B, M, Fu * L, device=dev)))
    w, bias = torch.randn(M, R, 3, device=dev) * 0.1, torch.randn(M, device=dev)
    sc = torch.randn(B, M, device=dev)
    dw, db = torch.empty(M, R, 3, device=dev), torch.empty(M, device=dev)
    fwd = [(lambda s: (lambda: ops.conv1d(s["h"], w, bias, pad=1, up=Fu, e_scale=sc.view(-1), e_bstride=M, res=s["skip"],
                                          out_pre=s["u"], out=s["y"])))(S[i & 1]) for i in range(20)]
    dgr = [(lambda s: (lambda: ops.conv1d(s["du"], w, None, pad=1, transposed=True, store=2, sp=Fu)))(S[i & 1]) for i in range(20)]
    wgr = [(lambda s: (lambda: ops.conv1d_wgrad(s["h"], s["du"], 3, pad=1, up=Fu, dw=dw, dbias=db)))(S[i & 1]) for i in range(20)]
    a_in, a_out = 4 * B * R * L / 1e6, 4 * B * M * Fu * L / 1e6
    print(f"[{B},{R},{L}] x{Fu} -> [{B},{M},{Fu * L}]  ({a_in:.1f} MB in, {a_out:.1f} MB out): "
          f"forward+merge {graph_time(fwd):5.1f} us ({a_in + 3 * a_out:.0f} MB)  data gradient {graph_time(dgr):5.1f} us "
          f"({a_in + a_out:.0f} MB)  weight gradient (2 launches) {graph_time(wgr):5.1f} us ({a_in + a_out:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
