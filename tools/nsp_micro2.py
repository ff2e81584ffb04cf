"""Kernel work: wide-N blocks under the 1x1 convs and the nearest-upsample convs (ADP_MM_NSP 1 / 2 / 4), isolated launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402
from conv_family_ab import timeit  # noqa: E402

B = 4
dev = torch.device("cuda:0")
cases = [  # R, M, L, KT, tr, up
    (64, 128, 16384, 1, True, 1), (128, 256, 4096, 1, True, 1), (256, 256, 2048, 1, True, 1), (512, 512, 1024, 1, True, 1),
    (512, 1024, 512, 1, True, 1), (1024, 1024, 256, 1, True, 1), (128, 64, 16384, 1, False, 1), (256, 128, 4096, 1, False, 1),
    (512, 256, 1024, 3, False, 2), (512, 512, 512, 3, False, 2), (256, 128, 2048, 3, False, 2), (128, 64, 4096, 3, False, 4),
]
for R, M, L, KT, tr, up in cases:
    x = torch.randn(B, R, L, device=dev)
    w = (torch.randn(R, M, KT, device=dev) if tr else torch.randn(M, R, KT, device=dev)) * 0.05
    res = torch.randn(B, M, L * up, device=dev)
    fl = 2 * B * M * R * KT * L * up
    line = f"R{R:5d} M{M:5d} L{L:6d} KT{KT} tr{int(tr)} up{up}:"
    for nsp in ("1", "2", "4"):
        os.environ["ADP_MM_NSP"] = nsp
        t = timeit(lambda: ops.conv1d(x, w, None, pad=(KT - 1) // 2, up=up, transposed=tr, res=res))
        line += f"  nsp{nsp} {t:6.1f} us {fl / t / 1e6:6.1f} TF"
    print(line, flush=True)
