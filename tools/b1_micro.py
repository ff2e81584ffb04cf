"""Kernel work: batch-1 launches of the deep ResnetItem convs under tile / K-split policies (environment knobs of
conv_mm.hip: ADP_MM_MIN_BLOCKS, ADP_MM_NSP_MIN_BLOCKS, ADP_MM_NSP, ADP_MM_KS_TARGET, ADP_MM_KS_MAX, ADP_MM_KS_MINCH).
Times include the split-K reduce launch.  usage: python tools/b1_micro.py [batch]"""
import os
import sys
from ctypes import byref

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import _C, ops  # noqa: E402
from conv_family_ab import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
KEYS = ("ADP_MM_MIN_BLOCKS", "ADP_MM_NSP_MIN_BLOCKS", "ADP_MM_NSP", "ADP_MM_KS_TARGET", "ADP_MM_KS_MAX", "ADP_MM_KS_MINCH")
SETTINGS = [
    ("default", {}),
    ("bm64>=64", {"ADP_MM_MIN_BLOCKS": "64", "ADP_MM_NSP": "1"}),
    ("bm64>=32", {"ADP_MM_MIN_BLOCKS": "32", "ADP_MM_NSP": "1"}),
    ("bm64>=16,ks16", {"ADP_MM_MIN_BLOCKS": "16", "ADP_MM_NSP": "1", "ADP_MM_KS_MAX": "16", "ADP_MM_KS_MINCH": "2"}),
    ("bm64>=32,nsp2>=16", {"ADP_MM_MIN_BLOCKS": "32", "ADP_MM_NSP": "2", "ADP_MM_NSP_MIN_BLOCKS": "16"}),
    ("bm64>=32,nsp2>=16,ks16", {"ADP_MM_MIN_BLOCKS": "32", "ADP_MM_NSP": "2", "ADP_MM_NSP_MIN_BLOCKS": "16", "ADP_MM_KS_MAX": "16",
                                "ADP_MM_KS_MINCH": "2"}),
    ("bm64>=16,nsp4>=8,ks16", {"ADP_MM_MIN_BLOCKS": "16", "ADP_MM_NSP": "4", "ADP_MM_NSP_MIN_BLOCKS": "8", "ADP_MM_KS_MAX": "16",
                               "ADP_MM_KS_MINCH": "2"}),
    ("default,ks16", {"ADP_MM_KS_MAX": "16", "ADP_MM_KS_MINCH": "2"}),
]
dev = torch.device("cuda:0")
for C, L in ((128, 4096), (256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128)):
    x = torch.randn(B, C, L, device=dev)
    w = torch.randn(C, C, 3, device=dev) * 0.05
    bias = torch.randn(C, device=dev)
    res = torch.randn(B, C, L, device=dev)
    fl = 2 * B * C * C * 3 * L
    for name, env in SETTINGS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        d = _C.ConvDesc(_C.ptr(x), None, _C.ptr(w), None, None, None, None, None, None, _C.ptr(x), None, B, C, C, L, C, L, 3, 1, 1,
                        1, 1, 0, 0, 1, 0, 1, 0)
        tile = _C.query("adp_conv1d_tile", byref(d))
        wsb = _C.query("adp_conv1d_ws_bytes", byref(d))
        ks = wsb // (4 * B * C * L) if wsb else 1
        t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res, gn=ops.GnPart()))
        t_t = timeit(lambda: ops.conv1d(x, w, None, pad=1, transposed=True))
        print(f"B{B} C{C:5d} L{L:5d} {name:26s} tile {tile % 1000000:6d} ks {ks:2d}: fwd {t_f:6.1f} us {fl / t_f / 1e6:6.1f} TF | dgrad "
              f"{t_t:6.1f} us {fl / t_t / 1e6:6.1f} TF", flush=True)
