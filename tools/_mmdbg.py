import os, sys, torch
sys.path.insert(0, "/root/repo")
from audio_diffusion_pytorch_amd import ops, _C
from tools.conv_family_ab import timeit
dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for C, L in ((1024, 256), (1024, 128), (512, 1024), (512, 512), (256, 2048)):
    B = 4
    x = torch.randn(B, C, L, device=dev); w = torch.randn(C, C, 3, device=dev) * 0.05
    bias = torch.randn(C, device=dev); res = torch.randn(B, C, L, device=dev)
    row = f"C{C} L{L}:"
    for rep in range(2):
        for tag in ("old", "new"):
            _C._testing_use_library(os.path.join(ROOT, "tools", "ab", f"lib_{tag}.so"), allow_cpu=False)
            t1 = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res))
            t2 = timeit(lambda: ops.conv1d(x, w, bias, pad=1, gn=ops.GnPart()))
            t3 = timeit(lambda: ops.conv1d(x, w, None, pad=1, transposed=True))
            row += f"  {tag}: fwd+res {t1:5.1f} fwd+gn {t2:5.1f} dgrad {t3:5.1f} |"
    print(row, flush=True)
