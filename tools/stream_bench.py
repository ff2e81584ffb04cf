"""conv_stream32 / conv_direct microbench (GPU box): the HBM-bound depth-0/1 ConvBlock convs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops
from tools.conv_bench import timeit

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    for C, L in ((32, 65536), (8, 262144)):
        x = torch.randn(B, C, L, device=dev); w = torch.randn(C, C, 3, device=dev) * 0.1
        bias = torch.randn(C, device=dev); res = torch.randn(B, C, L, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        stats = ops.gn_stats(x, 8)
        A = 4 * x.numel()
        t1 = timeit(lambda: ops.conv1d(x, w, bias, pad=1, prologue=1, pro_stats=stats, pro_gamma=gamma, pro_beta=beta, groups=8))
        t2 = timeit(lambda: ops.conv1d(x, w, bias, pad=1, prologue=1, pro_stats=stats, pro_gamma=gamma, pro_beta=beta, groups=8, res=res))
        t3 = timeit(lambda: ops.conv1d(x, w, None, pad=1, transposed=True))
        tc = timeit(lambda: ops.add(x, res))
        print(f"C{C} L{L} B{B}: conv1 {t1:6.1f} us {2*A/t1/1e3:6.0f} GB/s | conv2+res {t2:6.1f} us {3*A/t2/1e3:6.0f} GB/s | dgrad {t3:6.1f} us {2*A/t3/1e3:6.0f} GB/s | add(2R1W) {tc:6.1f} us {3*A/tc/1e3:6.0f} GB/s", flush=True)

if __name__ == "__main__":
    main()
