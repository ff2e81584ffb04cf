"""A/B of two builds of libadp_hip.so on one GPU box (kernel work): tools/ab/lib_old.so vs tools/ab/lib_new.so, the
headline training step replayed from a hipGraph, interleaved.  usage: python tools/ab_lib.py [rounds]"""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from audio_diffusion_pytorch_amd import _C  # noqa: E402
import audio_diffusion_pytorch_amd as adp  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = {"old": [], "new": []}
    for B in (4, 1):
        model = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS,
                                   items=bench.ITEMS).to(dev)
        x = torch.randn(B, 2, bench.LENGTH, device=dev)

        def zero():
            for p in model.parameters():
                p.grad = None

        def step():
            zero()
            model(x).backward()
        for r in range(rounds):
            for tag in ("old", "new"):
                _C._testing_use_library(os.path.join(ROOT, "tools", "ab", f"lib_{tag}.so"), allow_cpu=False)
                dt = bench._time(bench._graphed(step, zero), 20)
                print(f"batch {B} round {r} {tag}: {dt * 1e3:.3f} ms", flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
