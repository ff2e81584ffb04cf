"""Interleaved A/B/C... of environment settings on the headline training step (hipGraph replay) at one batch size.
usage: python tools/multi_ab.py BATCH ROUNDS "A=1,B=2" "A=0" ...   (an empty string = the defaults)"""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audio_diffusion_pytorch_amd as adp  # noqa: E402


def main():
    B, rounds = int(sys.argv[1]), int(sys.argv[2])
    settings = sys.argv[3:]
    names = sorted({kv.split("=")[0] for s in settings for kv in s.split(",") if kv})
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS,
                               items=bench.ITEMS).to(dev)
    x = torch.randn(B, 2, bench.LENGTH, device=dev)

    def zero():
        for p in model.parameters():
            p.grad = None

    def one():
        zero()
        model(x).backward()
    res = {s: [] for s in settings}
    for r in range(rounds):
        for s in settings:
            for n in names:
                os.environ.pop(n, None)
            for kv in s.split(","):
                if kv:
                    k, v = kv.split("=")
                    os.environ[k] = v
            replay = bench._graphed(one, zero)
            dt = bench._time(replay, 20)
            torch.cuda.synchronize()
            res[s].append(dt * 1e3)
            del replay
    for s in settings:
        print(f"batch {B} [{s or 'defaults'}]: " + " ".join(f"{t:.3f}" for t in res[s]) + f"  min {min(res[s]):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
