"""Interleaved A/B of one environment switch on the headline training step (hipGraph replay), batch 4 and 1 (GPU box).
usage: python tools/env_ab.py NAME A_VALUE B_VALUE [rounds]   e.g.  python tools/env_ab.py ADP_WGRAD_SIDE 0 1"""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audio_diffusion_pytorch_amd as adp  # noqa: E402


def main():
    name, va, vb = sys.argv[1:4]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    dev = torch.device("cuda:0")
    for B in (4, 1):
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
        grads = {}
        for r in range(rounds):
            for v in (va, vb):
                os.environ[name] = v
                replay = bench._graphed(one, zero)
                dt = bench._time(replay, 20)
                torch.cuda.synchronize()
                grads[v] = [p.grad.clone() for p in model.parameters()]
                print(f"batch {B} round {r} {name}={v}: {dt * 1e3:.3f} ms", flush=True)
        same = all(torch.equal(a, b) for a, b in zip(grads[va], grads[vb]))
        print(f"batch {B}: gradients of the two settings bit-identical: {same}", flush=True)
        del model
        torch.cuda.empty_cache()
    os.environ.pop(name, None)


if __name__ == "__main__":
    main()
