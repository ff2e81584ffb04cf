"""Eager steps of one of bench.py's extra layouts, for `rocprofv3 --kernel-trace --stats` (GPU box).
usage: python tools/cfg_prof.py {config4|readme_attention|batch1} [steps] [batch]
Prints the replayed (hipGraph) step time too, so a kernel table and the step it belongs to come from one process."""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audio_diffusion_pytorch_amd as adp  # noqa: E402

LAYOUTS = {
    "batch1": (dict(), False),
    "readme_attention": (dict(attentions=[0, 0, 0, 0, 0, 1, 1, 1, 1], attention_heads=8, attention_features=64), False),
    "config4": (dict(cross_attentions=[0, 0, 0, 1, 1, 1, 1, 1, 1], embedding_features=768, attention_heads=8,
                     attention_features=64), True),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    extra, use_emb = LAYOUTS[name]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS,
                           items=bench.ITEMS, **extra).to(dev)
    x = torch.randn(B, 2, bench.LENGTH, device=dev)
    kw = dict(embedding=torch.randn(B, 64, 768, device=dev)) if use_emb else {}

    def zero():
        for p in m.parameters():
            p.grad = None

    def step():
        zero()
        m(x, **kw).backward()
    for _ in range(steps + 1):  # one warm step + `steps` eager ones (what the kernel table averages over)
        step()
    torch.cuda.synchronize()
    if os.environ.get("ADP_CFG_PROF_REPLAY", "1") != "0":
        dt = bench._time(bench._graphed(step, zero), 10)
        print(f"{name} batch {B}: replayed step {dt * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
