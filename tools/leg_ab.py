"""Interleaved A/B of switches on one of bench.py's layouts, ONE process (GPU box): every setting re-captures the step's hipGraph.
usage: python tools/leg_ab.py {headline|batch1|config4|readme_attention} NAME=v1,v2 [NAME=...] [--rounds N] [--batch B]
  NAME = an environment variable read per call by the library (e.g. ADP_ATTN_MERGE=0,1), or unet.ATTR for a module switch of
  audio_diffusion_pytorch_amd/unet.py read at import (e.g. unet.MOD_LN_BWD_FUSE=0,1).  Settings are taken one at a time
  against the defaults (not the cross product)."""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import audio_diffusion_pytorch_amd as adp  # noqa: E402
from audio_diffusion_pytorch_amd import unet as unet_mod  # noqa: E402
from tools.cfg_prof import LAYOUTS  # noqa: E402


def apply(name, value):
    if name.startswith("unet."):
        setattr(unet_mod, name[5:], value not in ("0", "False", ""))
    elif value == "unset":
        os.environ.pop(name, None)
    else:
        os.environ[name] = value


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
    args = [a for a in args if not a.isdigit()]
    switches = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args[1:]]
    for leg in args[0].split(","):  # several layouts in one process: headline,batch1,config4
        B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else (4 if leg == "headline" else 1)
        run_leg(leg, B, switches, rounds)


def run_leg(leg, B, switches, rounds):
    extra, use_emb = LAYOUTS["batch1" if leg == "headline" else leg]
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
    for _ in range(2):
        step()
    for name, values in switches:
        for r in range(rounds):
            for v in values:
                apply(name, v)
                dt = bench._time(bench._graphed(step, zero), 20)
                print(f"{leg} batch {B} round {r} {name}={v}: {dt * 1e3:.3f} ms", flush=True)
        apply(name, values[-1] if name.startswith("unet.") else "unset")


if __name__ == "__main__":
    main()
