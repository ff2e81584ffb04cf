"""BASELINE config 3: VSampler.sample, num_steps=50, noise [B, 2, 2**18], inference only, one hipGraph-captured step
replayed per iteration.  Prints sampler steps/s (1 step = 1 U-Net forward + the rotation kernel)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--graph", type=int, default=1)
    a = ap.parse_args()
    import audio_diffusion_pytorch_amd as adp
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS,
                               items=bench.ITEMS, sampler_use_graph=bool(a.graph)).to(dev)
    noise = torch.randn(a.batch, 2, bench.LENGTH).to(dev)
    model.sample(noise, num_steps=2)  # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.sample(noise, num_steps=a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "sampler steps/s (VSampler, UNetV0 forward only)", "value": round(a.steps / dt, 2),
                      "ms_per_step": round(dt / a.steps * 1e3, 3), "batch": a.batch, "num_steps": a.steps,
                      "launch": "hipGraph replay" if a.graph else "eager", "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
