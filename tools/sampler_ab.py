"""VSampler step time with / without the hoisted conditioning table, one process (GPU box): BASELINE config 3 ([1,2,2**18], 50 steps).
usage: python tools/sampler_ab.py [batch] [ENV_NAME]   (default switch: ADP_SAMPLER_HOIST)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    noise = torch.randn(B, 2, bench.LENGTH, device=dev)
    outs = {}
    for r in range(2):
        name = sys.argv[2] if len(sys.argv) > 2 else "ADP_SAMPLER_HOIST"
        for hoist in ("0", "1"):
            os.environ[name] = hoist
            model.sampler._graph_cache.clear()  # (the cache key knows nothing of environment switches)
            model.sample(noise, num_steps=3)  # capture for this structure
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                out = model.sample(noise, num_steps=50)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100
            outs[hoist] = out
            print(f"batch {B} round {r} {name}={hoist}: {dt * 1e3:.3f} ms per sampler step", flush=True)
    print("samples bit-identical:", torch.equal(outs["0"], outs["1"]), flush=True)


if __name__ == "__main__":
    main()
