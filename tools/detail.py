"""Per-(kernel, shape) timing table of one instrumented fwd+bwd step (HIP events around every launch) for a model
variant -- kernel-work tool, run on the GPU box:  python tools/detail.py <out.txt> [--batch B] [--variant plain|attn|cfg4] [--fwd]"""
import argparse
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--variant", default="plain")
    ap.add_argument("--fwd", action="store_true", help="inference forward only (the sampler's step)")
    a = ap.parse_args()
    import audio_diffusion_pytorch_amd as adp
    dev = torch.device("cuda:0")
    extra = {"plain": {}, "attn": dict(attentions=[0, 0, 0, 0, 0, 1, 1, 1, 1], attention_heads=8, attention_features=64),
             "cfg4": dict(cross_attentions=[0, 0, 0, 1, 1, 1, 1, 1, 1], embedding_features=768, attention_heads=8,
                          attention_features=64)}[a.variant]
    torch.manual_seed(0)
    m = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=bench.CHANNELS, factors=bench.FACTORS,
                           items=bench.ITEMS, **extra).to(dev)
    x = torch.randn(a.batch, 2, bench.LENGTH, device=dev)
    kw = dict(embedding=torch.randn(a.batch, 64, 768, device=dev)) if a.variant == "cfg4" else {}
    if a.fwd:
        t = torch.full((a.batch,), 0.5, device=dev)

        def step():
            with torch.no_grad():
                m.net(x, t, **kw)
    else:
        def step():
            m(x, **kw).backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    recs = bench.profiled_step(m, step)
    det, total = {}, 0.0
    for call, kern, meta, ms in recs:
        e = det.setdefault(kern + " :: " + meta.get("shape", ""), [0, 0.0, 0, 0])
        e[0] += 1
        e[1] += ms
        e[2] += meta.get("flops", 0)
        e[3] += meta.get("bytes", 0)
        total += ms
    with open(a.out, "w") as f:
        f.write(f"# variant={a.variant} batch={a.batch} fwd_only={a.fwd}: {len(recs)} launches, {total:.3f} ms of kernels\n")
        for k, (n, ms, fl, by) in sorted(det.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{ms:8.3f} ms  n={n:3d}  avg {ms / n * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF  {by / ms / 1e6:7.1f} GB/s  {k}\n")
    print(open(a.out).read()[:3000])


if __name__ == "__main__":
    main()
