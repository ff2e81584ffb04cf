"""A/B of the conv families behind adp_conv1d / adp_conv1d_wgrad on the GPU box (kernel work, not part of the product):

  python tools/conv_family_ab.py micro [batch]   isolated launches of the wide ResnetItem shapes, forward / data gradient /
                                                 weight gradient, per ADP_CONV_WINO mode (0 = direct form, 1 = Winograd F(2,3)
                                                 in the MMA waves' registers)
  python tools/conv_family_ab.py step [rounds]   the headline training step replayed from a hipGraph under a list of
                                                 environment settings, interleaved
"""
import os

os.environ.setdefault("ADP_TRAIN_GRAPH", "0")  # these tools launch / capture the step themselves (graphed.py is the README loop's path)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_diffusion_pytorch_amd import ops  # noqa: E402


def timeit(fn, n=20):
    """us per call, n calls replayed from one hipGraph (eager event timing is host-bound below ~12 us per launch)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3


def setenv(env):
    for k in ("ADP_CONV_WINO", "ADP_WINO_MIN_R", "ADP_WINO_WGRAD_MIN_R", "ADP_ACT_MATERIALIZE_MIN_C", "ADP_MM_NSP"):
        os.environ.pop(k, None)
    os.environ.update(env)


def micro():
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    shapes = ((128, 4096), (256, 2048), (512, 1024), (512, 512), (1024, 256), (1024, 128))
    for C, L in shapes:
        x = torch.randn(B, C, L, device=dev)
        dy = torch.randn(B, C, L, device=dev)
        w = torch.randn(C, C, 3, device=dev) * 0.05
        bias = torch.randn(C, device=dev)
        res = torch.randn(B, C, L, device=dev)
        fl = 2 * B * C * C * 3 * L
        ref = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding=1) + res.double()
        for mode in ("0", "1"):
            setenv({"ADP_CONV_WINO": mode, "ADP_WINO_MIN_R": "64", "ADP_WINO_WGRAD_MIN_R": "64"})
            out = ops.conv1d(x, w, bias, pad=1, res=res)
            err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
            t_f = timeit(lambda: ops.conv1d(x, w, bias, pad=1, res=res))
            t_t = timeit(lambda: ops.conv1d(dy, w, None, pad=1, transposed=True))
            t_w = timeit(lambda: ops.conv1d_wgrad(x, dy, 3, pad=1))
            print(f"B{B} C{C:5d} L{L:5d} mode {mode}: fwd {t_f:6.1f} us {fl / t_f / 1e6:6.1f} TF | dgrad {t_t:6.1f} us "
                  f"{fl / t_t / 1e6:6.1f} TF | wgrad(+reduce) {t_w:6.1f} us {fl / t_w / 1e6:6.1f} TF | fwd err vs fp64 {err:.1e}",
                  flush=True)


def step():
    import bench
    import audio_diffusion_pytorch_amd as adp
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    settings = [
        ("direct", {"ADP_CONV_WINO": "0"}),
        ("wino", {}),
        ("wino conv only", {"ADP_WINO_WGRAD_MIN_R": "100000"}),
        ("wino wgrad only", {"ADP_WINO_MIN_R": "100000"}),
        ("wino all>=32", {"ADP_WINO_MIN_R": "32", "ADP_WINO_WGRAD_MIN_R": "32"}),
        ("wino all>=128", {"ADP_WINO_MIN_R": "128", "ADP_WINO_WGRAD_MIN_R": "128"}),
        ("wino all>=256", {"ADP_WINO_MIN_R": "256", "ADP_WINO_WGRAD_MIN_R": "256"}),
    ]
    extra = os.environ.get("AB_SETTINGS")
    if extra:  # name=K:V,K:V;name2=...
        settings = []
        for item in extra.split(";"):
            name, _, kv = item.partition("=")
            settings.append((name, dict(p.split(":") for p in kv.split(",") if p)))
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
        best = {}
        for r in range(rounds):
            for name, env in settings:
                setenv(env)
                dt = bench._time(bench._graphed(one, zero), 20) * 1e3
                best[name] = min(best.get(name, 1e9), dt)
                print(f"batch {B} round {r} {name:22s}: {dt:.3f} ms", flush=True)
        print(f"batch {B} best: " + ", ".join(f"{k} {v:.3f}" for k, v in best.items()), flush=True)
        del model
        torch.cuda.empty_cache()
    setenv({})


if __name__ == "__main__":
    {"micro": micro, "step": step}[sys.argv[1]]()
