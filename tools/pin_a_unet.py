"""Pins oracle/a_unet_restatement.py to the REAL a_unet the moment that package is importable.

The block arithmetic of the reference lives in the third-party package `a-unet` (/root/reference/setup.py:20, imported
at /root/reference/audio_diffusion_pytorch/components.py:5-24), which is absent from this image and not installable
offline -- the a_unet half of the oracle is therefore a recollection and its parity is UNPINNED.
This script is the recipe that closes the gap on any machine where `import a_unet` works:

  python tools/pin_a_unet.py              build the reference's own UNetV0 (components.py:34-105 over the real a_unet)
                                          for a set of small configurations, map its state_dict onto the restatement's,
                                          compare outputs and parameter gradients, and -- when they agree to 1e-5 --
                                          write tests/golden/a_unet_golden.pt (inputs, real outputs / gradients, the
                                          real weights) and tests/golden/a_unet_keymap.json (real key -> oracle key),
                                          which tests/test_pin_a_unet.py then checks on every machine, a_unet or not.
  python tools/pin_a_unet.py --self-test  the same machinery with the restatement standing in for a_unet (renamed
                                          keys), so the mapping / comparison code is exercised in this image too.

A mismatch names the first diverging configuration; the [switch] constants at the top of the restatement (and their
mirror unet.A_UNET_SEMANTICS in the product) are the places to flip.
"""
import importlib.util
import json
import os
import sys
import types
from collections import OrderedDict, defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden", "a_unet_golden.pt")
KEYMAP = os.path.join(ROOT, "tests", "golden", "a_unet_keymap.json")
REFERENCE_ROOT = "/root/reference"

# Small configurations with DISTINCT channel counts per depth (the shape + order matching below is then unambiguous),
# covering every item type UNetV0 can build.
CONFIGS = OrderedDict(
    plain=dict(in_channels=2, channels=[8, 16, 32], factors=[1, 2, 4], items=[1, 2, 1], modulation_features=24),
    attention=dict(in_channels=2, channels=[8, 16, 32], factors=[2, 2, 2], items=[1, 1, 2], modulation_features=24,
                   attentions=[0, 1, 1], cross_attentions=[0, 0, 1], attention_heads=2, attention_features=8,
                   embedding_features=12),
    upsampler_shape=dict(in_channels=4, out_channels=2, channels=[8, 24], factors=[1, 4], items=[1, 1],
                         modulation_features=16),
    skipcat=dict(in_channels=2, channels=[8, 16], factors=[2, 2], items=[1, 1], use_modulation=False,
                 use_time_conditioning=False),
)


def a_unet_available() -> bool:
    return importlib.util.find_spec("a_unet") is not None


def load_reference_unetv0():
    """The reference's UNetV0 factory (components.py) over the real a_unet.  components.py also imports torchaudio for
    its MelSpectrogram plugin; a stub module stands in when torchaudio is missing (never touched by UNetV0)."""
    if not a_unet_available():
        raise RuntimeError("a_unet is not importable here; run with --self-test, or install a-unet and re-run")
    if importlib.util.find_spec("torchaudio") is None:
        stub = types.ModuleType("torchaudio")
        stub.transforms = types.ModuleType("torchaudio.transforms")
        sys.modules.setdefault("torchaudio", stub)
        sys.modules.setdefault("torchaudio.transforms", stub.transforms)
    name = "_ref_adp_components"
    pkg = types.ModuleType(name)
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "audio_diffusion_pytorch")]
    sys.modules[name] = pkg
    spec = importlib.util.spec_from_file_location(f"{name}.components",
                                                  os.path.join(REFERENCE_ROOT, "audio_diffusion_pytorch", "components.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"{name}.components"] = mod
    spec.loader.exec_module(mod)
    return lambda **cfg: mod.UNetV0(dim=1, **cfg)


def match_state_dicts(real_sd, oracle_sd):
    """real key -> oracle key.  Parameters are matched by SHAPE, and inside one shape class by registration order
    (both trees register a depth's down items before its up items, conv1 before conv2, ...).  Raises when the shape
    multisets differ -- the first structural evidence that a [switch] is set wrongly."""
    by_shape_r, by_shape_o = defaultdict(list), defaultdict(list)
    for k, v in real_sd.items():
        by_shape_r[tuple(v.shape)].append(k)
    for k, v in oracle_sd.items():
        by_shape_o[tuple(v.shape)].append(k)
    if {s: len(v) for s, v in by_shape_r.items()} != {s: len(v) for s, v in by_shape_o.items()}:
        only_r = {s: len(v) for s, v in by_shape_r.items() if len(v) != len(by_shape_o.get(s, []))}
        only_o = {s: len(v) for s, v in by_shape_o.items() if len(v) != len(by_shape_r.get(s, []))}
        raise AssertionError(f"parameter shapes differ: real {only_r} vs restatement {only_o}")
    return {rk: ok for s in by_shape_r for rk, ok in zip(by_shape_r[s], by_shape_o[s])}


def example_inputs(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    L = 64 * max(1, int(torch.tensor(cfg["factors"]).prod().item()) // 8)
    kw = {}
    if cfg.get("embedding_features"):
        kw["embedding"] = torch.randn(2, 5, cfg["embedding_features"], generator=g)
    if not cfg.get("use_time_conditioning", True) and cfg.get("use_modulation", True):
        kw["features"] = torch.randn(2, cfg["modulation_features"], generator=g)
    x = torch.randn(2, cfg["in_channels"], L, generator=g)
    t = torch.tensor([0.25, 0.7]) if cfg.get("use_time_conditioning", True) else None
    gy = torch.randn(2, cfg.get("out_channels", cfg["in_channels"]), L, generator=g)
    return x, t, kw, gy


def run(net, x, t, kw, gy):
    for p in net.parameters():
        p.grad = None
    y = net(x, t, **kw) if t is not None else net(x, **kw)
    y.backward(gy)
    return y.detach(), OrderedDict((n, p.grad.detach().clone()) for n, p in net.named_parameters() if p.grad is not None)


def rel(a, b):
    d = b.abs().max().item()
    return (a - b).abs().max().item() / (d if d > 0 else 1.0)


def pin(real_factory, tol=1e-5, verbose=True, check_order=True):
    """Compares the restatement with `real_factory(**cfg)` on every configuration; returns the golden payload."""
    from oracle.a_unet_restatement import UNetV0Oracle
    payload = {"configs": {}, "tolerance": tol}
    for name, cfg in CONFIGS.items():
        torch.manual_seed(0)
        real = real_factory(**cfg)
        oracle = UNetV0Oracle(**cfg)
        real_sd = OrderedDict((k, v) for k, v in real.state_dict().items() if v.dtype.is_floating_point)
        keymap = match_state_dicts(real_sd, oracle.state_dict())
        oracle.load_state_dict({keymap[k]: v for k, v in real_sd.items()})
        x, t, kw, gy = example_inputs(cfg)
        y_r, g_r = run(real, x, t, kw, gy)
        y_o, g_o = run(oracle, x, t, kw, gy)
        e_y = rel(y_o, y_r)
        real_param_names = {n for n, _ in real.named_parameters()}
        e_g = max(rel(g_o[keymap[k]], g_r[k]) for k in g_r if k in real_param_names and keymap[k] in g_o)
        if verbose:
            print(f"[pin_a_unet] {name:16s} output rel err {e_y:.2e}   worst parameter-gradient rel err {e_g:.2e}")
        if not (e_y < tol and e_g < 100 * tol):
            raise AssertionError(f"configuration {name!r}: the restatement does not reproduce the real a_unet "
                                 f"(output {e_y:.2e}, gradients {e_g:.2e}); check the [switch] constants")
        # the product's positional checkpoint loader (UNetV0Net.a_unet_key_order): does the real registration order match the
        # recalled one?  (compared through the shape-and-order key map above, which does not depend on it)
        import audio_diffusion_pytorch_amd as adp
        recalled = adp.UNetV0(dim=1, **cfg).a_unet_key_order()
        real_order = [keymap[k] for k in real_sd]
        if verbose and check_order and recalled != real_order:
            first = next(i for i, (a, b) in enumerate(zip(recalled, real_order)) if a != b)
            print(f"[pin_a_unet] {name:16s} NOTE: a_unet registers entry {first} as {real_order[first]!r}, "
                  f"UNetV0Net.a_unet_key_order() has {recalled[first]!r} there -- fix the order in unet.py")
        payload["configs"][name] = {"cfg": cfg, "keymap": keymap, "key_order_matches": recalled == real_order, "state_dict": real_sd, "x": x, "t": t, "kw": kw,
                                    "gy": gy, "y": y_r, "grads": {k: v for k, v in list(g_r.items())[:8]}}
    return payload


def _self_test_factory(**cfg):
    """The restatement under foreign key names (same registration order), standing in for a_unet."""
    from oracle.a_unet_restatement import UNetV0Oracle

    class Renamed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.wrapped_net = UNetV0Oracle(**cfg)

        def forward(self, *a, **k):
            return self.wrapped_net(*a, **k)
    return Renamed()


def main():
    self_test = "--self-test" in sys.argv
    payload = pin(_self_test_factory if self_test else load_reference_unetv0(), check_order=not self_test)
    if self_test:
        print("[pin_a_unet] self-test ok (nothing written: the restatement compared with itself pins nothing)")
        return
    import a_unet
    payload["a_unet_version"] = getattr(a_unet, "__version__", "unknown")
    torch.save(payload, GOLDEN)
    with open(KEYMAP, "w") as f:
        json.dump({n: c["keymap"] for n, c in payload["configs"].items()}, f, indent=1)
    print(f"[pin_a_unet] PINNED against a_unet {payload['a_unet_version']}: wrote {GOLDEN} and {KEYMAP}; remove the "
          f"'PARITY UNPINNED' notes in oracle/a_unet_restatement.py and DESIGN.md section 2")


if __name__ == "__main__":
    main()
