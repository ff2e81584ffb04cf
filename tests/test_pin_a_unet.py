"""The recipe that pins the a_unet half of the oracle (tools/pin_a_unet.py): its machinery is exercised here with the
restatement standing in for a_unet; the live comparison and the committed-fixture check run wherever a_unet / the fixture
exist (neither does in the build image: a_unet is not installable offline -- SURVEY.md section 8c)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pin_a_unet  # noqa: E402


def test_pin_machinery_self_test():
    """State-dict matching by shape + registration order, weight transfer, output / gradient comparison: run end to end
    with the restatement under foreign key names."""
    payload = pin_a_unet.pin(pin_a_unet._self_test_factory, verbose=False)
    assert set(payload["configs"]) == set(pin_a_unet.CONFIGS)
    km = payload["configs"]["plain"]["keymap"]
    assert all(k.startswith("wrapped_net.") and k[len("wrapped_net."):] == v for k, v in km.items())


def test_pin_detects_a_different_arithmetic():
    """A stand-in "a_unet" whose arithmetic differs from the restatement's (here: GELU instead of SiLU inside the
    ResnetBlocks) has the same parameters but other outputs: the recipe must refuse to pin."""
    import torch.nn.functional as F

    def factory(**cfg):
        net = pin_a_unet._self_test_factory(**cfg)
        orig = F.silu

        class Patched(type(net)):
            def forward(self, *a, **k):
                F.silu = F.gelu
                try:
                    return super().forward(*a, **k)
                finally:
                    F.silu = orig
        net.__class__ = Patched
        return net
    with pytest.raises(AssertionError, match="does not reproduce"):
        pin_a_unet.pin(factory, verbose=False)


def test_shape_mismatch_is_reported():
    a = {"w": torch.zeros(3, 4), "b": torch.zeros(3)}
    b = {"x.w": torch.zeros(3, 4), "x.b": torch.zeros(4)}
    with pytest.raises(AssertionError, match="parameter shapes differ"):
        pin_a_unet.match_state_dicts(a, b)


@pytest.mark.skipif(not os.path.exists(pin_a_unet.GOLDEN), reason="tests/golden/a_unet_golden.pt has not been generated "
                    "(needs a machine with a_unet: python tools/pin_a_unet.py)")
def test_restatement_matches_committed_a_unet_fixture():
    from oracle.a_unet_restatement import UNetV0Oracle
    payload = torch.load(pin_a_unet.GOLDEN)
    for name, c in payload["configs"].items():
        oracle = UNetV0Oracle(**c["cfg"])
        oracle.load_state_dict({c["keymap"][k]: v for k, v in c["state_dict"].items()})
        y, grads = pin_a_unet.run(oracle, c["x"], c["t"], c["kw"], c["gy"])
        assert pin_a_unet.rel(y, c["y"]) < 1e-5, name
        for k, g in c["grads"].items():
            assert pin_a_unet.rel(grads[c["keymap"][k]], g) < 1e-3, (name, k)


@pytest.mark.skipif(not (pin_a_unet.a_unet_available() and os.path.isdir(pin_a_unet.REFERENCE_ROOT)),
                    reason="a_unet is not importable here (not installable offline)")
def test_restatement_matches_live_a_unet():
    pin_a_unet.pin(pin_a_unet.load_reference_unetv0(), verbose=False)
