"""Pins the oracle (test infrastructure) before it is trusted, and the product's host logic with it:
  1. oracle/vdiffusion.py vs the committed golden fixtures generated from the LIVE reference
     (tests/golden/make_golden.py) -- and vs the live reference itself when /root/reference is present;
  2. the product VDiffusion / VSampler / utils (running on the emulated kernels) vs the same fixtures;
  3. structural checks of the a_unet restatement against the reference's call-site contract.
"""
import os
import sys

import pytest
import torch

import audio_diffusion_pytorch_amd as adp
from audio_diffusion_pytorch_amd import utils as putils
from conftest import rel_err
from oracle import vdiffusion as ovd
from oracle.a_unet_restatement import UNetV0Oracle
from oracle.reference_loader import load_reference, reference_available

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import StubNet  # noqa: E402

GOLD = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vdiffusion_golden.pt"))


def test_golden_endpoint_constants():
    assert GOLD["linspace_51"][1].item() == pytest.approx(0.98, abs=1e-7)
    assert GOLD["alpha_51"][0].item() == pytest.approx(-4.3711388e-08, rel=1e-6)
    assert GOLD["beta_51"][0].item() == 1.0 and GOLD["alpha_51"][-1].item() == 1.0 and GOLD["beta_51"][-1].item() == 0.0
    a, b = ovd.alpha_beta(ovd.linear_schedule(51))
    assert torch.equal(a, GOLD["alpha_51"]) and torch.equal(b, GOLD["beta_51"])


def test_oracle_vdiffusion_matches_golden():
    net = StubNet()
    loss = ovd.v_loss(net, GOLD["vd_x"], GOLD["vd_noise"], GOLD["vd_sigmas"])
    assert torch.equal(loss.detach(), GOLD["vd_loss"])
    for steps in (1, 5, 50):
        out = ovd.v_sample(net, GOLD["vs_noise"], steps)
        assert torch.equal(out, GOLD[f"vs_out_{steps}"]), steps


def test_oracle_resample_matches_golden():
    w = GOLD["rs_in"]
    for f in (2, 4, 16):
        assert torch.equal(ovd.downsample(w, f), GOLD[f"rs_down_{f}"])
        assert torch.equal(ovd.upsample(w[..., :64], f), GOLD[f"rs_up_{f}"])


def test_product_resample_matches_golden(dev):
    """utils.resample on the polyphase HIP kernel (adp_resample) against the live reference's outputs (fixtures):
    same taps, same padding / interleave / crop index math; only the fp32 summation order differs from conv1d."""
    w = GOLD["rs_in"]
    for f in (2, 4, 16):
        down = putils.downsample(w.to(dev), f)
        assert down.shape == GOLD[f"rs_down_{f}"].shape
        assert rel_err(down, GOLD[f"rs_down_{f}"]) < 1e-5, f
        up = putils.upsample(w[..., :64].contiguous().to(dev), f)
        assert up.shape == GOLD[f"rs_up_{f}"].shape
        assert rel_err(up, GOLD[f"rs_up_{f}"]) < 1e-5, f
    # a length that is not a multiple of the factor, several rows, and the kernel bank equals the reference's bits
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 1000, generator=g)
    assert rel_err(putils.resample(x.to(dev), 3, 2), ovd.resample(x, 3, 2)) < 1e-5
    assert rel_err(putils.downsample(x.to(dev), 16), ovd.downsample(x, 16)) < 1e-5


def test_host_helpers_match_golden():
    assert putils.groupby("diffusion_", dict(diffusion_a=1, sampler_b=2, c=3)) == GOLD["groupby"]
    assert [putils.closest_power_2(v) for v in (3.0, 5.9, 6.1, 1000.0)] == GOLD["closest_power_2"]
    assert putils.default(None, 3) == 3 and putils.default(0, 3) == 0 and putils.default(None, lambda: 7) == 7


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_oracle_matches_live_reference():
    D, U = load_reference()
    net = StubNet()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 2, 96, generator=g)
    assert torch.equal(D.VSampler(net)(x, num_steps=7), ovd.v_sample(net, x, 7))
    sig = torch.tensor([0.21, 0.83])

    class Fixed(D.Distribution):
        def __call__(self, num_samples, device=torch.device("cpu")):
            return sig

    torch.manual_seed(3)
    noise = torch.randn_like(x)
    torch.manual_seed(3)
    assert torch.equal(D.VDiffusion(net, sigma_distribution=Fixed())(x), ovd.v_loss(net, x, noise, sig))
    w = torch.randn(1, 2, 300, generator=g)
    assert torch.equal(U.downsample(w, 3), ovd.downsample(w, 3)) and torch.equal(U.upsample(w, 3), ovd.upsample(w, 3))
    mask = torch.rand(2, 2, 96, generator=g) > 0.5
    torch.manual_seed(11)
    ref = D.VInpainter(net)(x, mask, num_steps=5, num_resamples=2)
    torch.manual_seed(11)
    assert torch.equal(ref, ovd.v_inpaint(net, x, mask, 5, 2))


def test_oracle_inpainter_matches_golden():
    """VInpainter: the oracle restatement is bit-equal to the live reference's fixtures."""
    net = StubNet()
    for steps, res in ((4, 3), (6, 1)):
        torch.manual_seed(77)
        assert torch.equal(ovd.v_inpaint(net, GOLD["vi_source"], GOLD["vi_mask"], steps, res), GOLD[f"vi_out_{steps}_{res}"])


def test_product_inpainter_matches_golden(dev, monkeypatch):
    """The product VInpainter (adp_v_inpaint_step: rotation + re-noise + masked blend in one kernel per resample)
    against the live reference's fixtures (diffusion.py:321-354), on the emulator and -- with -m gpu -- on the HIP
    kernel.  The reference's draws come from torch's CPU generator; the harness keeps them on the host stream so
    the device run sees the same noise (the product itself draws on the device)."""
    host_randn_like = torch.randn_like
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: host_randn_like(t.cpu(), **kw).to(t.device))
    net = StubNet().to(dev)
    for steps, res in ((4, 3), (6, 1)):
        torch.manual_seed(77)
        out = adp.VInpainter(net)(GOLD["vi_source"].to(dev), GOLD["vi_mask"].to(dev), num_steps=steps,
                                  num_resamples=res)
        assert out.device.type == dev.type
        assert rel_err(out, GOLD[f"vi_out_{steps}_{res}"]) < 1e-5


def test_product_vdiffusion_and_sampler_match_golden(dev):
    """The product VDiffusion / VSampler (fused kernels) against the reference-generated fixtures
    (diffusion.py:82-95, :172-190), on the emulator and -- with -m gpu -- on the HIP kernels, eager and through the
    hipGraph-replayed step."""
    net = StubNet().to(dev)

    class Fixed(adp.Distribution):
        def __call__(self, num_samples, device=torch.device("cpu")):
            return GOLD["vd_sigmas"][:num_samples].to(device)

    loss = adp.VDiffusion(net, sigma_distribution=Fixed())(GOLD["vd_x"].to(dev), noise=GOLD["vd_noise"].to(dev))
    assert loss.item() == pytest.approx(GOLD["vd_loss"].item(), rel=1e-5)
    for use_graph in ((False, True) if dev.type == "cuda" else (False,)):
        samp = adp.VSampler(net, use_graph=use_graph)
        for steps in (1, 5, 50):
            assert rel_err(samp(GOLD["vs_noise"].to(dev), num_steps=steps), GOLD[f"vs_out_{steps}"]) < 1e-5
    # custom loss_fn keeps working through autograd
    l1 = adp.VDiffusion(net, sigma_distribution=Fixed(), loss_fn=torch.nn.functional.l1_loss)
    l1(GOLD["vd_x"].to(dev), noise=GOLD["vd_noise"].to(dev)).backward()
    assert net.w.grad is not None


def test_restatement_structure():
    """Call-site contract of components.py:79-105: kwarg names accepted, output shape, item counts."""
    cfg = dict(in_channels=2, channels=[8, 16, 16], factors=[1, 4, 2], items=[1, 2, 3], attentions=[0, 0, 1],
               cross_attentions=[0, 1, 1], attention_heads=2, attention_features=4, embedding_features=6,
               modulation_features=16, resnet_groups=8, out_channels=3)
    net = UNetV0Oracle(**cfg)
    y = net(torch.randn(2, 2, 64), torch.rand(2), embedding=torch.randn(2, 5, 6))
    assert y.shape == (2, 3, 64)
    assert net.blocks[2].item_types == ["resnet", "modulation", "attention", "cross_attention"] * 3
    assert net.blocks[0].skip_adapter is not None and net.blocks[1].skip_adapter is None
    mine = adp.UNetV0(dim=1, **cfg)
    assert sum(p.numel() for p in mine.parameters()) == sum(p.numel() for p in net.parameters())
    with pytest.raises(AssertionError):
        adp.UNetV0(dim=1, in_channels=2, channels=[8, 16], factors=[1], items=[1, 1])


def test_product_semantics_match_oracle_switches():
    """The product hard-wires the recalled a_unet semantics in kernels and host code (unet.A_UNET_SEMANTICS says where);
    the oracle names the same choices as [switch] constants.  The two tables must agree: a switch flipped in the oracle
    (e.g. after tools/pin_a_unet.py compared it with a real a_unet) cannot silently diverge from the kernels."""
    import oracle.a_unet_restatement as rs
    from audio_diffusion_pytorch_amd.unet import A_UNET_SEMANTICS
    switches = {k: getattr(rs, k) for k in A_UNET_SEMANTICS}
    assert switches == A_UNET_SEMANTICS
    # every [switch] of the oracle's header block is covered by the product table
    src = open(rs.__file__).read()
    block = src[src.index("# [switch] constants"):src.index("ITEM_RESNET")]
    import re
    assert set(re.findall(r"^([A-Z][A-Z0-9_]+)\s*=", block, flags=re.M)) == set(A_UNET_SEMANTICS)


def test_c_abi_exports_every_declared_symbol():
    """include/adp.h <-> the built libraries: every declared function is exported (no compute calls)."""
    import ctypes
    import re
    from audio_diffusion_pytorch_amd import _C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "adp.h")).read()
    declared = set(re.findall(r"\b(adp_[a-z0-9_]+)\s*\(", header)) - {"adp_conv_desc", "adp_wgrad_desc"}
    assert declared == set(_C.SIGNATURES), declared ^ set(_C.SIGNATURES)
    if not os.path.exists(_C.LIB_PATH):
        pytest.skip("libadp_hip.so not built yet (run __graft_entry__.build())")
    lib = ctypes.CDLL(_C.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.adp_version() >= 100


def test_a_unet_checkpoint_order_loader(emul):
    """UNetV0Net.load_a_unet_state_dict: a checkpoint whose tensors come in a_unet's (recalled) NESTED registration order under
    foreign key names -- depth d + 1 between depth d's down and up items, a_unet's `blocks.N...` style names -- loads
    position by position and reproduces the restatement's output; a checkpoint of another structure is refused with the
    entry that does not fit.  (The order itself is unverifiable offline: tools/pin_a_unet.py checks it where a_unet exists.)"""
    import audio_diffusion_pytorch_amd as adp
    from oracle.a_unet_restatement import UNetV0Oracle
    cfg = dict(in_channels=2, channels=[8, 16, 16], factors=[2, 2, 2], items=[1, 2, 1], modulation_features=24,
               attentions=[0, 0, 1], attention_heads=2, attention_features=8)  # depths 1 and 2 share their shapes
    torch.manual_seed(3)
    oracle = UNetV0Oracle(**cfg)
    net = adp.UNetV0(dim=1, **cfg)
    order = net.a_unet_key_order()
    osd = oracle.state_dict()
    assert sorted(order) == sorted(osd.keys())
    assert order.index("blocks.1.items_down.0.gn1.weight") < order.index("blocks.2.down.weight") < \
        order.index("blocks.1.items_up.0.gn1.weight"), "depth 2 nests between depth 1's down and up items"
    def foreign_key(i, k):  # a_unet-style nesting names; the last component is the tensor's role as a_unet spells it
        return f"blocks.2.blocks.{i}." + ("weights" if k == "time_weights" else k.rsplit(".", 1)[-1])
    foreign = {foreign_key(i, k): osd[k].clone() for i, k in enumerate(order)}
    keymap = net.load_a_unet_state_dict(foreign)
    assert list(keymap.values()) == order
    x, t = torch.randn(2, 2, 64), torch.tensor([0.3, 0.8])
    with torch.no_grad():
        assert rel_err(net(x, t), oracle(x, t)) < 1e-4
    wrong = dict(foreign)
    k5 = list(wrong)[10]
    wrong[k5] = torch.zeros(3, 3)
    with pytest.raises(ValueError, match="expected"):
        net.load_a_unet_state_dict(wrong)
    # a GroupNorm's (weight, bias) pair in the other order has the same shapes: refused by the key's last component
    keys = list(foreign)
    i = order.index("blocks.0.items_down.0.gn1.weight")
    swapped = [(k, v) for k, v in foreign.items()]
    swapped[i], swapped[i + 1] = swapped[i + 1], swapped[i]
    with pytest.raises(ValueError, match="is a `bias` where"):
        net.load_a_unet_state_dict(dict(swapped))
    # the plugin wrappers: Time(Text(CFG(XUNet))) puts the T5 tensors and the fixed-embedding table between the time MLP and
    # the blocks; components.load_reference_state_dict peels them off
    from audio_diffusion_pytorch_amd.components import load_reference_state_dict
    cfg2 = dict(cfg, cross_attentions=[0, 0, 1], embedding_features=12, use_embedding_cfg=True, embedding_max_length=5,
                use_text_conditioning=True, text_embedder=torch.nn.Identity())
    torch.manual_seed(4)
    wrapped = adp.UNetV0(dim=1, **cfg2)
    src = adp.UNetV0(dim=1, **cfg2)
    core, order2 = src.net.net, src.net.net.a_unet_key_order()
    own = core.oracle_named_grads({n: p.detach() for n, p in core.named_parameters()})  # (name mapping only: bank split per item)
    n_time = sum(1 for k in order2 if k.startswith("time_"))
    tensors = [(foreign_key(i, k), own[k].clone()) for i, k in enumerate(order2)]
    ckpt = dict(tensors[:n_time] + [("embedder.transformer.shared.weight", torch.randn(7, 3)),
                                    ("blocks.0.fixed_embedding.weight", src.net.fixed_embedding.weight.detach().clone())]
                + tensors[n_time:])
    keymap = load_reference_state_dict(wrapped, ckpt)
    assert keymap["blocks.0.fixed_embedding.weight"] == "fixed_embedding.weight"
    assert keymap["embedder.transformer.shared.weight"].startswith("skipped")
    for (n, p), (_, q) in zip(wrapped.named_parameters(), src.named_parameters()):
        assert torch.equal(p, q), n
