"""Whole-path parity: UNetV0 forward/backward, VDiffusion loss, VSampler loop against the CPU oracle
(oracle/a_unet_restatement.py + oracle/vdiffusion.py) on identical weights, sigmas and noise.
Runs on the SIMT emulator (CPU suite, tiny configs) and on the real gfx950 library (-m gpu)."""
import pytest
import torch

import audio_diffusion_pytorch_amd as adp
from conftest import rel_err
from oracle import vdiffusion as ovd
from oracle.a_unet_restatement import UNetV0Oracle

TOL = 1e-3  # north_star: 1e-3 rel fp32, ||a-b||inf / ||b||inf per tensor

TINY = dict(in_channels=2, channels=[8, 16, 32], factors=[1, 4, 2], items=[1, 2, 1], modulation_features=64)
TINY_CTX = dict(in_channels=2, channels=[8, 16], factors=[2, 2], items=[1, 1], modulation_features=32,
                context_channels=[0, 3])


class FixedSigmas(adp.Distribution):
    def __init__(self, vals):
        self.vals = torch.tensor(vals, dtype=torch.float32)

    def __call__(self, num_samples, device=torch.device("cpu")):
        return self.vals[:num_samples].to(device)


def build_pair(cfg, dev, seed=0):
    torch.manual_seed(seed)
    oracle = UNetV0Oracle(**cfg)
    # make norms / biases non-trivial so their gradients are exercised
    with torch.no_grad():
        for n_, p in oracle.named_parameters():
            if n_.endswith("gn1.weight") or n_.endswith("gn2.weight") or "norm" in n_ and n_.endswith("weight"):
                p.add_(0.1 * torch.randn_like(p))
            if n_.endswith("gn1.bias") or n_.endswith("gn2.bias"):
                p.add_(0.1 * torch.randn_like(p))
    net = adp.UNetV0(dim=1, **cfg)
    net.load_oracle_state_dict(oracle.state_dict())
    return oracle, net.to(dev)


def compare_grads(net, oracle, tol=TOL):
    own = {n_: p.grad for n_, p in net.named_parameters()}
    assert all(g is not None for g in own.values())
    mapped = net.oracle_named_grads(own)
    worst = ("", 0.0)
    gmax = max(p.grad.abs().max().item() for p in oracle.parameters())
    for n_, p in oracle.named_parameters():
        assert n_ in mapped, n_
        # a conv bias feeding a GroupNorm with one channel per group has an identically-zero true gradient:
        # both sides then hold only cancellation roundoff, so the denominator is floored at 1e-3 x the largest
        # gradient magnitude of the model
        a, b = mapped[n_].detach().double().cpu(), p.grad.double()
        e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-3 * gmax)
        if e > worst[1]:
            worst = (n_, e)
    assert worst[1] < tol, worst


@pytest.mark.parametrize("B,L", [(2, 256)])
def test_unet_forward_backward_tiny(dev, B, L):
    oracle, net = build_pair(TINY, dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 2, L, generator=g)
    t = torch.tensor([0.3, 0.8])[:B]
    feats = 0.1 * torch.randn(B, TINY["modulation_features"], generator=g)
    y_ref = oracle(x, t, features=feats)
    y = net(x.to(dev), t.to(dev), features=feats.to(dev))
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)


def test_unet_inference_matches_training_forward(dev):
    oracle, net = build_pair(TINY, dev)
    x = torch.randn(1, 2, 128)
    t = torch.tensor([0.5])
    with torch.no_grad():
        y0 = net(x.to(dev), t.to(dev))
    y1 = net(x.to(dev), t.to(dev))
    assert y1.requires_grad
    assert torch.equal(y0.cpu(), y1.detach().cpu())
    assert rel_err(y0, oracle(x, t)) < TOL


def test_unet_inject_channels(dev):
    oracle, net = build_pair(TINY_CTX, dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 2, 64, generator=g)
    t = torch.tensor([0.2, 0.9])
    ctx = torch.randn(2, 3, 16, generator=g)
    y_ref = oracle(x, t, channels=[None, ctx])
    y = net(x.to(dev), t.to(dev), channels=[None, ctx.to(dev)])
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)


def test_vdiffusion_loss_and_grads(dev):
    """DiffusionModel(x) -> loss ; loss.backward(): config-1 shape of the path at tiny size."""
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas([0.3, 0.7]), **TINY)
    oracle = UNetV0Oracle(**TINY)
    model.net.load_oracle_state_dict(oracle.state_dict())
    model = model.to(dev)
    g = torch.Generator().manual_seed(3)
    x, noise = torch.randn(2, 2, 256, generator=g), torch.randn(2, 2, 256, generator=g)
    loss_ref = ovd.v_loss(oracle, x, noise, torch.tensor([0.3, 0.7]))
    loss_ref.backward()
    loss = model(x.to(dev), noise=noise.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < TOL * abs(loss_ref.item())
    compare_grads(model.net, oracle)


def test_vsampler_matches_oracle(dev):
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, sampler_use_graph=False, **TINY)
    oracle = UNetV0Oracle(**TINY)
    model.net.load_oracle_state_dict(oracle.state_dict())
    model = model.to(dev)
    noise = torch.randn(1, 2, 128)
    ref = ovd.v_sample(oracle, noise, 4)
    out = model.sample(noise.to(dev), num_steps=4)
    assert rel_err(out, ref) < TOL


def test_upsampler_append_channels(dev):
    """DiffusionUpsampler: AppendChannelsPlugin input concat read through two base pointers (config 5 shape)."""
    from oracle.a_unet_restatement import AppendChannelsOracle
    torch.manual_seed(0)
    cfg = dict(TINY)
    cfg.pop("in_channels")
    up = adp.DiffusionUpsampler(net_t=adp.UNetV0, in_channels=2, upsample_factor=4,
                                diffusion_sigma_distribution=FixedSigmas([0.4, 0.6]), **cfg)
    oracle_wrap = AppendChannelsOracle(lambda **kw: UNetV0Oracle(**kw), channels=2)(in_channels=2, **cfg)
    up.net.net.load_oracle_state_dict(oracle_wrap.net.state_dict())
    up = up.to(dev)
    g = torch.Generator().manual_seed(5)
    x, noise = torch.randn(2, 2, 256, generator=g), torch.randn(2, 2, 256, generator=g)
    re = ovd.upsample(ovd.downsample(x.clone(), 4), 4)
    loss_ref = ovd.v_loss(oracle_wrap, x, noise, torch.tensor([0.4, 0.6]), append_channels=re)
    loss_ref.backward()
    loss = up(x.to(dev), noise=noise.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < TOL * abs(loss_ref.item())
    compare_grads(up.net.net, oracle_wrap.net)


ATTN = dict(in_channels=2, channels=[8, 16, 32], factors=[1, 2, 2], items=[1, 1, 2], modulation_features=32,
            attentions=[0, 1, 1], cross_attentions=[0, 0, 1], attention_heads=2, attention_features=8,
            embedding_features=12)


def test_unet_attention_self_and_cross(dev):
    """README attention layout at tiny size (self attention + cross attention over an injected embedding:
    BASELINE config 4 feeds `embedding=` directly)."""
    oracle, net = build_pair(ATTN, dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 2, 96, generator=g)
    t = torch.tensor([0.15, 0.65])
    emb = torch.randn(2, 5, 12, generator=g).requires_grad_()
    y_ref = oracle(x, t, embedding=emb)
    emb_d = emb.detach().to(dev).requires_grad_()
    y = net(x.to(dev), t.to(dev), embedding=emb_d)
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)
    assert rel_err(emb_d.grad, emb.grad) < TOL
