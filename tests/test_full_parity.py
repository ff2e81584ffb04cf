"""BASELINE.json configs at FULL size on the MI355X against the CPU oracle (the reference CPU path):
config 2 (UNetV0 README channels, batch 4, [4,2,2**18], fwd+bwd: loss, v_pred and every parameter gradient at
1e-3 rel) and config 3 (VSampler on noise [1,2,2**18]; a few steps, eager and hipGraph-replayed)."""
import pytest
import torch

import audio_diffusion_pytorch_amd as adp
from conftest import rel_err
from oracle import vdiffusion as ovd
from oracle.a_unet_restatement import UNetV0Oracle
from test_unet import FixedSigmas, compare_grads

FULL = dict(in_channels=2, channels=[8, 32, 64, 128, 256, 512, 512, 1024, 1024], factors=[1, 4, 4, 4, 2, 2, 2, 2, 2],
            items=[1, 2, 2, 2, 2, 2, 2, 4, 4])


@pytest.fixture(scope="module")
def pair():
    torch.manual_seed(0)
    oracle = UNetV0Oracle(**FULL)
    return oracle


@pytest.mark.gpu
def test_config2_full_fwd_bwd(hip, pair):
    oracle = pair
    B = 4
    sig = [0.3, 0.55, 0.9, 0.05]
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig), **FULL)
    model.net.load_oracle_state_dict(oracle.state_dict())
    model = model.to(hip)
    g = torch.Generator().manual_seed(0)
    x, noise = torch.randn(B, 2, 2 ** 18, generator=g), torch.randn(B, 2, 2 ** 18, generator=g)
    for p in oracle.parameters():
        p.grad = None
    x_noisy, v_target = ovd.v_noise(x, noise, torch.tensor(sig))
    v_ref = oracle(x_noisy, torch.tensor(sig))
    loss_ref = torch.nn.functional.mse_loss(v_ref, v_target)
    loss_ref.backward()
    with torch.no_grad():
        v = model.net(x_noisy.to(hip), torch.tensor(sig).to(hip))
    assert rel_err(v, v_ref) < 1e-3
    loss = model(x.to(hip), noise=noise.to(hip))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-3 * abs(loss_ref.item())
    compare_grads(model.net, oracle)


@pytest.mark.gpu
def test_config3_sampler_full(hip, pair):
    oracle = pair
    steps = 3
    noise = torch.randn(1, 2, 2 ** 18, generator=torch.Generator().manual_seed(0))
    ref = ovd.v_sample(oracle, noise, steps)
    for use_graph in (False, True):
        model = adp.DiffusionModel(net_t=adp.UNetV0, sampler_use_graph=use_graph, **FULL)
        model.net.load_oracle_state_dict(oracle.state_dict())
        model = model.to(hip)
        out = model.sample(noise.to(hip), num_steps=steps)
        assert rel_err(out, ref) < 1e-3, use_graph
        out2 = model.sample(noise.to(hip), num_steps=steps)   # replays the cached graph
        assert torch.equal(out, out2)
