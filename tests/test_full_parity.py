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


@pytest.mark.gpu
def test_config3_sampler_50_steps(hip, pair):
    """BASELINE config 3 as stated: VSampler.sample with num_steps=50 on noise [1, 2, 2**18] (hipGraph-captured step
    replayed 50 times) against the CPU oracle's 50 steps (diffusion.py:183-188), 1e-3 rel on the final sample."""
    oracle = pair
    torch.set_num_threads(min(16, torch.get_num_threads()))
    noise = torch.randn(1, 2, 2 ** 18, generator=torch.Generator().manual_seed(0))
    ref = ovd.v_sample(oracle, noise, 50)
    model = adp.DiffusionModel(net_t=adp.UNetV0, **FULL)
    model.net.load_oracle_state_dict(oracle.state_dict())
    model = model.to(hip)
    out = model.sample(noise.to(hip), num_steps=50)
    assert out.shape == ref.shape and rel_err(out, ref) < 1e-3


@pytest.mark.gpu
def test_config5_upsampler_sample_at_batch_8(hip):
    """BASELINE config 5's sampling half as stated: `DiffusionUpsampler.sample(randn(8, 2, 2**14), num_steps=50)` ->
    [8, 2, 2**18] (models.py:160-165: polyphase upsample x16, host-drawn starting noise, 50 VSampler steps with the
    conditioning appended).  Seeded like a user script (torch.manual_seed): the product must draw the reference's
    starting noise.  The whole batch runs on the GPU; two of the eight samples are followed through the CPU oracle
    (samples are independent: no batch-coupled op exists on the path)."""
    from oracle.a_unet_restatement import AppendChannelsOracle
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    cfg = dict(FULL)
    cfg.pop("in_channels")
    up = adp.DiffusionUpsampler(net_t=adp.UNetV0, in_channels=2, upsample_factor=16, **cfg)
    oracle = AppendChannelsOracle(lambda **kw: UNetV0Oracle(**kw), channels=2)(in_channels=2, **cfg)
    up.net.net.load_oracle_state_dict(oracle.net.state_dict())
    up = up.to(hip)
    low = torch.randn(8, 2, 2 ** 14, generator=torch.Generator().manual_seed(7))
    torch.manual_seed(123)
    out = up.sample(low.to(hip), num_steps=50)
    assert out.shape == (8, 2, 2 ** 18)
    torch.manual_seed(123)
    start = torch.randn(8, 2, 2 ** 18)  # the reference's draw: utils.randn_like on the host (utils.py:123-125)
    cond = ovd.upsample(low, 16)
    for i in (0, 5):
        ref = ovd.v_sample(oracle, start[i:i + 1], 50, append_channels=cond[i:i + 1])
        assert rel_err(out[i:i + 1], ref) < 1e-3, i


ATTN_README = dict(attentions=[0, 0, 0, 0, 0, 1, 1, 1, 1], attention_heads=8, attention_features=64)
CROSS_CFG4 = dict(cross_attentions=[0, 0, 0, 1, 1, 1, 1, 1, 1], embedding_features=768, attention_heads=8,
                  attention_features=64)


@pytest.mark.gpu
@pytest.mark.parametrize("extra,use_emb", [(ATTN_README, False), (CROSS_CFG4, True)], ids=["readme-self-attn", "config4"])
def test_config4_attention_full_fwd_bwd(hip, extra, use_emb):
    """BASELINE config 4 (text-conditional UNetV0: cross attention at depths 3-8 over embedding [B, 64, 768]) and the
    README self-attention layout, at full size with one sample (the per-rank share of global batch 8 on 8 GPUs):
    loss, every parameter gradient and the embedding gradient against the CPU oracle at 1e-3."""
    cfg = dict(FULL, **extra)
    torch.manual_seed(0)
    oracle = UNetV0Oracle(**cfg)
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas([0.45]), **cfg)
    model.net.load_oracle_state_dict(oracle.state_dict())
    model = model.to(hip)
    g = torch.Generator().manual_seed(0)
    x, noise = torch.randn(1, 2, 2 ** 18, generator=g), torch.randn(1, 2, 2 ** 18, generator=g)
    kw_ref, kw = {}, {}
    if use_emb:
        emb = torch.randn(1, 64, 768, generator=g).requires_grad_()
        emb_d = emb.detach().to(hip).requires_grad_()
        kw_ref, kw = dict(embedding=emb), dict(embedding=emb_d)
    loss_ref = ovd.v_loss(oracle, x, noise, torch.tensor([0.45]), **kw_ref)
    loss_ref.backward()
    loss = model(x.to(hip), noise=noise.to(hip), **kw)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-3 * abs(loss_ref.item())
    compare_grads(model.net, oracle)
    if use_emb:
        assert rel_err(emb_d.grad, emb.grad) < 1e-3


@pytest.mark.gpu
def test_config5_upsampler_full(hip):
    """BASELINE config 5: DiffusionUpsampler(upsample_factor=16) on the full UNetV0 (in=4 via AppendChannels, out=2):
    training loss + gradients on [1, 2, 2**18] (reupsample = polyphase HIP resampler down x16 then up x16) and a
    2-step sample from [1, 2, 2**14] -> [1, 2, 2**18], against the CPU oracle."""
    from oracle.a_unet_restatement import AppendChannelsOracle
    torch.manual_seed(0)
    cfg = dict(FULL)
    cfg.pop("in_channels")
    up = adp.DiffusionUpsampler(net_t=adp.UNetV0, in_channels=2, upsample_factor=16, sampler_use_graph=False,
                                diffusion_sigma_distribution=FixedSigmas([0.6]), **cfg)
    oracle = AppendChannelsOracle(lambda **kw: UNetV0Oracle(**kw), channels=2)(in_channels=2, **cfg)
    up.net.net.load_oracle_state_dict(oracle.net.state_dict())
    up = up.to(hip)
    g = torch.Generator().manual_seed(1)
    x, noise = torch.randn(1, 2, 2 ** 18, generator=g), torch.randn(1, 2, 2 ** 18, generator=g)
    re_ref = ovd.upsample(ovd.downsample(x.clone(), 16), 16)
    assert rel_err(up.reupsample(x.to(hip)), re_ref) < 1e-4
    loss_ref = ovd.v_loss(oracle, x, noise, torch.tensor([0.6]), append_channels=re_ref)
    loss_ref.backward()
    loss = up(x.to(hip), noise=noise.to(hip))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-3 * abs(loss_ref.item())
    compare_grads(up.net.net, oracle.net)
    low = torch.randn(1, 2, 2 ** 14, generator=g)
    start = torch.randn(1, 2, 2 ** 18, generator=g)
    re_low = ovd.upsample(low, 16)
    ref = ovd.v_sample(oracle, start, 2, append_channels=re_low)
    out = up.sampler(start.to(hip), num_steps=2, append_channels=up_sample_input(up, low.to(hip)))
    assert out.shape == (1, 2, 2 ** 18) and rel_err(out, ref) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["config4", "config5"])
def test_configs_4_and_5_at_batch_8(hip, which):
    """BASELINE configs 4 and 5 at their stated batch of 8 on ONE GPU (the 8-GPU runs split this batch): loss and every
    parameter gradient against the CPU oracle at 1e-3.  Config 4: cross attention over embedding [8, 64, 768];
    config 5: DiffusionUpsampler(upsample_factor=16) training step on [8, 2, 2**18]."""
    from oracle.a_unet_restatement import AppendChannelsOracle
    B = 8
    sig = [0.05, 0.2, 0.35, 0.5, 0.65, 0.8, 0.95, 0.45]
    g = torch.Generator().manual_seed(2)
    x, noise = torch.randn(B, 2, 2 ** 18, generator=g), torch.randn(B, 2, 2 ** 18, generator=g)
    torch.manual_seed(0)
    if which == "config4":
        cfg = dict(FULL, **CROSS_CFG4)
        oracle = UNetV0Oracle(**cfg)
        model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas(sig), **cfg)
        model.net.load_oracle_state_dict(oracle.state_dict())
        emb = torch.randn(B, 64, 768, generator=g)
        loss_ref = ovd.v_loss(oracle, x, noise, torch.tensor(sig), embedding=emb)
        loss_ref.backward()
        model = model.to(hip)
        loss = model(x.to(hip), noise=noise.to(hip), embedding=emb.to(hip))
        unet, ounet = model.net, oracle
    else:
        cfg = dict(FULL)
        cfg.pop("in_channels")
        model = adp.DiffusionUpsampler(net_t=adp.UNetV0, in_channels=2, upsample_factor=16,
                                       diffusion_sigma_distribution=FixedSigmas(sig), **cfg)
        oracle = AppendChannelsOracle(lambda **kw: UNetV0Oracle(**kw), channels=2)(in_channels=2, **cfg)
        model.net.net.load_oracle_state_dict(oracle.net.state_dict())
        re_ref = ovd.upsample(ovd.downsample(x, 16), 16)
        loss_ref = ovd.v_loss(oracle, x, noise, torch.tensor(sig), append_channels=re_ref)
        loss_ref.backward()
        model = model.to(hip)
        loss = model(x.to(hip), noise=noise.to(hip))
        unet, ounet = model.net.net, oracle.net
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-3 * abs(loss_ref.item())
    compare_grads(unet, ounet)


@pytest.mark.gpu
def test_sampler_graph_cache_follows_kwargs(hip):
    """VSampler's hipGraph cache is keyed on the call STRUCTURE and owns static copies of the tensor kwargs: fresh
    `embedding` / `append_channels` tensors per call replay the same graph with the new values (never a stale or
    freed pointer), python scalars are part of the key, and the cache is LRU-bounded."""
    cfg = dict(in_channels=2, channels=[8, 32, 64], factors=[1, 4, 4], items=[1, 2, 2], modulation_features=64,
               cross_attentions=[0, 1, 1], attention_heads=2, attention_features=16, embedding_features=24,
               use_embedding_cfg=True, embedding_max_length=6)
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, **cfg).to(hip)
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(2, 2, 1024, generator=g).to(hip)

    def run(emb, use_graph, **kw):
        model.sampler.use_graph = use_graph
        return model.sample(noise, num_steps=3, embedding=emb, **kw)

    outs = []
    for i in range(3):  # a NEW embedding tensor of the same shape per call (the id()-reuse pattern of ADVICE r1)
        emb = torch.randn(2, 5, 24, generator=g).to(hip)
        eager = run(emb, False)
        graphed = run(emb, True)
        assert rel_err(graphed, eager) < 1e-5, i
        outs.append(graphed)
        del emb
    assert len(model.sampler._graph_cache) == 1
    assert rel_err(outs[1], outs[0]) > 1e-3 and rel_err(outs[2], outs[1]) > 1e-3  # the new values were really used
    emb = torch.randn(2, 5, 24, generator=g).to(hip)
    for scale in (1.5, 2.0, 2.5, 3.0, 3.5, 4.0):  # a python scalar baked into the capture -> its own entry
        assert rel_err(run(emb, True, embedding_scale=scale), run(emb, False, embedding_scale=scale)) < 1e-5, scale
    assert len(model.sampler._graph_cache) <= adp.VSampler.GRAPH_CACHE_ENTRIES
    # DiffusionUpsampler.sample creates its append_channels tensor inside every call
    ucfg = dict(channels=[8, 32, 64], factors=[1, 4, 4], items=[1, 2, 2], modulation_features=64)
    up = adp.DiffusionUpsampler(net_t=adp.UNetV0, in_channels=2, upsample_factor=4, **ucfg).to(hip)
    for i in range(3):
        low = torch.randn(1, 2, 256, generator=g).to(hip)
        gen = torch.Generator().manual_seed(10 + i)
        up.sampler.use_graph = True
        a = up.sample(low, num_steps=2, generator=gen)
        gen = torch.Generator().manual_seed(10 + i)
        up.sampler.use_graph = False
        b = up.sample(low, num_steps=2, generator=gen)
        assert a.shape == (1, 2, 1024) and rel_err(a, b) < 1e-5, i
    assert len(up.sampler._graph_cache) == 1


def up_sample_input(up, low):
    from audio_diffusion_pytorch_amd.utils import upsample
    return upsample(low, factor=up.upsample_factor)


@pytest.mark.gpu
def test_winograd_and_direct_form_agree_inside_the_unet(hip, monkeypatch):
    """The Winograd variants of the kernel-3 convs and weight gradients (the default: F(4,3) -- conv_tile32, conv_mm4,
    conv_tilek, wgrad_mm's W4 -- from 32 / 128 / 64 channels, F(2,3) below and for the shapes those kernels do not take)
    against the direct form (ADP_CONV_WINO=0) inside a whole U-Net step: loss, prediction and every parameter gradient on the same weights and
    inputs.  Both are plain fp32 arithmetic, so the bound is 1e-4 (ten times tighter than the path's parity tolerance)."""
    env = "ADP_CONV_WINO"
    cfg = dict(in_channels=2, channels=[8, 32, 256, 512], factors=[1, 4, 4, 2], items=[1, 1, 2, 2])
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_sigma_distribution=FixedSigmas([0.3, 0.7]), **cfg).to(hip)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 2, 2 ** 13, generator=g).to(hip)
    t = torch.tensor([0.3, 0.7], device=hip)

    def run():
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(5)  # the same noise draw in VDiffusion
        loss = model(x)
        loss.backward()
        with torch.no_grad():
            v = model.net(x, t)
        return loss.detach().clone(), v.clone(), {n: p.grad.clone() for n, p in model.named_parameters()}

    monkeypatch.setenv(env, "0")
    loss0, v0, g0 = run()
    monkeypatch.setenv(env, "1")
    loss1, v1, g1 = run()
    assert not torch.equal(v0, v1), "the other family was meant to run (different rounding)"
    assert abs(loss1.item() - loss0.item()) <= 1e-5 * abs(loss0.item())
    assert rel_err(v1, v0) < 1e-4
    # (a conv bias that feeds a GroupNorm has a true gradient of zero -- both runs hold cancellation noise there -- so the
    # denominator is floored at 1e-3 x the largest gradient of the model, as in test_unet.compare_grads)
    gmax = max(v.abs().max().item() for v in g0.values())
    worst = max(((g1[n] - g0[n]).abs().max().item() / max(g0[n].abs().max().item(), 1e-3 * gmax), n) for n in g0)
    assert worst[0] < 1e-4, worst
