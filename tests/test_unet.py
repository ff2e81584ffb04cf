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


@pytest.mark.parametrize("B,L", [(2, 256), (18, 32)])  # 18 > the 16 rows one conditioning-Linear launch holds
def test_unet_forward_backward_tiny(dev, B, L):
    oracle, net = build_pair(TINY, dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 2, L, generator=g)
    t = torch.linspace(0.3, 0.8, B)
    feats = 0.1 * torch.randn(B, TINY["modulation_features"], generator=g)
    y_ref = oracle(x, t, features=feats)
    y = net(x.to(dev), t.to(dev), features=feats.to(dev))
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)


def test_unet_materialised_activation_path(dev, monkeypatch):
    """ResnetBlocks at and above ACT_MATERIALIZE_MIN_C channels (default 128: depths 3-8 of the README model) write
    SiLU(GroupNorm(x)) once (from the producing conv's partial statistics where they nest into the groups, from a pass
    over the tensor otherwise) and feed plain convs / weight gradients; the threshold is lowered here so that the tiny
    CPU-sized model takes that path at every depth."""
    from audio_diffusion_pytorch_amd import unet
    monkeypatch.setattr(unet, "ACT_MATERIALIZE_MIN_C", 8)
    cfg = dict(in_channels=2, channels=[8, 32, 64], factors=[1, 4, 2], items=[1, 2, 1], modulation_features=64)
    oracle, net = build_pair(cfg, dev)
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(2, 2, 512, generator=g), torch.tensor([0.2, 0.9])
    y_ref = oracle(x, t)
    y = net(x.to(dev), t.to(dev))
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
    ctx = torch.randn(2, 3, 16, generator=g).requires_grad_()
    ctx_d = ctx.detach().to(dev).requires_grad_()
    y_ref = oracle(x, t, channels=[None, ctx])
    y = net(x.to(dev), t.to(dev), channels=[None, ctx_d])
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)
    assert rel_err(ctx_d.grad, ctx.grad) < TOL  # both InjectChannelsItems of the depth feed the context gradient


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


@pytest.mark.parametrize("batch,steps", [(1, 4), (3, 7)])   # (3 x 7 = 21 table rows: more than one 16-row pass of the bank)
def test_vsampler_hoisted_conditioning(dev, batch, steps, monkeypatch):
    """VSampler forms the conditioning rows of ALL steps in one batched pass in front of its loop (UNetV0Net.conditioning_table)
    and hands each call its rows: the table equals the per-call conditioning, and so does the sample
    (ADP_SAMPLER_HOIST=0 = the per-step path); with `features` the per-step path is kept."""
    from audio_diffusion_pytorch_amd import unet as unet_mod
    # (bit for bit on the emulator; on the GPU a 16-row pass of the bank and a 1-row pass may round differently)
    same = torch.equal if torch.device(dev).type == "cpu" else (lambda a, b: rel_err(a, b) < 1e-5)
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, sampler_use_graph=False, **TINY).to(dev)
    net = model.net
    noise = torch.randn(batch, 2, 128).to(dev)
    sig, _ = model.sampler._tables(steps, batch, noise.device)
    table = net.conditioning_table(sig[:steps])
    assert table.shape == (steps * batch, net.bank_total)
    for i in (0, steps - 1):
        run = unet_mod._Run(net, False)
        assert same(run.conditioning(sig[i], None), table[i * batch:(i + 1) * batch])
    calls = []
    orig = type(net).conditioning_table
    monkeypatch.setattr(type(net), "conditioning_table", lambda self, t: (calls.append(1), orig(self, t))[1])
    out = model.sample(noise, num_steps=steps)
    assert len(calls) == 1
    monkeypatch.setenv("ADP_SAMPLER_HOIST", "0")
    ref = model.sample(noise, num_steps=steps)
    assert len(calls) == 1 and same(out, ref)
    with pytest.raises(RuntimeError):  # the precomputed rows are an inference-only input
        net(noise.requires_grad_(), sig[0], conditioning=table[:batch])


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


def test_upsampler_sample_draws_the_references_noise(dev):
    """`DiffusionUpsampler.sample(low)` in a seeded script: the reference draws the starting noise on the HOST from the
    global generator (utils.randn_like, utils.py:123-125; models.py:164).  The product does the same, so
    torch.manual_seed(s) reproduces the reference's sample on any backend; an explicit CPU generator likewise."""
    from oracle.a_unet_restatement import AppendChannelsOracle
    torch.manual_seed(0)
    cfg = dict(TINY)
    cfg.pop("in_channels")
    up = adp.DiffusionUpsampler(net_t=adp.UNetV0, in_channels=2, upsample_factor=4, **cfg)
    oracle = AppendChannelsOracle(lambda **kw: UNetV0Oracle(**kw), channels=2)(in_channels=2, **cfg)
    up.net.net.load_oracle_state_dict(oracle.net.state_dict())
    up = up.to(dev)
    low = torch.randn(2, 2, 64, generator=torch.Generator().manual_seed(9))
    cond = ovd.upsample(low, 4)
    torch.manual_seed(77)
    out = up.sample(low.to(dev), num_steps=3)
    torch.manual_seed(77)
    ref = ovd.v_sample(oracle, torch.randn(cond.shape), 3, append_channels=cond)
    assert rel_err(out, ref) < TOL
    out_g = up.sample(low.to(dev), generator=torch.Generator().manual_seed(5), num_steps=3)
    ref_g = ovd.v_sample(oracle, torch.randn(cond.shape, generator=torch.Generator().manual_seed(5)), 3, append_channels=cond)
    assert rel_err(out_g, ref_g) < TOL


ATTN = dict(in_channels=2, channels=[8, 16, 32], factors=[1, 2, 2], items=[1, 1, 2], modulation_features=32,
            attentions=[0, 1, 1], cross_attentions=[0, 0, 1], attention_heads=2, attention_features=8,
            embedding_features=12)


@pytest.mark.parametrize("batch,bank,fuse", [(2, "1", True), (1, "1", True), (2, "0", True), (2, "1", False)])
def test_unet_attention_self_and_cross(dev, batch, bank, fuse, monkeypatch):
    """README attention layout at tiny size (self attention + cross attention over an injected embedding:
    BASELINE config 4 feeds `embedding=` directly).  bank = the context side of the four cross-attention
    items as one folded weight bank (attention.CtxBank: slice views at batch 1, row copies otherwise) or item by item;
    fuse = the ModulationItem in front of an attention item forms that item's LayerNorm(s) in its own launch."""
    from audio_diffusion_pytorch_amd import unet as unet_mod
    monkeypatch.setenv("ADP_CTX_BANK", bank)
    monkeypatch.setattr(unet_mod, "MOD_LN_FUSE", fuse)
    monkeypatch.setattr(unet_mod, "MOD_LN_BWD_FUSE", batch == 2)  # (the batch-1 case keeps the separate backward launches)
    oracle, net = build_pair(ATTN, dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(batch, 2, 96, generator=g)
    t = torch.tensor([0.15, 0.65][:batch])
    emb = torch.randn(batch, 5, 12, generator=g).requires_grad_()
    y_ref = oracle(x, t, embedding=emb)
    emb_d = emb.detach().to(dev).requires_grad_()
    y = net(x.to(dev), t.to(dev), embedding=emb_d)
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)
    assert rel_err(emb_d.grad, emb.grad) < TOL
    assert (getattr(net.net if hasattr(net, "net") else net, "_ctx_tables", None) is not None) == (bank == "1")


@pytest.mark.parametrize("batch,fuse_bwd", [(1, True), (2, True), (2, False)])
def test_unet_cross_attention_behind_modulation(dev, batch, fuse_bwd, monkeypatch):
    """BASELINE config 4's layout at tiny size: CrossAttentionItems directly behind the ModulationItems (no self attention in
    between).  The Modulation forms the item's LayerNorm in its own launch and, with fuse_bwd, runs that LayerNorm's backward
    inside its own (adp_modulation_ln_fwd / adp_modulation_ln_bwd_partial); both settings against the oracle."""
    from audio_diffusion_pytorch_amd import unet as unet_mod
    monkeypatch.setattr(unet_mod, "MOD_LN_BWD_FUSE", fuse_bwd)
    cfg = dict(ATTN, attentions=[0, 0, 0], cross_attentions=[0, 1, 1])
    oracle, net = build_pair(cfg, dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(batch, 2, 96, generator=g)
    t = torch.tensor([0.15, 0.65][:batch])
    emb = torch.randn(batch, 5, 12, generator=g).requires_grad_()
    y_ref = oracle(x, t, embedding=emb)
    emb_d = emb.detach().to(dev).requires_grad_()
    y = net(x.to(dev), t.to(dev), embedding=emb_d)
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)
    assert rel_err(emb_d.grad, emb.grad) < TOL
    inner = net.net if hasattr(net, "net") else net
    assert getattr(inner, "_ln_deferred", 0) == (6 if fuse_bwd else 0)  # 2 sides x (1 + 2) cross-attention items


def test_classifier_free_guidance(dev):
    """UNetV0(use_embedding_cfg=True): training-time batch masking (gradients of the U-Net, of the incoming embedding
    and of the fixed positional table) and guided sampling (batched [x|x] evaluation + fused mix) against the
    ClassifierFreeGuidanceOracle restatement (components.py:66-69)."""
    from oracle.a_unet_restatement import ClassifierFreeGuidanceOracle
    torch.manual_seed(0)
    inner = UNetV0Oracle(**ATTN)
    oracle = ClassifierFreeGuidanceOracle(inner, embedding_max_length=7, embedding_features=12)
    net = adp.UNetV0(dim=1, use_embedding_cfg=True, embedding_max_length=7, **ATTN)
    assert isinstance(net, adp.ClassifierFreeGuidanceNet)
    net.net.load_oracle_state_dict(inner.state_dict())
    with torch.no_grad():
        net.fixed_embedding.weight.copy_(oracle.fixed_embedding.weight)
    net = net.to(dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 2, 96, generator=g)
    t = torch.tensor([0.15, 0.65, 0.4])
    emb = torch.randn(3, 5, 12, generator=g).requires_grad_()
    bm = torch.tensor([True, False, True])
    # training path: masked batch elements see the fixed embedding
    y_ref = oracle(x, t, embedding=emb, embedding_mask_proba=0.5, batch_mask=bm)
    emb_d = emb.detach().to(dev).requires_grad_()
    y = net(x.to(dev), t.to(dev), embedding=emb_d, embedding_mask_proba=0.5, batch_mask=bm.to(dev))
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net.net, inner)
    assert rel_err(emb_d.grad, emb.grad) < TOL
    assert rel_err(net.fixed_embedding.weight.grad, oracle.fixed_embedding.weight.grad) < TOL
    # guided sampling path: one batched evaluation, out_masked + (out - out_masked) * scale
    with torch.no_grad():
        s_ref = oracle(x, t, embedding=emb.detach(), embedding_scale=5.0)
        s = net(x.to(dev), t.to(dev), embedding=emb.detach().to(dev), embedding_scale=5.0)
    assert rel_err(s, s_ref) < TOL
    # through the model API (README.md:62-76): kwargs reach the plugin via DiffusionModel.sample
    model = adp.DiffusionModel(net_t=adp.UNetV0, use_embedding_cfg=True, embedding_max_length=7,
                               sampler_use_graph=False, **ATTN)
    model.net.net.load_oracle_state_dict(inner.state_dict())
    with torch.no_grad():
        model.net.fixed_embedding.weight.copy_(oracle.fixed_embedding.weight)
    model = model.to(dev)
    out = model.sample(x.to(dev), num_steps=2, embedding=emb.detach().to(dev), embedding_scale=3.0)
    ref = ovd.v_sample(lambda xx, tt, **kw: oracle(xx, tt, embedding=emb.detach(), embedding_scale=3.0), x, 2)
    assert rel_err(out, ref) < TOL


def test_diffusion_autoencoder(dev):
    """DiffusionAE (models.py:70-131): encoder latent injected at inject_depth, loss + encoder gradient through the
    inject conv's data gradient, decode() with the closest_power_2 noise length rule."""
    class Enc(adp.EncoderBase):
        def __init__(self):
            super().__init__()
            self.out_channels, self.downsample_factor = 3, 4
            self.conv = torch.nn.Conv1d(2, 3, kernel_size=4, stride=4)

        def forward(self, x, with_info=False):
            z = torch.tanh(self.conv(x))
            return (z, {"z": z}) if with_info else z

    torch.manual_seed(0)
    cfg = dict(channels=[8, 16], factors=[2, 2], items=[1, 1], modulation_features=32)
    enc = Enc()
    ae = adp.DiffusionAE(net_t=adp.UNetV0, in_channels=2, encoder=enc, inject_depth=1,
                         diffusion_sigma_distribution=FixedSigmas([0.3, 0.7]), sampler_use_graph=False, **cfg)
    oracle = UNetV0Oracle(in_channels=2, context_channels=[0, 3], **cfg)
    ae.net.load_oracle_state_dict(oracle.state_dict())
    ae = ae.to(dev)
    import copy
    enc_ref = copy.deepcopy(enc).cpu()
    g = torch.Generator().manual_seed(2)
    x, noise = torch.randn(2, 2, 64, generator=g), torch.randn(2, 2, 64, generator=g)
    z_ref = enc_ref(x)
    loss_ref = ovd.v_loss(oracle, x, noise, torch.tensor([0.3, 0.7]), channels=[None, z_ref])
    loss_ref.backward()
    loss, info = ae(x.to(dev), with_info=True, noise=noise.to(dev))
    loss.backward()
    assert "z" in info and abs(loss.item() - loss_ref.item()) < TOL * abs(loss_ref.item())
    compare_grads(ae.net, oracle)
    assert rel_err(ae.encoder.conv.weight.grad, enc_ref.conv.weight.grad) < TOL
    z = enc_ref(x).detach()
    out = ae.decode(z.to(dev), num_steps=2, generator=None)
    assert out.shape == (2, 2, 64)  # closest_power_2(16 * 4)
    # the starting noise comes from the latent's device and the caller's generator, like the reference's decode
    # (models.py:117-125): same seed -> same sample, and it equals sampling from that draw explicitly
    g1, g2 = torch.Generator(device=dev).manual_seed(5), torch.Generator(device=dev).manual_seed(5)
    o1 = ae.decode(z.to(dev), num_steps=2, generator=g1)
    start = torch.randn((2, 2, 64), device=dev, dtype=z.dtype, generator=g2)
    o2 = ae.sample(start, channels=[None, z.to(dev)], num_steps=2)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("L", [100, 36])
def test_unet_ragged_lengths(dev, L):
    """Lengths that are not multiples of the 4-position vector width (25 / 9 positions at the deepest level): the
    dispatcher leaves the 16-byte-load kernel families for the generic ones; results and gradients still match."""
    cfg = dict(in_channels=2, channels=[8, 32, 64], factors=[1, 2, 2], items=[1, 1, 1], modulation_features=32)
    oracle, net = build_pair(cfg, dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 2, L, generator=g)
    t = torch.tensor([0.1, 0.5, 0.9])
    y_ref = oracle(x, t)
    y = net(x.to(dev), t.to(dev))
    assert rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)


def test_unet_odd_channel_counts(dev):
    """Channel counts that are multiples of resnet_groups but not of the 32-channel MFMA chunk (24, 40): generic
    conv / weight-gradient kernels; factor 4 down/upsampling; out_channels != in_channels (skip adapter)."""
    cfg = dict(in_channels=3, out_channels=2, channels=[8, 24, 40], factors=[1, 4, 2], items=[1, 1, 2],
               modulation_features=48)
    oracle, net = build_pair(cfg, dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 64, generator=g)
    t = torch.tensor([0.25, 0.75])
    y_ref = oracle(x, t)
    y = net(x.to(dev), t.to(dev))
    assert y.shape == (2, 2, 64) and rel_err(y, y_ref) < TOL
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y.backward(gy.to(dev))
    compare_grads(net, oracle)


def test_parked_weight_gradients_are_final_at_the_block_hook(dev):
    """Split weight gradients park their second stage (ops.WgradPark) and are summed per block side; when the data-parallel
    hook is told that a block's slice of the flat gradient buffer is final, nothing of that block may still be parked -- and
    the gradients equal the oracle's with and without the hook."""
    cfg = dict(in_channels=2, channels=[32, 32], factors=[1, 2], items=[1, 2], modulation_features=32)
    oracle, net = build_pair(cfg, dev)
    inner = net.net if hasattr(net, "net") else net
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(2, 2, 2048, generator=g), torch.tensor([0.2, 0.7])
    y_ref = oracle(x, t)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    seen = {"parked": 0, "calls": 0}
    from audio_diffusion_pytorch_amd import ops as _ops
    add0, addc0 = _ops.WgradPark.add, _ops.WgradPark.add_call

    def counting_add(self, *a):
        seen["parked"] += 1
        return add0(self, *a)

    def counting_add_call(self, *a):
        seen["parked"] += 1
        return addc0(self, *a)

    parks = []
    flush0 = _ops.WgradPark.flush

    def tracking_flush(self):
        parks.append(self)
        return flush0(self)

    def hook(flat, a, b):
        seen["calls"] += 1
        assert all(not p.items and not p.calls for p in parks), \
            "a block's range was announced while weight gradients were still parked"

    _ops.WgradPark.add, _ops.WgradPark.add_call, _ops.WgradPark.flush = counting_add, counting_add_call, tracking_flush
    try:
        for use_hook in (False, True):
            inner._grad_ready_hook = hook if use_hook else None
            net.zero_grad(set_to_none=True)
            y = net(x.to(dev), t.to(dev))
            y.backward(gy.to(dev))
            compare_grads(net, oracle)
    finally:
        _ops.WgradPark.add, _ops.WgradPark.add_call, _ops.WgradPark.flush = add0, addc0, flush0
        inner._grad_ready_hook = None
    assert seen["parked"] > 0, "the test shape must take the split matrix-core weight gradient"
    assert seen["calls"] > 0



def test_deepcopy_of_a_net_keeps_working(emul):
    """An EMA-style copy.deepcopy of a net that has already run (its per-net caches -- parameter list, id() -> name map, flat
    gradient offsets -- are built) must not reuse the original's caches: the copy's step gives the original's gradients."""
    import copy
    torch.manual_seed(0)
    net = adp.UNetV0(dim=1, **TINY)
    x, t = torch.randn(2, 2, 64), torch.tensor([0.3, 0.7])
    net(x, t).square().mean().backward()
    ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    twin = copy.deepcopy(net)
    for p in twin.parameters():
        p.grad = None
    twin(x, t).square().mean().backward()
    for n, p in twin.named_parameters():
        assert torch.equal(p.grad, ref[n]), n
    assert all(a is not b for a, b in zip(net.parameters(), twin.parameters()))


def test_replaced_parameter_objects_are_followed(emul):
    """load_state_dict(assign=True) (also to_empty / overwrite-on-conversion) puts NEW Parameter objects into the modules after
    the net's per-net caches were built: the next step must key its gradients on the new objects (the kernels read them through
    the module attributes), not on the orphaned ones."""
    torch.manual_seed(0)
    net = adp.UNetV0(dim=1, **TINY)
    x, t = torch.randn(2, 2, 64), torch.tensor([0.3, 0.7])
    net(x, t).square().mean().backward()  # builds the caches
    torch.manual_seed(1)
    donor = adp.UNetV0(dim=1, **TINY)
    donor(x, t).square().mean().backward()
    ref = {n: p.grad.clone() for n, p in donor.named_parameters()}
    old = list(net.parameters())
    net.load_state_dict({k: v.clone() for k, v in donor.state_dict().items()}, assign=True)
    new = list(net.parameters())
    assert all(a is not b for a, b in zip(old, new)), "assign=True is expected to replace the Parameter objects"
    for p in new:
        p.requires_grad_(True)
        p.grad = None
    net(x, t).square().mean().backward()
    for n, p in net.named_parameters():
        assert p.grad is not None and torch.equal(p.grad, ref[n]), n
