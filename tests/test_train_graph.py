"""The README loop `loss = model(x); loss.backward()` (/root/reference/README.md:36-39) served by the graph-replayed step
(audio_diffusion_pytorch_amd/graphed.py): same numbers as the eager launches, autograd semantics kept."""
import copy

import pytest
import torch

import audio_diffusion_pytorch_amd as adp
from audio_diffusion_pytorch_amd import graphed

TINY = dict(in_channels=2, channels=[8, 32, 64], factors=[1, 4, 4], items=[1, 2, 2], modulation_features=128)


def _model(dev, seed=0, **extra):
    torch.manual_seed(seed)
    return adp.DiffusionModel(net_t=adp.UNetV0, **TINY, **extra).to(dev)


def _zero(m):
    for p in m.parameters():
        p.grad = None


def test_tracked_parameters_follows_replaced_objects():
    torch.manual_seed(0)
    net = adp.UNetV0(dim=1, **TINY)
    a = graphed.tracked_parameters(net)
    assert [id(p) for p in a] == [id(p) for p in net.parameters()]
    assert graphed.tracked_parameters(net) is a  # cached
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()}, assign=True)
    b = graphed.tracked_parameters(net)
    assert [id(p) for p in b] == [id(p) for p in net.parameters()] and b is not a


def test_cpu_tensors_never_take_the_graph_path(emul):
    m = _model(emul)
    x = torch.randn(1, 2, 256)
    m(x).backward()
    assert graphed.GRAPHS_OF.get(m.diffusion) is None


@pytest.mark.gpu
def test_readme_loop_replays_and_matches_the_eager_step(hip):
    m_g, m_e = _model(hip), _model(hip, diffusion_use_graph=False)
    xs = [torch.randn(2, 2, 4096, device=hip) for _ in range(3)]
    for m in (m_g, m_e):
        torch.manual_seed(123)
        torch.cuda.manual_seed(123)
        m.losses, m.grads = [], []
        for x in xs:
            _zero(m)
            loss = m(x)
            loss.backward()
            m.losses.append(loss.item())
            m.grads.append([p.grad.clone() for p in m.parameters()])
    g = graphed.GRAPHS_OF[m_g.diffusion]
    assert g.captures == 1 and g.replays == 3
    assert graphed.GRAPHS_OF.get(m_e.diffusion) is None
    # same seed -> the replayed graphs draw the eager loop's sigmas / noise; identical kernels -> identical numbers
    for i in range(3):
        assert m_g.losses[i] == m_e.losses[i], (i, m_g.losses, m_e.losses)
        for a, b in zip(m_g.grads[i], m_e.grads[i]):
            assert torch.equal(a, b)
    assert len({round(v, 9) for v in m_g.losses}) == 3, "every replay must draw fresh sigmas / noise"


@pytest.mark.gpu
def test_replayed_step_keeps_autograd_semantics(hip):
    m = _model(hip)
    x, noise = torch.randn(2, 2, 4096, device=hip), torch.randn(2, 2, 4096, device=hip)
    ref = _model(hip, diffusion_use_graph=False)

    def run(model, scale, accumulate):
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        _zero(model)
        for _ in range(accumulate):
            (model(x, noise=noise) * scale).backward()
        return [p.grad.clone() for p in model.parameters()]

    # loss scaling: the incoming gradient is an input of the backward graph
    for a, b in zip(run(m, 0.25, 1), run(ref, 0.25, 1)):
        assert torch.allclose(a, b, rtol=1e-6, atol=0)
    # gradient accumulation over two backward passes without zeroing in between (the first pass's .grad is a view of the
    # graph's static buffer: it must be moved before the second replay overwrites it)
    for a, b in zip(run(m, 1.0, 2), run(ref, 1.0, 2)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-9)
    # parameter hooks fire (torch's own DistributedDataParallel hangs its reducer on them)
    seen = []
    p0 = next(m.parameters())
    h = p0.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.abs().sum().item()))
    _zero(m)
    m(x).backward()
    h.remove()
    assert len(seen) == 1 and seen[0] > 0
    # a stale forward is refused instead of silently differentiating the wrong activations
    l1 = m(x)
    m(x)
    with pytest.raises(RuntimeError, match="no longer the last one"):
        l1.backward()
    # no_grad / CPU-side validation loops stay eager and leave the graphs alone
    g = graphed.GRAPHS_OF[m.diffusion]
    n = g.replays
    with torch.no_grad():
        m(x)
    assert g.replays == n
    # an EMA copy after a capture: deep-copyable, and the copy captures its own step
    twin = copy.deepcopy(m)
    _zero(twin)
    twin(x).backward()
    assert graphed.GRAPHS_OF[twin.diffusion] is not g
    # moved / replaced parameters: the stale graph is dropped, not replayed
    caps = g.captures
    with torch.no_grad():
        for p in m.parameters():
            p.data = p.data.clone()
    _zero(m)
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    m(x, noise=noise).backward()
    assert g.captures == caps + 1
    for a, b in zip([p.grad for p in m.parameters()], run(ref, 1.0, 1)):
        assert torch.allclose(a, b, rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_replayed_step_with_conditioning_kwargs(hip):
    """Tensor kwargs (config 5's appended channels, config 4's embedding) become static inputs refreshed per call."""
    torch.manual_seed(0)
    up_g = adp.DiffusionUpsampler(net_t=adp.UNetV0, upsample_factor=4, **TINY).to(hip)
    up_e = copy.deepcopy(up_g)
    up_e.diffusion.use_graph = False
    for x in (torch.randn(2, 2, 4096, device=hip), torch.randn(2, 2, 4096, device=hip)):
        out = []
        for m in (up_g, up_e):
            torch.manual_seed(5)
            torch.cuda.manual_seed(5)
            _zero(m)
            loss = m(x)
            loss.backward()
            out.append((loss.item(), [p.grad.clone() for p in m.parameters()]))
        assert out[0][0] == out[1][0]
        for a, b in zip(out[0][1], out[1][1]):
            assert torch.equal(a, b)
    assert graphed.GRAPHS_OF[up_g.diffusion].captures == 1


@pytest.mark.gpu
def test_training_with_an_optimizer_follows_the_eager_trajectory(hip):
    """Five AdamW steps of the README loop (zero_grad / model(x) / backward / step), replayed vs launched eagerly, same seeds:
    the parameters must end up bit-identical -- the optimizer reads gradients that alias the graph's static buffer, zero_grad drops
    them, the next replay overwrites them."""
    xs = [torch.randn(2, 2, 4096, device=hip) for _ in range(5)]
    finals = []
    for use_graph in (True, False):
        m = _model(hip, seed=3, diffusion_use_graph=use_graph)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        losses = []
        for x in xs:
            opt.zero_grad()
            loss = m(x)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        finals.append(([p.detach().clone() for p in m.parameters()], losses))
    assert finals[0][1] == finals[1][1]
    for a, b in zip(finals[0][0], finals[1][0]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_replayed_step_of_a_guided_text_conditional_net(hip):
    """UNetV0(use_embedding_cfg=True) with cross attention: `embedding` becomes a static input, the Bernoulli batch mask is drawn
    inside the captured step (fresh per replay), the fixed-embedding table OUTSIDE the U-Net gets its gradient through autograd
    inside graph B."""
    cfg = dict(in_channels=2, channels=[8, 32, 64], factors=[1, 4, 4], items=[1, 2, 2], modulation_features=64,
               cross_attentions=[0, 1, 1], attention_heads=2, attention_features=16, embedding_features=24,
               use_embedding_cfg=True, embedding_max_length=6)
    out = []
    emb = torch.randn(2, 6, 24, device=hip)
    x = torch.randn(2, 2, 2048, device=hip)
    for use_graph in (True, False):
        torch.manual_seed(5)
        m = adp.DiffusionModel(net_t=adp.UNetV0, diffusion_use_graph=use_graph, **cfg).to(hip)
        torch.manual_seed(9)
        torch.cuda.manual_seed(9)
        steps = []
        for _ in range(2):
            _zero(m)
            loss = m(x, embedding=emb, embedding_mask_proba=0.5)
            loss.backward()
            steps.append((loss.item(), [None if p.grad is None else p.grad.clone() for p in m.parameters()]))
        out.append(steps)
        if use_graph:
            assert graphed.GRAPHS_OF[m.diffusion].captures == 1 and graphed.GRAPHS_OF[m.diffusion].replays == 2
    for (la, ga), (lb, gb) in zip(out[0], out[1]):
        assert la == lb
        for a, b in zip(ga, gb):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)
