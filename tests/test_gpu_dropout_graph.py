"""-m gpu: dropout consistency (same counter-based masks in forward and backward, fresh masks per step) and
hipGraph replay of the whole training step with the device-resident seed."""
import pytest
import torch

from tests.util import hhi_args, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]


def _model(cuda, compute="f32", impl="fused", p=0.5):
    from egot2_amd import hhi_ttm
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=p))
    m.load_state_dict(seeded_state_dict(m, 9))
    return m.to(cuda).set_compute(compute, impl).train().enable_device_seed()


def _loss(m, feats, target, seed_value):
    m._egx_seed_dev.fill_(seed_value)      # the forward advances it once: same start -> same masks
    logits = m.forward_features(*feats)
    return torch.nn.functional.cross_entropy(logits, target, weight=torch.tensor(CE_W, device=logits.device))


@pytest.mark.parametrize("impl", ["fused"])
def test_dropout_masks_are_identical_in_forward_and_backward(egx_lib, cuda, impl):
    """With the seed pinned the loss is a deterministic function of the parameters (fixed masks), so central finite
    differences must reproduce the analytic gradient at every dropout site (attention, both residual branches, FFN
    hidden, positional) — they only do if backward regenerates exactly the forward's masks."""
    m = _model(cuda, impl=impl)
    feats = [f.to(cuda) for f in seeded_feats(4, [(6, 15, 256)] * 3)]
    target = torch.tensor([0, 1, 1, 0, 1, 0], device=cuda)
    a = _loss(m, feats, target, 1234)
    b = _loss(m, feats, target, 1234)
    c = _loss(m, feats, target, 99)
    assert a.item() == b.item(), "same seed must give the same masks"
    assert abs(a.item() - c.item()) > 1e-6, "a different seed must give different masks"
    m.zero_grad()
    _loss(m, feats, target, 1234).backward()
    probes = [("task_embed", (0, 1, 5)), ("ln.weight", (7,)), ("proj_lam.bias", (3,)),
              ("transformer_encoder.layers.0.norm1.weight", (11,)), ("transformer_encoder.layers.0.linear2.bias", (2,)),
              ("transformer_encoder.layers.0.self_attn.in_proj_bias", (140,)), ("linear_head.1.bias", (1,))]
    named = dict(m.named_parameters())
    eps = 2e-2
    for name, idx in probes:
        pr = named[name]
        g = pr.grad[idx].item()
        with torch.no_grad():
            old = pr[idx].item()
            pr[idx] = old + eps
            lp = _loss(m, feats, target, 1234).item()
            pr[idx] = old - eps
            lm = _loss(m, feats, target, 1234).item()
            pr[idx] = old
        fd = (lp - lm) / (2 * eps)
        # 5 %: a step of 2e-2 crosses a few of the 6 * 45 * 2048 ReLU kinks (measured 4.5 % on norm1.weight with the round-3
        # masks); masks that differed between forward and backward give errors of order 100 %
        assert abs(fd - g) < 5e-2 * max(abs(g), abs(fd)) + 2e-4, f"{name}{idx}: finite difference {fd} vs gradient {g}"


def test_keep_rate_and_expectation(egx_lib, cuda):
    """Inverted dropout: averaged over many seeds the training-mode logits approach the eval-mode logits."""
    m = _model(cuda, p=0.3)
    feats = [f.to(cuda) for f in seeded_feats(5, [(32, 15, 256)] * 3)]
    with torch.no_grad():
        acc = torch.zeros(32, 2, device=cuda)
        n = 200
        for s in range(n):
            m._egx_seed_dev.fill_(1000 + s)
            acc += m.forward_features(*feats)
        m.eval()
        ref = m.forward_features(*feats)
    # LayerNorm/softmax are nonlinear, so only the rough level has to agree
    assert (acc / n - ref).abs().mean().item() < 0.25 * ref.abs().mean().item() + 0.05


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_whole_step_graph_replay(egx_lib, cuda, compute):
    """Forward + weighted CE + backward captured once with torch.cuda.graph and replayed: gradients stay finite,
    differ between replays (fresh dropout masks from the device seed) and match eager execution at p = 0."""
    m = _model(cuda, compute=compute, p=0.5)
    feats = [f.to(cuda) for f in seeded_feats(6, [(16, 15, 256)] * 3)]
    target = torch.randint(0, 2, (16,), device=cuda)
    w = torch.tensor(CE_W, device=cuda)
    params = list(m.parameters())

    def step():
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(m.forward_features(*feats), target, weight=w)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    g.replay()
    torch.cuda.synchronize()
    g1 = m.transformer_encoder.layers[0].linear1.weight.grad.clone()
    l1 = loss.item()
    g.replay()
    torch.cuda.synchronize()
    g2 = m.transformer_encoder.layers[0].linear1.weight.grad.clone()
    assert torch.isfinite(g1).all() and torch.isfinite(g2).all()
    assert (g1 - g2).abs().max().item() > 0, "replays must draw different dropout masks"
    assert l1 != loss.item()


def test_wide_path_step_graph_replay_draws_fresh_masks(egx_lib, cuda):
    """VERDICT r3 item 8a groundwork: the wide bf16 path (and with it the EgoT2-g / LTA configurations meant for 8 GPUs) takes
    its dropout keys from a table derived ON THE STREAM from the device-resident seed, so forward + backward capture into one
    hipGraph whose replays draw fresh masks; the first replay equals an eager step from the same seed."""
    from types import SimpleNamespace as NS
    from egot2_amd import hoi_lta
    cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=4, NUM_ACTIONS_TO_PREDICT=3),
             MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=2, TRANSLATION_INPUT_FEATURES=256, TRANSLATION_DROPOUT=0.3,
                      NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
    m.load_state_dict(seeded_state_dict(m, 9))
    m = m.to(cuda).set_compute("bf16", "wide").train().enable_device_seed()
    feats = [f.to(cuda) for f in seeded_feats(10, [(4, 4, 8192), (4, 4, 8192), (4, 4, 256), (4, 4, 2048)])]
    params = [p for p in m.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        o = m.forward_features(*feats)
        loss = o[0].square().mean() + o[1].square().mean()
        loss.backward()
        return loss

    # (warm-up, eager reference and capture all run on ONE side stream: an autograd graph kept alive from the default stream
    # would leave AccumulateGrad nodes bound to it and break the capture)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m._egx_seed_dev.fill_(777)
        eager = step().item()
        g_eager = m.transformer.layers[0].linear1.weight.grad.clone()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        loss = step()
    m._egx_seed_dev.fill_(777)
    g.replay()
    torch.cuda.synchronize()
    l1, g1 = loss.item(), m.transformer.layers[0].linear1.weight.grad.clone()
    assert abs(l1 - eager) < 1e-6 * max(1.0, abs(l1)) and torch.equal(g1, g_eager)      # deterministic path: same seed, same step
    g.replay()
    torch.cuda.synchronize()
    g2 = m.transformer.layers[0].linear1.weight.grad.clone()
    assert torch.isfinite(g2).all() and (g1 - g2).abs().max().item() > 0 and loss.item() != l1


def test_capturing_training_dropout_with_a_host_seed_is_refused(egx_lib, cuda):
    """A host seed is baked into a captured graph: every replay would repeat the same masks. The library refuses that capture (fused, tiled and
    wide implementations alike) instead of training silently on one mask; with the device-resident seed the same capture goes through, and
    p = 0 needs no seed at all."""
    from egot2_amd import hhi_ttm, _lib
    feats = [f.to(cuda) for f in seeded_feats(5, [(4, 15, 256)] * 3)]

    def try_capture(model):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model.forward_features(*feats).sum().backward()        # eager warm-up: fine with any seed
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            model.forward_features(*feats).sum().backward()
        g.replay()
        torch.cuda.synchronize()

    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.3)).to(cuda).set_compute("f32s").train()
    with pytest.raises(_lib.EgxError, match="cannot be captured in a hipGraph"):
        try_capture(m)
    torch.cuda.synchronize()
    try_capture(m.enable_device_seed())                                    # device seed: allowed
    m0 = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.0)).to(cuda).set_compute("f32s").train()
    m0.pos_embed.dropout.p = 0.0
    try_capture(m0)                                                        # no dropout: nothing to bake in
