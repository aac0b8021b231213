"""-m gpu: round-6 step fusions of the per-clip path — the weighted cross entropy evaluated in the forward's head epilogue (egx_ce), the
persistent packed-weight cache (egx_config.weight_cache) and the device seed advanced by the backward (egx_config.advance_seed = 2).
Each is checked against the path it replaces (separate CE launch, packing launch in every forward, seed advanced by the forward) and, for
the loss, against the fp64 oracle (HHI/tasks/ttm/video_task_2loader.py:21-22,34 = nn.CrossEntropyLoss(weight=[0.266, 0.734]))."""
import os

import pytest
import torch

from tests.util import hhi_args, max_err, rel_err, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]


def _model(cuda, compute="f32s", p=0.0, tasks=3, layers=1):
    from egot2_amd import hhi_ttm
    cls = hhi_ttm.TaskFusionMFTransformer3Task if tasks == 3 else hhi_ttm.TaskFusionMFTransformer2Task
    m = cls(hhi_args(dropout=p, num_layers=layers))
    m.load_state_dict(seeded_state_dict(m, 21))
    m = m.to(cuda).set_compute(compute).train()
    m.pos_embed.dropout.p = 0.0 if p == 0.0 else m.pos_embed.dropout.p
    return m


def _same(a, b, rtol):
    """||a - b|| <= rtol ||b|| + a floor for gradients that cancel to nearly zero (a head bias: the sum over the clips of d_logits), whose
    relative error under another atomic-add order is noise over ~0."""
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).norm().item() <= rtol * b.norm().item() + 2e-7 * (b.numel() ** 0.5)


def _grads(m):
    return {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("compute,B,cut", [("f32s", 256, None), ("f32s", 7, "1"), ("f32s", 7, "0"), ("f32", 33, None), ("bf16", 256, None),
                                           ("bf16", 5, "1")])
def test_fused_cross_entropy_equals_separate_launch_and_oracle(egx_lib, cuda, compute, B, cut):
    """(logits, loss) from one forward call: same logits and gradients as forward + egx_weighted_ce, loss within 1e-6, and the loss against
    the fp64 oracle. Labels outside [0, C) (ignore_index = -100) carry neither loss, weight nor gradient. EGX_FFN_SLICES=1: one workgroup
    per clip, so that EGX_FFN_CUT picks the launch whose epilogue is under test (ffn_fwd_kernel / fused_fwd_kernel)."""
    from egot2_amd import functional as F_egx
    from oracle import translator_ref as tr
    m = _model(cuda, compute)
    feats = [f.to(cuda) for f in seeded_feats(31, [(B, 15, 256)] * 3)]
    g = torch.Generator().manual_seed(5)
    target = torch.randint(0, 2, (B,), generator=g)
    with_oracle = compute != "bf16" and B <= 33
    if not with_oracle:
        target[B // 2] = -100           # (the oracle's weighted_ce has no ignore_index)
    target = target.to(cuda)
    w = torch.tensor(CE_W, device=cuda)
    with _Env(EGX_FFN_CUT=cut, EGX_FFN_SLICES="1" if cut is not None else None):
        m.zero_grad()
        logits_a = m.forward_features(*feats)
        loss_a = F_egx.weighted_cross_entropy(logits_a, target, w)
        loss_a.backward()
        ga = _grads(m)
        m.zero_grad()
        logits_b, loss_b = m.forward_features(*feats, target=target, class_weight=w)
        loss_b.backward()
        gb = _grads(m)
    torch.cuda.synchronize()
    assert torch.equal(logits_a, logits_b)
    assert abs(loss_a.item() - loss_b.item()) < 2e-6 * max(1.0, abs(loss_a.item()))
    # (atomic accumulation order differs run to run; bf16: d_logits differing in the last bit flips roundings of bf16 operands downstream)
    for k in ga:
        assert _same(gb[k], ga[k], 5e-3 if compute == "bf16" else 2e-5), k
    ref = torch.nn.functional.cross_entropy(logits_b.double().cpu(), target.cpu(), weight=torch.tensor(CE_W, dtype=torch.float64))
    assert abs(ref.item() - loss_b.item()) < 1e-5
    if with_oracle:
        sd64 = {k: v.detach().double().cpu().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in m.state_dict().items()}
        out = tr.ttm_forward(sd64, 4, *[f.double().cpu() for f in feats])
        tr.weighted_ce(out, target.cpu(), CE_W).backward()
        assert max_err(logits_b, out.detach()) < 1e-3
        name = "transformer_encoder.layers.0.linear1.weight"
        assert rel_err(gb[name], sd64[name].grad) < 1e-2


def test_fused_cross_entropy_upstream_gradients(egx_lib, cuda):
    """The loss's upstream gradient reaches the kernels as a device scalar (3 * loss), a gradient into the logits themselves is added by
    torch ops, an unused loss leaves the plain logits path; no class weights = plain mean cross entropy."""
    m = _model(cuda, "f32s")
    B = 9
    feats = [f.to(cuda) for f in seeded_feats(32, [(B, 15, 256)] * 3)]
    target = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1, 1], device=cuda)

    def run(fused, fn):
        m.zero_grad()
        if fused:
            logits, loss = m.forward_features(*feats, target=target)
        else:
            logits = m.forward_features(*feats)
            loss = torch.nn.functional.cross_entropy(logits, target)
        fn(logits, loss).backward()
        return _grads(m)

    for fn in (lambda z, l: 3.0 * l, lambda z, l: l + (z * z).sum() * 0.1, lambda z, l: (z * z).sum()):
        a, b = run(False, fn), run(True, fn)
        for k in a:
            assert _same(b[k], a[k], 2e-5), k


def test_fused_cross_entropy_without_valid_label_is_nan_like_torch(egx_lib, cuda):
    m = _model(cuda, "f32s")
    feats = [f.to(cuda) for f in seeded_feats(33, [(3, 15, 256)] * 3)]
    target = torch.full((3,), -100, dtype=torch.int64, device=cuda)
    with torch.no_grad():
        _, loss = m.forward_features(*feats, target=target, class_weight=torch.tensor(CE_W, device=cuda))
    assert torch.isnan(loss).item()


def test_fused_cross_entropy_on_other_implementations(egx_lib, cuda):
    """Tiled kernels (S > 48) and the deterministic mode append the egx_weighted_ce launch inside the library call: same API, same numbers."""
    from egot2_amd import functional as F_egx
    m = _model(cuda, "f32s")
    B = 4
    feats = [f.to(cuda) for f in seeded_feats(34, [(B, 20, 256)] * 3)]      # S = 60: tiled
    target = torch.tensor([1, 0, 0, 1], device=cuda)
    w = torch.tensor(CE_W, device=cuda)
    for det in (False, True):
        m.set_deterministic(det)
        m.zero_grad()
        z = m.forward_features(*feats)
        F_egx.weighted_cross_entropy(z, target, w).backward()
        ga = _grads(m)
        m.zero_grad()
        z2, loss = m.forward_features(*feats, target=target, class_weight=w)
        loss.backward()
        gb = _grads(m)
        assert torch.equal(z, z2)
        for k in ga:
            assert _same(gb[k], ga[k], 2e-5), k
    m.set_deterministic(False)
    assert F_egx.last_encoder_impl() == "tiled"


@pytest.mark.parametrize("compute", ["f32s", "bf16", "f32"])
def test_weight_cache_skips_the_packing_launch_and_follows_the_weights(egx_lib, cuda, compute):
    """Second forward with unchanged weights: a cache hit, one library launch less, bit-identical logits. An in-place torch update, a change of
    the dropout probability (the FFN keep-scale rides on the packed W1) and note_weights_changed() each force a re-pack; the results follow
    a model without the cache."""
    from egot2_amd import functional as F_egx
    m = _model(cuda, compute, p=0.0)
    ref = _model(cuda, compute, p=0.0)
    nb = 130        # (above the sliced mode of small batches, whose flag words the packing launch zeroes: that launch then stays)
    feats = [f.to(cuda) for f in seeded_feats(35, [(nb, 15, 256)] * 3)]
    target = torch.randint(0, 2, (nb,), generator=torch.Generator().manual_seed(1)).to(cuda)
    m.enable_weight_cache()
    wc = m._egx_wcache

    def step(mm):
        mm.zero_grad()
        z = mm.forward_features(*feats)
        torch.nn.functional.cross_entropy(z, target).backward()
        return z.detach().clone(), _grads(mm)

    egx_lib.egx_launch_count(1)
    z1, g1 = step(m)
    n1 = egx_lib.egx_launch_count(1)
    z2, g2 = step(m)
    n2 = egx_lib.egx_launch_count(1)
    assert (wc.packs, wc.hits) == (1, 1) and n2 == n1 - 1
    zr, gr = step(ref)
    assert torch.equal(z1, z2) and torch.equal(z1, zr)
    for k in gr:
        assert _same(g2[k], gr[k], 2e-5), k
    # in-place update through torch: version counters move
    with torch.no_grad():
        for mm in (m, ref):
            mm.transformer_encoder.layers[0].linear1.weight.mul_(1.25)
            mm.proj_lam.weight.add_(0.01)
    z3, _ = step(m)
    zr3, _ = step(ref)
    assert wc.packs == 2 and torch.equal(z3, zr3) and not torch.equal(z3, z1)
    # raw-pointer style update (what FusedAdam does): invisible to torch, announced by note_weights_changed()
    with torch.no_grad():
        for mm in (m, ref):
            mm.transformer_encoder.layers[0].linear2.weight.data.mul_(0.5)       # .data: no version bump
    F_egx.note_weights_changed()
    z4, _ = step(m)
    zr4, _ = step(ref)
    assert wc.packs == 3 and torch.equal(z4, zr4)
    # the FFN dropout keep-scale is part of the packed W1 / W2^T
    m.dp_rate = ref.dp_rate = 0.25
    m.enable_device_seed(); ref.enable_device_seed()
    m._egx_seed_dev.fill_(77); ref._egx_seed_dev.fill_(77)
    z5, _ = step(m)
    zr5, _ = step(ref)
    assert wc.packs == 4 and torch.equal(z5, zr5)
    m.eval(); ref.eval()
    with torch.no_grad():
        assert torch.equal(m.forward_features(*feats), ref.forward_features(*feats))
    assert wc.packs == 5                                    # eval: keep-scale 1 again


def test_frozen_cache_graph_step_has_no_launch_in_front_of_the_forward(egx_lib, cuda):
    """bench.py's step: frozen weight cache + fused cross entropy + device seed advanced by the backward, captured as one hipGraph. The graph
    holds two launches less than the round-5 step (packing, cross entropy), every replay draws fresh masks (the seed moves exactly one LCG
    step per replay), and a replay from a pinned seed equals the eager round-5 step from the same seed."""
    from egot2_amd import functional as F_egx
    B = 64
    feats = [f.to(cuda) for f in seeded_feats(36, [(B, 15, 256)] * 3)]
    target = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(2)).to(cuda)
    w = torch.tensor(CE_W, device=cuda)
    one = F_egx.unit_grad(cuda)
    lcg = lambda s: (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)      # noqa: E731
    s64 = lambda x: x - (1 << 64) if x >= (1 << 63) else x                           # noqa: E731  (the seed tensor is int64)

    with _Env(EGX_FFN_SLICES="1"):
        old = _model(cuda, "f32s", p=0.5).enable_device_seed()
        new = _model(cuda, "f32s", p=0.5).enable_device_seed().enable_weight_cache(frozen=True)

        def step_old():
            old.zero_grad()
            loss = F_egx.weighted_cross_entropy(old.forward_features(*feats), target, w)
            loss.backward(gradient=one)
            return loss

        def step_new():
            for p in new.parameters():
                p.grad = None
            loss = new.forward_features(*feats, target=target, class_weight=w)[1]
            loss.backward(gradient=one)
            return loss

        egx_lib.egx_launch_count(1)
        step_old()
        n_old = egx_lib.egx_launch_count(1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step_new()
            step_new()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        egx_lib.egx_launch_count(1)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            loss_new = step_new()
        n_new = egx_lib.egx_launch_count(1)
        assert n_new == n_old - 2, (n_old, n_new)

        s0 = 4242
        new._egx_seed_dev.fill_(s64(lcg(s0)))        # the round-5 forward advances BEFORE it uses the seed; the new step uses it as it is
        old._egx_seed_dev.fill_(s0)
        gr.replay()
        l_old = step_old()
        torch.cuda.synchronize()
        assert new._egx_seed_dev.item() % (1 << 64) == lcg(lcg(s0)) and old._egx_seed_dev.item() % (1 << 64) == lcg(s0)
        assert abs(loss_new.item() - l_old.item()) < 1e-5
        g_old, g_new = _grads(old), _grads(new)
        for k in g_old:
            assert _same(g_new[k], g_old[k], 2e-5), k
        a = loss_new.item()
        gr.replay()
        torch.cuda.synchronize()
        assert loss_new.item() != a, "a replay must draw fresh dropout masks"


@pytest.mark.parametrize("compute,B", [("f32s", 256), ("bf16", 256), ("f32s", 40), ("f32", 130)])
def test_default_backward_is_bit_reproducible(egx_lib, cuda, compute, B):
    """Round 6: every cross-workgroup sum of the per-clip backward runs in a fixed order in the DEFAULT mode (small_dw tiles + tail_reduce_kernel
    instead of float atomics): the same inputs, weights and dropout seed give bit-identical gradients run to run — at the bench batch in cut mode
    (f32s), with the one-launch kernels (bf16, exact fp32) and in the sliced mode of small batches (B = 40). Until round 5 that took
    set_deterministic(True) and three slow reduction passes (+8 % on the step)."""
    m = _model(cuda, compute, p=0.5).enable_device_seed()
    feats = [f.to(cuda) for f in seeded_feats(55, [(B, 15, 256)] * 3)]
    target = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(9)).to(cuda)
    w = torch.tensor(CE_W, device=cuda)
    runs = []
    for _ in range(3):
        m._egx_seed_dev.fill_(31337)
        m.zero_grad()
        _, loss = m.forward_features(*feats, target=target, class_weight=w)
        loss.backward()
        torch.cuda.synchronize()
        runs.append(_grads(m))
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k


# ---- egx_token_ce: the ASD task's lossAV evaluated by the encoder's own launches ------------------------------------------------------------------
def _asd(cuda, compute, p=0.0, layers=2, cache=True):
    from egot2_amd import hhi_asd
    m = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(dropout=p, num_layers=layers))
    m.load_state_dict(seeded_state_dict(m, 31))
    m = m.to(cuda).set_compute(compute).train()
    if p == 0.0:
        m.pos_embed.dropout.p = 0.0
    if cache:
        m.enable_weight_cache()
    head = hhi_asd.lossAV(128)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        head.FC.weight.copy_(torch.randn(2, 128, generator=g) * 0.2)
        head.FC.bias.copy_(torch.randn(2, generator=g) * 0.1)
    return m, head.to(cuda)


@pytest.mark.parametrize("compute,B,T,p", [("bf16", 256, 15, 0.0), ("bf16", 130, 9, 0.0), ("f32s", 200, 15, 0.0), ("f32", 140, 16, 0.0), ("f32s", 256, 15, 0.3),
                                           ("bf16", 160, 15, 0.1)])
def test_fused_lossav_equals_the_two_launch_head(egx_lib, cuda, compute, B, T, p):
    """model.forward_features(..., lossav=head, labels=y) (egx_token_ce: classifier + weighted CE + scores in the launch that normalises the last
    layer's tokens; d tokens and the classifier's gradients rebuilt by the backward's first launch) against head(model.forward_features(...), y)
    (egx_linear_ce_fwd / _bwd on the returned tokens): same scores, labels and counts, loss within 1e-6, every gradient of the translator and of
    the classifier within rounding of the other summation order; with a loss scale as the upstream gradient; with dropout under the same seed."""
    from egot2_amd import functional as F_egx
    feats = [f.to(cuda) for f in seeded_feats(77, [(B, T, 256)] * 3)]
    y = torch.randint(0, 2, (B * T,), generator=torch.Generator().manual_seed(3)).to(cuda)
    y[5] = -100     # ignored frame: no loss, no weight, no gradient
    out = {}
    for fused in (True, False):
        m, head = _asd(cuda, compute, p=p)
        m._egx_seed = lambda: 0x5EED77       # (p > 0: both forms draw the same masks)
        if fused:
            nloss, score, label, correct = m.forward_features(*feats, lossav=head, labels=y)
            assert F_egx.last_encoder_impl() == "fused"
        else:
            nloss, score, label, correct = head(m.forward_features(*feats), y)
        (nloss * 3.0).backward()
        torch.cuda.synchronize()
        out[fused] = (nloss.detach(), score.detach(), label.detach(), correct.detach(), _grads(m), _grads(head))
    a, b = out[True], out[False]
    assert abs(a[0].item() - b[0].item()) <= 1e-6 * max(1.0, abs(b[0].item()))
    assert max_err(a[1], b[1]) < 1e-6 and torch.equal(a[2], b[2]) and a[3].item() == b[3].item()
    tol = 5e-3 if compute == "bf16" else 2e-5
    assert set(a[4]) == set(b[4]) and set(a[5]) == set(b[5]) == {"FC.weight", "FC.bias"}
    bad = {k: rel_err(a[4][k], b[4][k]) for k in b[4] if not _same(a[4][k], b[4][k], tol)}
    bad.update({k: rel_err(a[5][k], b[5][k]) for k in b[5] if not _same(a[5][k], b[5][k], 2e-5)})
    assert not bad, bad


def test_fused_lossav_against_the_oracle(egx_lib, cuda):
    """The fused lossAV of the ASD translator (two layers, f32s) against the fp64 oracle: model_taskspecific.py:139-158 + tasks/asd/loss.py:11-30."""
    from oracle import translator_ref as tr
    B, T = 140, 15
    m, head = _asd(cuda, "f32s")
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    feats = seeded_feats(78, [(B, T, 256)] * 3)
    y = torch.randint(0, 2, (B * T,), generator=torch.Generator().manual_seed(4))
    nloss, score, label, correct = m.forward_features(*[f.to(cuda) for f in feats], lossav=head, labels=y.to(cuda))
    nloss.backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    W64 = head.FC.weight.detach().cpu().double().requires_grad_(True)
    b64 = head.FC.bias.detach().cpu().double().requires_grad_(True)
    x = tr.asd_forward(sd64, 4, *[f.double() for f in feats])
    z = x @ W64.t() + b64
    ref = tr.weighted_ce(z, y, [1.0, 4.0])
    ref.backward()
    assert abs(nloss.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))
    pr = torch.softmax(z.detach(), dim=-1)
    assert max_err(score, pr) < 1e-4
    assert correct.item() == float((torch.round(pr)[:, 1] == y.double()).sum().item())
    assert rel_err(head.FC.weight.grad, W64.grad) < 1e-3 and rel_err(head.FC.bias.grad, b64.grad) < 1e-3
    named = dict(m.named_parameters())
    errs = {k: rel_err(named[k].grad, v.grad) for k, v in sd64.items() if v.grad is not None and k in named and named[k].grad is not None and v.grad.norm() > 0}
    assert len(errs) > 20 and max(errs.values()) < 1e-3, {k: e for k, e in errs.items() if e >= 1e-3}


def test_fused_lossav_falls_back_where_the_kernels_do_not_fuse_it(egx_lib, cuda):
    """Without a weight cache, in deterministic mode, on a sliced small batch or a long (tiled) clip the same call composes the encoder with
    egx_linear_ce_fwd / _bwd: same values as the explicit two-step form."""
    for kind in ("nocache", "small", "long"):
        B, T = (6, 15) if kind == "small" else (3, 60) if kind == "long" else (140, 15)
        feats = [f.to(cuda) for f in seeded_feats(79, [(B, T, 256)] * 3)]
        y = torch.randint(0, 2, (B * T,), generator=torch.Generator().manual_seed(6)).to(cuda)
        res = []
        for fused_call in (True, False):
            m, head = _asd(cuda, "f32s", cache=kind != "nocache")
            if fused_call:
                nloss, score, label, correct = m.forward_features(*feats, lossav=head, labels=y)
            else:
                nloss, score, label, correct = head(m.forward_features(*feats), y)
            nloss.backward()
            torch.cuda.synchronize()
            res.append((nloss.detach(), score.detach(), _grads(m), _grads(head)))
        assert abs(res[0][0].item() - res[1][0].item()) <= 1e-6 and max_err(res[0][1], res[1][1]) < 1e-6, kind
        assert all(_same(res[0][2][k], res[1][2][k], 2e-5) for k in res[1][2]), kind
        assert all(_same(res[0][3][k], res[1][3][k], 2e-5) for k in res[1][3]), kind


def test_fused_lossav_step_replays_in_a_graph(egx_lib, cuda):
    """The C3 bench step (fused lossAV, frozen weight cache, device seed) captured once and replayed: the loss of every replay equals the eager
    step's on the same seed state, and the arrival counter leaves the control block clean (a second model sharing nothing still gets its loss)."""
    from egot2_amd import synth
    wl = synth.make_workload("c3", cuda, batch=160, frames=15, dtype="bf16", dropout=0.0)
    model = wl["model"]
    model.pos_embed.dropout.p = 0.0
    model.enable_weight_cache(frozen=True)
    params = [p for p in model.parameters() if p.requires_grad] + list(getattr(model, "extra_params", []))
    def step():
        for p in params:
            p.grad = None
        loss = wl["loss_fn"]()
        loss.backward()
        return loss
    eager = [step().item() for _ in range(3)]
    assert abs(eager[0] - eager[2]) < 1e-6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        with torch.cuda.graph(g, stream=s):
            out = step()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        assert abs(out.item() - eager[0]) < 1e-6
