"""-m gpu: the tiled d = 128 path (EGX_IMPL_TILED: 48 < S <= 512) — the per-clip kernels over 48-token tiles with the attention of
the whole clip between the launches — against the fp64 oracle. These are the reference's REAL TTM / ASD batch shapes: segments
of 15 .. 150 frames per task (HHI/dataset/ttm/data_loader_2task.py:119,150-162), a batch truncated to its shortest member
(HHI/utils/ttm/utils.py:232-241), B * T ~ 400 (HHI/dataset/ttm/sampler.py:41), validation one clip of up to 150 frames.
Tolerances are those of the per-clip kernels (tests/test_gpu_translator.py): logits 1e-3 (f32s) / 1e-2 (bf16), gradients
1e-2 / 8e-2 relative; train-mode cases are compared with the oracle under the SAME dropout masks (tests/dropmask.py)."""
import numpy as np
import pytest
import torch

from oracle import translator_ref as tr
from tests import dropmask as dm
from tests.util import hhi_args, rel_err, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]
TOL = {"f32s": (1e-3, 1e-2), "bf16": (1e-2, 8e-2)}


def _sd64(sd):
    return {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}


@pytest.mark.parametrize("compute", ["f32s", "bf16"])
@pytest.mark.parametrize("n_tasks,B,T,L,p", [
    (3, 5, 17, 1, 0.0),       # S = 51: one full tile + 3 tokens
    (3, 17, 23, 2, 0.0),      # S = 69 (B * T ~ 400 as the reference's sampler makes them)
    (3, 13, 30, 1, 0.5),      # S = 90, train mode p = 0.5 (+ 0.1 PE) under the oracle's masks
    (2, 7, 60, 2, 0.0),       # two tasks, S = 120
    (3, 6, 60, 1, 0.5),       # S = 180 (the 256-key attention instantiation)
    (3, 3, 150, 1, 0.0),      # S = 450: the longest training / validation segment
    (3, 1, 150, 2, 0.5),      # validation batch of ONE clip, two layers, train-mode masks
    (3, 2, 32, 1, 0.0),       # S = 96: exactly two tiles
    (2, 2, 160, 1, 0.5),      # S = 320: the largest clip whose f32s planes are ONE chunk of the LDS operand
    (3, 2, 107, 1, 0.5),      # S = 321 (padded 352): two chunks of 192 rows in f32s, the last K-block half empty, masks across the chunk edge
])
def test_tiled_ttm_translator_vs_oracle(egx_lib, cuda, compute, n_tasks, B, T, L, p):
    from egot2_amd import functional as F_egx, hhi_ttm
    cls = hhi_ttm.TaskFusionMFTransformer3Task if n_tasks == 3 else hhi_ttm.TaskFusionMFTransformer2Task
    model = cls(hhi_args(num_layers=L, dropout=p))
    sd = seeded_state_dict(model, seed=500 + n_tasks + B + T)
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute(compute).train()          # impl "auto": must pick the tiled path by itself
    seed = 0x71ED0000 + 977 * B + T
    p_pos = 0.1 if p > 0 else 0.0
    model.pos_embed.dropout.p = p_pos
    model._egx_seed = lambda: seed
    feats = seeded_feats(70 + B + T, [(B, T, 256)] * n_tasks)
    target = torch.from_numpy(np.random.default_rng(B).integers(0, 2, B)).long()
    logits = model.forward_features(*[f.to(cuda) for f in feats])
    assert F_egx.last_encoder_impl() == "tiled"
    loss = torch.nn.functional.cross_entropy(logits, target.to(cuda), weight=torch.tensor(CE_W, device=cuda))
    loss.backward()
    torch.cuda.synchronize()
    masks = dm.encoder_masks(seed, "tiled", B, [T] * n_tasks, 128, 4, 2048, L, p, p_pos) if p > 0 else None
    sd64 = _sd64(sd)
    ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
    ref_loss = tr.weighted_ce(ref, target, CE_W)
    ref_loss.backward()
    tol_logit, tol_grad = TOL[compute]
    err = ((logits.detach().double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
    assert err < tol_logit, err
    assert abs(loss.item() - ref_loss.item()) < tol_logit * max(1.0, abs(ref_loss.item()))
    named = dict(model.named_parameters())
    errs = {k: rel_err(named[k].grad, v.grad) for k, v in sd64.items() if v.grad is not None}
    assert set(errs) == set(k for k, q in named.items() if q.grad is not None)
    bad = {k: v for k, v in errs.items() if not v < tol_grad}
    assert not bad, bad


@pytest.mark.parametrize("compute", ["f32s", "bf16"])
@pytest.mark.parametrize("B,T,L,p", [(9, 30, 2, 0.1), (2, 150, 1, 0.0)])
def test_tiled_asd_translator_vs_oracle(egx_lib, cuda, compute, B, T, L, p):
    """HHI/models/asd TaskFusionMFTransformer3Task at real lengths: token order asd, ttm, lam, per-frame output = the first T
    tokens of every clip (the slice is taken on the host outside the per-clip kernels)."""
    from egot2_amd import functional as F_egx, hhi_asd
    model = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(dropout=p, num_layers=L))
    sd = seeded_state_dict(model, seed=90 + B)
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute(compute).train()
    seed = 0xA5D1 + B
    p_pos = 0.1 if p > 0 else 0.0
    model.pos_embed.dropout.p = p_pos
    model._egx_seed = lambda: seed
    feats = seeded_feats(61 + B, [(B, T, 256)] * 3)
    out = model.forward_features(*[f.to(cuda) for f in feats])
    assert F_egx.last_encoder_impl() == "tiled"
    w = torch.from_numpy(np.random.default_rng(3).standard_normal((B * T, 128))).double() / B
    (out.double() * w.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    masks = dm.encoder_masks(seed, "tiled", B, [T] * 3, 128, 4, 2048, L, p, p_pos) if p > 0 else None
    sd64 = _sd64(sd)
    ref = tr.asd_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
    (ref * w).sum().backward()
    tol_logit, tol_grad = TOL[compute]
    if compute == "bf16":       # per-token outputs: the 1e-2 bar in the L2 sense (tests/test_gpu_dropout_parity.py)
        assert rel_err(out, ref.detach()) < 1e-2
        tol_logit = 4e-2
    err = ((out.detach().double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
    assert err < tol_logit, err
    named = dict(model.named_parameters())
    errs = {k: rel_err(named[k].grad, v.grad) for k, v in sd64.items() if v.grad is not None}
    bad = {k: v for k, v in errs.items() if not v < tol_grad}
    assert not bad, bad


def test_tiled_matches_per_clip_kernels_on_shared_work(egx_lib, cuda):
    """Same clips through the tiled path (forced: T = 20 -> S = 60) and, truncated to S = 45, nothing in common — so instead:
    tiled vs generic on identical inputs in eval mode, and tiled is deterministic run to run in forward."""
    from egot2_amd import hhi_ttm
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(num_layers=2))
    m.load_state_dict(seeded_state_dict(m, 3))
    m = m.to(cuda).eval()
    feats = [f.to(cuda) for f in seeded_feats(4, [(6, 20, 256)] * 3)]
    with torch.no_grad():
        a = m.set_compute("f32s", "tiled").forward_features(*feats)
        b = m.set_compute("f32s", "tiled").forward_features(*feats)
        g = m.set_compute("f32", "generic").forward_features(*feats)
    assert torch.equal(a, b)
    assert (a - g).abs().max().item() < 1e-4


@pytest.mark.parametrize("compute", ["f32s", "bf16"])
def test_tiled_deterministic_mode_repeats_bit_for_bit(egx_lib, cuda, compute):
    """set_deterministic() on the tiled path (two layers, train-mode masks, the pooled head whose backward runs inside the first backward
    tile launch and leaves its parameter gradients in the partial row of a clip's first tile): logits and every gradient repeat bit for
    bit, and agree with the default (atomic) mode to summation-order noise."""
    from egot2_amd import functional as F_egx, hhi_ttm
    B, T = 21, 40                                   # S = 120: three tiles per clip, the last one half empty
    feats = [f.to(cuda) for f in seeded_feats(91, [(B, T, 256)] * 3)]
    target = (torch.arange(B, device=cuda) * 5) % 2
    w = torch.tensor(CE_W, device=cuda)

    def run(det):
        m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(num_layers=2, dropout=0.3))
        m.load_state_dict(seeded_state_dict(m, 17))
        m = m.to(cuda).set_compute(compute).set_deterministic(det).train()
        m._egx_seed = lambda: 0x7D37
        logits = m.forward_features(*feats)
        assert F_egx.last_encoder_impl() == "tiled"
        torch.nn.functional.cross_entropy(logits, target, weight=w).backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0])
    assert set(a[1]) == set(c[1]) and any(k.startswith("mlp_head") or "head" in k for k in a[1])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
        assert torch.allclose(a[1][k], c[1][k], rtol=3e-3, atol=2e-6), k


def test_tiled_refuses_what_it_cannot_run(egx_lib, cuda):
    from egot2_amd import hhi_ttm, _lib
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args())
    m = m.to(cuda).eval()
    short = [f.to(cuda) for f in seeded_feats(4, [(2, 15, 256)] * 3)]
    with pytest.raises(_lib.EgxError, match="tiled"):
        m.set_compute("f32s", "tiled").forward_features(*short)          # S = 45: the per-clip kernels' shape
    long_ = [f.to(cuda) for f in seeded_feats(4, [(2, 20, 256)] * 3)]
    with pytest.raises(_lib.EgxError, match="tiled"):
        m.set_compute("f32", "tiled").forward_features(*long_)           # exact-fp32 MFMA stays on the generic kernels
