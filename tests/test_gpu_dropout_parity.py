"""-m gpu: TRAIN-mode parity at p > 0 — the mode bench.py times — against the fp64 oracle fed the SAME dropout masks.

tests/dropmask.py restates the library's counter-based generator (site_key / rand_quad / thresholds) and each implementation's
(row, column) keying in Python; the masks a given seed produces are handed to oracle/translator_ref.py's `masks` arguments
(whose five sites are pinned to torch's own train-mode arithmetic and to the live reference class on CPU:
tests/test_oracle_dropout.py). Logits within 1e-3 (f32 / f32s) / 1e-2 (bf16), every parameter gradient within 1e-2 / 8e-2
relative — the p = 0 tolerances. A mis-scaled site (keep-scale folded into packed weights but not into a bias, positional
dropout drawn with the wrong p, a mask keyed differently in forward and backward) is an O(1) error here."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import translator_ref as tr
from tests import dropmask as dm
from tests.util import hhi_args, rel_err, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]
TOL = {"f32": (1e-3, 1e-2), "f32s": (1e-3, 1e-2), "bf16": (1e-2, 8e-2)}


def _pin_seed(model, seed):
    """Host-seed path: every forward draws its keys from `seed` (TranslatorMixin._egx_seed is the per-step counter)."""
    model._egx_seed = lambda: seed
    return model


def _sd64(sd):
    return {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}


def _check(logits, ref, named, sd64, tol_logit, tol_grad, skip=()):
    err = ((logits.detach().double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
    assert err < tol_logit, f"outputs differ from the masked oracle by {err}"
    errs = {}
    for k, p in named.items():
        if k in skip:
            continue
        g = sd64[k].grad
        if g is None:
            continue
        assert p.grad is not None, k
        errs[k] = rel_err(p.grad, g)
    bad = {k: v for k, v in errs.items() if not v < tol_grad}
    assert not bad, f"rel grad errs over {tol_grad}: {bad}"
    return err, errs


@pytest.mark.parametrize("compute", ["f32", "f32s", "bf16"])
@pytest.mark.parametrize("impl,n_tasks,B,T,L,device_seed", [
    ("fused", 3, 256, 15, 1, False),      # BASELINE.json configs[1] = the bench line (C2)
    ("fused", 3, 256, 15, 1, True),       # ... with the device-resident seed the bench's hipGraph uses
    ("fused", 2, 32, 15, 1, False),       # configs[0] (C1)
    ("fused", 3, 5, 16, 2, False),        # S = 48, two layers
    ("fused", 3, 3, 7, 3, False),         # ragged tiles, three layers
    ("generic", 3, 6, 15, 2, False),
    ("generic", 3, 4, 23, 1, False),      # S = 69: what a real TTM batch looks like (T > 16)
])
def test_ttm_train_mode_matches_oracle_under_the_same_masks(egx_lib, cuda, compute, impl, n_tasks, B, T, L, device_seed):
    if impl == "generic" and compute == "f32s":
        pytest.skip("f32s is f32 outside the fused kernels")
    from egot2_amd import hhi_ttm
    p, p_pos, seed = 0.5, 0.1, 0x5EED0000 + B * 131 + T
    cls = hhi_ttm.TaskFusionMFTransformer3Task if n_tasks == 3 else hhi_ttm.TaskFusionMFTransformer2Task
    model = cls(hhi_args(dropout=p, num_layers=L))
    sd = seeded_state_dict(model, seed=400 + n_tasks + B)
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute(compute, impl).train()
    if device_seed:
        model.enable_device_seed()
        model._egx_seed_dev.fill_(seed)
        eff_seed = dm.lcg(seed)             # the training forward advances the device seed once before drawing
    else:
        _pin_seed(model, seed)
        eff_seed = seed
    feats = seeded_feats(50 + B, [(B, T, 256)] * n_tasks)
    target = torch.from_numpy(np.random.default_rng(B).integers(0, 2, B)).long()
    logits = model.forward_features(*[f.to(cuda) for f in feats])
    loss = torch.nn.functional.cross_entropy(logits, target.to(cuda), weight=torch.tensor(CE_W, device=cuda))
    loss.backward()
    torch.cuda.synchronize()
    masks = dm.encoder_masks(eff_seed, impl, B, [T] * n_tasks, 128, 4, 2048, L, p, p_pos)
    sd64 = _sd64(sd)
    ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
    ref_loss = tr.weighted_ce(ref, target, CE_W)
    ref_loss.backward()
    tol_logit, tol_grad = TOL[compute]
    if compute == "bf16" and L >= 3:
        # three bf16 layers at p = 0.5 on three clips of 7 frames: every gradient sits at 8.0e-2 .. 9.2e-2 (the same shape at
        # p = 0 is the worst bf16 case of test_ttm_translator_vs_oracle, 7.4e-2; the 2x keep-scale doubles what a rounding
        # of a kept unit moves). A mask or scale error is O(1).
        tol_grad = 1.2e-1
    _check(logits, ref, dict(model.named_parameters()), sd64, tol_logit, tol_grad)
    assert abs(loss.item() - ref_loss.item()) < tol_logit * max(1.0, abs(ref_loss.item()))
    # and the masks matter: the eval-mode oracle is far away
    ev = tr.ttm_forward(tr.to_dtype(sd, torch.float64), 4, *[f.double() for f in feats])
    assert (ev - ref.detach()).abs().max().item() > 10 * tol_logit


@pytest.mark.parametrize("compute", ["bf16", "f32s"])
@pytest.mark.parametrize("B,T,L", [(256, 15, 2), (7, 16, 2)])
def test_asd_train_mode_matches_oracle_under_the_same_masks(egx_lib, cuda, compute, B, T, L):
    """BASELINE.json configs[2] (C3): ASD 3-task translator, token order asd, ttm, lam, per-frame output, p = 0.1 (+0.1 PE)."""
    from egot2_amd import hhi_asd
    p, p_pos, seed = 0.1, 0.1, 0xA5D0 + B
    model = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(dropout=p, num_layers=L))
    sd = seeded_state_dict(model, seed=77 + B)
    model.load_state_dict(sd)
    model = _pin_seed(model.to(cuda).set_compute(compute, "fused").train(), seed)
    feats = seeded_feats(60 + B, [(B, T, 256)] * 3)                 # forward_features(ttm_out, lam_out, asd_out)
    out = model.forward_features(*[f.to(cuda) for f in feats])
    w = torch.from_numpy(np.random.default_rng(3).standard_normal((B * T, 128))).double() / B
    (out.double() * w.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    masks = dm.encoder_masks(seed, "fused", B, [T] * 3, 128, 4, 2048, L, p, p_pos)
    sd64 = _sd64(sd)
    ref = tr.asd_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
    (ref * w).sum().backward()
    tol_logit, tol_grad = TOL[compute]
    if compute == "bf16":
        # per-TOKEN outputs (B * T * 128 LayerNorm outputs of magnitude up to ~4), not two logits per clip: the 1e-2 bar is held
        # in the L2 sense; the worst single element of the 491 520 at B = 256 after two bf16 layers is 2e-2 (the same
        # masks in f32s: 1e-6, so it is rounding, not a mask or scale error)
        assert rel_err(out, ref.detach()) < 1e-2
        tol_logit = 4e-2
    _check(out, ref, dict(model.named_parameters()), sd64, tol_logit, tol_grad)


def _lta_cfg(n, d, heads, layers, p):
    return NS(FORECASTING=NS(NUM_INPUT_CLIPS=n, NUM_ACTIONS_TO_PREDICT=3),
              MODEL=NS(TRANSLATION_HEADS=heads, TRANSLATION_LAYERS=layers, TRANSLATION_INPUT_FEATURES=d, TRANSLATION_DROPOUT=p,
                       NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))


@pytest.mark.parametrize("impl,compute,n,d,device_seed", [("wide", "bf16", 8, 256, False), ("wide", "bf16", 32, 768, False),
                                                          ("generic", "f32", 4, 256, False), ("wide", "bf16", 8, 256, True),
                                                          ("wide", "bf16", 40, 256, False), ("wide", "bf16", 36, 512, True)])
def test_lta4_train_mode_matches_oracle_under_the_same_masks(egx_lib, cuda, impl, compute, n, d, device_seed):
    """The wide bf16 path (configs[3] flavour: 4 x n tokens, 8 heads, learned positions, identity action segment) and the
    generic kernels at p = 0.3 on all four encoder-layer sites. n = 40 / 36: S = 160 / 144 > 128, the wide path's online-softmax
    attention kernels (head dim 32 / 64), whose dropout rows are keyed bh * 512 + query."""
    from egot2_amd import hoi_lta
    B, L, p, seed = 3, 2, 0.3, 0x17A4 + n
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(n, d, 8, L, p))
    sd = seeded_state_dict(m, 9 + n)
    m.load_state_dict(sd)
    m = _pin_seed(m.to(cuda).set_compute(compute, impl).train(), seed)
    if device_seed:         # the wide path reads its keys from a table derived on the stream from the device-resident seed
        m.enable_device_seed()
        m._egx_seed_dev.fill_(seed)
        seed = dm.lcg(seed)
    feats = seeded_feats(10 + n, [(B, n, 8192), (B, n, 8192), (B, n, d), (B, n, 2048)])
    outs = m.forward_features(*[f.to(cuda) for f in feats])
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    (lin(outs[0]) + lin(outs[1])).backward()
    torch.cuda.synchronize()
    masks = dm.encoder_masks(seed, impl, B, [n] * 4, d, 8, 2048, L, p)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.lta4_forward(sd64, 8, *[f.double() for f in feats], [5, 7], masks=masks)
    (lin(ref[0]) + lin(ref[1])).backward()
    tol_logit, tol_grad = TOL[compute]
    named = dict(m.named_parameters())
    for o, r in zip(outs, ref):
        assert (o.detach().cpu().double() - r.detach()).abs().max().item() < tol_logit * max(1.0, r.detach().abs().max().item())
    errs = {k: rel_err(named[k].grad, v.grad) for k, v in sd64.items() if v.grad is not None and k in named}
    bad = {k: v for k, v in errs.items() if not v < tol_grad}
    assert not bad, bad


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_pnr3_feature_dropout_matches_oracle_under_the_same_masks(egx_lib, cuda, compute):
    """The shipped PNR / OSCC recipe's sites: feature dropout on every projected segment (`self.dp(proj_k(.))`,
    HOI/models/pnr/video_model_transfer_3task.py:249-252) + the encoder layer's four, 8 heads of 16, d_ff = 2 d."""
    from egot2_amd import hoi_pnr
    B, L, d, p, p_feat, seed = 4, 2, 128, 0.2, 0.3, 0xF0A7
    cfg = NS(DATA=NS(TASK="state_change_detection"),
             MODEL=NS(TRANSLATION_INPUT_FEATURES=d, TRANSLATION_LAYERS=L, FEAT_DROPOUT_RATE=p_feat, TRANSFORMER_DROPOUT_RATE=p))
    m = hoi_pnr.TaskFusionMFTransformer3TaskDropout(cfg)
    m.load_state_dict(seeded_state_dict(m, 31))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}     # `ln` is shared with linear_head.0: read back what was loaded
    m = _pin_seed(m.to(cuda).set_compute(compute).train(), seed)
    feats = seeded_feats(32, [(B, 16, 8192), (B, 16, 8192), (B, 8, 2048), (B, 8, 256)])
    out = m.forward_features(*[f.to(cuda) for f in feats])
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    lin(out).backward()
    torch.cuda.synchronize()
    from egot2_amd import functional as F_egx
    impl = F_egx.last_encoder_impl()
    masks = dm.encoder_masks(seed, impl, B, [16, 16, 8, 8], d, 8, 2 * d, L, p, 0.0, p_feat)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.pnr3_forward(sd64, 8, *[f.double() for f in feats], masks=masks).unsqueeze(2)
    lin(ref).backward()
    tol_logit, tol_grad = TOL[compute]
    named = dict(m.named_parameters())
    # linear_head.0.* is the shared `ln`: named_parameters lists it once, as ln.*
    _check(out, ref, {k: v for k, v in named.items()}, sd64, tol_logit, tol_grad)
