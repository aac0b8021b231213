"""-m gpu: the per-clip kernels on the HOI EgoT2-s shapes: 8 heads of 16, d_ff = 2 d, up to 6 layers, feature
dropout on the projections, a LEARNED positional table that needs its gradient, 8192-wide PNR / OSCC features — the shipped
PNR / OSCC recipe (HOI/configs/pnr/ts_pnr.yaml:28-34, HOI/models/pnr/video_model_transfer_3task.py:212-258) and the action
recognition translators (HOI/models/lta/lta_models_transfer.py). Against the fp64 oracle, eval-equivalent (p = 0) and in
train mode under the SAME dropout masks (tests/dropmask.py)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import translator_ref as tr
from tests import dropmask as dm
from tests.util import rel_err, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
TOL = {"f32": (1e-3, 1e-2), "f32s": (1e-3, 1e-2), "bf16": (1e-2, 8e-2)}


def _pnr3(cuda, L, p, p_feat, compute, impl="fused"):
    from egot2_amd import hoi_pnr
    cfg = NS(DATA=NS(TASK="state_change_detection"),
             MODEL=NS(TRANSLATION_INPUT_FEATURES=128, TRANSLATION_LAYERS=L, FEAT_DROPOUT_RATE=p_feat, TRANSFORMER_DROPOUT_RATE=p))
    m = hoi_pnr.TaskFusionMFTransformer3TaskDropout(cfg)
    m.load_state_dict(seeded_state_dict(m, 61 + L))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}       # `ln` is shared with linear_head.0
    return m.to(cuda).set_compute(compute, impl).train(), sd


@pytest.mark.parametrize("compute", ["f32", "f32s", "bf16"])
@pytest.mark.parametrize("B,L,p,p_feat", [(5, 2, 0.0, 0.0), (3, 6, 0.0, 0.0), (4, 2, 0.2, 0.3), (257, 1, 0.1, 0.1), (2, 6, 0.1, 0.2)])
def test_pnr_oscc_recipe_on_the_per_clip_kernels(egx_lib, cuda, compute, B, L, p, p_feat):
    from egot2_amd import functional as F_egx
    m, sd = _pnr3(cuda, L, p, p_feat, compute)
    seed = 0xB00 + 17 * B + L
    m._egx_seed = lambda: seed
    feats = seeded_feats(33 + B, [(B, 16, 8192), (B, 16, 8192), (B, 8, 2048), (B, 8, 256)])
    out = m.forward_features(*[f.to(cuda) for f in feats])
    assert F_egx.last_encoder_impl() == "fused"
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    lin(out).backward()
    torch.cuda.synchronize()
    masks = dm.encoder_masks(seed, "fused", B, [16, 16, 8, 8], 128, 8, 256, L, p, 0.0, p_feat) if (p > 0 or p_feat > 0) else None
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.pnr3_forward(sd64, 8, *[f.double() for f in feats], masks=masks).unsqueeze(2)
    lin(ref).backward()
    tol_out, tol_grad = TOL[compute]
    if compute == "bf16" and L >= 6:
        tol_grad = 1.5e-1       # six bf16 layers: rounding compounds (the same depth at p = 0 in f32s: 1e-6)
    err = ((out.detach().double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
    assert err < tol_out, err
    named = dict(m.named_parameters())
    errs = {k: rel_err(named[k].grad, sd64[k].grad) for k in named if sd64[k].grad is not None}
    assert "pe" in errs and "proj1.weight" in errs
    bad = {k: v for k, v in errs.items() if not v < tol_grad}
    assert not bad, bad


def test_auto_now_picks_the_per_clip_kernels_for_the_hoi_d128_translators(egx_lib, cuda):
    """impl = auto: the PNR / OSCC translator (S = 48, h = 8) and the 3-task action-recognition translator leave the generic
    kernels; outputs agree with the forced generic path."""
    from egot2_amd import functional as F_egx
    m, _ = _pnr3(cuda, 2, 0.0, 0.0, "f32", impl="auto")
    feats = [f.to(cuda) for f in seeded_feats(9, [(3, 16, 8192), (3, 16, 8192), (3, 8, 2048), (3, 8, 256)])]
    a = m.forward_features(*feats)
    assert F_egx.last_encoder_impl() == "fused"
    b = m.set_compute("f32", "generic").forward_features(*feats)
    assert F_egx.last_encoder_impl() == "generic"
    assert (a - b).abs().max().item() < 1e-4
