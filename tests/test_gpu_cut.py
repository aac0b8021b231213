"""-m gpu: cut mode of the per-clip kernels (round 5, egot2_amd/csrc/ffn_cut.hip): the clip kernels are cut at the FFN, whose loops run as
launches of their own with eight waves per clip (two per SIMD). Compared with the one-launch kernels of the same library
(EGX_FFN_CUT=0) under the same masks, and with the fp64 oracle. One workgroup per clip is forced (EGX_FFN_SLICES=1) so that small
batches take the path a 256-clip batch takes by default."""
import os
from contextlib import contextmanager

import pytest
import torch

from oracle import translator_ref as tr
from tests import dropmask as dm
from tests.util import hhi_args, rel_err, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]


@contextmanager
def _env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(cuda, compute, B, T, L, p, cut, seed=0xC07, d_ff=None):
    from egot2_amd import functional as F_egx, hhi_ttm
    with _env(EGX_FFN_SLICES=1, EGX_FFN_CUT=int(cut)):
        m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=p, num_layers=L))
        sd = seeded_state_dict(m, 31)
        m.load_state_dict(sd)
        m.pos_embed.dropout.p = 0.1 if p > 0 else 0.0
        m = m.to(cuda).set_compute(compute, "fused").train()
        m._egx_seed = lambda: seed
        feats = seeded_feats(32 + B, [(B, T, 256)] * 3)
        target = torch.arange(B, device=cuda) % 2
        egx = __import__("egot2_amd._lib", fromlist=["load"]).load()
        egx.egx_launch_count(1)
        logits = m.forward_features(*[f.to(cuda) for f in feats])
        n_fwd = egx.egx_launch_count(1)
        loss = torch.nn.functional.cross_entropy(logits, target, weight=torch.tensor(CE_W, device=cuda))
        loss.backward()
        torch.cuda.synchronize()
        assert F_egx.last_encoder_impl() == "fused"
        return (logits.detach().double().cpu(), {k: q.grad.double().cpu() for k, q in m.named_parameters() if q.grad is not None}, sd, feats,
                target.cpu(), n_fwd)


@pytest.mark.parametrize("compute", ["f32", "f32s", "bf16"])
@pytest.mark.parametrize("B,T,L,p", [(7, 15, 1, 0.5), (32, 15, 1, 0.0), (5, 16, 2, 0.5), (33, 11, 3, 0.3), (1, 7, 1, 0.5), (3, 1, 2, 0.5)])
def test_cut_launches_equal_the_one_launch_kernels(egx_lib, cuda, compute, B, T, L, p):
    """Same library, same masks (the dropout keys do not depend on the cut): the only arithmetic difference is the order in which the
    FFN hidden blocks are summed (eight waves instead of four), so the fp32-grade modes agree to rounding and bf16 to a few bf16 ulps."""
    lo, go, *_, n_one = _run(cuda, compute, B, T, L, p, cut=False)
    lc, gc, *_, n_cut = _run(cuda, compute, B, T, L, p, cut=True)
    assert n_cut == n_one + 2 * L - 1, (n_one, n_cut)        # weight packing + two launches per layer instead of one launch for all
    tol_l, tol_g = (1e-5, 1e-3 if L > 1 else 1e-4) if compute != "bf16" else (2e-2, 5e-2)
    assert (lc - lo).abs().max().item() < tol_l * max(1.0, lo.abs().max().item())
    assert set(gc) == set(go)
    bad = {k: rel_err(gc[k], go[k]) for k in go if not rel_err(gc[k], go[k]) < tol_g}
    assert not bad, bad


@pytest.mark.parametrize("compute,tol_l,tol_g", [("f32", 1e-3, 1e-2), ("f32s", 1e-3, 1e-2), ("bf16", 1e-2, 8e-2)])
@pytest.mark.parametrize("B,L", [(256, 1), (9, 2)])
def test_cut_launches_match_the_oracle_under_the_same_masks(egx_lib, cuda, compute, tol_l, tol_g, B, L):
    """Train mode (p = 0.5, positional dropout 0.1) against the fp64 oracle fed the masks of the counter-based generator: the bench
    workload (C2 at B = 256: what `bench.py` times) and a two-layer stack (the layer hand-over through xin / dxin)."""
    T, p, seed = 15, 0.5, 0xC07
    logits, grads, sd, feats, target, _ = _run(cuda, compute, B, T, L, p, cut=True, seed=seed)
    masks = dm.encoder_masks(seed, "fused", B, [T] * 3, 128, 4, 2048, L, p, 0.1)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
    torch.nn.functional.cross_entropy(ref, target, weight=torch.tensor(CE_W, dtype=torch.float64)).backward()
    assert (logits - ref.detach()).abs().max().item() < tol_l * max(1.0, ref.detach().abs().max().item())
    tg = tol_g * (1.5 if (compute == "bf16" and L > 1) else 1.0)
    bad = {k: rel_err(grads[k], v.grad) for k, v in sd64.items() if v.grad is not None and k in grads and not rel_err(grads[k], v.grad) < tg}
    assert not bad, bad


@pytest.mark.parametrize("compute", ["f32", "f32s", "bf16"])
def test_cut_asd_translator_tokens_out(egx_lib, cuda, compute):
    """The ASD translator (first-T token slice leaves the kernel, no pooled head; two layers) cut vs one launch."""
    from egot2_amd import hhi_asd
    res = {}
    for cut in (0, 1):
        with _env(EGX_FFN_SLICES=1, EGX_FFN_CUT=cut):
            m = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(dropout=0.3, num_layers=2))
            m.load_state_dict(seeded_state_dict(m, 5))
            m = m.to(cuda).set_compute(compute, "fused").train()
            m._egx_seed = lambda: 77
            feats = [f.to(cuda) for f in seeded_feats(6, [(20, 15, 256)] * 3)]
            out = m.forward_features(*feats)
            out.square().sum().backward()
            torch.cuda.synchronize()
            res[cut] = (out.detach().double().cpu(), {k: q.grad.double().cpu() for k, q in m.named_parameters() if q.grad is not None})
    tol_l, tol_g = (1e-5, 1e-3) if compute != "bf16" else (2e-2, 5e-2)
    assert (res[1][0] - res[0][0]).abs().max().item() < tol_l * max(1.0, res[0][0].abs().max().item())
    bad = {k: rel_err(res[1][1][k], v) for k, v in res[0][1].items() if not rel_err(res[1][1][k], v) < tol_g}
    assert not bad, bad


@pytest.mark.parametrize("compute", ["f32s", "bf16"])
def test_cut_pnr_recipe_eight_heads_six_layers(egx_lib, cuda, compute):
    """The shipped PNR / OSCC recipe (8 heads of 16, six layers, d_ff = 256 = ONE hidden block per wave, feature dropout, learned
    positions; HOI/configs/pnr/ts_pnr.yaml:28-34) through the cut launches vs the one-launch kernels."""
    from types import SimpleNamespace as NS
    from egot2_amd import hoi_pnr
    res = {}
    for cut in (0, 1):
        with _env(EGX_FFN_SLICES=1, EGX_FFN_CUT=cut):
            cfg = NS(DATA=NS(TASK="state_change_detection"),
                     MODEL=NS(TRANSLATION_INPUT_FEATURES=128, TRANSLATION_LAYERS=6, FEAT_DROPOUT_RATE=0.2, TRANSFORMER_DROPOUT_RATE=0.1))
            m = hoi_pnr.TaskFusionMFTransformer3TaskDropout(cfg)
            m.load_state_dict(seeded_state_dict(m, 71))
            m = m.to(cuda).set_compute(compute).train()
            m._egx_seed = lambda: 99
            feats = [f.to(cuda) for f in seeded_feats(72, [(6, 16, 8192), (6, 16, 8192), (6, 8, 2048), (6, 8, 256)])]
            out = m.forward_features(*feats)
            out.square().sum().backward()
            torch.cuda.synchronize()
            res[cut] = (out.detach().double().cpu(), {k: q.grad.double().cpu() for k, q in m.named_parameters() if q.grad is not None})
    tol_l, tol_g = (1e-4, 2e-3) if compute != "bf16" else (3e-2, 1e-1)
    assert (res[1][0] - res[0][0]).abs().max().item() < tol_l * max(1.0, res[0][0].abs().max().item())
    assert set(res[1][1]) == set(res[0][1])
    bad = {k: rel_err(res[1][1][k], v) for k, v in res[0][1].items() if not rel_err(res[1][1][k], v) < tol_g}
    assert not bad, bad


def test_cut_mode_is_deterministic(egx_lib, cuda):
    """Fixed-order sums everywhere in the cut launches: with egx_config.deterministic the gradients repeat bit for bit."""
    from egot2_amd import hhi_ttm
    outs = []
    with _env(EGX_FFN_SLICES=1, EGX_FFN_CUT=1):
        for _ in range(3):
            m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5, num_layers=2))
            m.load_state_dict(seeded_state_dict(m, 3))
            m = m.to(cuda).set_compute("f32s", "fused").set_deterministic(True).train()
            m._egx_seed = lambda: 4242
            feats = [f.to(cuda) for f in seeded_feats(4, [(40, 15, 256)] * 3)]
            logits = m.forward_features(*feats)
            logits.square().sum().backward()
            torch.cuda.synchronize()
            outs.append((logits.detach().clone(), {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None}))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0])
        assert all(torch.equal(o[1][k], v) for k, v in outs[0][1].items())

