"""-m gpu: size extremes. Long sequences (T = 150 per task as in validation, S = 450) on the shape-generic kernels against
the oracle; very large clip batches on the fused kernels (intermediates above 2 GiB) against the same clips run in
small batches."""
import pytest
import torch

from oracle import translator_ref as tr
from tests.util import hhi_args, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]


@pytest.mark.parametrize("compute,impl", [("f32", "generic"), ("f32s", "tiled")])
def test_long_validation_sequence_matches_oracle(egx_lib, cuda, compute, impl):
    """batch_size = 1, T = 150: S = 450 tokens, far beyond the per-clip kernels' 48: the shape-generic
    kernels in exact fp32, and the tiled MFMA path (fp32-grade split arithmetic) held to the same fp32 tolerances."""
    from egot2_amd import functional as F_egx, hhi_ttm
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(num_layers=1))
    sd = seeded_state_dict(m, 21)
    m.load_state_dict(sd)
    m = m.to(cuda).set_compute(compute).train()
    m.pos_embed.dropout.p = 0.0
    feats = seeded_feats(22, [(1, 150, 256)] * 3)
    target = torch.tensor([1])
    logits = m.forward_features(*[f.to(cuda) for f in feats])
    assert F_egx.last_encoder_impl() == impl
    torch.nn.functional.cross_entropy(logits, target.to(cuda), weight=torch.tensor(CE_W, device=cuda)).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats])
    tr.weighted_ce(ref, target, CE_W).backward()
    assert (logits.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-3
    for k, p in m.named_parameters():
        g, r = p.grad.detach().cpu().double(), sd64[k].grad
        assert (g - r).norm().item() <= 1e-2 * r.norm().item() + 1e-6, k


@pytest.mark.parametrize("B,compute", [(6144, "f32"), (6144, "f32s"), (6144, "bf16"), (6143, "bf16"), (6143, "f32s")])
def test_huge_batch_on_fused_kernels(egx_lib, cuda, B, compute):
    """B = 6144 clips: H tiles 2.4 GB + dH tiles 2.4 GB (size_t arithmetic everywhere). Logits of the big batch must
    equal the logits of the same clips in batches of 256 (clips are independent), gradients must equal the sum."""
    from egot2_amd import hhi_ttm
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(num_layers=1))
    m.load_state_dict(seeded_state_dict(m, 23))
    m = m.to(cuda).set_compute(compute).train()
    m.pos_embed.dropout.p = 0.0
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(B, 15, 256, generator=g).to(cuda) for _ in range(3)]
    w = torch.randn(B, 2, generator=g).to(cuda)
    big = m.forward_features(*feats)
    (big * w).sum().backward()
    big_grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    outs = []
    for i in range(0, B, 768):
        o = m.forward_features(*[f[i:i + 768] for f in feats])
        (o * w[i:i + 768]).sum().backward()           # accumulates into .grad
        outs.append(o.detach())
    small = torch.cat(outs)
    assert torch.isfinite(big).all()
    assert (big.detach() - small).abs().max().item() < 1e-5      # clips are independent: same arithmetic per clip in every mode
    for k, p in m.named_parameters():
        a, b = big_grads[k], p.grad
        assert (a - b).norm().item() <= 2e-3 * b.norm().item() + 1e-4, k


def test_c4_real_dimensions_match_oracle(egx_lib, cuda):
    """BASELINE.json configs[3] at its real sizes (n = 32 clips per task -> S = 128, d = 768, 8 heads of 96, 4 layers,
    8192-wide PNR/OSCC features), B = 2: the shape-generic kernels (head dim 96, chunked attention) against the oracle."""
    from types import SimpleNamespace as NS
    from egot2_amd import hoi_lta
    cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=32, NUM_ACTIONS_TO_PREDICT=3),
             MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=4, TRANSLATION_INPUT_FEATURES=768, TRANSLATION_DROPOUT=0.0,
                      NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
    sd = seeded_state_dict(m, 33)
    m.load_state_dict(sd)
    m = m.to(cuda).train()
    B = 2
    feats = seeded_feats(34, [(B, 32, 8192), (B, 32, 8192), (B, 32, 768), (B, 32, 2048)])
    outs = m.forward_features(*[f.to(cuda) for f in feats])
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    (lin(outs[0]) + lin(outs[1])).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.lta4_forward(sd64, 8, *[f.double() for f in feats], [5, 7])
    (lin(ref[0]) + lin(ref[1])).backward()
    for o, r in zip(outs, ref):
        assert (o.detach().cpu().double() - r.detach()).abs().max().item() < 1e-3 * max(1.0, r.abs().max().item())
    worst = 0.0
    for k, p in m.named_parameters():
        r = sd64[k].grad
        if r is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            continue
        err = (p.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-9)
        worst = max(worst, err)
        assert err < 1e-2, f"{k}: {err}"
