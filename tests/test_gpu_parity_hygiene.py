"""-m gpu: parity cases VERDICT r5 item 7 asked for — the HHI EgoT2-g encoder + decoder at the bench batch under a checker, a fixed-seed slice of
the randomised sweep (tools/fuzz_parity.py) as part of the suite, and the bench shape in bf16 in the RECIPE mode (p = 0.5) under the oracle's masks."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests.util import hhi_args, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c5_hhi_encoder_and_decoder_at_bench_batch_against_the_oracle_on_sampled_clips(egx_lib, cuda):
    """BASELINE.json configs[4], HHI flavour (EgoT2-g: d = 256, 4 heads, 3 layers, S = 45, two decoder tokens; task_prompt_model.py:230-269) at the
    batch `bench.py --config c5hhi` times, B = 256, in bf16. Clips are independent units: the loss reads the vocabulary logits of three sampled
    clips, the fp64 oracle runs on those three clips, and the memory rows, the logits and EVERY parameter gradient of the 256-clip launch (its
    grids, token splits, grouped weight gradients and slab reductions are those of B = 256) must match it."""
    from egot2_amd import hhi_multitask
    from egot2_amd.synth import HHI_G_VOCAB
    from oracle import translator_ref as tr
    B, T, pick = 256, 15, [1, 130, 255]
    m = hhi_multitask.TaskTranslationPromptTransformer(hhi_args(hidden_dim=256, num_heads=4, num_layers=3, dropout=0.0), HHI_G_VOCAB)
    sd = seeded_state_dict(m, 77)
    m.load_state_dict(sd)
    m.pos_embed.dropout.p = 0.0
    m = m.to(cuda).set_compute("bf16").train()
    feats = seeded_feats(903, [(B, T, 256)] * 3)
    g = torch.Generator().manual_seed(4)
    y = torch.stack([torch.full((B,), HHI_G_VOCAB["ttm"]), torch.randint(5, 7, (B,), generator=g)], dim=1)
    mem = m.encode_features("ttm", *[f.to(cuda) for f in feats])              # (S, B, 256)
    logits = m.decode(y.to(cuda), mem)                                          # (sy, B, V)
    assert mem.shape == (3 * T, B, 256) and logits.shape[1] == B
    idx = torch.tensor(pick, device=cuda)
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    lin(logits[:, idx].contiguous()).backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    rmem = tr.hhi_g_encode(sd64, 4, "ttm", *[f[pick].double() for f in feats])
    rlog = tr.g_decode(sd64, 4, y[pick], rmem)
    lin(rlog).backward()
    assert (mem[:, idx].detach().cpu().double() - rmem.detach()).abs().max().item() < 1e-2 * max(1.0, rmem.detach().abs().max().item())
    assert (logits[:, idx].detach().cpu().double() - rlog.detach()).abs().max().item() < 1.5e-2 * max(1.0, rlog.detach().abs().max().item())
    named = dict(m.named_parameters())
    errs = {k: ((named[k].grad.cpu().double() - v.grad).norm() / (v.grad.norm() + 1e-12)).item() for k, v in sd64.items()
            if v.grad is not None and k in named and named[k].grad is not None and v.grad.norm() > 0}
    # three bf16 encoder layers + three bf16 decoder layers deep, summed over only 3 x 45 token rows: the documented bf16 bound of the deep stacks
    # (1.2e-1 / 1.5e-1 for three / six layers; measured here: 0.12 on the token-preparation biases — six bf16 layers from the loss —, 0.007 on the vocabulary head)
    bad = {k: e for k, e in errs.items() if not e < 1.5e-1}
    assert len(errs) > 40 and not bad, bad
    assert errs["fc.weight"] < 2e-2      # (one bf16 GEMM deep: the per-layer error that the stack accumulates)


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fixed_seed_slice_of_the_randomised_parity_sweep(egx_lib, cuda, capsys):
    """40 cases of tools/fuzz_parity.py with a fixed generator seed (the d = 128 translators: TTM 2 / 3-task and ASD, B, T, layers, compute mode,
    dropout, deterministic mode and the kernel-selection switches drawn at random; per-clip kernels with / without the cut, sliced small batches,
    tiled long clips), each against the fp64 oracle under the same masks at the sweep's tolerances. Until round 5 the sweep was a builder-run
    tool; this slice runs with `-m gpu`."""
    fz = _fuzz()
    rng = np.random.default_rng(20260603)
    results = [fz.one_case(rng, cuda, 5000 + i) for i in range(40)]
    out = capsys.readouterr().out
    assert all(results), "\n".join(line for line in out.splitlines() if "FAIL" in line or "EXC" in line)
    kinds = {(line.split()[2], line.split()[3]) for line in out.splitlines() if line.startswith("[")}
    assert len(kinds) >= 6, kinds       # the slice covers several (model, compute) pairings


def test_bench_shape_bf16_in_the_recipe_mode_under_the_oracles_masks(egx_lib, cuda):
    """BASELINE.json configs[1] at its own size (B = 256, S = 45) in bf16 with the README recipe's dropout (p = 0.5, + 0.1 on the positional
    encoding), against the fp64 oracle fed the SAME masks. What bf16 delivers in this mode is a measured bound, not the p = 0 figure: kept
    elements carry a factor 2, so the bf16 rounding noise of the logits doubles. The north_star's 1e-2 is asserted where it holds (the
    batch RMS and 98 % of the logits) and the measured maximum (<= 2.5e-2, profiles/r05_fuzz_parity.txt: 1.6 - 2.0e-2) is asserted as
    such; DESIGN.md section 3 and INTEGRATION.md state this as the limitation of bf16 in the recipe mode."""
    from egot2_amd import functional as F_egx, hhi_ttm
    from oracle import translator_ref as tr
    from tests import dropmask as dm
    B, T, p = 256, 15, 0.5
    model = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(num_layers=1, dropout=p))
    sd = seeded_state_dict(model, seed=4242)
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute("bf16").train()
    seed = 0x5EED1234
    model._egx_seed = lambda: seed
    feats = seeded_feats(4243, [(B, T, 256)] * 3)
    out = model.forward_features(*[f.to(cuda) for f in feats])
    impl = F_egx.last_encoder_impl()
    masks = dm.encoder_masks(seed, impl, B, [T] * 3, 128, 4, 2048, 1, p, 0.1)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
    err = (out.detach().double().cpu() - ref).abs()
    scale = ref.abs().clamp(min=1.0)
    rel = err / scale
    assert (rel.pow(2).mean().sqrt().item()) < 1e-2
    assert (rel < 1e-2).double().mean().item() >= 0.98
    assert rel.max().item() < 2.5e-2
