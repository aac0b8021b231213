"""-m gpu: the HIP translator classes against the golden fixtures generated from the REAL reference (outputs and
gradient digests), through the C ABI of libegot2x.so. fp32: outputs within 1e-3, gradients within 1e-2 relative
(BASELINE.json:north_star); bf16: 1e-2 / 6e-2 (measured worst gradient-norm error 5.6e-2 on the 4-token
EgoT2-g action fixture, <= 1.1e-2 on most; tools/bf16_err_report.py)."""
import numpy as np
import pytest
import torch

from tests.test_oracle_golden import (CE_W, FIXTURES, build_ours, check_against_fixture, fixture_feats, g_targets,
                                      load_fixture)
from tests.util import seeded_state_dict

pytestmark = pytest.mark.gpu


def hip_run(c, model, feats):
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device).view_as(t)).sum()  # noqa: E731
    if c["kind"] == "ttm":
        out = model.forward_features(*feats)
        target = torch.from_numpy(np.random.default_rng(c["fseed"]).integers(0, 2, c["B"])).long().to(out.device)
        return {"out": out}, torch.nn.functional.cross_entropy(out, target, weight=torch.tensor(CE_W, device=out.device))
    if c["kind"] == "asd":
        out = model.forward_features(*feats)
        return {"out": out}, lin(out)
    if c["kind"] == "hhig":
        outs, loss = {}, 0
        for task in ("lam", "ttm", "asd"):
            lamf = feats[1][:, :7].contiguous() if task == "lam" else feats[1]
            enc = model.encode_features(task, lamf, feats[0], feats[2])
            outs[f"out_{task}"] = enc
            dec = model.decode(g_targets(c, task, enc.shape[1], 7).to(enc.device), enc)      # HIP decoder (row F1)
            outs[f"dec_{task}"] = dec
            loss = loss + lin(enc) + lin(dec)
        return outs, loss
    if c["kind"] == "lta4":
        o = model.forward_features(*feats)
        return {"out_verb": o[0], "out_noun": o[1]}, lin(o[0]) + lin(o[1])
    if c["kind"] == "hoig":
        outs = {"out_pnr": model.encode_features("pnr", *feats[:4]), "out_lta": model.encode_features("lta_verb", *feats[4:])}
        for task, sy in (("pnr", 2), ("lta", 4)):
            enc = outs[f"out_{task}"]
            outs[f"dec_{task}"] = model.decode(g_targets(c, task, enc.shape[1], 12, sy).to(enc.device), enc)
        return outs, sum(lin(v) for v in outs.values())
    if c["kind"] == "pnr3":
        out = model.forward_features(*feats)
        return {"out": out}, lin(out)
    if c["kind"] in ("ar3", "ar2", "lta2"):
        o = model.forward_features(*feats)
        return {"out_verb": o[0], "out_noun": o[1]}, lin(o[0]) + lin(o[1])
    if c["kind"] == "pnrvit":
        out = model.forward_features(*feats)
        return {"out": out}, lin(out)
    if c["kind"] == "hoig2":
        enc = model.encode_features(*feats)
        outs = {"out": enc, "dec": model.decode(g_targets(c, "pnr", enc.shape[1], 12, 2).to(enc.device), enc)}
        return outs, sum(lin(v) for v in outs.values())
    if c["kind"] == "hoiga":
        outs = {"out_lta": model.encode_features("lta_verb", feats[0], feats[1]),
                "out_action": model.encode_features("action_verb", feats[0][:, 0:1].contiguous())}
        for task, sy in (("lta", 3), ("action", 2)):
            enc = outs[f"out_{task}"]
            outs[f"dec_{task}"] = model.decode(g_targets(c, task, enc.shape[1], 12, sy).to(enc.device), enc)
        return outs, sum(lin(v) for v in outs.values())
    raise KeyError(c["kind"])


# "f32s" (fp32 operands split into three bf16 parts, six bf16 MFMA products per K-block) is held to the fp32 tolerances: it
# is implemented by the fused d = 128 kernels and means "f32" everywhere else, so only the fixtures those kernels run are repeated.
@pytest.mark.parametrize("compute,tol_out,tol_grad", [("f32", 1e-3, 1e-2), ("bf16", 1e-2, 6e-2), ("f32s", 1e-3, 1e-2)])
@pytest.mark.parametrize("name", FIXTURES)
def test_hip_matches_reference_fixture(egx_lib, cuda, name, compute, tol_out, tol_grad):
    c, z = load_fixture(name)
    if compute == "f32s" and c["kind"] not in ("ttm", "asd", "pnr3", "ar3", "ar2"):
        pytest.skip("f32s differs from f32 only on the per-clip d = 128 kernels (HHI TTM / ASD, HOI PNR / OSCC and action recognition)")
    model = build_ours(c)
    model.load_state_dict(seeded_state_dict(model, c["wseed"]))
    model = model.to(cuda).set_compute(compute).train()
    if hasattr(model, "pos_embed"):
        model.pos_embed.dropout.p = 0.0
    feats = [f.to(cuda) for f in fixture_feats(c)]
    outs, loss = hip_run(c, model, feats)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    # bf16: the fused decoder (row F1, egx_decoder_*) stacks L more bf16 layers on the bf16 encoder memory, whose own error is
    # already 0.5-0.7 % of the output scale: its logits are held to 1.5e-2 (measured 0.7-1.2 %)
    check_against_fixture(z, outs, loss, grads, tol_out, tol_grad, tol_dec=1.5e-2 if compute == "bf16" else None)
