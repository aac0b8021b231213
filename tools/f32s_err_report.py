"""Error of the three compute modes of the fused kernels against the fp64 oracle (development report, GPU box).
usage: python tools/f32s_err_report.py"""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import translator_ref as tr
from tests.util import hhi_args, rel_err, seeded_feats, seeded_state_dict
from egot2_amd import hhi_ttm
dev = torch.device("cuda:0"); CE_W = [0.266, 0.734]
for (n_tasks, B, T, L) in [(3, 8, 15, 1), (2, 32, 15, 1), (3, 256, 15, 1), (3, 6, 16, 2), (3, 1, 16, 4), (2, 3, 7, 3), (3, 257, 3, 1)]:
    sd = None
    for comp in ("f32", "f32s", "bf16"):
        cls = hhi_ttm.TaskFusionMFTransformer3Task if n_tasks == 3 else hhi_ttm.TaskFusionMFTransformer2Task
        model = cls(hhi_args(num_layers=L)); sd = seeded_state_dict(model, seed=100 + n_tasks + B); model.load_state_dict(sd)
        model = model.to(dev).set_compute(comp, "fused").train(); model.pos_embed.dropout.p = 0.0
        feats = seeded_feats(7 + B, [(B, T, 256)] * n_tasks)
        target = torch.from_numpy(np.random.default_rng(B).integers(0, 2, B)).long()
        logits = model.forward_features(*[f.to(dev) for f in feats])
        torch.nn.functional.cross_entropy(logits, target.to(dev), weight=torch.tensor(CE_W, device=dev)).backward()
        sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
        ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats]); tr.weighted_ce(ref, target, CE_W).backward()
        named = dict(model.named_parameters())
        errs = {k: rel_err(named[k].grad, v.grad) for k, v in sd64.items() if v.grad is not None}
        w = max(errs, key=errs.get)
        le = ((logits.double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
        print(f"{comp:5s} {(n_tasks, B, T, L)} logit err {le:.2e}  worst grad {w} {errs[w]:.2e}  median {float(np.median(list(errs.values()))):.2e}", flush=True)
