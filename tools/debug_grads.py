import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egot2_amd import hhi_ttm
from oracle import translator_ref as tr
from tests.util import hhi_args, seeded_feats, seeded_state_dict, rel_err
dev = torch.device("cuda:0")
CE_W = [0.266, 0.734]
def run(impl, n_tasks, B, T, L, compute="f32"):
    cls = hhi_ttm.TaskFusionMFTransformer3Task if n_tasks == 3 else hhi_ttm.TaskFusionMFTransformer2Task
    model = cls(hhi_args(num_layers=L))
    sd = seeded_state_dict(model, seed=100 + n_tasks + B)
    model.load_state_dict(sd)
    model = model.to(dev).set_compute(compute, impl).train()
    model.pos_embed.dropout.p = 0.0
    feats = seeded_feats(7 + B, [(B, T, 256)] * n_tasks)
    target = torch.from_numpy(np.random.default_rng(B).integers(0, 2, B)).long()
    logits = model.forward_features(*[f.to(dev) for f in feats])
    loss = torch.nn.functional.cross_entropy(logits, target.to(dev), weight=torch.tensor(CE_W, device=dev))
    loss.backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats])
    tr.weighted_ce(ref, target, CE_W).backward()
    errs = {k: rel_err(p.grad, sd64[k].grad) for k, p in model.named_parameters()}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f"{impl:8s} K={n_tasks} B={B} T={T} L={L} {compute}: logit err {(logits.double().cpu()-ref.detach()).abs().max().item():.2e}  worst grads", [(k[-28:], f"{v:.1e}") for k, v in worst])
for cfg in [(3, 6, 16, 2), (3, 6, 16, 1), (3, 6, 15, 2), (3, 6, 15, 1), (3, 64, 16, 2), (3, 64, 15, 2)]:
    for impl in ("generic", "fused"):
        run(impl, *cfg)
