import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_oracle_golden import FIXTURES, build_ours, fixture_feats, load_fixture
from tests.test_gpu_golden import hip_run
from tests.util import seeded_state_dict
dev = torch.device("cuda:0")
for name in FIXTURES:
    for impl in (["auto", "generic"] if name[:3] not in ("ttm", "asd") else ["auto"]):
        c, z = load_fixture(name)
        model = build_ours(c)
        model.load_state_dict(seeded_state_dict(model, c["wseed"]))
        model = model.to(dev).set_compute("bf16", impl).train()
        if hasattr(model, "pos_embed"):
            model.pos_embed.dropout.p = 0.0
        feats = [f.to(dev) for f in fixture_feats(c)]
        outs, loss = hip_run(c, model, feats)
        loss.backward()
        worst = 0
        for k, v in outs.items():
            ref = torch.from_numpy(z[k]).double()
            worst = max(worst, (v.detach().double().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
        gw = 0
        for k, p in model.named_parameters():
            if p.grad is not None and ("gnorm/" + k) in z.files:
                n_ref = float(z["gnorm/" + k])
                gw = max(gw, abs(p.grad.double().norm().item() - n_ref) / (n_ref + 1e-12))
        print(f"{name:24s} {impl:8s} out err {worst:.4f}  worst grad-norm err {gw:.4f}")
