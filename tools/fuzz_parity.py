"""Randomised parity sweep of the d = 128 translators against the fp64 oracle (GPU box, repo root) — development aid.

    python3 tools/fuzz_parity.py [--wide] [seconds, default 300] [rng seed]          (--wide: the bf16 wide path through the HOI LTA 4-task translator; --pnr: the PNR / OSCC recipe; --hhig: EgoT2-g HHI encoder + decoder; --hoig: EgoT2-g HOI encoder)

Draws (model, tasks, B, T, layers, compute mode, dropout, deterministic, env knobs) at random, lets the library pick its implementation
(per-clip kernels with / without the cut at the FFN, sliced small batches, tiled long clips), runs forward + weighted CE + backward and
compares logits / loss / every parameter gradient with oracle/translator_ref.py under the SAME dropout masks (tests/dropmask.py), at the
tolerances of tests/test_gpu_translator.py. Prints one line per case and a summary; exit code 1 if any case is out of tolerance.
The point is the shapes nobody wrote a test for: ragged tile counts, one-clip batches, chunk edges of the tiled attention, B just
above / below the slicing thresholds.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import translator_ref as tr
from tests import dropmask as dm
from tests.util import hhi_args, rel_err, seeded_feats, seeded_state_dict
import egot2_amd.functional as _F_tuning; _F_tuning.reload_tuning_each_call = True   # the kernel-selection switches are flipped inside this process

class ReluSpy:
    """Smallest |ReLU pre-activation| / rms the oracle's forward sees. A hidden unit whose pre-activation is within fp32 rounding of zero
    (~1e-6 of the rms after a K = 128 dot product) can come out on the other side of the kink in fp32 arithmetic: its gradient contribution
    flips, and on a batch of a hundred tokens that ONE unit is 1e-2 of a weight gradient's norm (case `--pnr 300 99` #70: exact fp32 2e-6, f32s
    1.7e-2, smallest pre-activation 1.7e-6). Such cases are judged at 5e-2 and marked `kink`."""
    def __enter__(self):
        self.orig, self.ratio = torch.relu, float("inf")

        def spy(x):
            self.ratio = min(self.ratio, (x.detach().abs().min() / x.detach().pow(2).mean().sqrt().clamp(min=1e-30)).item())
            return self.orig(x)
        torch.relu = spy
        return self

    def __exit__(self, *a):
        torch.relu = self.orig


CE_W = [0.266, 0.734]
TOL = {"f32": (1e-3, 1e-2), "f32s": (1e-3, 1e-2), "bf16": (1.5e-2, 1.2e-1)}


def one_case(rng, cuda, idx):
    from egot2_amd import functional as F_egx, hhi_ttm, hhi_asd
    kind = rng.choice(["ttm3", "ttm3", "ttm2", "asd"])
    compute = rng.choice(["f32s", "f32s", "bf16", "f32"])
    L = int(rng.choice([1, 1, 2, 3]))
    p = float(rng.choice([0.0, 0.1, 0.5]))
    # sequence length classes: per-clip kernels (S <= 48), tiled (S <= 512)
    n_tasks = 2 if kind == "ttm2" else 3
    cls = rng.choice(["short", "short", "mid", "long"])
    T = int({"short": rng.integers(1, 48 // n_tasks + 1), "mid": rng.integers(48 // n_tasks + 1, 61),
             "long": rng.integers(61, 512 // n_tasks + 1)}[cls])
    S = n_tasks * T
    if compute == "f32" and S > 48:
        compute = "f32s"                      # exact fp32 MFMA beyond 48 tokens is the generic path: not what this sweep is about
    budget = 2600 if S > 200 else 6000        # tokens: keeps the fp64 oracle autograd in seconds
    B = int(rng.integers(1, max(2, min(70, budget // S)) + 1))
    det = bool(rng.random() < 0.2)
    env = {}
    if S <= 48 and rng.random() < 0.4:
        env["EGX_FFN_CUT"] = str(int(rng.integers(0, 2)))
    if S <= 48 and rng.random() < 0.3:
        env["EGX_FFN_SLICES"] = str(int(rng.choice([1, 2, 4, 8])))
    for k, v in env.items():
        os.environ[k] = v
    try:
        if kind == "asd":
            model = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(num_layers=L, dropout=p))
        else:
            model = (hhi_ttm.TaskFusionMFTransformer3Task if n_tasks == 3 else hhi_ttm.TaskFusionMFTransformer2Task)(hhi_args(num_layers=L, dropout=p))
        sd = seeded_state_dict(model, seed=1000 + idx)
        model.load_state_dict(sd)
        model = model.to(cuda).set_compute(compute).set_deterministic(det).train()
        seed = 0x5EED0000 + 7919 * idx
        p_pos = 0.1 if p > 0 else 0.0
        model.pos_embed.dropout.p = p_pos
        model._egx_seed = lambda: seed
        feats = seeded_feats(3000 + idx, [(B, T, 256)] * n_tasks)
        out = model.forward_features(*[f.to(cuda) for f in feats])
        impl = F_egx.last_encoder_impl()
        slices = F_egx.last_encoder_slices() if hasattr(F_egx, "last_encoder_slices") else 1
        sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
        masks = dm.encoder_masks(seed, impl, B, [T] * n_tasks, 128, 4, 2048, L, p, p_pos) if p > 0 else None
        if kind == "asd":
            gen = torch.Generator().manual_seed(idx)
            gout = torch.randn(out.shape, generator=gen)
            (out * gout.to(cuda)).sum().backward()
            with ReluSpy() as spy:
                ref = tr.asd_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
            (ref * gout.double()).sum().backward()
            loss_err = 0.0
        else:
            target = torch.from_numpy(np.random.default_rng(idx).integers(0, 2, B)).long()
            loss = torch.nn.functional.cross_entropy(out, target.to(cuda), weight=torch.tensor(CE_W, device=cuda))
            loss.backward()
            with ReluSpy() as spy:
                ref = tr.ttm_forward(sd64, 4, *[f.double() for f in feats], masks=masks)
            ref_loss = tr.weighted_ce(ref, target, CE_W)
            ref_loss.backward()
            loss_err = abs(loss.item() - ref_loss.item()) / max(1.0, abs(ref_loss.item()))
        torch.cuda.synchronize()
        tol_o, tol_g = TOL[compute]
        if L >= 3 and compute == "bf16":
            tol_g = 1.5e-1
        kink = compute != "bf16" and spy.ratio < 4e-6
        if kink:
            tol_g = 5e-2
        if compute == "bf16" and p >= 0.5:
            tol_o = 2.5e-2                    # kept elements carry a factor 2: bf16 rounding noise of the logits doubles
        if kind == "asd" and compute == "bf16":
            tol_o = 4e-2 if p < 0.5 else 8e-2 # per-token outputs, max norm (tests/test_gpu_tiled.py asserts the L2 error < 1e-2)
        err_o = ((out.detach().double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
        named = dict(model.named_parameters())
        errs = {k: rel_err(named[k].grad, v.grad) for k, v in sd64.items() if v.grad is not None and named[k].grad is not None}
        missing = [k for k, v in sd64.items() if v.grad is not None and v.grad.abs().max() > 0 and named[k].grad is None]
        if compute == "bf16":
            # d(head bias) = sum_clips d(logits) cancels to nearly zero on balanced predictions: its RELATIVE error in bf16 is noise / ~0
            # (f32s on the same path: <= 2e-5). Judged by the head weight's gradient instead.
            errs.pop("linear_head.1.bias", None)
        worst = max(errs.items(), key=lambda kv: kv[1]) if errs else ("-", 0.0)
        ok = err_o < tol_o and loss_err < tol_o and worst[1] < tol_g and not missing and all(np.isfinite(v) for v in errs.values())
        print(f"[{idx:4d}] {'ok  ' if ok else 'FAIL'} {kind} {compute:4s} B={B:3d} T={T:3d} S={S:3d} L={L} p={p} det={int(det)} {impl}"
              f"{'/' + str(slices) if slices and slices > 1 else ''} {env} out {err_o:.2e} loss {loss_err:.2e} grad {worst[1]:.2e} ({worst[0]})"
              f"{' kink %.1e' % spy.ratio if kink else ''}{' MISSING ' + str(missing) if missing else ''}", flush=True)
        return ok
    finally:
        for k in env:
            os.environ.pop(k, None)


def one_wide_case(rng, cuda, idx):
    """HOI LTA 4-task translator (learned positions, 8192-wide PNR / OSCC features, d >= 256) in bf16: the wide path, p = 0."""
    from types import SimpleNamespace as NS
    from egot2_amd import functional as F_egx, hoi_lta
    d, heads = [(256, 4), (256, 8), (512, 8), (768, 8), (384, 4), (512, 4), (1024, 8)][int(rng.integers(0, 7))]
    L = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.choice([rng.integers(1, 33), rng.integers(33, 121)]))          # clips per task: S = 4 n (<= 128: one-pass attention; beyond: online softmax)
    B = int(rng.integers(1, max(2, min(9, 900 // (4 * n))) + 1))
    cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=n, NUM_ACTIONS_TO_PREDICT=3),
             MODEL=NS(TRANSLATION_HEADS=heads, TRANSLATION_LAYERS=L, TRANSLATION_INPUT_FEATURES=d, TRANSLATION_DROPOUT=0.0,
                      NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
    sd = seeded_state_dict(m, 2000 + idx)
    m.load_state_dict(sd)
    m = m.to(cuda).set_compute("bf16").train()
    feats = seeded_feats(4000 + idx, [(B, n, 8192), (B, n, 8192), (B, n, d), (B, n, 2048)])
    outs = m.forward_features(*[f.to(cuda) for f in feats])
    impl = F_egx.last_encoder_impl()
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    (lin(outs[0]) + lin(outs[1])).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.lta4_forward(sd64, heads, *[f.double() for f in feats], [5, 7])
    (lin(ref[0]) + lin(ref[1])).backward()
    torch.cuda.synchronize()
    err_o = max(((o.detach().cpu().double() - r.detach()).abs().max() / max(1.0, r.abs().max().item())).item() for o, r in zip(outs, ref))
    errs = {}
    for k, q in m.named_parameters():
        r = sd64[k].grad
        if r is not None and q.grad is not None:
            errs[k] = (q.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-12)
    missing = [k for k, q in m.named_parameters() if sd64[k].grad is not None and sd64[k].grad.abs().max() > 0 and q.grad is None]
    worst = max(errs.items(), key=lambda kv: kv[1])
    tol_g = 6e-2 if L < 3 else 1.0e-1
    if B * 4 * n < 128:
        tol_g = max(tol_g, 9e-2)              # a hundred tokens: few terms per weight-gradient element, bf16 operand noise averages out less
    ok = err_o < 1e-2 and worst[1] < tol_g and not missing and all(np.isfinite(v) for v in errs.values())
    print(f"[{idx:4d}] {'ok  ' if ok else 'FAIL'} lta4 bf16 B={B:2d} n={n:3d} S={4 * n:3d} d={d} h={heads} L={L} {impl} out {err_o:.2e} grad {worst[1]:.2e} ({worst[0]})"
          f"{' MISSING ' + str(missing) if missing else ''}", flush=True)
    return ok


def one_pnr_case(rng, cuda, idx):
    """The shipped PNR / OSCC EgoT2-s recipe on the per-clip kernels: S = 16 + 16 + 8 + 8, 8 heads of 16, d_ff = 256, 1-6 layers, feature dropout on the
    projections, learned positions with their gradient, 8192-wide features; train mode under the oracle's masks."""
    from types import SimpleNamespace as NS
    from egot2_amd import functional as F_egx, hoi_pnr
    compute = rng.choice(["f32s", "f32s", "bf16", "f32"])
    L = int(rng.integers(1, 7))
    p = float(rng.choice([0.0, 0.1, 0.3]))
    p_feat = float(rng.choice([0.0, 0.2]))
    B = int(rng.choice([rng.integers(1, 12), rng.integers(12, 80), rng.integers(80, 300)]))
    det = bool(rng.random() < 0.2)
    env = {}
    if rng.random() < 0.4:
        env["EGX_FFN_CUT"] = str(int(rng.integers(0, 2)))
    if rng.random() < 0.3:
        env["EGX_FFN_SLICES"] = str(int(rng.choice([1, 2, 4, 8])))
    for k, v in env.items():
        os.environ[k] = v
    try:
        cfg = NS(DATA=NS(TASK="state_change_detection"),
                 MODEL=NS(TRANSLATION_INPUT_FEATURES=128, TRANSLATION_LAYERS=L, FEAT_DROPOUT_RATE=p_feat, TRANSFORMER_DROPOUT_RATE=p))
        m = hoi_pnr.TaskFusionMFTransformer3TaskDropout(cfg)
        m.load_state_dict(seeded_state_dict(m, 5000 + idx))
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}       # `ln` is shared with linear_head.0
        m = m.to(cuda).set_compute(compute).set_deterministic(det).train()
        seed = 0xB0000 + 131 * idx
        m._egx_seed = lambda: seed
        feats = seeded_feats(6000 + idx, [(B, 16, 8192), (B, 16, 8192), (B, 8, 2048), (B, 8, 256)])
        out = m.forward_features(*[f.to(cuda) for f in feats])
        impl, slices = F_egx.last_encoder_impl(), F_egx.last_encoder_slices()
        lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
        lin(out).backward()
        torch.cuda.synchronize()
        masks = dm.encoder_masks(seed, impl, B, [16, 16, 8, 8], 128, 8, 256, L, p, 0.0, p_feat) if (p > 0 or p_feat > 0) else None
        sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        with ReluSpy() as spy:
            ref = tr.pnr3_forward(sd64, 8, *[f.double() for f in feats], masks=masks).unsqueeze(2)
        lin(ref).backward()
        tol_o, tol_g = {"f32": (1e-3, 1e-2), "f32s": (1e-3, 1e-2), "bf16": (1.5e-2, 8e-2)}[compute]
        if compute == "bf16" and L >= 4:
            tol_g = 1.5e-1
        kink = compute != "bf16" and spy.ratio < 4e-6
        if kink:
            tol_g = 5e-2
        err_o = ((out.detach().double().cpu() - ref.detach()).abs() / ref.detach().abs().clamp(min=1.0)).max().item()
        named = dict(m.named_parameters())
        errs = {k: rel_err(named[k].grad, sd64[k].grad) for k in named if sd64[k].grad is not None and named[k].grad is not None}
        missing = [k for k in named if sd64[k].grad is not None and sd64[k].grad.abs().max() > 0 and named[k].grad is None]
        worst = max(errs.items(), key=lambda kv: kv[1])
        ok = err_o < tol_o and worst[1] < tol_g and not missing and all(np.isfinite(v) for v in errs.values())
        print(f"[{idx:4d}] {'ok  ' if ok else 'FAIL'} pnr3 {compute:4s} B={B:3d} L={L} p={p} p_feat={p_feat} det={int(det)} {impl}{'/' + str(slices) if slices > 1 else ''} {env} "
              f"out {err_o:.2e} grad {worst[1]:.2e} ({worst[0]}){' kink %.1e' % spy.ratio if kink else ''}{' MISSING ' + str(missing) if missing else ''}", flush=True)
        return ok
    finally:
        for k in env:
            os.environ.pop(k, None)


def one_hhig_case(rng, cuda, idx):
    """EgoT2-g HHI (d = 256, task prompt): encode_features() on the wide path + decode() (fused decoder up to 64 memory tokens, composed kernels beyond),
    bf16 against the fp64 oracle: memory, vocabulary logits, every parameter gradient. p = 0."""
    from types import SimpleNamespace as NS
    from egot2_amd import functional as F_egx, hhi_multitask
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    task = str(rng.choice(["ttm", "lam", "asd"]))
    L = int(rng.choice([1, 2, 3]))
    heads = int(rng.choice([4, 4, 8]))
    T = int(rng.choice([rng.integers(1, 22), rng.integers(22, 151)]))
    B = int(rng.integers(1, max(2, min(12, 1400 // (3 * T))) + 1))
    args = NS(hidden_dim=256, num_heads=heads, num_layers=L, dropout=0.0, lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None)
    m = hhi_multitask.TaskTranslationPromptTransformer(args, vocab)
    sd = seeded_state_dict(m, 7000 + idx)
    m.load_state_dict(sd)
    m.pos_embed.dropout.p = 0.0
    m = m.to(cuda).set_compute("bf16").train()
    feats = seeded_feats(8000 + idx, [(B, T, 256)] * 3)
    mem = m.encode_features(task, *[f.to(cuda) for f in feats])
    impl = F_egx.last_encoder_impl()
    nb = mem.shape[1]
    y = torch.stack([torch.full((nb,), vocab[task]), torch.randint(5, 7, (nb,), generator=torch.Generator().manual_seed(idx))], dim=1)
    logits = m.decode(y.to(cuda), mem)
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    lin(logits).backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    rmem = tr.hhi_g_encode(sd64, heads, task, *[f.double() for f in feats])
    rlog = tr.g_decode(sd64, heads, y, rmem)
    lin(rlog).backward()
    e_mem = (mem.detach().cpu().double() - rmem.detach()).abs().max().item() / max(1.0, rmem.detach().abs().max().item())
    e_log = (logits.detach().cpu().double() - rlog.detach()).abs().max().item() / max(1.0, rlog.detach().abs().max().item())
    named = dict(m.named_parameters())
    errs = {k: ((named[k].grad.cpu().double() - v.grad).norm() / (v.grad.norm() + 1e-12)).item() for k, v in sd64.items()
            if v.grad is not None and k in named and named[k].grad is not None and v.grad.norm() > 0}
    missing = [k for k, v in sd64.items() if v.grad is not None and k in named and v.grad.norm() > 0 and named[k].grad is None]
    worst = max(errs.items(), key=lambda kv: kv[1])
    # a dozen target rows: a decoder bias gradient is a sum over 2 .. 24 rows, and the ~0.4 % of hidden units whose pre-activation bf16 noise carries across
    # the ReLU kink are not averaged out (relative error ~ sqrt(fraction flipped)): 1.2e-1, 2e-1 at three layers; the suite's 8e-2 holds from ~50 rows
    tol_g = (8e-2 if L < 3 else 1.2e-1) if 2 * nb >= 64 else (1.2e-1 if L < 3 else 2e-1)
    ok = e_mem < 4e-2 and e_log < 4e-2 and worst[1] < tol_g and not missing and len(errs) > 20 and all(np.isfinite(v) for v in errs.values())
    print(f"[{idx:4d}] {'ok  ' if ok else 'FAIL'} hhig bf16 {task} B={B:2d} T={T:3d} rows={nb} h={heads} L={L} {impl} mem {e_mem:.2e} logits {e_log:.2e} grad {worst[1]:.2e} ({worst[0]})"
          f"{' MISSING ' + str(missing) if missing else ''}", flush=True)
    return ok


def one_hoig_case(rng, cuda, idx):
    """EgoT2-g HOI encoder (HOI/models/multitask/video_model_builder.py): both prompt layouts of encode() — 'pnr' (16 + 16 + 8 + 8 tokens, SlowFast pathways
    projected separately) and 'lta_verb' (per-clip PNR / OSCC frames + action + LTA features, 4 n tokens) — in bf16 (wide path) or fp32 (generic kernels)."""
    from egot2_amd import functional as F_egx
    from tests.test_oracle_golden import build_ours, fixture_feats
    compute = str(rng.choice(["bf16", "bf16", "f32"]))
    d, h = [(512, 8), (256, 4), (256, 8), (512, 4)][int(rng.integers(0, 4))]
    c = dict(kind="hoig", B=int(rng.integers(1, 9)), n=int(rng.integers(1, 13)), L=int(rng.choice([1, 2, 3])), d=d, h=h, wseed=9000 + idx, fseed=9500 + idx)
    model = build_ours(c)
    sd = seeded_state_dict(model, c["wseed"])
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute(compute).train()
    model.pos_embed.dropout.p = 0.0
    feats = fixture_feats(c)
    fd = [f.to(cuda) for f in feats]
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    outs = [model.encode_features("pnr", *fd[:4])]
    impl = F_egx.last_encoder_impl()
    outs.append(model.encode_features("lta_verb", *fd[4:]))
    (lin(outs[0]) + lin(outs[1])).backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    f64 = [f.double() for f in feats]
    refs = [tr.hoi_g_encode(sd64, h, "pnr", *f64[:4]), tr.hoi_g_encode(sd64, h, "lta_verb", *f64[4:])]
    (lin(refs[0]) + lin(refs[1])).backward()
    err_o = max((o.detach().cpu().double() - r.detach()).abs().max().item() / max(1.0, r.abs().max().item()) for o, r in zip(outs, refs))
    errs = {}
    for k, q in model.named_parameters():
        r = sd64[k].grad if k in sd64 else None
        if r is not None and q.grad is not None:
            errs[k] = (q.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-12)
    worst = max(errs.items(), key=lambda kv: kv[1])
    tol_o, tol_g = (1e-3, 1e-2) if compute == "f32" else (1e-2, 6e-2 if c["L"] < 3 else 1e-1)
    if compute == "bf16" and c["B"] * 48 < 150:
        tol_g = max(tol_g, 9e-2)
    ok = err_o < tol_o and worst[1] < tol_g and len(errs) >= 25 and all(np.isfinite(v) for v in errs.values())
    print(f"[{idx:4d}] {'ok  ' if ok else 'FAIL'} hoig {compute:4s} B={c['B']} n={c['n']:2d} d={d} h={h} L={c['L']} {impl} out {err_o:.2e} grad {worst[1]:.2e} ({worst[0]}) [{len(errs)} grads]", flush=True)
    return ok


def main():
    wide = "--wide" in sys.argv
    pnr = "--pnr" in sys.argv
    hhig = "--hhig" in sys.argv
    hoig = "--hoig" in sys.argv
    argv = [a for a in sys.argv[1:] if a not in ("--wide", "--pnr", "--hhig", "--hoig")]
    secs = float(argv[0]) if len(argv) > 0 else 300.0
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 12345)
    cuda = torch.device("cuda", 0)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        try:
            ok = (one_wide_case if wide else one_pnr_case if pnr else one_hhig_case if hhig else one_hoig_case if hoig else one_case)(rng, cuda, n)
        except Exception as e:      # noqa: BLE001  (an EgxError for an unsupported pairing is a finding too: print and go on)
            print(f"[{n:4d}] EXC  {type(e).__name__}: {str(e)[:300]}", flush=True)
            ok = False
        bad += 0 if ok else 1
        n += 1
    print(f"{n} cases, {bad} out of tolerance / raised, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
