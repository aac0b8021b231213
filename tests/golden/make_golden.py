#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by running the REAL reference translator classes
(/root/reference, imported through oracle/ref_harness.py with import-time stubs only) on seeded weights and
features. Only runnable where /root/reference exists; the fixtures (inputs are re-derived from seeds, outputs and
gradient digests are stored) travel with the repo.

    python tests/golden/make_golden.py            # regenerates every *.npz here

Each fixture stores: the config (JSON), the weight/feature seeds, the reference outputs, a scalar loss, and for every
trainable parameter the gradient's L2 norm, sum and first 64 entries (full gradient when it has <= 4096 entries);
ttm3_B4_T15_L1 (the bench configuration at B = 4) also the WHOLE gradients of linear1 / linear2 / in_proj as scaled fp16.
Weights: tests/util.seeded_state_dict(model, seed); features: tests/util.seeded_feats(seed, shapes).
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CE_W = [0.266, 0.734]

HHI_CASES = [
    dict(name="ttm3_B4_T15_L1", kind="ttm", n_tasks=3, B=4, T=15, L=1, d=128, h=4, wseed=11, fseed=12),
    dict(name="ttm2_B4_T15_L1", kind="ttm", n_tasks=2, B=4, T=15, L=1, d=128, h=4, wseed=21, fseed=22),
    dict(name="ttm3_B3_T23_L2", kind="ttm", n_tasks=3, B=3, T=23, L=2, d=128, h=4, wseed=31, fseed=32),
    dict(name="asd3_B3_T15_L2", kind="asd", n_tasks=3, B=3, T=15, L=2, d=128, h=4, wseed=41, fseed=42),
    dict(name="hhig_B2_T15_L3_d256", kind="hhig", n_tasks=3, B=2, T=15, L=3, d=256, h=4, wseed=51, fseed=52),
]
HOI_CASES = [
    # REAL forward(x_lta, x_pnr): F per-frame PNR features per clip, averaged by the reference's encode_clips_pnr
    dict(name="lta4_B3_n4_L2_d256", kind="lta4", B=3, n=4, F=3, L=2, d=256, h=8, classes=[5, 7], z=3, wseed=61, fseed=62),
    # HOI EgoT2-s (row F3): real recipes ts_pnr.yaml (d=128, 8 heads -> head dim 16, d_ff = 2d) and ts_ar.yaml
    dict(name="pnr3_B2_L2_d128", kind="pnr3", B=2, L=2, d=128, h=8, task="state_change_detection", wseed=71, fseed=72),
    dict(name="pnr3kf_B2_L1_d256", kind="pnr3", B=2, L=1, d=256, h=8, task="keyframe_localization", wseed=73, fseed=74),
    dict(name="ar3_B2_L2_d128", kind="ar3", B=2, L=2, d=128, h=8, classes=[11, 70], wseed=81, fseed=82),
    dict(name="ar2_B3_L1_d128", kind="ar2", B=3, L=1, d=128, h=4, n=2, classes=[11, 70], wseed=91, fseed=92),
    # HOI EgoT2-g encoder (config C5): REAL encode() of the 6-task model on the 48-token and on the per-clip 'lta' layout
    dict(name="hoig_B2_n3_L2_d256", kind="hoig", B=2, n=3, L=2, d=256, h=8, wseed=95, fseed=96),
    # row F3 remainder: LTA 2-task translator, pre-LN / GELU PNR translator (simple_vit), 2-task and action EgoT2-g encoders
    dict(name="lta2_B3_n3_L2_d256", kind="lta2", B=3, n=3, L=2, d=256, h=8, classes=[5, 7], z=3, wseed=101, fseed=102),
    dict(name="pnrvit_B2_kf", kind="pnrvit", B=2, task="keyframe_localization", wseed=103, fseed=104),
    dict(name="hoig2_B2_L2_d256", kind="hoig2", B=2, L=2, d=256, h=8, wseed=105, fseed=106),
    dict(name="hoiga_B3_L2_d128", kind="hoiga", B=3, L=2, d=128, h=4, wseed=107, fseed=108),
]
# row F4 (producer side): the REAL PNR / OSCC head (`ResNetKeyframeLocalizationHead`) on a seeded res5 map — `middle=True` rows
# for the keyframe head (kt = 1) and the state-change head (kt = T), and the `projection` + activation + permute path
HEAD_CASES = [
    dict(name="pnrhead_N2_T3", kind="pnrhead", N=2, C=2048, T=3, H=8, W=8, classes_kf=17, classes_sc=2, wseed=111, fseed=112),
]


def g_targets(c, task, batch, vocab_size, sy=2):
    """Deterministic (batch, sy) int64 decoder inputs for the EgoT2-g fixtures."""
    import numpy as np
    import torch
    import zlib
    rng = np.random.default_rng([c["fseed"], zlib.crc32(task.encode())])
    return torch.from_numpy(rng.integers(0, vocab_size, (batch, sy))).long()


def sd_keys(m):
    return json.dumps({k: list(v.shape) for k, v in m.state_dict().items()})


def digest(grads, out, full=()):
    """`full`: substrings of parameter names whose WHOLE gradient is stored too (fp16 of g / max|g| + the scale: 2.5e-4 of the
    largest entry per element) - a one-hop per-element check of the large weight gradients against the reference."""
    import numpy as np
    for k, g in grads.items():
        g = g.detach().double().reshape(-1)
        out[f"gnorm/{k}"] = g.norm().numpy()
        out[f"gsum/{k}"] = g.sum().numpy()
        out[f"ghead/{k}"] = (g if g.numel() <= 4096 else g[:64]).float().numpy()
        if any(f in k for f in full):
            scale = g.abs().max().clamp_min(1e-30)
            out[f"gfull_scale/{k}"] = scale.numpy()
            out[f"gfull/{k}"] = (g / scale).numpy().astype(np.float16).reshape(tuple(grads[k].shape))


def run_hhi():
    import numpy as np
    import torch
    from oracle import ref_harness as rh
    from tests.util import seeded_feats, seeded_state_dict
    for c in HHI_CASES:
        args = rh.hhi_args(hidden_dim=c["d"], num_heads=c["h"], dropout=0.0, num_layers=c["L"])
        if c["kind"] == "ttm":
            m = rh.ref_ttm(c["n_tasks"], args)
        elif c["kind"] == "asd":
            m = rh.ref_asd(args)
        else:
            m = rh.ref_hhi_g(args)
        m.load_state_dict(seeded_state_dict(m, c["wseed"]))
        m.train()
        m.pos_embed.dropout.p = 0.0
        feats = seeded_feats(c["fseed"], [(c["B"], c["T"], 256)] * c["n_tasks"])  # order: ttm, lam[, asd]
        out = {"config": np.array(json.dumps(c)), "sd_keys": np.array(sd_keys(m))}
        if c["kind"] == "ttm":
            logits = rh.ref_ttm_forward(m, *feats)
            target = torch.from_numpy(np.random.default_rng(c["fseed"]).integers(0, 2, c["B"])).long()
            loss = torch.nn.functional.cross_entropy(logits, target, weight=torch.tensor(CE_W))
            out["out"] = logits.detach().numpy()
            out["target"] = target.numpy()
        elif c["kind"] == "asd":
            B, T = c["B"], c["T"]
            y = m(feats[1], torch.zeros(B, T, 1, 1), feats[0], feats[2])    # (B*T, d): video->lam, audio->ttm, audio_asd->asd
            loss = (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum()
            out["out"] = y.detach().numpy()
        else:
            B, T = c["B"], c["T"]
            loss = 0
            for task in ("lam", "ttm", "asd"):
                lamf = feats[1][:, :7] if task == "lam" else feats[1]
                enc = m.encode(lamf, torch.zeros(B, T, 1, 1), feats[0], feats[2], task)   # video->lam, audio->ttm, audio_asd->asd
                out[f"out_{task}"] = enc.detach().numpy()
                loss = loss + (enc * torch.linspace(-1, 1, enc.numel()).view_as(enc)).sum()
                # row F1: the sequence decoder + vocabulary head on that memory (target = prompt token + one answer token)
                tgt = g_targets(c, task, enc.shape[1], len(m.vocab))
                dec = m.decode(tgt, enc)                                                   # (sy, batch, |V|)
                out[f"dec_{task}"] = dec.detach().numpy()
                loss = loss + (dec * torch.linspace(-1, 1, dec.numel()).view_as(dec)).sum()
        m.zero_grad()
        loss.backward()
        out["loss"] = loss.detach().numpy()
        # the bench configuration's fixture carries the FULL gradients of its three large matrices (FFN, in-projection)
        full = ("linear1.weight", "linear2.weight", "in_proj_weight") if c["name"] == "ttm3_B4_T15_L1" else ()
        digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, out, full)
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), **out)
        print("wrote", c["name"], "loss", float(loss))


def hoi_feat_shapes(c):
    B = c["B"]
    if c["kind"] == "lta4":
        return [(B, c["n"], c["F"], 8192), (B, c["n"], c["d"]), (B, c["n"], 2048)]      # PNR frames, action, lta
    if c["kind"] == "pnr3":
        return [(B, 16, 8192), (B, 16, 8192), (B, 8, 2048), (B, 8, 256)]           # pnr, oscc, slow, fast
    if c["kind"] == "ar3":
        return [(B, 8, 2048), (B, 8, 256), (B, 16, 8192), (B, 16, 8192)]           # slow, fast, pnr, oscc
    if c["kind"] == "ar2":
        return [(B, 8, 2048), (B, 8, 256), (B, c["n"], 2048)]                      # slow, fast, lta
    if c["kind"] == "hoig":     # pnr, oscc, slow, fast (48-token layout); per-clip pnr frames, action, lta ('lta' layout)
        n, d = c["n"], c["d"]
        return [(B, 16, 8192), (B, 16, 8192), (B, 8, 2048), (B, 8, 256), (B, n, 1, 8192), (B, n, d), (B, n, 2048)]
    if c["kind"] == "lta2":
        return [(B, c["n"], c["d"]), (B, c["n"], 2048)]                            # action, lta
    if c["kind"] in ("pnrvit", "hoig2"):
        return [(B, 16, 8192), (B, 16, 8192)]                                      # pnr, oscc
    if c["kind"] == "hoiga":
        return [(B, 2, c["d"]), (B, 2, c["d"])]                                    # action, lta (both already d wide)
    raise KeyError(c["kind"])


def run_hoi():
    import numpy as np
    import torch
    from oracle import ref_harness as rh
    from tests.util import seeded_feats, seeded_state_dict
    lin = lambda o: (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum()  # noqa: E731
    for c in HOI_CASES:
        feats = seeded_feats(c["fseed"], hoi_feat_shapes(c))
        if c["kind"] == "lta4":
            cfg = rh.hoi_cfg(d=c["d"], heads=c["h"], layers=c["L"], n_clips=c["n"], num_classes=c["classes"], z=c["z"])
            m = rh.ref_lta4(cfg)
        elif c["kind"] == "pnr3":
            m = rh.ref_pnr3(rh.hoi_s_cfg(d=c["d"], layers=c["L"], task=c["task"]))
        elif c["kind"] == "hoig":
            m = rh.ref_hoi_g(rh.hoi_g_args(hidden_dim=c["d"], num_heads=c["h"], num_layers=c["L"]))
        elif c["kind"] == "ar3":
            m = rh.ref_ar3(rh.hoi_s_cfg(d=c["d"], layers=c["L"], heads=c["h"], num_classes=c["classes"]))
        elif c["kind"] == "lta2":
            m = rh.ref_lta2(rh.hoi_cfg(d=c["d"], heads=c["h"], layers=c["L"], n_clips=c["n"], num_classes=c["classes"], z=c["z"]))
        elif c["kind"] == "pnrvit":
            m = rh.ref_pnrvit(c["task"])
        elif c["kind"] == "hoig2":
            m = rh.ref_hoi_g2(rh.hoi_g_args(hidden_dim=c["d"], num_heads=c["h"], num_layers=c["L"]))
        elif c["kind"] == "hoiga":
            m = rh.ref_hoi_ga(rh.hoi_g_args(hidden_dim=c["d"], num_heads=c["h"], num_layers=c["L"]))
        else:
            m = rh.ref_ar2(rh.hoi_s_cfg(d=c["d"], layers=c["L"], heads=c["h"], num_classes=c["classes"], n_clips=c["n"]))
        m.load_state_dict(seeded_state_dict(m, c["wseed"]))
        m.train()
        if c["kind"] == "lta2":         # REAL forward(x)
            outs = m([feats[0], feats[1]])
            named = {"out_verb": outs[0], "out_noun": outs[1]}
        elif c["kind"] == "pnrvit":     # REAL forward(x) over pass-through PNR / OSCC backbones
            named = {"out": m([feats[0], feats[1]])}
        elif c["kind"] == "hoig2":      # REAL encode(video_pnr) + decode
            m.pos_embed.dropout.p = 0.0
            enc = m.encode([feats[0], feats[1]])
            named = {"out": enc, "dec": m.decode(g_targets(c, "pnr", enc.shape[1], len(m.vocab), 2), enc)}
        elif c["kind"] == "hoiga":      # REAL encode(video, task) on both branches + decode
            m.pos_embed.dropout.p = 0.0
            named = {"out_lta": m.encode([feats[0], feats[1]], "lta_verb"), "out_action": m.encode([feats[0], feats[1]], "action_verb")}
            for task, sy in (("lta", 3), ("action", 2)):
                enc = named[f"out_{task}"]
                named[f"dec_{task}"] = m.decode(g_targets(c, task, enc.shape[1], len(m.vocab), sy), enc)
        elif c["kind"] == "hoig":
            m.pos_embed.dropout.p = 0.0
            named = {"out_pnr": rh.hoi_g_encode_other(m, "pnr", *feats[:4]),
                     "out_lta": rh.hoi_g_encode_lta(m, *feats[4:])}
            for task, sy in (("pnr", 2), ("lta", 4)):       # row F1: decoder + vocabulary head, 2- and 4-token targets
                enc = named[f"out_{task}"]
                named[f"dec_{task}"] = m.decode(g_targets(c, task, enc.shape[1], len(m.vocab), sy), enc)
        elif c["kind"] == "lta4":
            outs = m([feats[1], feats[2]], feats[0])      # REAL forward(x_lta, x_pnr)
            named = {"out_verb": outs[0], "out_noun": outs[1]}
        elif c["kind"] == "pnr3":       # the REAL forward(x1, x2) over pass-through backbones
            named = {"out": m([feats[0], feats[1]], [rh.pathway5d(feats[2]), rh.pathway5d(feats[3])])}
        elif c["kind"] == "ar3":        # REAL forward(x_action, x_pnr)
            outs = m([rh.pathway5d(feats[0]), rh.pathway5d(feats[1])], [feats[2], feats[3]])
            named = {"out_verb": outs[0], "out_noun": outs[1]}
        else:                           # REAL forward(x): clips 0..n-1 feed the LTA pathway, the last clip SlowFast
            B, n = c["B"], c["n"]
            x_slow = torch.zeros(B, n + 1, 2048, 8, 1, 1)
            x_fast = torch.zeros(B, n + 1, 256, 8, 1, 1)
            x_slow[:, -1] = rh.pathway5d(feats[0])
            x_fast[:, -1] = rh.pathway5d(feats[1])
            x_slow[:, :n, :, 0, 0, 0] = feats[2]
            outs = m([x_slow, x_fast])
            named = {"out_verb": outs[0], "out_noun": outs[1]}
        loss = sum(lin(o) for o in named.values())
        m.zero_grad()
        loss.backward()
        out = {"config": np.array(json.dumps(c)), "sd_keys": np.array(sd_keys(m)), "loss": loss.detach().numpy()}
        out.update({k: v.detach().numpy() for k, v in named.items()})
        digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, out)
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), **out)
        print("wrote", c["name"], "loss", float(loss))


def run_head():
    """HOI/models/pnr/head_helper.py:293-381 as built at HOI/models/pnr/video_model_builder.py:310-322 (keyframe localisation:
    pool (1, 7, 7), softmax over the frames) and :350-362 (state change: pool (T, 7, 7), softmax over the classes)."""
    import numpy as np
    import torch
    from oracle import ref_harness as rh
    from tests.util import seeded_feats, seeded_state_dict
    lin = lambda o: (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum()  # noqa: E731
    for c in HEAD_CASES:
        fmap = seeded_feats(c["fseed"], [(c["N"], c["C"], c["T"], c["H"], c["W"])])[0]
        out = {"config": np.array(json.dumps(c))}
        for tag, classes, pool, act in (("kf", c["classes_kf"], (1, 7, 7), "softmax_1"), ("sc", c["classes_sc"], (c["T"], 7, 7), "softmax_2")):
            head = rh.ref_pnr_head(classes, pool, act)
            head.load_state_dict(seeded_state_dict(head, c["wseed"]))
            out[f"sd_keys_{tag}"] = np.array(sd_keys(head))
            head.train()
            mid = head([fmap], middle=True)                     # (N, T', 8192)
            out[f"mid_{tag}"] = mid.detach().numpy()
            if tag == "kf":                                     # encode_clips_pnr: one row per input clip (lta_models_lta_transfer.py:335-345)
                out["mid_kf_clipmean"] = mid.mean(dim=1).detach().numpy()
            y = head([fmap])                                    # train mode: projection, no activation, permute(0, 2, 1)
            out[f"proj_train_{tag}"] = y.detach().numpy()
            head.zero_grad()
            lin(y).backward()
            digest({f"{tag}/{k}": p.grad for k, p in head.named_parameters()}, out)
            head.eval()
            with torch.no_grad():
                out[f"proj_eval_{tag}"] = head([fmap]).numpy()
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), **out)
        print("wrote", c["name"])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        {"hhi": run_hhi, "hoi": lambda: (run_hoi(), run_head())}[sys.argv[1]]()
    else:  # HHI and HOI share top-level package names -> one process per tree
        for tree in ("hhi", "hoi"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), tree])
