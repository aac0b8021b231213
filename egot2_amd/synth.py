"""Synthetic workloads for the BASELINE.json configurations: the model class, random
backbone features of the named shape, the task's loss and the algorithmic FLOP count. Shared by bench.py,
tools/run_ttm_synth.py and the tests; nothing here touches the oracle.

Each workload is a dict:
  model        the nn.Module (reference class name / constructor), already on `device`, in train mode
  feats        list of feature tensors (B, T_k, d_in_k) resident on `device`
  loss_fn      () -> scalar loss (forward + task loss); `.backward()` of it is the full backward
  params       trainable parameters
  flops        (forward, backward) algorithmic FLOPs per step (GEMM 2mnk; softmax / LN / elementwise excluded)
  describe     text for the bench line
"""
from __future__ import annotations

from argparse import Namespace
from types import SimpleNamespace as NS
from typing import Callable, Dict, List, Sequence, Tuple

import torch

HHI_G_VOCAB = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}          # HHI/utils/utils.py:12-18
HOI_G_VOCAB = {'</s>': 0, '<unk>': 1, 'pnr': 2, 'oscc': 3, 'action_verb': 4, 'action_noun': 5, 'lta_verb': 6,
               'lta_noun': 7, '0': 8, '1': 9, '2': 10, '3': 11}


def hhi_args(hidden_dim=128, num_heads=4, dropout=0.0, num_layers=1, **kw):
    """The argparse fields the HHI translator constructors read (HHI/configs/ttm/config.py:45-55), backbone-less."""
    a = Namespace(lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None, nofreeze=True,
                  hidden_dim=hidden_dim, num_heads=num_heads, dropout=dropout, num_layers=num_layers, hidden_dim2=512)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def lta4_cfg(n_clips=32, d=768, heads=8, layers=4, dropout=0.1):
    """yacs-shaped tree read by TaskFusionMFTransformerLTA4Task (HOI/models/lta/lta_models_lta_transfer.py:259-313)."""
    return NS(FORECASTING=NS(NUM_INPUT_CLIPS=n_clips, NUM_ACTIONS_TO_PREDICT=20),
              MODEL=NS(TRANSLATION_HEADS=heads, TRANSLATION_LAYERS=layers, TRANSLATION_INPUT_FEATURES=d,
                       TRANSLATION_DROPOUT=dropout, NUM_CLASSES=[115, 478], DROPOUT_RATE=0.0, HEAD_ACT="softmax"),
              TEST=NS(NO_ACT=False))


def encoder_flops(B: int, segs: Sequence[Tuple[int, int, bool]], d: int, d_ff: int, L: int, extra_fwd: float = 0.0):
    """segs = [(T, d_in, projected)]. Returns (forward, backward) FLOPs, SURVEY.md §8d: backward = 2 x forward minus the
    dX of the feature projections (frozen features)."""
    S = sum(t for t, _, _ in segs)
    N = B * S
    proj = sum(2.0 * B * t * k * d for t, k, pj in segs if pj)
    layer = 2.0 * N * d * 3 * d + 4.0 * B * S * S * d + 2.0 * N * d * d + 4.0 * N * d * d_ff
    fwd = proj + L * layer + extra_fwd
    return fwd, 2.0 * fwd - proj


def _randn(gen, shape, device):
    return torch.randn(*shape, generator=gen).to(device)


def make_workload(name: str, device, *, batch: int = 256, frames: int = 15, layers: int | None = None,
                  dtype: str | None = None, impl: str = "auto", dropout: float | None = None, seed: int = 1234,
                  encoder_only: bool = False, feat_dtype: str = "f32", feat_frames: int = 1, feat_source: str = "tensor",
                  fused_ce: bool = True) -> Dict:
    from . import functional as F_egx
    from .train import CrossEntropyLoss
    name = name.lower()
    gen = torch.Generator().manual_seed(seed)
    B, T = batch, frames
    torch.manual_seed(0)
    if name in ("c1", "c2"):
        from . import hhi_ttm
        K = 2 if name == "c1" else 3
        if name == "c1" and batch == 256:
            B = 32                                   # B=32, the run_ttm.py plumbing case
        L = layers or 1
        p = 0.5 if dropout is None else dropout      # README.md:81,84 recipe
        cls = hhi_ttm.TaskFusionMFTransformer2Task if K == 2 else hhi_ttm.TaskFusionMFTransformer3Task
        model = cls(hhi_args(hidden_dim=128, num_heads=4, dropout=p, num_layers=L))
        model = model.to(device).set_compute(dtype or "f32", impl).train()
        feats = [_randn(gen, (B, T, 256), device) for _ in range(K)]
        target = torch.randint(0, 2, (B,), generator=gen).to(device)
        crit = CrossEntropyLoss(torch.FloatTensor([0.266, 0.734])).to(device)     # video_task_2loader.py:21-22
        if fused_ce:        # the criterion evaluated inside the forward (egx_ce: the head epilogue of the launch that writes the logits)
            cw = crit.weight
            loss_fn = lambda: model.forward_features(*feats, target=target, class_weight=cw)[1]   # noqa: E731
        else:
            loss_fn = lambda: crit(model.forward_features(*feats), target)   # noqa: E731
        segs = [(T, 256, True)] * K
        fl = encoder_flops(B, segs, 128, 2048, L, extra_fwd=2.0 * B * 128 * 2)
        desc = (f"configs[{1 if K == 3 else 0}]: TTM {K}-task translator, {L} layer d=128 h=4 d_ff=2048, B={B}/GPU T={T} "
                f"S={K * T}, synthetic N(0,1) features, random-init weights, train mode dropout={p} (+0.1 on PE), weighted CE"
                + (" (evaluated in the forward's head epilogue)" if fused_ce else ""))
        d, S = 128, K * T
    elif name == "c3":
        from . import hhi_asd
        L = layers or 2
        p = 0.1 if dropout is None else dropout
        model = hhi_asd.TaskFusionMFTransformer3Task(hhi_args(hidden_dim=128, num_heads=4, dropout=p, num_layers=L))
        model = model.to(device).set_compute(dtype or "bf16", impl).train()
        feats = [_randn(gen, (B, T, 256), device) for _ in range(3)]
        target = torch.randint(0, 2, (B * T,), generator=gen).to(device)
        # the classifier lives outside the model: lossAV = Linear(dim, 2) + CE(weight [1, 4]) + scores, HHI/tasks/asd/loss.py:11-30
        # (video_task_taskspecific.py:24,33: nloss, _, _, prec = self.lossAV.forward(outsAV, labels))
        head = hhi_asd.lossAV(128).to(device)
        if fused_ce:    # lossAV evaluated by the encoder's own launches where the per-clip kernels can (egx_token_ce; elsewhere the same two launches)
            loss_fn = lambda: model.forward_features(*feats, lossav=head, labels=target)[0]   # noqa: E731
        else:
            loss_fn = lambda: head(model.forward_features(*feats), target)[0]   # noqa: E731
        segs = [(T, 256, True)] * 3
        fl = encoder_flops(B, segs, 128, 2048, L)
        desc = (f"configs[2]: ASD 3-task translator, {L} layers d=128 h=4 d_ff=2048, B={B}/GPU T={T} S={3 * T}, per-frame "
                f"output (B*T, d) + lossAV (FC + weighted CE + scores" + (", evaluated by the encoder launches" if fused_ce else ", one launch each way") + f"), dropout={p}")
        d, S = 128, 3 * T
        model.extra_params = list(head.FC.parameters())
    elif name == "pnr":
        # the shipped PNR / OSCC EgoT2-s recipe (HOI/configs/pnr/ts_pnr.yaml:28-34; video_model_transfer_3task.py:212-258):
        # 16 + 16 + 8 + 8 = 48 tokens, d = 128, 8 heads of 16, d_ff = 256, 6 layers, feature dropout, learned positions
        from . import hoi_pnr
        L = layers or 6
        p = 0.1 if dropout is None else dropout
        cfg = NS(DATA=NS(TASK="state_change_detection"),
                 MODEL=NS(TRANSLATION_INPUT_FEATURES=128, TRANSLATION_LAYERS=L, FEAT_DROPOUT_RATE=p, TRANSFORMER_DROPOUT_RATE=p))
        model = hoi_pnr.TaskFusionMFTransformer3TaskDropout(cfg).to(device).set_compute(dtype or "f32s", impl).train()
        feats = [torch.randn(B, 16, 8192, device=device), torch.randn(B, 16, 8192, device=device),
                 _randn(gen, (B, 8, 2048), device), _randn(gen, (B, 8, 256), device)]
        target = torch.randint(0, 2, (B,), generator=gen).to(device)
        loss_fn = lambda: torch.nn.functional.cross_entropy(model.forward_features(*feats).squeeze(2), target)   # noqa: E731
        segs = [(16, 8192, True), (16, 8192, True), (8, 2048, True), (8, 256, True)]
        fl = encoder_flops(B, segs, 128, 256, L, extra_fwd=2.0 * B * 128 * 2)
        desc = (f"HOI PNR / OSCC EgoT2-s translator (ts_pnr.yaml recipe): S=48 (16+16+8+8), d=128 h=8 d_ff=256, {L} layers, B={B}/GPU, "
                f"8192-wide PNR / OSCC features, feature + encoder dropout {p}, learned positions, CE")
        d, S = 128, 48
    elif name == "c4":
        from . import hoi_lta
        L = layers or 4
        n = 32
        p = 0.1 if dropout is None else dropout
        model = hoi_lta.TaskFusionMFTransformerLTA4Task(lta4_cfg(n, 768, 8, L, p))
        model = model.to(device).set_compute(dtype or "bf16", impl).train()
        # feature hand-off (row F4): --feat-dtype bf16 = the backbones write packed bf16 features (the action stream feeds
        # the LayerNorm directly and stays fp32); --feat-frames F = per-frame PNR / OSCC features, temporal mean fused
        fr = max(int(feat_frames), 1)
        feats = [torch.randn(B, n * fr, 8192, device=device), torch.randn(B, n * fr, 8192, device=device),
                 _randn(gen, (B, n, 768), device), _randn(gen, (B, n, 2048), device)]
        if feat_dtype == "bf16":
            feats = [feats[0].bfloat16(), feats[1].bfloat16(), feats[2], feats[3].bfloat16()]
        producer = None
        if feat_source == "sink":
            # producer side of row F4: the PNR / OSCC token rows come out of egx_pool_pack (the backbones' pooling head fused
            # with the per-clip temporal mean and the bf16 cast) into a FeatureSink; the translator reads the packed bf16
            # streams in place. The maps of ONE token position stand in for all n (a backbone would produce n different ones).
            import time
            from .feature_sink import FeatureSink
            assert fr == 1, "--feat-source sink pools the frames inside the producer kernel"
            F_frames = 4
            sink = FeatureSink(device, torch.bfloat16)
            fmaps = [torch.randn(B, 2048, F_frames, 8, 8, device=device) for _ in range(2)]
            for k, nm in enumerate(("pnr", "oscc")):
                sink.alloc(nm, B, n, 8192)

            def produce():
                for k, nm in enumerate(("pnr", "oscc")):
                    for i in range(n):
                        sink.put_pooled_map(nm, fmaps[k], (1, 7, 7), token=i, frames_mean=True)
            produce()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            produce()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            nbytes = 2.0 * n * (fmaps[0].numel() * 4 + B * 8192 * 2)
            producer = {"kernel": "egx::pool_pack_kernel", "launches": 2 * n, "maps_per_launch": B,
                        "map_shape": [2048, F_frames, 8, 8], "us_per_launch": dt / (2 * n) * 1e6,
                        "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / dt / 1e9,
                        "note": "AvgPool3d(1,7,7) + permute + mean over frames + bf16 cast of one res5 map batch per launch, into token "
                                "row i of the packed (B, n, 8192) stream; run before the timed region (frozen-backbone side)"}
            feats = [sink.get("pnr"), sink.get("oscc"), feats[2], feats[3].bfloat16()]
            feat_dtype = "bf16 (FeatureSink)"
        tv = torch.randint(0, 115, (B * 20,), generator=gen).to(device)
        tn = torch.randint(0, 478, (B * 20,), generator=gen).to(device)

        def loss_fn():
            verbs, nouns = (model.forward_frame_features(*feats, frames_per_clip=fr) if fr > 1
                            else model.forward_features(*feats))      # (B, 20, 115), (B, 20, 478)
            # hundreds of classes x thousands of rows: outside the (B, few-class) shape the fused CE kernel is built for
            ce = torch.nn.functional.cross_entropy
            return ce(verbs.reshape(-1, 115), tv) + ce(nouns.reshape(-1, 478), tn)
        segs = [(n, 8192, True), (n, 8192, True), (n, 768, False), (n, 2048, True)]
        fl = encoder_flops(B, segs, 768, 2048, L, extra_fwd=2.0 * B * 768 * 593 * 20)
        desc = (f"configs[3]: HOI LTA 4-task translator (PNR+OSCC+AR+LTA), n={n} clips/task S={4 * n} d=768 h=8 d_ff=2048, "
                f"{L} layers, B={B}/GPU, MultiTaskHead 20x593 + CE, dropout={p}, features {feat_dtype}"
                + (f", {fr} frames per clip pooled in the hand-off" if fr > 1 else ""))
        d, S = 768, 4 * n
    elif name in ("c5", "c5hhi", "c5hoi"):
        L = layers or 3
        p = 0.1 if dropout is None else dropout
        if name == "c5hhi":
            from . import hhi_multitask
            model = hhi_multitask.TaskTranslationPromptTransformer(hhi_args(hidden_dim=256, num_heads=4, num_layers=L, dropout=p), HHI_G_VOCAB)
            feats = [_randn(gen, (B, T, 256), device) for _ in range(3)]
            segs = [(T, 256, True)] * 3
            d, S, V, task, ntok = 256, 3 * T, len(HHI_G_VOCAB), "ttm", 2
            y = torch.stack([torch.full((B,), HHI_G_VOCAB["ttm"]), torch.randint(5, 7, (B,), generator=gen),
                             torch.zeros(B, dtype=torch.long)], dim=1).to(device)
        else:
            from . import hoi_multitask
            model = hoi_multitask.TaskTranslationPromptTransformer6Task(NS(hidden_dim=512, num_heads=8, num_layers=L, dropout=p), HOI_G_VOCAB)
            feats = [_randn(gen, (B, 16, 8192), device), _randn(gen, (B, 16, 8192), device), _randn(gen, (B, 8, 2048), device),
                     _randn(gen, (B, 8, 256), device)]
            segs = [(16, 8192, True), (16, 8192, True), (8, 2048, True), (8, 256, True)]
            d, S, V, task, ntok = 512, 48, len(HOI_G_VOCAB), "pnr", 2
            y = torch.stack([torch.full((B,), HOI_G_VOCAB["pnr"]), torch.randint(8, 12, (B,), generator=gen),
                             torch.zeros(B, dtype=torch.long)], dim=1).to(device)
        model = model.to(device).set_compute(dtype or "bf16", impl).train()

        def loss_fn():
            mem = model.encode_features(task, *feats)
            if encoder_only:
                return mem.sum()
            logits = model.decode(y[:, :-1], mem)                       # (sy, B, V)
            return F_egx.weighted_cross_entropy(logits.permute(1, 0, 2).reshape(-1, V), y[:, 1:].reshape(-1))
        fl = encoder_flops(B, segs, d, 2048, L)
        desc = (f"configs[4]: EgoT2-g {'HHI (3 tasks, d=256 h=4)' if name == 'c5hhi' else 'HOI (4 backbones / 6 tasks, d=512 h=8)'} "
                f"encoder S={S} {L} layers" + ("" if encoder_only else f" + {ntok}-token sequence decoder + vocabulary CE")
                + f", B={B}/GPU, dropout={p}; FLOPs counted for the encoder only")
    else:
        raise ValueError(f"unknown workload {name!r} (c1, c2, c3, c4, c5hhi, c5hoi)")
    params = [q for q in model.parameters() if q.requires_grad] + list(getattr(model, "extra_params", []))
    return {"name": name, "model": model, "feats": feats, "loss_fn": loss_fn, "params": params, "flops": fl,
            "describe": desc, "B": B, "S": S, "d": d, "segs": segs, "L": L, "compute": model.egx_compute,
            "batch_arg": batch, "frames": frames, "layers_arg": layers or 0, "encoder_only": bool(encoder_only),
            "producer": locals().get("producer"), "dff": 256 if name == "pnr" else 2048}
