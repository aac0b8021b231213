"""-m gpu: producer side of the feature hand-off: egx_pool_pack against the torch ops the reference's
PNR / OSCC head runs for `middle=True` (HOI/models/pnr/head_helper.py:353-373) and against encode_clips_pnr's temporal mean
(HOI/models/lta/lta_models_lta_transfer.py:335-345); backbone stub -> FeatureSink -> translator against today's `forward()`
and the fp64 oracle; the on-disk Stage-II feature cache."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn

from oracle import translator_ref as tr
from tests.util import seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu


def _head_ref(fmap, pool):
    """The reference head's `middle=True` arithmetic: the oracle's restatement (fp64), itself pinned to the REAL
    `ResNetKeyframeLocalizationHead` by tests/golden/pnrhead_*.npz (tests/test_oracle_golden.py)."""
    return tr.pnr_head_forward(fmap.double(), pool)


def test_pool_pack_and_drop_in_head_match_the_real_reference_head(egx_lib, cuda):
    """One hop to the reference: `egx_pool_pack` / `PooledFeatureHead` against the outputs the REAL head
    (HOI/models/pnr/head_helper.py:293-381, imported by tests/golden/make_golden.py) produced on the same seeded res5 map —
    `middle=True` rows for kt = 1 and kt = T, the per-clip temporal mean of encode_clips_pnr, the projection + permute path in
    train mode (with the projection's gradients) and with the eval-mode activation. fp32 throughout."""
    import json
    from egot2_amd.feature_sink import FeatureSink, PooledFeatureHead
    from tests.test_oracle_golden import HEAD_FIXTURES, head_fixture_cases, load_fixture
    assert HEAD_FIXTURES, "tests/golden/pnrhead_*.npz is missing"
    lin = lambda o: (o * torch.linspace(-1, 1, o.numel(), device=o.device).view_as(o)).sum()  # noqa: E731
    for name in HEAD_FIXTURES:
        c, z = load_fixture(name)
        fmap = seeded_feats(c["fseed"], [(c["N"], c["C"], c["T"], c["H"], c["W"])])[0].to(cuda)
        for tag, classes, pool, act_dim in head_fixture_cases(c):
            ref_mid = torch.from_numpy(z[f"mid_{tag}"])
            # (1) the kernel itself, all rows
            sink = FeatureSink(cuda, torch.float32)
            sink.alloc("x", c["N"], ref_mid.shape[1], ref_mid.shape[2]).fill_(float("nan"))
            got = sink.put_pooled_map("x", fmap, pool).cpu()
            assert (got - ref_mid).abs().max().item() < 2e-6, (name, tag)
            # (2) the drop-in head: same constructor arguments and state_dict keys as the reference head
            head = PooledFeatureHead([c["C"]], classes, [list(pool)], dropout_rate=0.0, act_func=f"softmax_{act_dim}")
            assert {k: list(v.shape) for k, v in head.state_dict().items()} == json.loads(str(z[f"sd_keys_{tag}"]))
            head.load_state_dict(seeded_state_dict(head, c["wseed"]))
            head = head.to(cuda).train()
            mid = head([fmap], middle=True)
            assert mid.dtype == torch.float32 and (mid.cpu() - ref_mid).abs().max().item() < 2e-6
            y = head([fmap])
            ref_y = torch.from_numpy(z[f"proj_train_{tag}"])
            assert tuple(y.shape) == tuple(ref_y.shape) and (y.detach().cpu() - ref_y).abs().max().item() < 1e-4 * (1 + ref_y.abs().max().item())
            lin(y).backward()
            torch.cuda.synchronize()
            for k, p in head.named_parameters():
                g = p.grad.double().cpu().reshape(-1)
                n_ref = float(z[f"gnorm/{tag}/{k}"])
                assert abs(g.norm().item() - n_ref) < 1e-4 * n_ref, (name, tag, k)
                h_ref = torch.from_numpy(z[f"ghead/{tag}/{k}"]).double()
                assert (g[:h_ref.numel()] - h_ref).abs().max().item() < 1e-4 * (1 + h_ref.abs().max().item()), (name, tag, k)
            head.eval()
            with torch.no_grad():
                ye = head([fmap])
            assert (ye.cpu() - torch.from_numpy(z[f"proj_eval_{tag}"])).abs().max().item() < 1e-5
        # (3) encode_clips_pnr: AvgPool + permute + per-clip `.mean(dim=1)` fused into token row i of a packed (N, n, 8192) stream
        ref_cm = torch.from_numpy(z["mid_kf_clipmean"])
        sink = FeatureSink(cuda, torch.float32)
        sink.alloc("pnr", c["N"], 3, ref_cm.shape[1]).fill_(float("nan"))
        for i in range(3):
            sink.put_pooled_map("pnr", fmap, (1, 7, 7), token=i, frames_mean=True)
        got = sink.get("pnr").cpu()
        assert all((got[:, i] - ref_cm).abs().max().item() < 2e-6 for i in range(3))
        s16 = FeatureSink(cuda, torch.bfloat16)      # what the wide path's projection GEMM reads in place: one bf16 rounding of the same rows
        s16.alloc("pnr", c["N"], 1, ref_cm.shape[1])
        got16 = s16.put_pooled_map("pnr", fmap, (1, 7, 7), token=0, frames_mean=True).float().cpu()[:, 0]
        assert (got16 - ref_cm).abs().max().item() <= 2 ** -8 * ref_cm.abs().max().item() + 1e-6


@pytest.mark.parametrize("N,C,T,H,W,pool", [(3, 2048, 4, 8, 8, (1, 7, 7)), (2, 192, 3, 7, 7, (1, 6, 6)), (2, 100, 5, 8, 8, (5, 7, 7)),
                                            (1, 64, 1, 9, 9, (1, 8, 8)), (4, 2048, 2, 8, 8, (2, 7, 7))])
def test_pool_pack_matches_the_reference_heads_torch_ops(egx_lib, cuda, N, C, T, H, W, pool):
    from egot2_amd.feature_sink import FeatureSink
    g = torch.Generator().manual_seed(N * 31 + C + T)
    fmap = torch.randn(N, C, T, H, W, generator=g).to(cuda)
    ref = _head_ref(fmap.cpu(), pool)                                   # (N, T', H'W'C)
    rows, row_len = ref.shape[1], ref.shape[2]
    # (1) all rows, fp32 out: exact up to summation order
    s32 = FeatureSink(cuda, torch.float32)
    s32.alloc("x", N, rows, row_len)
    out = s32.put_pooled_map("x", fmap, pool)
    assert (out.double().cpu() - ref).abs().max().item() < 1e-5
    # (2) bf16 out: one rounding
    s16 = FeatureSink(cuda, torch.bfloat16)
    s16.alloc("x", N, rows, row_len)
    out16 = s16.put_pooled_map("x", fmap, pool)
    assert (out16.double().cpu() - ref).abs().max().item() < 2 ** -8 * ref.abs().max().item() + 1e-6
    # (3) bf16 MAP in (a backbone running in bf16): compare with the reference on the rounded map
    ref_b = _head_ref(fmap.bfloat16().float().cpu(), pool)
    out_b = FeatureSink(cuda, torch.float32)
    out_b.alloc("x", N, rows, row_len)
    assert (out_b.put_pooled_map("x", fmap.bfloat16(), pool).double().cpu() - ref_b).abs().max().item() < 1e-5
    # (4) fused temporal mean into token rows of a (N, n_clips, row_len) stream: encode_clips_pnr's `.mean(dim=1)` per input clip
    if pool[0] == 1:
        n_clips = 3
        sink = FeatureSink(cuda, torch.float32)
        sink.alloc("pnr", N, n_clips, row_len).fill_(float("nan"))
        maps = [fmap, fmap.flip(1), fmap * 0.5]
        for i, m in enumerate(maps):
            sink.put_pooled_map("pnr", m.contiguous(), pool, token=i, frames_mean=True)
        want = torch.stack([_head_ref(m.cpu(), pool).mean(dim=1) for m in maps], dim=1)
        assert (sink.get("pnr").double().cpu() - want).abs().max().item() < 1e-5


class _StubBackbone(nn.Module):
    """A frozen PNR / OSCC backbone reduced to what matters at the boundary: `model([clip], middle=True)` runs a (fixed) stem that
    yields the res5 map and then the head module under the reference's attribute name. `flip` = the OSCC stand-in of the
    fixtures (channel-reversed map) so that the two streams differ."""

    def __init__(self, head, flip=False):
        super().__init__()
        self.Keyframe_localisation_head = head
        self.flip = flip

    def forward(self, x, middle=False):
        fmap = x[0]
        if self.flip:
            fmap = fmap.flip(1).contiguous()
        return self.Keyframe_localisation_head([fmap], middle=middle)


class _TorchHead(nn.Module):
    """Today's path: the reference head's ops in torch (fp32 tensors all the way)."""

    def __init__(self, pool):
        super().__init__()
        self.pathway0_avgpool = nn.AvgPool3d(pool, stride=1)
        self.projection = nn.Linear(8192, 17)

    def forward(self, inputs, middle=False):
        x = self.pathway0_avgpool(inputs[0]).permute((0, 2, 3, 4, 1))
        x = x.reshape(x.shape[0], x.shape[1], 2048 * 2 * 2)
        return x if middle else self.projection(x).permute(0, 2, 1)


class _ActionStub(nn.Module):
    def forward(self, x, *a, **k):
        return x[0]


class _LtaStub(nn.Module):
    def forward(self, x, *a, **k):
        return x[1].transpose(0, 1)


def _lta_model(cuda, n, d, L):
    from egot2_amd import hoi_lta
    cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=n, NUM_ACTIONS_TO_PREDICT=3),
             MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=L, TRANSLATION_INPUT_FEATURES=d, TRANSLATION_DROPOUT=0.0,
                      NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
    sd = seeded_state_dict(m, 41)
    m.load_state_dict(sd)
    pool = (1, 7, 7)
    m.pnr_model = _StubBackbone(_TorchHead(pool))
    m.oscc_model = _StubBackbone(_TorchHead(pool), flip=True)
    m.action_model, m.lta_model = _ActionStub(), _LtaStub()
    return m.to(cuda).set_compute("bf16").train(), sd


def test_backbone_stub_through_the_sink_equals_todays_forward_and_the_oracle(egx_lib, cuda):
    """forward(x_lta, x_pnr) with the torch heads (fp32 (B, T', 8192) per clip, `.mean(dim=1)`, torch.stack) vs the same model
    after enable_feature_sink(): the heads become PooledFeatureHead and write bf16 token rows into the packed stream. Both are
    compared with the fp64 oracle on the torch-pooled features; the parameter gradients of both paths must agree."""
    B, n, F, d, L = 3, 4, 3, 256, 2
    g = torch.Generator().manual_seed(7)
    x_pnr = torch.randn(B, n, 2048, F, 8, 8, generator=g).to(cuda)
    act, lta = [f.to(cuda) for f in seeded_feats(8, [(B, n, d), (B, n, 2048)])]
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    res = {}
    for mode in ("torch_heads", "sink"):
        m, sd = _lta_model(cuda, n, d, L)
        if mode == "sink":
            m.enable_feature_sink(torch.bfloat16)
            from egot2_amd.feature_sink import PooledFeatureHead
            assert isinstance(m.pnr_model.Keyframe_localisation_head, PooledFeatureHead)
        out = m([act, lta], x_pnr)
        (lin(out[0]) + lin(out[1])).backward()
        torch.cuda.synchronize()
        res[mode] = ([o.detach().double().cpu() for o in out],
                     {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None and "_model." not in k})
        if mode == "sink":
            assert m._sink.get("pnr").dtype == torch.bfloat16 and tuple(m._sink.get("pnr").shape) == (B, n, 8192)
    # oracle on the reference arithmetic of the producer side
    pooled = torch.stack([_head_ref(x_pnr[:, i].cpu(), (1, 7, 7)).mean(dim=1) for i in range(n)], dim=1)
    pooled_o = torch.stack([_head_ref(x_pnr[:, i].flip(1).cpu(), (1, 7, 7)).mean(dim=1) for i in range(n)], dim=1)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = tr.lta4_forward(sd64, 8, pooled, pooled_o, act.double().cpu(), lta.double().cpu(), [5, 7])
    for mode in res:
        for o, r in zip(res[mode][0], ref):
            assert (o - r).abs().max().item() < 1e-2 * max(1.0, r.abs().max().item()), mode
    for o_a, o_b in zip(res["torch_heads"][0], res["sink"][0]):
        assert (o_a - o_b).abs().max().item() < 1e-2 * max(1.0, o_a.abs().max().item())
    ga, gb = res["torch_heads"][1], res["sink"][1]
    assert set(ga) == set(gb)
    for k in ga:
        assert (ga[k] - gb[k]).norm().item() <= 5e-2 * ga[k].norm().item() + 1e-7, k      # two bf16 roundings of the 8192-wide operand


def test_two_forwards_before_backward_keep_their_own_sink_streams(egx_lib, cuda):
    """the sink stream is refilled through a raw pointer; a second forward before the first one's backward must not
    overwrite the features the first one's projections saved. Summed micro-batch gradients = gradients of the two runs apart."""
    B, n, F, d, L = 2, 4, 2, 256, 1
    g = torch.Generator().manual_seed(17)
    xa = torch.randn(B, n, 2048, F, 8, 8, generator=g).to(cuda)
    xb = torch.randn(B, n, 2048, F, 8, 8, generator=g).to(cuda)
    act, lta = [f.to(cuda) for f in seeded_feats(18, [(B, n, d), (B, n, 2048)])]
    lin = lambda out: sum((t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum() for t in out)  # noqa: E731
    m, _ = _lta_model(cuda, n, d, L)
    m.enable_feature_sink(torch.bfloat16)
    grads = []
    for x in (xa, xb):
        m.zero_grad()
        lin(m([act, lta], x)).backward()
        grads.append(m.proj_pnr.weight.grad.clone())
    m.zero_grad()
    la = lin(m([act, lta], xa))
    pa = m._sink.get("pnr").data_ptr()
    lb = lin(m([act, lta], xb))                 # second forward BEFORE the first backward
    assert m._sink.get("pnr").data_ptr() != pa
    (la + lb).backward()
    want = grads[0] + grads[1]
    assert (m.proj_pnr.weight.grad - want).norm().item() <= 2e-3 * want.norm().item()
    with torch.no_grad():                       # inference keeps reusing one buffer
        m([act, lta], xa); p1 = m._sink.get("pnr").data_ptr()
        m([act, lta], xb); assert m._sink.get("pnr").data_ptr() == p1


def test_pooled_head_projection_path_and_cache_roundtrip(egx_lib, cuda, tmp_path):
    """middle=False of the drop-in head (pool -> Linear(8192, classes) -> permute) vs torch; FeatureCache: save packed streams per
    clip, load a batch back into a sink, translate -> identical logits to the uncached call."""
    from egot2_amd.feature_sink import FeatureCache, FeatureSink, PooledFeatureHead
    head = PooledFeatureHead([2048], 17, [[1, 7, 7]], act_func="none").to(cuda).train()
    fmap = torch.randn(2, 2048, 3, 8, 8, generator=torch.Generator().manual_seed(3)).to(cuda)
    y = head([fmap], middle=False)
    want = torch.nn.functional.linear(_head_ref(fmap.cpu(), (1, 7, 7)), head.projection.weight.double().cpu(), head.projection.bias.double().cpu()).permute(0, 2, 1)
    assert tuple(y.shape) == (2, 17, 3) and (y.double().cpu() - want).abs().max().item() < 1e-3
    m, _ = _lta_model(cuda, 4, 256, 1)
    m.eval()
    B = 3
    feats = {"pnr": torch.randn(B, 4, 8192, device=cuda).bfloat16(), "oscc": torch.randn(B, 4, 8192, device=cuda).bfloat16(),
             "action": torch.randn(B, 4, 256, device=cuda), "lta": torch.randn(B, 4, 2048, device=cuda).bfloat16()}
    cache = FeatureCache(str(tmp_path / "stage2"))
    for b in range(B):
        assert not cache.has(f"clip{b}")
        cache.save(f"clip{b}", {k: v[b:b + 1] for k, v in feats.items()})
    sink = FeatureSink(cuda)
    got = cache.load_batch([f"clip{b}" for b in range(B)], sink)
    assert all(torch.equal(got[k], feats[k]) and got[k].dtype == feats[k].dtype for k in feats)
    with torch.no_grad():
        a = m.forward_features(feats["pnr"], feats["oscc"], feats["action"], feats["lta"])
        b_ = m.forward_features(got["pnr"], got["oscc"], got["action"], got["lta"])
    assert all(torch.equal(x, y_) for x, y_ in zip(a, b_))
