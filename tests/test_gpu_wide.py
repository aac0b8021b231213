"""-m gpu: the wide bf16 path (BASELINE.json configs[3] HOI LTA 4-task d = 768 and configs[4] EgoT2-g encoders d = 256 / 512):
bf16-storage MFMA GEMMs (NT / TN), MFMA attention forward / backward and the whole translator against the oracle.

Unit tolerances: the kernels take bf16 operands and accumulate in fp32, so against an fp32 / fp64 product of the SAME
bf16-rounded operands they must agree to accumulation-order noise (1e-3 relative to the result's scale); bf16 OUTPUTS add
one rounding (2^-9 relative). Model level: BASELINE.json north_star, outputs within 1e-2 of the fp64 oracle."""
import ctypes as C

import pytest
import torch

from oracle import translator_ref as tr
from tests.util import seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("M,N,K,bias,relu,res", [(135, 256, 256, True, False, False), (1024, 768, 2048, True, True, False),
                                                 (300, 2304, 768, False, False, True), (128, 128, 64, True, False, True),
                                                 (4097, 512, 8192, True, False, False), (77, 2048, 768, True, True, True),
                                                 # round 3 tile variants: 256 x 192 ping-pong (N = 768 / 1536, >= 256 tiles), 256 x 256
                                                 # ping-pong, 64 x 64 (<= 256 tiles of 128 x 128), each with ragged M
                                                 (32768 - 37, 768, 768, True, False, True), (16384, 1536, 256, True, True, False),
                                                 (32768, 1024, 128, False, False, True), (512, 512, 2048, True, False, True),
                                                 (11520 - 5, 256, 1024, True, True, False)])
def test_wide_gemm_nt(egx_lib, cuda, M, N, K, bias, relu, res):
    """C = A B^T (+ bias) (ReLU) (+ residual): ragged M (clamped loads, masked stores), every epilogue, both output types,
    every tile variant (the stores go through LDS since round 3)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(cuda).bfloat16()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda).bfloat16()
    b = torch.randn(N, generator=g).to(cuda) if bias else None
    R = torch.randn(M, N, generator=g).to(cuda) if res else None
    Cf = torch.empty(M, N, device=cuda)
    Cb = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    scratch = torch.empty(egx_lib.egx_wide_gemm_scratch(0, M, N, K), dtype=torch.uint8, device=cuda)
    rc = egx_lib.egx_wide_gemm(0, _ptr(A), _ptr(B), _ptr(Cf), _ptr(Cb), M, N, K, _ptr(b), int(relu), _ptr(R), _ptr(scratch), _stream())
    assert rc == 0, egx_lib.egx_last_error()
    ref = A.double() @ B.double().T
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp(min=0)
    if res:
        ref = ref + R.double()
    scale = ref.abs().max().item()
    assert (Cf.double() - ref).abs().max().item() < 1e-3 * scale
    assert (Cb.double() - ref).abs().max().item() < 6e-3 * scale          # + one bf16 rounding of the output


@pytest.mark.parametrize("M,N,K", [(256, 128, 135), (768, 2048, 4133), (128, 8192, 64), (2304, 768, 1000)])
def test_wide_gemm_tn(egx_lib, cuda, M, N, K):
    """C = A^T B over K tokens (weight-gradient shape): ragged K (zero-page rows), split-K slabs, transposed LDS reads."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = torch.randn(K, M, generator=g).to(cuda).bfloat16()
    B = torch.randn(K, N, generator=g).to(cuda).bfloat16()
    Cf = torch.empty(M, N, device=cuda)
    scratch = torch.empty(egx_lib.egx_wide_gemm_scratch(2, M, N, K), dtype=torch.uint8, device=cuda)
    rc = egx_lib.egx_wide_gemm(2, _ptr(A), _ptr(B), _ptr(Cf), None, M, N, K, None, 0, None, _ptr(scratch), _stream())
    assert rc == 0, egx_lib.egx_last_error()
    ref = A.double().T @ B.double()
    assert (Cf.double() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()
    # bitwise reproducible: fixed-order slab reduction, no atomics
    C2 = torch.empty_like(Cf)
    egx_lib.egx_wide_gemm(2, _ptr(A), _ptr(B), _ptr(C2), None, M, N, K, None, 0, None, _ptr(scratch), _stream())
    assert torch.equal(Cf, C2)


def _attn_ref(qkv, d_out, B, S, H, d):
    """fp64 attention on the bf16-rounded operands, with autograd."""
    x = qkv.double().view(B, S, 3, H, d // H).requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)      # (B, H, S, dh)
    s = q @ k.transpose(-1, -2) / (d // H) ** 0.5
    lse = torch.logsumexp(s, dim=-1)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * S, d)
    o.backward(d_out.double())
    return o.detach(), lse.detach(), x.grad.reshape(B * S, 3 * d)


@pytest.mark.parametrize("B,S,H,d", [(3, 128, 8, 768), (2, 45, 4, 256), (5, 48, 8, 512), (2, 17, 4, 128), (1, 1, 2, 256),
                                     (2, 100, 2, 256), (2, 64, 8, 256), (3, 65, 4, 384),
                                     (2, 129, 4, 256), (3, 450, 4, 256), (2, 300, 8, 256), (1, 480, 8, 512), (2, 225, 8, 512), (40, 150, 4, 256)])
def test_wide_attention_fwd_bwd(egx_lib, cuda, B, S, H, d):
    """S^T = K Q^T orientation, softmax, O^T = V^T P^T and both backward passes against fp64 autograd. Covers head dims
    32 / 64 / 96 / 128, both tile counts (S <= 64 and S <= 128), ragged last tiles and a single token; S > 128: the online-softmax
    kernels (head dim 32 / 64, split and unsplit grids)."""
    g = torch.Generator().manual_seed(S * 7 + d)
    qkv = torch.randn(B * S, 3 * d, generator=g).to(cuda).bfloat16()
    d_out = torch.randn(B * S, d, generator=g).to(cuda).bfloat16()
    out = torch.empty(B * S, d, device=cuda, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=cuda)
    dqkv = torch.full((B * S, 3 * d), float("nan"), device=cuda, dtype=torch.bfloat16)
    delta = torch.empty(B, H, S, device=cuda)
    assert egx_lib.egx_wide_attention_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, S, H, d, 0.0, 0, _stream()) == 0, egx_lib.egx_last_error()
    assert egx_lib.egx_wide_attention_bwd(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(d_out), _ptr(dqkv), _ptr(delta), B, S, H, d, 0.0, 0, _stream()) == 0, egx_lib.egx_last_error()
    ro, rl, rg = _attn_ref(qkv, d_out, B, S, H, d)
    assert (lse.double() - rl).abs().max().item() < 2e-3 * max(1.0, rl.abs().max().item())
    assert (out.double() - ro).abs().max().item() < 1.5e-2 * ro.abs().max().item()       # P and O rounded to bf16
    assert torch.isfinite(dqkv.float()).all()                                            # every element written
    err = (dqkv.double() - rg).norm().item() / rg.norm().item()
    assert err < 1.5e-2, err


@pytest.mark.parametrize("S", [64, 200])
def test_wide_attention_dropout_is_consistent(egx_lib, cuda, S):
    """p > 0: the backward regenerates the forward's mask. Checked by linearity: d/d(eps) out(v + eps * dv) == out_dv exactly
    matches the backward's dV contraction when the same mask is used (V enters linearly), and the keep rate is 1 - p."""
    B, H, d, p = 2, 4, 256, 0.25
    g = torch.Generator().manual_seed(11)
    delta = torch.empty(B, H, S, device=cuda)
    qkv = torch.randn(B * S, 3 * d, generator=g).to(cuda).bfloat16()
    out = torch.empty(B * S, d, device=cuda, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=cuda)
    egx_lib.egx_wide_attention_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, S, H, d, p, 1234, _stream())
    # v := 1 (so out = sum of dropped P per query = sum_k P ks): its mean over queries is ~1 (inverted dropout)
    q1 = qkv.clone()
    q1[:, 2 * d:] = 1.0
    egx_lib.egx_wide_attention_fwd(_ptr(q1), _ptr(out), _ptr(lse), B, S, H, d, p, 1234, _stream())
    rowsum = out.float()[:, ::d // H]              # one column per head: sum_k P_drop[q, k]
    assert abs(rowsum.mean().item() - 1.0) < 0.05 and rowsum.std().item() > 0.01
    # backward with dO := 1 and V := 1: dV[k] = sum_q P_drop[q, k]; summed over keys it equals sum_q rowsum[q] exactly-ish
    d_out = torch.ones(B * S, d, device=cuda, dtype=torch.bfloat16)
    dqkv = torch.empty(B * S, 3 * d, device=cuda, dtype=torch.bfloat16)
    egx_lib.egx_wide_attention_bwd(_ptr(q1), _ptr(out), _ptr(lse), _ptr(d_out), _ptr(dqkv), _ptr(delta), B, S, H, d, p, 1234, _stream())
    dv = dqkv.float()[:, 2 * d::d // H].view(B, S, H)
    rs = rowsum.view(B, S, H)
    assert torch.allclose(dv.sum(1), rs.sum(1), rtol=2e-2, atol=2e-2)


def _lta_cfg(n, d, heads, layers, nz=3, classes=(5, 7)):
    from types import SimpleNamespace as NS
    return NS(FORECASTING=NS(NUM_INPUT_CLIPS=n, NUM_ACTIONS_TO_PREDICT=nz),
              MODEL=NS(TRANSLATION_HEADS=heads, TRANSLATION_LAYERS=layers, TRANSLATION_INPUT_FEATURES=d, TRANSLATION_DROPOUT=0.0,
                       NUM_CLASSES=list(classes), DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))


def _grad_errs(model, sd64):
    errs = {}
    for k, p in model.named_parameters():
        r = sd64[k].grad
        if r is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            continue
        errs[k] = (p.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-12)
    return errs


@pytest.mark.parametrize("B", [2, 5])
def test_c4_real_dimensions_bf16_on_wide_path(egx_lib, cuda, B):
    """BASELINE.json configs[3] at its real sizes in its named precision (bf16): n = 32 clips per task -> S = 128, d = 768,
    8 heads of 96, 4 layers, 8192-wide PNR / OSCC features. Outputs within 1e-2 (north_star); gradients: the bf16 path
    rounds every GEMM operand to 8 bits, so per-parameter relative errors of a few 1e-2 are expected - 6e-2 bound."""
    from egot2_amd import hoi_lta, _lib
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(32, 768, 8, 4))
    sd = seeded_state_dict(m, 33)
    m.load_state_dict(sd)
    m = m.to(cuda).set_compute("bf16").train()
    feats = seeded_feats(34 + B, [(B, 32, 8192), (B, 32, 8192), (B, 32, 768), (B, 32, 2048)])
    outs = m.forward_features(*[f.to(cuda) for f in feats])
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    (lin(outs[0]) + lin(outs[1])).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.lta4_forward(sd64, 8, *[f.double() for f in feats], [5, 7])
    (lin(ref[0]) + lin(ref[1])).backward()
    for o, r in zip(outs, ref):
        assert (o.detach().cpu().double() - r.detach()).abs().max().item() < 1e-2 * max(1.0, r.abs().max().item())
    errs = _grad_errs(m, sd64)
    bad = {k: v for k, v in errs.items() if not v < 6e-2}
    assert not bad, bad


def test_wide_path_is_selected_and_generic_agrees(egx_lib, cuda):
    """impl = auto picks the wide kernels for bf16 at d = 256; forcing impl = generic (fp32-storage kernels, bf16 MFMA
    operands) must give the same outputs and gradients to bf16 noise; impl = wide on an unsupported shape raises."""
    from egot2_amd import hoi_lta, _lib
    from egot2_amd._lib import Config, Segment
    cfg = Config(256, 8, 2048, 2, 1, 1e-5, 1, 0, 0.0, 0.0, 0.0)
    segs = (Segment * 1)()
    segs[0].T, segs[0].d_in, segs[0].proj_w = 16, 2048, 1
    assert egx_lib.egx_encoder_impl(C.byref(cfg), segs, 4) == _lib.EGX_IMPL_WIDE
    cfg.compute = 0
    assert egx_lib.egx_encoder_impl(C.byref(cfg), segs, 4) == _lib.EGX_IMPL_GENERIC       # fp32 stays on the exact kernels
    cfg.compute, cfg.impl = 1, _lib.EGX_IMPL_WIDE
    segs[0].T = 200                                                                       # 128 < S <= 480: the online-softmax kernels
    assert egx_lib.egx_encoder_impl(C.byref(cfg), segs, 4) == _lib.EGX_IMPL_WIDE
    segs[0].T = 600
    assert egx_lib.egx_encoder_impl(C.byref(cfg), segs, 4) == -1 and b"wide" in egx_lib.egx_last_error()

    from egot2_amd import functional as F_egx
    for n in (4, 40):                                   # S = 16 and S = 160 (online-softmax attention)
        res = {}
        for impl in ("wide", "generic"):
            m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(n, 256, 8, 2))
            m.load_state_dict(seeded_state_dict(m, 5))
            m = m.to(cuda).set_compute("bf16", impl).train()
            feats = [f.to(cuda) for f in seeded_feats(6, [(3, n, 8192), (3, n, 8192), (3, n, 256), (3, n, 2048)])]
            o = m.forward_features(*feats)
            assert F_egx.last_encoder_impl() == impl
            (o[0].sum() + (o[1] * o[1]).sum()).backward()
            res[impl] = (torch.cat([t.detach().flatten() for t in o]), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        ow, og = res["wide"][0], res["generic"][0]
        assert (ow - og).abs().max().item() < 2e-2 * max(1.0, og.abs().max().item())
        for k, gg in res["generic"][1].items():
            gw = res["wide"][1][k]
            assert (gw - gg).norm().item() <= 8e-2 * gg.norm().item() + 1e-6, (n, k)


def test_wide_path_feeds_gradients_into_identity_segments(egx_lib, cuda):
    """The LTA translators' action stream comes from a SlowFast whose head is TRAINABLE (reference
    lta_models_lta_transfer.py:296-302, 357): it needs d(feature). On the wide path that gradient is the token-prep LayerNorm
    backward's fp32 input gradient; the encoder must stay on the wide kernels (round 2 silently fell back to the generic
    ones) and agree with them."""
    from egot2_amd import hoi_lta, functional as F_egx
    res = {}
    for impl in ("auto", "generic"):
        m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(4, 256, 8, 2))
        m.load_state_dict(seeded_state_dict(m, 5))
        m = m.to(cuda).set_compute("bf16", impl).train()
        feats = [f.to(cuda) for f in seeded_feats(6, [(3, 4, 8192), (3, 4, 8192), (3, 4, 256), (3, 4, 2048)])]
        feats[2].requires_grad_()
        egx_lib.egx_launch_count(1)
        o = m.forward_features(*feats)
        (o[0].sum() + (o[1] * o[1]).sum()).backward()
        res[impl] = (feats[2].grad.clone(), m.transformer.layers[0].linear1.weight.grad.clone())
    # the auto run used the wide kernels: its weight gradient is bit-identical to a run without the feature gradient
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(4, 256, 8, 2))
    m.load_state_dict(seeded_state_dict(m, 5))
    m = m.to(cuda).set_compute("bf16", "wide").train()
    feats = [f.to(cuda) for f in seeded_feats(6, [(3, 4, 8192), (3, 4, 8192), (3, 4, 256), (3, 4, 2048)])]
    o = m.forward_features(*feats)
    (o[0].sum() + (o[1] * o[1]).sum()).backward()
    assert torch.equal(res["auto"][1], m.transformer.layers[0].linear1.weight.grad)
    ga, gg = res["auto"][0], res["generic"][0]
    assert (ga - gg).norm().item() <= 8e-2 * gg.norm().item() + 1e-6


def test_wide_path_dropout_training_is_reproducible(egx_lib, cuda):
    """Train-mode dropout on the wide path: same seed -> bit-identical outputs and gradients (counter-based masks shared by
    forward and backward, fixed-order reductions everywhere: no atomics on this path); eval mode differs and has no noise."""
    from egot2_amd import hoi_lta
    cfg = _lta_cfg(4, 256, 8, 2)
    cfg.MODEL.TRANSLATION_DROPOUT = 0.3
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
    m.load_state_dict(seeded_state_dict(m, 9))
    m = m.to(cuda).set_compute("bf16", "wide").train()
    feats = [f.to(cuda) for f in seeded_feats(10, [(4, 4, 8192), (4, 4, 8192), (4, 4, 256), (4, 4, 2048)])]
    runs = []
    for _ in range(2):
        torch.manual_seed(77)
        m._egx_step = 0
        m.zero_grad(set_to_none=True)
        o = m.forward_features(*feats)
        (o[0].sum() + o[1].sum()).backward()
        runs.append((o[0].detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    for k in runs[0][1]:
        if k.startswith("head."):       # MultiTaskHead runs on the generic linear kernels (atomic bias-gradient sums)
            assert torch.allclose(runs[0][1][k], runs[1][1][k], rtol=1e-4, atol=1e-6), k
        else:
            assert torch.equal(runs[0][1][k], runs[1][1][k]), k
    m.eval()
    with torch.no_grad():
        e = m.forward_features(*feats)[0]
    assert not torch.equal(e, runs[0][0]) and torch.isfinite(e).all()


def test_feature_handoff_bf16_and_frame_pooling(egx_lib, cuda):
    """SURVEY.md 8f row F4. (1) Features handed over in bf16 are used in place by the projection GEMMs: results are
    BIT-IDENTICAL to fp32 features holding the same (bf16-representable) values, which go through the cast pass.
    (2) Per-frame PNR / OSCC features with the temporal mean fused into the hand-off equal the pooled-then-handed-over
    path (HOI encode_clips_pnr `.mean(dim=1)`) to bf16 rounding. (3) The fp32-storage kernels refuse packed features."""
    from egot2_amd import hoi_lta, _lib
    B, n, F = 3, 4, 16
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(n, 256, 8, 2))
    m.load_state_dict(seeded_state_dict(m, 15))
    m = m.to(cuda).set_compute("bf16").train()
    frames = [f.to(cuda) for f in seeded_feats(16, [(B, n * F, 8192), (B, n * F, 8192)])]
    act, lta = [f.to(cuda) for f in seeded_feats(17, [(B, n, 256), (B, n, 2048)])]
    pooled = [f.view(B, n, F, 8192).mean(2) for f in frames]

    def run(fn):
        m.zero_grad(set_to_none=True)
        o = fn()
        (o[0].sum() + (o[1] * o[1]).sum()).backward()
        return torch.cat([t.detach().flatten() for t in o]), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    # (1) bf16 in place == fp32 holding the same values
    p16 = [f.bfloat16() for f in pooled]
    l16 = lta.bfloat16()
    o_a, g_a = run(lambda: m.forward_features(p16[0], p16[1], act, l16))
    o_b, g_b = run(lambda: m.forward_features(p16[0].float(), p16[1].float(), act, l16.float()))
    assert torch.equal(o_a, o_b)
    for k in g_a:
        if not k.startswith("head."):
            assert torch.equal(g_a[k], g_b[k]), k
    # (2) fused temporal mean (fp32 and bf16 frames) vs pooled features
    o_ref, g_ref = run(lambda: m.forward_features(pooled[0], pooled[1], act, lta))
    for fr in (frames, [f.bfloat16() for f in frames]):
        o_f, g_f = run(lambda: m.forward_frame_features(fr[0], fr[1], act, lta, frames_per_clip=F))
        assert (o_f - o_ref).abs().max().item() < 1e-2 * max(1.0, o_ref.abs().max().item())
        errs = {k: (g_f[k] - g_ref[k]).norm().item() / (g_ref[k].norm().item() + 1e-12) for k in g_ref}
        bad = {k: v for k, v in errs.items() if not v < 5e-2}     # two different bf16 roundings of the projection operands: ~3e-2
        assert not bad, bad
    # (3) fp32 compute (generic kernels): no packed features
    m.set_compute("f32")
    with pytest.raises(_lib.EgxError, match="wide bf16 path"):
        m.forward_features(p16[0], p16[1], act, l16)


@pytest.mark.parametrize("frame_dtype", ["f32", "bf16"])
def test_frame_feature_handoff_against_oracle_and_reference_fixture(egx_lib, cuda, frame_dtype):
    """Row F4 against the ORACLE and the REFERENCE: per-frame PNR / OSCC features go in through forward_frame_features (the
    temporal mean of encode_clips_pnr fused into the projection operand cast, HOI/models/lta/lta_models_lta_transfer.py:
    335-345) and must match (a) the fp64 oracle run on `frames.mean(2)` — outputs 1e-2, every parameter gradient 6e-2 (the
    bf16 tolerances) — and (b) the lta4 fixture, which the reference's REAL forward(x_lta, x_pnr) produced from the same
    frames."""
    from tests.test_oracle_golden import build_ours, check_against_fixture, load_fixture, lta4_frames
    c, z = load_fixture("lta4_B3_n4_L2_d256")
    B, n, F = c["B"], c["n"], c["F"]
    model = build_ours(c)
    sd = seeded_state_dict(model, c["wseed"])
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute("bf16").train()
    frames, action, lta = lta4_frames(c)
    fr_pnr = frames.reshape(B, n * F, 8192).to(cuda)
    fr_oscc = frames.flip(-1).reshape(B, n * F, 8192).contiguous().to(cuda)
    if frame_dtype == "bf16":
        fr_pnr, fr_oscc = fr_pnr.bfloat16(), fr_oscc.bfloat16()
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    o = model.forward_frame_features(fr_pnr, fr_oscc, action.to(cuda), lta.to(cuda), frames_per_clip=F)
    loss = lin(o[0]) + lin(o[1])
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    # (b) the reference's own forward on the same frames
    check_against_fixture(z, {"out_verb": o[0], "out_noun": o[1]}, loss, grads, 1e-2, 6e-2)
    # (a) fp64 oracle on the pooled features (what encode_clips_pnr hands to the projections)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    f64 = frames.double()
    if frame_dtype == "bf16":
        f64 = frames.bfloat16().double()       # the oracle sees the values the producer actually handed over
    r = tr.lta4_forward(sd64, c["h"], f64.mean(2), f64.flip(-1).mean(2), action.double(), lta.double(), c["classes"])
    (lin(r[0]) + lin(r[1])).backward()
    for a, b in zip(o, r):
        assert (a.detach().cpu().double() - b.detach()).abs().max().item() < 1e-2 * max(1.0, b.abs().max().item())
    errs = {k: (g.detach().cpu().double() - sd64[k].grad).norm().item() / (sd64[k].grad.norm().item() + 1e-12)
            for k, g in grads.items() if sd64[k].grad is not None}
    assert len(errs) > 20
    bad = {k: v for k, v in errs.items() if not v < 6e-2}
    assert not bad, bad


def test_packed_identity_segment_is_refused(egx_lib, cuda):
    """A segment WITHOUT a projection feeds the shared LayerNorm directly in fp32: bf16 or frame-pooled features for it
    have no cast pass to ride on, and every implementation (wide included) must refuse them instead of reading bf16 bytes
    as fp32 (the action stream of the LTA translators is such a segment)."""
    from egot2_amd import _lib, hoi_lta
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(4, 256, 8, 2))
    m.load_state_dict(seeded_state_dict(m, 5))
    pnr, oscc, act, lta = [f.to(cuda) for f in seeded_feats(6, [(3, 4, 8192), (3, 4, 8192), (3, 4, 256), (3, 4, 2048)])]
    for impl in ("auto", "wide"):
        m = m.to(cuda).set_compute("bf16", impl).train()
        with pytest.raises(_lib.EgxError, match="need a projection"):
            m.forward_features(pnr, oscc, act.bfloat16(), lta)
        with pytest.raises((_lib.EgxError, AssertionError)):
            m._translate([pnr, oscc, act.repeat(1, 2, 1), lta], [m.proj_pnr, m.proj_oscc, None, m.proj_lta], pools=[1, 1, 2, 1])


@pytest.mark.parametrize("compute,tol_out,tol_grad", [("f32", 1e-3, 1e-2), ("bf16", 1e-2, 6e-2)])
def test_hoi_egot2g_encoder_real_dimensions(egx_lib, cuda, compute, tol_out, tol_grad):
    """BASELINE.json configs[4], HOI EgoT2-g at its real width: d = 512, 8 heads of 64, 3 layers
    (HOI/models/multitask/video_model_builder.py:70-77); both prompt layouts of encode() (pnr: 16 + 16 + 8 + 8 = 48 tokens with
    the SlowFast pathways projected separately; lta: per-clip PNR / OSCC frames + action + LTA features) against the oracle.
    bf16 runs on the wide path, f32 on the shape-generic kernels."""
    from tests.test_oracle_golden import build_ours, fixture_feats
    c = dict(kind="hoig", B=3, n=3, L=3, d=512, h=8, wseed=211, fseed=212)
    model = build_ours(c)
    sd = seeded_state_dict(model, c["wseed"])
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute(compute).train()
    model.pos_embed.dropout.p = 0.0
    feats = fixture_feats(c)
    fd = [f.to(cuda) for f in feats]
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    outs = [model.encode_features("pnr", *fd[:4]), model.encode_features("lta_verb", *fd[4:])]
    assert outs[0].shape == (48, 3, 512)
    (lin(outs[0]) + lin(outs[1])).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    f64 = [f.double() for f in feats]
    refs = [tr.hoi_g_encode(sd64, 8, "pnr", *f64[:4]), tr.hoi_g_encode(sd64, 8, "lta_verb", *f64[4:])]
    (lin(refs[0]) + lin(refs[1])).backward()
    for o, r in zip(outs, refs):
        assert (o.detach().cpu().double() - r.detach()).abs().max().item() < tol_out * max(1.0, r.abs().max().item())
    errs = {}
    for k, p in model.named_parameters():
        r = sd64[k].grad if k in sd64 else None
        if r is None or p.grad is None:
            continue
        errs[k] = (p.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-12)
    assert len(errs) > 30
    bad = {k: v for k, v in errs.items() if not v < tol_grad}
    assert not bad, bad


def test_c4_at_bench_batch_against_the_oracle_on_sampled_clips(egx_lib, cuda):
    """BASELINE.json configs[3] at the batch `bench.py --config c4` times (B = 256, S = 128, d = 768, 8 heads of
    96, 4 layers, 8192-wide features) UNDER A CHECKER. Clips are independent units, so the fp64 oracle is run on three sampled clips
    only: the wide path's outputs for those rows of the 256-clip launch must match it, and with a loss that reads only those clips'
    outputs every parameter gradient of the full-size launch (its grids, token splits and slab reductions are those of B = 256; the
    other 253 clips contribute exact zeros) must equal the oracle's gradient on the three clips."""
    from egot2_amd import hoi_lta
    B, pick = 256, [0, 101, 255]
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(_lta_cfg(32, 768, 8, 4))
    sd = seeded_state_dict(m, 33)
    m.load_state_dict(sd)
    m = m.to(cuda).set_compute("bf16").train()
    feats = seeded_feats(901, [(B, 32, 8192), (B, 32, 8192), (B, 32, 768), (B, 32, 2048)])
    outs = m.forward_features(*[f.to(cuda) for f in feats])
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    idx = torch.tensor(pick, device=cuda)
    (lin(outs[0][idx]) + lin(outs[1][idx])).backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = tr.lta4_forward(sd64, 8, *[f[pick].double() for f in feats], [5, 7])
    (lin(ref[0]) + lin(ref[1])).backward()
    for o, r in zip(outs, ref):
        assert (o[idx].detach().cpu().double() - r.detach()).abs().max().item() < 1e-2 * max(1.0, r.abs().max().item())
    errs = _grad_errs(m, sd64)
    bad = {k: v for k, v in errs.items() if not v < 6e-2}
    assert len(errs) > 40 and not bad, bad


def test_c5_hoi_encoder_at_bench_batch_against_the_oracle_on_sampled_clips(egx_lib, cuda):
    """The same full-size check for BASELINE.json configs[4] (HOI EgoT2-g encoder, d = 512, 8 heads of 64, 3 layers, 48 tokens) at
    B = 256 in bf16: memory rows and parameter gradients of the 256-clip launch against the fp64 oracle on three sampled clips."""
    from tests.test_oracle_golden import build_ours
    B, pick = 256, [3, 128, 254]
    c = dict(kind="hoig", B=B, n=3, L=3, d=512, h=8, wseed=211, fseed=212)
    model = build_ours(c)
    sd = seeded_state_dict(model, c["wseed"])
    model.load_state_dict(sd)
    model = model.to(cuda).set_compute("bf16").train()
    model.pos_embed.dropout.p = 0.0
    feats = seeded_feats(902, [(B, 16, 8192), (B, 16, 8192), (B, 8, 2048), (B, 8, 256)])
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    out = model.encode_features("pnr", *[f.to(cuda) for f in feats])            # (48, B, 512)
    assert out.shape == (48, B, 512)
    idx = torch.tensor(pick, device=cuda)
    lin(out[:, idx].contiguous()).backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    ref = tr.hoi_g_encode(sd64, 8, "pnr", *[f[pick].double() for f in feats])
    lin(ref).backward()
    assert (out[:, idx].detach().cpu().double() - ref.detach()).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
    errs = {}
    for k, p in model.named_parameters():
        r = sd64[k].grad if k in sd64 else None
        if r is not None and p.grad is not None:
            errs[k] = (p.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-12)
    bad = {k: v for k, v in errs.items() if not v < 6e-2}
    assert len(errs) > 30 and not bad, bad
