"""-m gpu: the decoder-specific kernels of row F1 (csrc/decoder.hip) through their autograd Functions, against plain
torch fp64 math: few-query attention (causal self / cross), embedding + positional encoding, ReLU-fused linear,
LayerNorm(x + res), and the counter-based dropout of the decoder (same masks in forward and backward)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_attention(q, k, v, H, causal):
    B, Sq, d = q.shape
    Sk = k.shape[1]
    dh = d // H
    qh = q.reshape(B, Sq, H, dh).permute(0, 2, 1, 3)
    kh = k.reshape(B, Sk, H, dh).permute(0, 2, 1, 3)
    vh = v.reshape(B, Sk, H, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(Sq, Sk, dtype=torch.bool)), float("-inf"))
    return (torch.softmax(s, dim=-1) @ vh).permute(0, 2, 1, 3).reshape(B, Sq, d)


@pytest.mark.parametrize("B,sy,H,dh", [(3, 1, 4, 64), (5, 2, 4, 64), (2, 5, 8, 64), (4, 8, 2, 128), (7, 3, 8, 16)])
def test_self_attention_small_matches_torch(egx_lib, cuda, B, sy, H, dh):
    from egot2_amd import functional as F_egx
    d = H * dh
    g = torch.Generator().manual_seed(B * 100 + sy)
    qkv = torch.randn(B * sy, 3 * d, generator=g)
    w = torch.randn(B * sy, d, generator=g)
    x = qkv.to(cuda).requires_grad_(True)
    out = F_egx.SelfAttnSmallFn.apply(x, B, sy, H, True, 0.0, 0, 1)
    (out * w.to(cuda)).sum().backward()
    r = qkv.double().requires_grad_(True)
    r3 = r.view(B, sy, 3 * d)
    ref = _ref_attention(r3[..., :d], r3[..., d:2 * d], r3[..., 2 * d:], H, True).reshape(B * sy, d)
    (ref * w.double()).sum().backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-5
    assert (x.grad.cpu().double() - r.grad).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,sy,S,H,dh", [(4, 2, 45, 4, 64), (6, 4, 12, 8, 64), (30, 2, 3, 4, 64), (2, 1, 64, 8, 32), (3, 8, 48, 4, 128),
                                         (3, 2, 65, 4, 64), (5, 2, 450, 4, 64), (2, 8, 180, 8, 32), (2, 5, 1000, 2, 128), (9, 3, 128, 4, 64)])
def test_cross_attention_small_matches_torch(egx_lib, cuda, B, sy, S, H, dh):
    from egot2_amd import functional as F_egx
    d = H * dh
    g = torch.Generator().manual_seed(B + 17 * S)
    q = torch.randn(B * sy, d, generator=g)
    kv = torch.randn(B * S, 2 * d, generator=g)
    w = torch.randn(B * sy, d, generator=g)
    qd, kvd = q.to(cuda).requires_grad_(True), kv.to(cuda).requires_grad_(True)
    out = F_egx.CrossAttnSmallFn.apply(qd, kvd, B, sy, S, H, 0.0, 0, 3)
    (out * w.to(cuda)).sum().backward()
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    kv3 = kvr.view(B, S, 2 * d)
    ref = _ref_attention(qr.view(B, sy, d), kv3[..., :d], kv3[..., d:], H, False).reshape(B * sy, d)
    (ref * w.double()).sum().backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-5
    assert (qd.grad.cpu().double() - qr.grad).abs().max().item() < 1e-5
    assert (kvd.grad.cpu().double() - kvr.grad).abs().max().item() < 1e-5


@pytest.mark.parametrize("S", [20, 200])
def test_attention_dropout_masks_match_between_forward_and_backward(egx_lib, cuda, S):
    """With a fixed (seed, site) the output is a deterministic, piecewise-smooth function of q: central differences must
    reproduce the analytic gradient, which they only do if the backward regenerates the forward's mask."""
    from egot2_amd import functional as F_egx
    B, sy, H, dh = 3, 2, 4, 32
    d = H * dh
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * sy, d, generator=g).to(cuda)
    kv = torch.randn(B * S, 2 * d, generator=g).to(cuda)
    w = torch.randn(B * sy, d, generator=g).to(cuda)
    f = lambda qq, seed: (F_egx.CrossAttnSmallFn.apply(qq, kv, B, sy, S, H, 0.5, seed, 3) * w).sum()  # noqa: E731
    assert f(q, 11).item() == f(q, 11).item()
    assert abs(f(q, 11).item() - f(q, 12).item()) > 1e-6
    keep = (F_egx.CrossAttnSmallFn.apply(q, torch.cat((kv[:, :d], torch.ones_like(kv[:, d:])), 1), B, sy, S, H, 0.5, 11, 3)).mean().item()
    assert 0.8 < keep < 1.2            # V = 1: each output = sum of kept probabilities / (1 - p), expectation 1
    qg = q.clone().requires_grad_(True)
    f(qg, 11).backward()
    for idx in [(0, 3), (2, 77), (5, 120)]:
        eps = 1e-2
        qp, qm = q.clone(), q.clone()
        qp[idx] += eps
        qm[idx] -= eps
        fd = (f(qp, 11).item() - f(qm, 11).item()) / (2 * eps)
        assert abs(fd - qg.grad[idx].item()) < 2e-2 * max(abs(fd), 1e-2), (idx, fd, qg.grad[idx].item())


def test_embed_pos_and_scatter_gradient(egx_lib, cuda):
    from egot2_amd import functional as F_egx
    V, d, B, sy = 9, 64, 6, 3
    g = torch.Generator().manual_seed(2)
    emb = torch.randn(V, d, generator=g)
    pe = torch.randn(10, 1, d, generator=g)
    tok = torch.randint(0, V, (B, sy), generator=g)
    tok[0, 0] = tok[1, 1] = tok[2, 2] = 4                      # repeated token: the scatter must accumulate
    w = torch.randn(B * sy, d, generator=g)
    e = emb.to(cuda).requires_grad_(True)
    out = F_egx.EmbedPosFn.apply(tok.to(cuda), e, pe.to(cuda)[:, 0, :], math.sqrt(d), 0.0, 0)
    (out * w.to(cuda)).sum().backward()
    er = emb.double().requires_grad_(True)
    ref = (er[tok] * math.sqrt(d) + pe.double()[:sy, 0, :]).reshape(B * sy, d)
    (ref * w.double()).sum().backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-5
    assert (e.grad.cpu().double() - er.grad).abs().max().item() < 1e-4


def test_relu_linear_and_layernorm_residual(egx_lib, cuda):
    from egot2_amd import functional as F_egx
    g = torch.Generator().manual_seed(3)
    x, res = torch.randn(10, 128, generator=g), torch.randn(10, 128, generator=g)
    W, b = torch.randn(256, 128, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    lw, lb = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    w2 = torch.randn(10, 256, generator=g)
    dev = [t.to(cuda).requires_grad_(True) for t in (x, res, W, b, lw, lb)]
    y = F_egx.layer_norm_residual(dev[0], dev[1], dev[4], dev[5], 1e-5)
    h = F_egx.linear(y, dev[2], dev[3], "f32", relu=True)
    (h * w2.to(cuda)).sum().backward()
    ref = [t.double().requires_grad_(True) for t in (x, res, W, b, lw, lb)]
    yr = torch.nn.functional.layer_norm(ref[0] + ref[1], (128,), ref[4], ref[5], 1e-5)
    hr = torch.relu(yr @ ref[2].T + ref[3])
    (hr * w2.double()).sum().backward()
    assert (h.detach().cpu().double() - hr.detach()).abs().max().item() < 1e-4
    for a, r in zip(dev, ref):
        assert (a.grad.cpu().double() - r.grad).abs().max().item() < 1e-3 * max(1.0, r.grad.abs().max().item())


@pytest.mark.parametrize("S,compute", [(45, "f32"), (45, "bf16"), (180, "bf16"), (180, "f32")])
def test_decoder_trains_with_dropout(egx_lib, cuda, S, compute):
    """Train-mode decode (dropout on every site): finite, deterministic per seed, and gradients reach every decoder
    parameter; eval-mode decode is independent of the seed. bf16 = the fused decoder (S = 180: its long-memory cross-attention
    kernel with dropout on the probabilities), f32 = the composed one."""
    from types import SimpleNamespace as NS
    from egot2_amd import hhi_multitask
    from tests.util import seeded_state_dict
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    args = NS(hidden_dim=256, num_heads=4, num_layers=2, dropout=0.3, lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None)
    m = hhi_multitask.TaskTranslationPromptTransformer(args, vocab)
    m.load_state_dict(seeded_state_dict(m, 4))
    m = m.to(cuda).set_compute(compute).train()
    m._egx_seed = lambda: 99
    mem = torch.randn(S, 5, 256, device=cuda)
    y = torch.randint(0, 7, (5, 2), device=cuda)
    out = m.decode(y, mem)
    assert out.shape == (2, 5, 7) and torch.isfinite(out).all()
    assert torch.equal(out, m.decode(y, mem)), "same seed, same masks"
    m._egx_seed = lambda: 100
    assert not torch.equal(out, m.decode(y, mem)), "another seed draws other masks"
    out.square().sum().backward()
    missing = [n for n, p in m.named_parameters() if ("transformer_decoder" in n or n.startswith(("fc.", "embedding."))) and p.grad is None]
    assert not missing, missing
    m.eval()
    with torch.no_grad():
        a, b = m.decode(y, mem), m.decode(y, mem)
    assert torch.equal(a, b)


@pytest.mark.parametrize("task,T", [("ttm", 60), ("ttm", 150), ("asd", 50)])
def test_egot2g_hhi_long_sequences_encode_and_decode_match_the_oracle(egx_lib, cuda, task, T):
    """EgoT2-g HHI on real-length TTM / ASD inputs (3 x T tokens of memory, up to 450): the encoder runs the wide path's
    online-softmax attention (S > 128), the decoder's cross-attention the chunked long-memory kernel (S > 64; the fused decoder
    stops at 64 memory tokens). bf16 model against the fp64 oracle: memory, vocabulary logits and parameter gradients."""
    from types import SimpleNamespace as NS
    from egot2_amd import functional as F_egx, hhi_multitask
    from oracle import translator_ref as tr
    from tests.util import seeded_feats, seeded_state_dict
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    args = NS(hidden_dim=256, num_heads=4, num_layers=2, dropout=0.0, lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None)
    m = hhi_multitask.TaskTranslationPromptTransformer(args, vocab)
    sd = seeded_state_dict(m, 14)
    m.load_state_dict(sd)
    m.pos_embed.dropout.p = 0.0          # (the reference's PositionalEncoding keeps its default p = 0.1 whatever --dropout says)
    m = m.to(cuda).set_compute("bf16").train()
    B = 3
    feats = seeded_feats(15 + T, [(B, T, 256)] * 3)
    mem = m.encode_features(task, *[f.to(cuda) for f in feats])
    assert F_egx.last_encoder_impl() == "wide"
    nb = mem.shape[1]
    y = torch.stack([torch.full((nb,), vocab[task]), torch.randint(5, 7, (nb,), generator=torch.Generator().manual_seed(T))], dim=1)
    logits = m.decode(y.to(cuda), mem)
    lin = lambda t: (t * torch.linspace(-1, 1, t.numel(), device=t.device, dtype=t.dtype).view_as(t)).sum()  # noqa: E731
    lin(logits).backward()
    torch.cuda.synchronize()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    rmem = tr.hhi_g_encode(sd64, 4, task, *[f.double() for f in feats])
    rlog = tr.g_decode(sd64, 4, y, rmem)
    lin(rlog).backward()
    assert (mem.detach().cpu().double() - rmem.detach()).abs().max().item() < 4e-2 * max(1.0, rmem.detach().abs().max().item())
    assert (logits.detach().cpu().double() - rlog.detach()).abs().max().item() < 4e-2 * max(1.0, rlog.detach().abs().max().item())
    named = dict(m.named_parameters())
    errs = {k: ((named[k].grad.cpu().double() - v.grad).norm() / (v.grad.norm() + 1e-12)).item() for k, v in sd64.items()
            if v.grad is not None and k in named and v.grad.norm() > 0}
    bad = {k: e for k, e in errs.items() if not e < 8e-2}
    assert len(errs) > 20 and not bad, bad


@pytest.mark.parametrize("S,sy,heads", [(45, 2, 4), (180, 2, 4), (450, 3, 4), (200, 5, 8), (1024, 2, 4)])
def test_fused_decoder_matches_the_composed_decoder(egx_lib, cuda, S, sy, heads):
    """egx_decoder_fwd / _bwd (one call per direction, bf16 GEMMs; memories beyond 64 tokens: dec_attn_long_kernel) against the
    decoder composed from the unit operators (fp32 target-side GEMMs, egx_small_attention_*) at p = 0: logits, d(memory) and
    every parameter gradient."""
    from types import SimpleNamespace as NS
    from egot2_amd import hhi_multitask
    from tests.util import seeded_state_dict
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    args = NS(hidden_dim=256, num_heads=heads, num_layers=2, dropout=0.0, lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None)
    B = 4
    g = torch.Generator().manual_seed(S + sy)
    mem0 = torch.randn(S, B, 256, generator=g)
    y = torch.randint(0, 7, (B, sy), generator=g).to(cuda)
    w = torch.randn(sy, B, 7, generator=g).to(cuda)
    res = {}
    for mode in ("fused", "composed"):
        m = hhi_multitask.TaskTranslationPromptTransformer(args, vocab)
        m.load_state_dict(seeded_state_dict(m, 4))
        m.pos_embed.dropout.p = 0.0
        m = m.to(cuda).set_compute("bf16").train()
        m.egx_composed_decoder = mode == "composed"
        mem = mem0.to(cuda).requires_grad_(True)
        out = m.decode(y, mem)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (out.detach().double(), mem.grad.double(), {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None})
    (of, gf, pf), (oc, gc, pc) = res["fused"], res["composed"]
    assert (of - oc).abs().max().item() < 3e-2 * max(1.0, oc.abs().max().item())
    assert (gf - gc).norm().item() < 6e-2 * gc.norm().item()          # d(memory) passes through bf16 d(kv) rows on the fused path
    assert set(pf) == set(pc) and len(pf) > 30
    bad = {k: ((pf[k] - pc[k]).norm() / (pc[k].norm() + 1e-12)).item() for k in pc if pc[k].norm() > 0}
    bad = {k: e for k, e in bad.items() if not e < 8e-2}
    assert not bad, bad


def test_fused_decoder_under_graph_capture_matches_eager(egx_lib, cuda):
    """The fused decoder forks its K | V projections and weight gradients onto a library-owned side stream and joins them
    back by events; captured into a hipGraph (what bench.py replays) the side stream becomes a parallel branch. Replays must
    reproduce the eager results: logits, d(memory) and every parameter gradient (eval mode: no dropout seed in play)."""
    from types import SimpleNamespace as NS
    from egot2_amd import hhi_multitask
    from tests.util import seeded_state_dict
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    args = NS(hidden_dim=256, num_heads=4, num_layers=2, dropout=0.0, lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None)
    m = hhi_multitask.TaskTranslationPromptTransformer(args, vocab)
    m.load_state_dict(seeded_state_dict(m, 9))
    m = m.to(cuda).eval()
    m.set_compute("bf16")
    from egot2_amd import functional as F_egx
    dec = m.transformer_decoder
    assert F_egx.decoder_supported("bf16", 256, 4, dec.layers[0].linear1.out_features, 2, 45, len(dec.layers)), "the fused decoder must be the path under test"
    mem = torch.randn(45, 6, 256, device=cuda, requires_grad=True)
    y = torch.randint(0, 7, (6, 2), device=cuda)
    params = [p for n, p in m.named_parameters() if "transformer_decoder" in n or n.startswith(("fc.", "embedding."))]

    def step():
        for p in params:
            p.grad = None
        mem.grad = None
        out = m.decode(y, mem)
        out.square().sum().backward()
        return out

    ref_out = step().detach().clone()
    ref = [p.grad.detach().clone() for p in params] + [mem.grad.detach().clone()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()                               # warm-up on the capture stream (workspaces, side stream creation)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
        static = [p.grad for p in params] + [mem.grad]
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(out, ref_out, rtol=0, atol=1e-5 * ref_out.abs().max().item() + 1e-6)
    for a, r in zip(static, ref):
        assert torch.allclose(a, r, rtol=0, atol=2e-3 * r.abs().max().item() + 1e-6), (a - r).abs().max().item()


def test_fused_decoder_training_dropout_under_capture_needs_the_device_seed(egx_lib, cuda):
    """a host seed baked into a captured graph would repeat the same masks on every replay, so training-mode
    dropout under capture is refused - unless the model keeps its seed in device memory (enable_device_seed): then the
    decoder derives its keys on the stream and every replay draws fresh masks."""
    from types import SimpleNamespace as NS
    from egot2_amd import hhi_multitask, _lib
    from tests.util import seeded_state_dict
    vocab = {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    args = NS(hidden_dim=256, num_heads=4, num_layers=2, dropout=0.3, lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None)
    m = hhi_multitask.TaskTranslationPromptTransformer(args, vocab)
    m.load_state_dict(seeded_state_dict(m, 9))
    m = m.to(cuda).train()
    m.set_compute("bf16")
    mem = torch.randn(45, 6, 256, device=cuda)
    y = torch.randint(0, 7, (6, 2), device=cuda)

    def fwd():
        return m.decode(y, mem)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception, match="hipGraph|captur"):
        with torch.cuda.graph(g):
            fwd()
    torch.cuda.synchronize()
    m.enable_device_seed()
    with torch.cuda.stream(side):
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    m._egx_seed_dev.fill_(5)
    g.replay()
    torch.cuda.synchronize()
    a = out.detach().clone()
    m._egx_seed_dev.fill_(6)            # (the encoder's forward advances the seed in a real step)
    g.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and (a - out).abs().max().item() > 0
    m._egx_seed_dev.fill_(5)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, out)
