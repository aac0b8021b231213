"""CPU: pins the oracle's explicit-mask arguments (oracle/translator_ref.py `masks`) to the TRAIN-mode arithmetic of the stock
torch.nn modules the reference instantiates, and tests/dropmask.py's generator against hand-computed words.

The stock module (oracle/stock_module.py = the reference class minus backbones; the live reference class when
/root/reference is present) runs in .train() with torch.nn.functional.dropout replaced by a function that hands out the
masks in call order — PositionalEncoding per segment, then per layer attention probabilities / dropout1 / FFN hidden /
dropout2 — so every site, its position in the arithmetic and its layout (the reference is sequence-first) is checked
against what torch itself does, not against our reading of it."""
import contextlib
import math

import numpy as np
import pytest
import torch

from oracle import translator_ref as tr
from oracle.stock_module import StockTTMTranslator
from tests import dropmask as dm
from tests.util import seeded_feats, seeded_state_dict


@contextlib.contextmanager
def explicit_dropout(queue):
    """Patch F.dropout (nn.Dropout, and the attention probabilities) to pop masks from `queue` = [(shape, fn(x) -> mask)]."""
    import torch.nn.functional as F
    real_dropout, real_sdpa = F.dropout, F.scaled_dot_product_attention

    def fake_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        name, get = queue.pop(0)
        m = get(x)
        assert m.shape == x.shape, f"{name}: mask {tuple(m.shape)} vs activation {tuple(x.shape)}"
        return x * m.to(x.dtype)

    def fake_sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        assert attn_mask is None and not is_causal
        s = (q @ k.transpose(-1, -2)) * (scale if scale is not None else 1.0 / math.sqrt(q.shape[-1]))
        return fake_dropout(torch.softmax(s, dim=-1), dropout_p, dropout_p > 0.0) @ v

    F.dropout, F.scaled_dot_product_attention = fake_dropout, fake_sdpa
    try:
        yield
    finally:
        F.dropout, F.scaled_dot_product_attention = real_dropout, real_sdpa


def _queue_for_ttm(masks, seg_T, B, H, L):
    """Call order of the seq-first reference forward."""
    q = []
    off = 0
    for T in seg_T:
        pm = masks["pos"][:, off:off + T]
        q.append(("pos", lambda x, pm=pm: pm.permute(1, 0, 2)))               # PositionalEncoding sees (T, B, d)
        off += T
    S = sum(seg_T)
    for layer in range(L):
        lm = masks["layers"][layer]

        def attn(x, lm=lm):
            a = lm["attn"]                                                     # (B, H, S, S)
            return a if x.dim() == 4 else a.reshape(B * H, S, S)               # need_weights path: (B*H, S, S)
        q.append(("attn", attn))
        q.append(("res1", lambda x, lm=lm: lm["res1"].permute(1, 0, 2)))      # (S, B, d)
        q.append(("ffn", lambda x, lm=lm: lm["ffn"].permute(1, 0, 2)))
        q.append(("res2", lambda x, lm=lm: lm["res2"].permute(1, 0, 2)))
    return q


@pytest.mark.parametrize("L,B,T", [(1, 5, 7), (2, 3, 4)])
def test_oracle_masks_are_torchs_train_mode_sites(L, B, T):
    d, H, d_ff, p, p_pos = 128, 4, 2048, 0.5, 0.1
    torch.manual_seed(0)
    m = StockTTMTranslator(3, d, H, p, L).double().train()
    sd = {k: v.double() for k, v in seeded_state_dict(m, 21).items()}
    m.load_state_dict(sd)
    feats = [f.double() for f in seeded_feats(3, [(B, T, 256)] * 3)]
    target = torch.arange(B) % 2
    masks = dm.encoder_masks(0xC0FFEE, "fused", B, [T] * 3, d, H, d_ff, L, p, p_pos)
    queue = _queue_for_ttm(masks, [T] * 3, B, H, L)
    with explicit_dropout(queue):
        logits = m(*feats)
        loss = tr.weighted_ce(logits, target, [0.266, 0.734])
        loss.backward()
    assert not queue, f"unused masks: {[n for n, _ in queue]}"
    sd64 = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    ref = tr.ttm_forward(sd64, H, *feats, masks=masks)
    tr.weighted_ce(ref, target, [0.266, 0.734]).backward()
    assert (logits - ref).abs().max().item() < 1e-10
    # the masks do something: eval-mode logits differ
    assert (tr.ttm_forward(sd64, H, *feats) - ref).abs().max().item() > 1e-3
    for k, prm in m.named_parameters():
        g = sd64[k].grad
        assert (prm.grad - g).abs().max().item() < 1e-9 * (1 + g.abs().max().item()), k


@pytest.mark.reference
def test_oracle_masks_against_the_live_reference_class():
    """Same check with the REAL reference class (imported with import-time stubs only) instead of the stock restatement."""
    from oracle import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("/root/reference not present")
    B, T, L, d, H, p = 4, 6, 2, 128, 4, 0.5
    model = rh.ref_ttm(3, rh.hhi_args(hidden_dim=d, num_heads=H, dropout=p, num_layers=L)).double().train()
    sd = {k: v.double() for k, v in seeded_state_dict(model, 5).items()}
    model.load_state_dict(sd)
    feats = [f.double() for f in seeded_feats(8, [(B, T, 256)] * 3)]
    masks = dm.encoder_masks(77, "fused", B, [T] * 3, d, H, 2048, L, p, 0.1)
    queue = _queue_for_ttm(masks, [T] * 3, B, H, L)
    with explicit_dropout(queue):
        logits = rh.ref_ttm_forward(model, *feats)
    assert not queue
    ref = tr.ttm_forward(sd, H, *feats, masks=masks)
    assert (logits - ref).abs().max().item() < 1e-10


def test_generator_known_answers():
    """site_key / rand_quad against values worked out by hand with Python integers (no numpy)."""
    def rq(key, row, cq):
        k0, k1 = key & 0xFFFFFFFF, key >> 32
        x = ((row * 0x9E3779B1 + k1) & 0xFFFFFFFF) ^ ((cq * 0x85EBCA77 + k0) & 0xFFFFFFFF)
        x ^= x >> 16
        p = x * 0x7FEB352D
        y = (p & 0xFFFFFFFF) ^ (p >> 32)
        y ^= y >> 15
        z = (y * 0x846CA68B) & 0xFFFFFFFF
        z ^= z >> 16
        return z, y
    key = dm.site_key(1234, 1, dm.SITE_FFN)
    assert key & 1 and key < 2 ** 64
    rows = np.array([0, 1, 77, 16383], dtype=np.int64)
    cqs = np.array([0, 3, 511], dtype=np.int64)
    z, y = dm.rand_quad(key, rows[:, None], cqs[None, :])
    for i, r in enumerate(rows):
        for j, c in enumerate(cqs):
            assert (int(z[i, j]), int(y[i, j])) == rq(key, int(r), int(c))
    # keep rate and scale
    ks = dm.keep_scale(key, np.arange(4096), np.arange(512), 0.5)
    assert abs((ks > 0).double().mean().item() - 0.5) < 5e-3 and ks.max().item() == 2.0
    assert dm.drop_threshold(0.1) == 6554 and dm.drop_threshold(0.5) == 32768 and dm.drop_threshold(0.0) == 0
    assert dm.lcg(0) == 1442695040888963407
