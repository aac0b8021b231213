"""ORACLE — test infrastructure only. Never imported by egot2_amd (the product path).

CPU restatement of the reference's Stage-II translator arithmetic, written out op by op in plain torch tensor
math (no nn.Transformer*, no nn.MultiheadAttention, no F.layer_norm), batch-first, dtype-generic (fp32 to mirror
the reference, fp64 as a tight yardstick). Gradients come from torch.autograd over this restatement.

Parity status: PINNED by generated fixtures, not by reference tests (the reference has none).
tests/golden/make_golden.py imports the real reference classes from /root/reference (import-time stubs only) and
records inputs / logits / loss / gradient digests; tests/test_oracle_golden.py checks this file against them.

Follows (paths relative to the reference checkout):
  token preparation      HHI/models/ttm/model_taskspecific.py:131-151 (PositionalEncoding), :222-226 (encode_prepare)
  token packing          :238-241 (order ttm, lam, asd); HHI/models/asd/model_taskspecific.py:151-154 (asd, ttm, lam)
  encoder layer          torch.nn.TransformerEncoderLayer as constructed at :212-215 (post-LN, ReLU, d_ff=2048,
                         eps=1e-5): x = LN1(x + MHA(x)); x = LN2(x + W2 relu(W1 x))
  TTM head               :243-244 (mean over tokens, LayerNorm, Linear(d, 2))
  ASD output             HHI/models/asd/model_taskspecific.py:155-157 (first T tokens of every clip)
  EgoT2-g encoder        HHI/models/multitask/task_prompt_model.py:224-258
  HOI translators        HOI/models/lta/lta_models_lta_transfer.py:355-363, HOI/models/pnr/video_model_transfer_3task.py:249-257
  LTA head               HOI/models/lta/head_helper.py:261-290 (training branch: Linear per future action)
  PNR / OSCC head        HOI/models/pnr/head_helper.py:353-381 (AvgPool3d stride 1 -> permute -> (N, T', 8192) rows -> Linear);
                         the per-clip `.mean(dim=1)` of HOI/models/lta/lta_models_lta_transfer.py:335-345
  loss                   HHI/tasks/ttm/video_task_2loader.py:21-22,34 (CrossEntropyLoss(weight=[0.266, 0.734]))
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)  # biased, as F.layer_norm
    return (x - mean) / torch.sqrt(var + eps) * w + b


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def sinusoid_table(max_len: int, d: int, dtype=torch.float32) -> torch.Tensor:
    """(max_len, d) — same values as the reference's `pe` buffer (computed in fp32 like the reference, then cast)."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def self_attention(x: torch.Tensor, in_w, in_b, out_w, out_b, n_heads: int, attn_mask=None) -> torch.Tensor:
    """x: (B, S, d). Packed in-projection rows = [Wq; Wk; Wv]; per-head softmax(QK^T / sqrt(d_h)) V; out-projection.
    attn_mask (B, H, S, S): keep-scale of the dropout on the probabilities (F.multi_head_attention_forward: dropout(softmax))."""
    B, S, d = x.shape
    dh = d // n_heads
    qkv = linear(x, in_w, in_b)  # (B, S, 3d)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]

    def heads(t):
        return t.reshape(B, S, n_heads, dh).permute(0, 2, 1, 3)  # (B, H, S, dh)

    q, k, v = heads(q), heads(k), heads(v)
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    scores = scores - scores.max(dim=-1, keepdim=True).values
    p = torch.exp(scores)
    p = p / p.sum(dim=-1, keepdim=True)
    if attn_mask is not None:
        p = p * attn_mask
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    return linear(o, out_w, out_b)


def _m(x: torch.Tensor, masks, name: str) -> torch.Tensor:
    """Apply an explicit dropout keep-scale (0 or 1 / (1 - p) per element) when the caller supplies one."""
    if masks is None or masks.get(name) is None:
        return x
    return x * masks[name].to(x.dtype)


def encoder_layer(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, n_heads: int, eps: float = 1e-5,
                  masks=None) -> torch.Tensor:
    """`masks` (train mode with EXPLICIT masks; None = eval): dict of keep-scales at nn.TransformerEncoderLayer's four dropout
    sites — "attn" (B, H, S, S) on the softmax output, "res1" (B, S, d) = dropout1 on the attention branch, "ffn" (B, S, d_ff)
    = dropout on the activated hidden, "res2" (B, S, d) = dropout2 on the FFN branch (torch 1.12 transformer.py
    _sa_block / _ff_block as constructed at HHI/models/ttm/model_taskspecific.py:211-215)."""
    g = lambda k: sd[prefix + k]  # noqa: E731
    a = self_attention(x, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"),
                       g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), n_heads,
                       None if masks is None else masks.get("attn"))
    x = layer_norm(x + _m(a, masks, "res1"), g("norm1.weight"), g("norm1.bias"), eps)
    h = _m(torch.relu(linear(x, g("linear1.weight"), g("linear1.bias"))), masks, "ffn")
    f = linear(h, g("linear2.weight"), g("linear2.bias"))
    return layer_norm(x + _m(f, masks, "res2"), g("norm2.weight"), g("norm2.bias"), eps)


def encoder(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, n_layers: int, n_heads: int, masks=None) -> torch.Tensor:
    """masks: None, or {"layers": [per-layer dict for encoder_layer, ...]}."""
    for i in range(n_layers):
        x = encoder_layer(x, sd, f"{prefix}layers.{i}.", n_heads, masks=None if masks is None else masks["layers"][i])
    return x


def n_layers_of(sd: Dict[str, torch.Tensor], prefix: str) -> int:
    n = 0
    while f"{prefix}layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def encode_prepare(feat: torch.Tensor, proj_w, proj_b, ln_w, ln_b, task_vec, pe_rows, pos_mask=None) -> torch.Tensor:
    """(B, T, d_in) -> (B, T, d): LN_shared(proj(feat)) + task_embed[k] + pe[0:T] (position restarts per task).
    pos_mask (B, T, d): keep-scale of PositionalEncoding's dropout, applied to the SUM (reference :149-151)."""
    x = feat if proj_w is None else linear(feat, proj_w, proj_b)
    x = layer_norm(x, ln_w, ln_b)
    if task_vec is not None:
        x = x + task_vec
    if pe_rows is not None:
        x = x + pe_rows
    if pos_mask is not None:
        x = x * pos_mask.to(x.dtype)
    return x


# ---- HHI: TTM translators ------------------------------------------------------------------------
def hhi_tokens(sd, feats: Sequence[torch.Tensor], names: Sequence[str], task_ids: Sequence[int], masks=None) -> torch.Tensor:
    """masks["pos"] (B, S, d) in PACKED token order: segment k takes its own T_k rows."""
    pe = sd["pos_embed.pe"][:, 0, :]
    xs = []
    off = 0
    for f, n, k in zip(feats, names, task_ids):
        T = f.shape[1]
        pm = None if masks is None or masks.get("pos") is None else masks["pos"][:, off:off + T]
        xs.append(encode_prepare(f, sd[f"proj_{n}.weight"], sd[f"proj_{n}.bias"], sd["ln.weight"], sd["ln.bias"],
                                 sd["task_embed"][0, k], pe[:T], pm))
        off += T
    return torch.cat(xs, dim=1)


def ttm_forward(sd, n_heads: int, ttm_out, lam_out, asd_out=None, masks=None) -> torch.Tensor:
    """TaskFusionMFTransformer{2,3}Task.forward on backbone features -> (B, 2) logits.
    masks = {"pos": ..., "layers": [...]}: train-mode forward under explicit dropout masks (tests/dropmask.py)."""
    feats, names, ids = [ttm_out, lam_out], ["ttm", "lam"], [0, 1]
    if asd_out is not None:
        feats.append(asd_out); names.append("asd"); ids.append(2)
    x = hhi_tokens(sd, feats, names, ids, masks)
    x = encoder(x, sd, "transformer_encoder.", n_layers_of(sd, "transformer_encoder."), n_heads, masks)
    pooled = x.mean(dim=1)
    y = layer_norm(pooled, sd["linear_head.0.weight"], sd["linear_head.0.bias"])
    return linear(y, sd["linear_head.1.weight"], sd["linear_head.1.bias"])


def asd_forward(sd, n_heads: int, ttm_out, lam_out, asd_out, masks=None) -> torch.Tensor:
    """HHI/models/asd TaskFusionMFTransformer3Task: token order asd, ttm, lam (task ids 2, 0, 1); returns the
    encoded ASD block as (B*T, d)."""
    x = hhi_tokens(sd, [asd_out, ttm_out, lam_out], ["asd", "ttm", "lam"], [2, 0, 1], masks)
    x = encoder(x, sd, "transformer_encoder.", n_layers_of(sd, "transformer_encoder."), n_heads, masks)
    B, T = asd_out.shape[0], asd_out.shape[1]
    return x[:, :T, :].reshape(B * T, -1)


def hhi_g_encode(sd, n_heads: int, task: str, lam_feat, ttm_feat=None, asd_feat=None) -> torch.Tensor:
    """TaskTranslationPromptTransformer.encode -> memory in the reference's (S, B, d) / (3, B*T, d) layout."""
    if task == "lam":
        x = hhi_tokens(sd, [lam_feat], ["lam"], [0])
    else:
        x = hhi_tokens(sd, [lam_feat, ttm_feat, asd_feat], ["lam", "ttm", "asd"], [0, 1, 2])
    x = encoder(x, sd, "transformer_encoder.", n_layers_of(sd, "transformer_encoder."), n_heads)
    if task == "asd":
        B, S, d = x.shape
        T = S // 3
        return torch.stack([x[:, 0:T].reshape(-1, d), x[:, T:2 * T].reshape(-1, d), x[:, 2 * T:3 * T].reshape(-1, d)], dim=0)
    return x.permute(1, 0, 2)


def weighted_ce(logits: torch.Tensor, target: torch.Tensor, weight: Sequence[float]) -> torch.Tensor:
    """nn.CrossEntropyLoss(weight=w): sum_i w[y_i] * nll_i / sum_i w[y_i]."""
    w = torch.tensor(list(weight), dtype=logits.dtype)
    z = logits - logits.max(dim=-1, keepdim=True).values
    logp = z - torch.log(torch.exp(z).sum(dim=-1, keepdim=True))
    nll = -logp.gather(1, target[:, None])[:, 0]
    wi = w[target]
    return (wi * nll).sum() / wi.sum()


# ---- HOI translators ----------------------------------------------------------------------------
def hoi_tokens(sd, feats: Sequence[torch.Tensor], proj_names: Sequence[Optional[str]], masks=None) -> torch.Tensor:
    """cat(proj_k(feat_k)) -> shared LN -> + learned pe (1, S, d).
    masks["feat"][k] (B, T_k, d): keep-scale of the feature dropout `self.dp(proj_k(feat_k))`
    (HOI/models/pnr/video_model_transfer_3task.py:249-252), None entries = no dropout on that segment."""
    xs = [f if n is None else linear(f, sd[f"{n}.weight"], sd[f"{n}.bias"]) for f, n in zip(feats, proj_names)]
    if masks is not None and masks.get("feat") is not None:
        xs = [x if m is None else x * m.to(x.dtype) for x, m in zip(xs, masks["feat"])]
    x = torch.cat(xs, dim=1)
    return layer_norm(x, sd["ln.weight"], sd["ln.bias"]) + sd["pe"]


def lta4_forward(sd, n_heads: int, feat_pnr, feat_oscc, feat_action, feat_lta, num_classes: Sequence[int], masks=None):
    """TaskFusionMFTransformerLTA4Task.forward (training-mode head, no activation) on features:
    pnr/oscc (B, n, 8192), action (B, n, d), lta (B, n, 2048) -> [(B, Z, n_verbs), (B, Z, n_nouns)]."""
    x = hoi_tokens(sd, [feat_pnr, feat_oscc, feat_action, feat_lta], ["proj_pnr", "proj_oscc", None, "proj_lta"], masks)
    x = encoder(x, sd, "transformer.", n_layers_of(sd, "transformer."), n_heads, masks)
    pooled = x.mean(dim=1)
    z = 0
    outs = []
    while f"head.projections.{z}.weight" in sd:
        outs.append(linear(pooled, sd[f"head.projections.{z}.weight"], sd[f"head.projections.{z}.bias"]))
        z += 1
    y = torch.stack(outs, dim=1)
    return list(torch.split(y, list(num_classes), dim=-1))


def pnr3_forward(sd, n_heads: int, pnr_feat, oscc_feat, slow_feat, fast_feat, masks=None) -> torch.Tensor:
    """TaskFusionMFTransformer3TaskDropout.forward on features (eval / p = 0, or train mode under explicit `masks`): the
    shared `ln` is also the first element of linear_head."""
    x = hoi_tokens(sd, [pnr_feat, oscc_feat, slow_feat, fast_feat], ["proj1", "proj2", "proj3_slow", "proj3_fast"], masks)
    x = encoder(x, sd, "transformer.", n_layers_of(sd, "transformer."), n_heads, masks)
    pooled = x.mean(dim=1)
    y = layer_norm(pooled, sd["ln.weight"], sd["ln.bias"])
    return linear(y, sd["linear_head.1.weight"], sd["linear_head.1.bias"])


def ar_forward(sd, n_heads: int, feats: Sequence[torch.Tensor], proj_names: Sequence[str]):
    """HOI/models/lta/lta_models_transfer.py:125-137 (3-task) and :227-235 (2-task AR) on features: cat(proj_k(feat_k))
    -> ln + pe -> encoder -> token mean -> two heads that both start with the SAME `ln`
    (`linear_head{1,2} = Sequential(self.ln, Linear)`); only `ln.*` is read so that its gradient sums all three uses."""
    x = hoi_tokens(sd, list(feats), list(proj_names))
    x = encoder(x, sd, "transformer.", n_layers_of(sd, "transformer."), n_heads)
    y = layer_norm(x.mean(dim=1), sd["ln.weight"], sd["ln.bias"])
    return [linear(y, sd["linear_head1.1.weight"], sd["linear_head1.1.bias"]),
            linear(y, sd["linear_head2.1.weight"], sd["linear_head2.1.bias"])]


def hoi_g_encode(sd, n_heads: int, task: str, feat_pnr, feat_oscc, a, b) -> torch.Tensor:
    """TaskTranslationPromptTransformer6Task.encode (HOI/models/multitask/video_model_builder.py:320-347) on features.
    'lta' prompts: a = action features (B, n, d) used unprojected (task id 2), b = LTA features (B, n, 2048) (task id 3);
    otherwise a / b are the pooled SlowFast slow / fast pathways, projected separately, concatenated to ONE 16-token
    block that shares task id 2 and one position run. Returns the decoder memory (S, B, d)."""
    pe = sd["pos_embed.pe"][:, 0, :]
    lw, lb, te = sd["ln.weight"], sd["ln.bias"], sd["task_embed"]
    x1 = encode_prepare(feat_pnr, sd["proj_pnr.weight"], sd["proj_pnr.bias"], lw, lb, te[0, 0], pe[:feat_pnr.shape[1]])
    x2 = encode_prepare(feat_oscc, sd["proj_oscc.weight"], sd["proj_oscc.bias"], lw, lb, te[0, 1], pe[:feat_oscc.shape[1]])
    if "lta" in task:
        x3 = encode_prepare(a, None, None, lw, lb, te[0, 2], pe[:a.shape[1]])
        x4 = encode_prepare(b, sd["proj_lta.weight"], sd["proj_lta.bias"], lw, lb, te[0, 3], pe[:b.shape[1]])
        x = torch.cat((x1, x2, x3, x4), dim=1)
    else:
        f3 = torch.cat((linear(a, sd["proj_action_slow.weight"], sd["proj_action_slow.bias"]),
                        linear(b, sd["proj_action_fast.weight"], sd["proj_action_fast.bias"])), dim=1)
        x3 = encode_prepare(f3, None, None, lw, lb, te[0, 2], pe[:f3.shape[1]])
        x = torch.cat((x1, x2, x3), dim=1)
    x = encoder(x, sd, "transformer_encoder.", n_layers_of(sd, "transformer_encoder."), n_heads)
    return x.permute(1, 0, 2)


def lta2_forward(sd, n_heads: int, feat_action, feat_lta, num_classes: Sequence[int]):
    """TaskFusionMFTransformer2Task.forward of the LTA task (HOI/models/lta/lta_models_lta_transfer.py:505-513) on features:
    action (B, n, d) used as is, lta (B, n, 2048) through proj_lta (identity when the translator is 2048 wide)."""
    x = hoi_tokens(sd, [feat_action, feat_lta], [None, "proj_lta" if "proj_lta.weight" in sd else None])
    x = encoder(x, sd, "transformer.", n_layers_of(sd, "transformer."), n_heads)
    pooled = x.mean(dim=1)
    z, outs = 0, []
    while f"head.projections.{z}.weight" in sd:
        outs.append(linear(pooled, sd[f"head.projections.{z}.weight"], sd[f"head.projections.{z}.bias"]))
        z += 1
    return list(torch.split(torch.stack(outs, dim=1), list(num_classes), dim=-1))


def gelu(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() (exact form): x * Phi(x)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def vit_forward(sd, heads: int, pnr_feat, oscc_feat) -> torch.Tensor:
    """TaskFusionMFTransformer.forward (HOI/models/pnr/video_model_transfer.py:58-66) over simple_vit.Transformer
    (HOI/models/pnr/simple_vit.py:55-107) on features: tokens = cat(proj1(pnr), proj2(oscc)) + pe (NO token-prep LayerNorm);
    per block x = to_out(softmax(q k^T * dim_head^-0.5) v) + x with q, k, v = to_qkv(norm(x)) (no biases, inner width
    heads * dim_head), then x = W2 gelu(W1 norm(x) + b1) + b2 + x; head = Linear(LayerNorm(mean_s x))."""
    x = torch.cat((linear(pnr_feat, sd["proj1.weight"], sd["proj1.bias"]), linear(oscc_feat, sd["proj2.weight"], sd["proj2.bias"])), dim=1) + sd["pe"]
    B, S, d = x.shape
    i = 0
    while f"transformer.layers.{i}.0.to_qkv.weight" in sd:
        pa, pf = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1.net."
        h = layer_norm(x, sd[pa + "norm.weight"], sd[pa + "norm.bias"])
        qkv = linear(h, sd[pa + "to_qkv.weight"], None)
        inner = qkv.shape[-1] // 3
        dh = inner // heads
        q, k, v = [t.reshape(B, S, heads, dh).permute(0, 2, 1, 3) for t in qkv.split(inner, dim=-1)]
        dots = (q @ k.transpose(-1, -2)) * dh ** -0.5
        dots = dots - dots.max(dim=-1, keepdim=True).values
        p = torch.exp(dots)
        p = p / p.sum(dim=-1, keepdim=True)
        o = (p @ v).permute(0, 2, 1, 3).reshape(B, S, inner)
        x = linear(o, sd[pa + "to_out.weight"], None) + x
        h = layer_norm(x, sd[pf + "0.weight"], sd[pf + "0.bias"])
        x = linear(gelu(linear(h, sd[pf + "1.weight"], sd[pf + "1.bias"])), sd[pf + "3.weight"], sd[pf + "3.bias"]) + x
        i += 1
    y = layer_norm(x.mean(dim=1), sd["linear_head.0.weight"], sd["linear_head.0.bias"])
    return linear(y, sd["linear_head.1.weight"], sd["linear_head.1.bias"])


def hoi_g2_encode(sd, n_heads: int, feat_pnr, feat_oscc) -> torch.Tensor:
    """TaskTranslationPromptTransformer2Task.encode (HOI/models/multitask/video_model_builder_2task.py:128-140) on features."""
    pe = sd["pos_embed.pe"][:, 0, :]
    lw, lb, te = sd["ln.weight"], sd["ln.bias"], sd["task_embed"]
    x1 = encode_prepare(feat_pnr, sd["proj_pnr.weight"], sd["proj_pnr.bias"], lw, lb, te[0, 0], pe[:feat_pnr.shape[1]])
    x2 = encode_prepare(feat_oscc, sd["proj_oscc.weight"], sd["proj_oscc.bias"], lw, lb, te[0, 1], pe[:feat_oscc.shape[1]])
    x = encoder(torch.cat((x1, x2), dim=1), sd, "transformer_encoder.", n_layers_of(sd, "transformer_encoder."), n_heads)
    return x.permute(1, 0, 2)


def hoi_ga_encode(sd, n_heads: int, task: str, feat_action, feat_lta=None) -> torch.Tensor:
    """TaskTranslationPromptTransformerActionTask.encode (HOI/models/multitask/video_model_builder_action.py:117-133) on
    features: 'lta' prompts -> ln(cat(action, lta)) + learned pe (1, 4, d); otherwise ONE action token with task embedding 0
    and sinusoid position 0."""
    lw, lb = sd["ln.weight"], sd["ln.bias"]
    if "lta" in task:
        x = layer_norm(torch.cat((feat_action, feat_lta), dim=1), lw, lb) + sd["pe"]
    else:
        x = encode_prepare(feat_action, None, None, lw, lb, sd["task_embed"][0, 0], sd["pos_embed.pe"][:feat_action.shape[1], 0, :])
    x = encoder(x, sd, "transformer_encoder.", n_layers_of(sd, "transformer_encoder."), n_heads)
    return x.permute(1, 0, 2)


# ---- EgoT2-g sequence decoder + vocabulary head ---------------------------------------------
def attention(q_in: torch.Tensor, kv_in: torch.Tensor, in_w, in_b, out_w, out_b, n_heads: int, causal: bool) -> torch.Tensor:
    """nn.MultiheadAttention math, batch-first: q_in (B, Sq, d), kv_in (B, Sk, d); packed in-projection rows
    [Wq; Wk; Wv]; `causal` adds the reference's lower-triangular additive mask (get_tgt_mask)."""
    B, Sq, d = q_in.shape
    Sk = kv_in.shape[1]
    dh = d // n_heads
    q = linear(q_in, in_w[:d], in_b[:d]).reshape(B, Sq, n_heads, dh).permute(0, 2, 1, 3)
    k = linear(kv_in, in_w[d:2 * d], in_b[d:2 * d]).reshape(B, Sk, n_heads, dh).permute(0, 2, 1, 3)
    v = linear(kv_in, in_w[2 * d:], in_b[2 * d:]).reshape(B, Sk, n_heads, dh).permute(0, 2, 1, 3)
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        keep = torch.tril(torch.ones(Sq, Sk, dtype=torch.bool))
        scores = scores.masked_fill(~keep, float("-inf"))
    scores = scores - scores.max(dim=-1, keepdim=True).values
    p = torch.exp(scores)
    p = p / p.sum(dim=-1, keepdim=True)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, Sq, d)
    return linear(o, out_w, out_b)


def decoder_layer(x, mem, sd, prefix: str, n_heads: int, eps: float = 1e-5):
    """Post-LN nn.TransformerDecoderLayer as subclassed by CustomDecoderLayer
    (HHI/models/multitask/task_prompt_model.py:163-172, HOI/models/multitask/video_model_builder.py:20-30):
    x = norm1(x + SA(x, causal)); x = norm2(x + CA(x, memory)); x = norm3(x + FFN(x))."""
    g = lambda k: sd[prefix + k]  # noqa: E731
    a = attention(x, x, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"), g("self_attn.out_proj.weight"),
                  g("self_attn.out_proj.bias"), n_heads, True)
    x = layer_norm(x + a, g("norm1.weight"), g("norm1.bias"), eps)
    c = attention(x, mem, g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias"),
                  g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"), n_heads, False)
    x = layer_norm(x + c, g("norm2.weight"), g("norm2.bias"), eps)
    f = linear(torch.relu(linear(x, g("linear1.weight"), g("linear1.bias"))), g("linear2.weight"), g("linear2.bias"))
    return layer_norm(x + f, g("norm3.weight"), g("norm3.bias"), eps)


def g_decode(sd, n_heads: int, y: torch.Tensor, memory: torch.Tensor) -> torch.Tensor:
    """decode() of the EgoT2-g models (task_prompt_model.py:260-269 / video_model_builder.py:150-159), dropout off:
    y (B, sy) int64 prompt/target tokens, memory (S, B, d) from encode() -> (sy, B, |V|) vocabulary logits."""
    d = sd["embedding.weight"].shape[1]
    sy = y.shape[1]
    x = sd["embedding.weight"][y] * math.sqrt(d) + sd["pos_embed.pe"][:sy, 0, :]        # (B, sy, d)
    mem = memory.permute(1, 0, 2)
    for i in range(n_layers_of(sd, "transformer_decoder.")):
        x = decoder_layer(x, mem, sd, f"transformer_decoder.layers.{i}.", n_heads)
    return linear(x, sd["fc.weight"], sd["fc.bias"]).permute(1, 0, 2)


def pnr_head_forward(fmap: torch.Tensor, pool: Sequence[int], proj_w=None, proj_b=None, middle: bool = True,
                     act_dim: Optional[int] = None, clip_mean: bool = False) -> torch.Tensor:
    """`ResNetKeyframeLocalizationHead.forward(inputs, middle)` for its single pathway (HOI/models/pnr/head_helper.py:353-381),
    producer side of the feature hand-off, written out as window sums (no nn.AvgPool3d):
      fmap (N, C, T, H, W) res5 map -> AvgPool3d(pool, stride=1) -> permute (N, T', H', W', C) -> rows (N, T', H'W'C)   [:359-372]
      middle: return the rows [:373-374] (clip_mean: their `.mean(dim=1)`, lta_models_lta_transfer.py:335-345)
      else  : Linear(8192, classes) on the rows [:375], softmax over `act_dim` when given (eval mode, :378-379),
              permute(0, 2, 1) [:381]."""
    kt, kh, kw = (int(v) for v in pool)
    N, C, T, H, W = fmap.shape
    To, Ho, Wo = T - kt + 1, H - kh + 1, W - kw + 1
    acc = torch.zeros((N, C, To, Ho, Wo), dtype=fmap.dtype)
    for a in range(kt):
        for b in range(kh):
            for c in range(kw):
                acc = acc + fmap[:, :, a:a + To, b:b + Ho, c:c + Wo]
    x = (acc / float(kt * kh * kw)).permute(0, 2, 3, 4, 1).reshape(N, To, Ho * Wo * C)
    if middle:
        return x.mean(dim=1) if clip_mean else x
    y = linear(x, proj_w, proj_b)
    if act_dim is not None:
        y = y - y.max(dim=act_dim, keepdim=True).values
        y = torch.exp(y)
        y = y / y.sum(dim=act_dim, keepdim=True)
    return y.permute(0, 2, 1)


def to_dtype(sd: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
