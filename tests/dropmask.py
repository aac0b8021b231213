"""Python restatement of libegot2x's counter-based dropout generator (egot2_amd/csrc/common.h: site_key, rand_quad,
drop_threshold, drop_scale) and of the (row, column) keying each encoder implementation uses at each dropout site.

Test infrastructure: it hands the masks the HIP kernels draw for a given seed to the oracle (oracle/translator_ref.py `masks`
arguments), so that the TRAIN-mode arithmetic (p = 0.5 + 0.1 positional: the mode bench.py times) is compared with the fp64
oracle element by element instead of through expectations. Nothing here is imported by the product path.

Sites (nn.TransformerEncoderLayer + PositionalEncoding + the HOI feature dropout; reference
HHI/models/ttm/model_taskspecific.py:149-151,211-215, HOI/models/pnr/video_model_transfer_3task.py:249-252):
    FEAT  key layer = segment index; row = b * T_k + t (row of the segment's projection GEMM); col = feature
    POS   key layer = 0;             row = b * S + s (packed token);  col = feature
    ATTN  key layer = l;             row = (b * H + h) * RS + query;  col = key        RS = 64 fused, 128 wide, S generic / tiled
    RES1  key layer = l;             row = b * S + s;                 col = feature
    FFN   key layer = l;             row = b * 64 + s (fused) / b * S + s (others); col = hidden unit
    RES2  key layer = l;             row = b * S + s;                 col = feature
"""
from __future__ import annotations

import numpy as np
import torch

SITE_FEAT, SITE_POS, SITE_ATTN, SITE_RES1, SITE_FFN, SITE_RES2 = 1, 2, 3, 4, 5, 6
M64 = (1 << 64) - 1
M32 = np.uint64(0xFFFFFFFF)


def lcg(seed: int) -> int:
    """egx_seed_advance / pack_weights_kernel: the device seed's step (one per training forward)."""
    return (seed * 6364136223846793005 + 1442695040888963407) & M64


def site_key(seed: int, layer: int, site: int) -> int:
    k = ((seed & M64) * 0x9E3779B97F4A7C15 + ((layer << 8 | site) * 0xD1B54A32D192ED03)) & M64
    k ^= k >> 29
    return k | 1


def drop_threshold(p: float) -> int:
    t = float(np.float32(p)) * 65536.0       # the C side receives p as a float
    if t <= 0:
        return 0
    if t >= 65536.0:
        return 65536
    u = int(t + 0.5)
    return u if u else 1


def inv_keep(p: float) -> float:
    p32 = np.float32(p)
    return float(np.float32(1.0) / (np.float32(1.0) - p32)) if p32 < 1 else 0.0


def rand_quad(key: int, row: np.ndarray, colquad: np.ndarray):
    """-> (z, y) uint64 arrays holding 32-bit words: columns 4c, 4c+1 = low / high half of z; 4c+2, 4c+3 = of y."""
    k0, k1 = np.uint64(key & 0xFFFFFFFF), np.uint64(key >> 32)
    row = row.astype(np.uint64)
    cq = colquad.astype(np.uint64)
    x = ((row * np.uint64(0x9E3779B1) + k1) & M32) ^ ((cq * np.uint64(0x85EBCA77) + k0) & M32)
    x ^= x >> np.uint64(16)
    p = x * np.uint64(0x7FEB352D)
    y = (p & M32) ^ (p >> np.uint64(32))
    y ^= y >> np.uint64(15)
    z = (y * np.uint64(0x846CA68B)) & M32
    z ^= z >> np.uint64(16)
    return z, y


def keep_scale(key: int, rows: np.ndarray, cols: np.ndarray, p: float) -> torch.Tensor:
    """rows (...,) and cols (C,) -> float64 tensor (..., C) of 0 / (1 / (1 - p))."""
    thresh = drop_threshold(p)
    if thresh == 0:
        return torch.ones(rows.shape + cols.shape, dtype=torch.float64)
    r = rows.reshape(rows.shape + (1,))
    c = cols.reshape((1,) * rows.ndim + cols.shape)
    z, y = rand_quad(key, r, c >> 2)
    w = np.where((c & 2) != 0, y, z)
    v = np.where((c & 1) != 0, w >> np.uint64(16), w & np.uint64(0xFFFF))
    return torch.from_numpy(np.where(v >= np.uint64(thresh), inv_keep(p), 0.0))


ATTN_ROW_STRIDE = {"fused": 64, "wide": 128}      # others (generic, tiled): S; wide with S > 128: 512


def encoder_masks(seed: int, impl: str, B: int, seg_T, d: int, H: int, d_ff: int, L: int, p_drop: float, p_pos: float = 0.0,
                  p_feat: float = 0.0, feat_proj=None):
    """The keep-scales an `impl` ("fused" | "generic" | "wide" | "tiled") encoder call with host seed `seed` applies, as the
    oracle's `masks` dict. seg_T: tokens per segment in packed order; feat_proj[k]: segment k has a projection (FEAT dropout
    applies to projected segments only)."""
    S = int(sum(seg_T))
    b = np.arange(B, dtype=np.int64)
    s = np.arange(S, dtype=np.int64)
    tok_rows = b[:, None] * S + s[None, :]                                      # (B, S)
    cols_d = np.arange(d, dtype=np.int64)
    masks = {"layers": []}
    if p_pos > 0:
        masks["pos"] = keep_scale(site_key(seed, 0, SITE_POS), tok_rows, cols_d, p_pos)
    if p_feat > 0:
        masks["feat"] = []
        for k, T in enumerate(seg_T):
            if feat_proj is not None and not feat_proj[k]:
                masks["feat"].append(None)
                continue
            rows = b[:, None] * T + np.arange(T)[None, :]
            masks["feat"].append(keep_scale(site_key(seed, k, SITE_FEAT), rows, cols_d, p_feat))
    rs = 512 if (impl == "wide" and S > 128) else ATTN_ROW_STRIDE.get(impl, S)
    ffn_rows = b[:, None] * 64 + s[None, :] if impl == "fused" else tok_rows
    h = np.arange(H, dtype=np.int64)
    attn_rows = (b[:, None, None] * H + h[None, :, None]) * rs + s[None, None, :]   # (B, H, S)
    for layer in range(L):
        if p_drop <= 0:
            masks["layers"].append(None)
            continue
        masks["layers"].append({
            "attn": keep_scale(site_key(seed, layer, SITE_ATTN), attn_rows, s, p_drop),
            "res1": keep_scale(site_key(seed, layer, SITE_RES1), tok_rows, cols_d, p_drop),
            "ffn": keep_scale(site_key(seed, layer, SITE_FFN), ffn_rows, np.arange(d_ff, dtype=np.int64), p_drop),
            "res2": keep_scale(site_key(seed, layer, SITE_RES2), tok_rows, cols_d, p_drop),
        })
    return masks
