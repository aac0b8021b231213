// Attention of the tiled d = 128 path (48 < S <= 512: the reference's real TTM / ASD batches are 15 .. 150 frames per task,
// HHI/dataset/ttm/data_loader_2task.py:119,150-162; validation runs one clip of up to 3 x 150 tokens). Replaces the
// bmm / softmax / dropout / bmm chain inside F.multi_head_attention_forward as called by nn.TransformerEncoderLayer
// (HHI/models/ttm/model_taskspecific.py:211-215) for 4 heads of 32.
//
// One workgroup (8 waves) per (clip, head, range of 16-row tiles); the other operand of the whole clip (K | V, or Q | dO) sits in
// LDS as fp32 rows (S <= 512); a wave owns one 16-query (or 16-key) tile at a time and walks the 32-row blocks of the LDS
// operand with two score tiles live (forward: online softmax). Operand convention, compute modes (bf16 / split-bf16 "f32s") and
// the feature-major chaining (S^T = K Q^T puts the probabilities where the next MFMA wants its B operand) are those of fused_dev.h.
//   forward      S^T = K Q^T, P = softmax, O^T = V^T P^T                        -> attn_o (Ntok, 128), lse (B, 4, S)
//   backward A   per query tile: dS^T = P (mask dP^T - delta), dQ^T = K^T dS^T  -> dqkv[:, 0:128], delta (B, 4, S)
//   backward B   per key tile:   dV^T = dO^T P, dK^T = Q^T dS                   -> dqkv[:, 128:384]
// delta = rowsum(dO . O) (equal to sum_k P mask dP). Dropout masks are regenerated from (seed, layer, SITE_ATTN) with the
// generic kernels' keying: row = (clip * 4 + head) * S + query, column = key.
#include "fused.h"
#include "fused_dev.h"

namespace egx {

namespace {
constexpr int TA_LD = FDH + 4;          // row stride of the token-major K / V / Q / dO blocks
constexpr float TA_SCALE = 0.17677669529663687f;   // 1 / sqrt(32)
constexpr float TA_C2 = TA_SCALE * 1.4426950408889634f;     // scores are kept in log2 units: P = exp2(c2 q.k - lse2), one v_exp per element

// VALU budget of the softmax / mask code (the kernels are VALU-bound: ~10 instructions per score against 1/8 .. 3/4 of an MFMA):
//  * the query-side (or key-side) fragment is scaled by c2 once, so a score needs no multiply;
//  * the key < S test runs in the last K-block only (wave-uniform branch);
//  * the dropout keep-scale 1 / (1 - p) rides on the staged V (forward, dQ) or dO (dK / dV) rows: a kept element is selected, not multiplied;
//  * the 1 / sqrt(d_h) of dS is applied to the finished dQ / dK tiles.
__device__ __forceinline__ float4 scale4(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
template <int CM>
__device__ __forceinline__ Frag<CM> load_frag_scaled(const float* p, int q, float s) {
    return make_frag<CM>(scale4(*reinterpret_cast<const float4*>(p + 4 * q), s), scale4(*reinterpret_cast<const float4*>(p + 16 + 4 * q), s));
}
// 0 / 1 keep decision of element (row, col) (drop_scale without the scale)
__device__ __forceinline__ bool drop_keep(uint64_t key, uint32_t row, uint32_t col, uint32_t thresh) {
    const uint2 h = rand_quad(key, row, col >> 2);
    const uint32_t w = (col & 2u) ? h.y : h.x;
    const uint32_t v = (col & 1u) ? (w >> 16) : (w & 0xffffu);
    return v >= thresh;
}

__device__ __forceinline__ uint64_t attn_key(const TiledAttnParams& p) {
    return p.seed_ptr ? site_key(*p.seed_ptr, (uint32_t)p.layer, SITE_ATTN) : p.drop_key;
}

// rows [0, S) of one 32-wide column block of the clip's Q | K | V grid -> token-major LDS rows, rows S .. SKP - 1 zero
// (four requests in flight per thread: a load-store-load chain pays a memory round trip per element)
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int ld_src, int S, int SKP, float scale = 1.f) {
    const int n = SKP * 8, step = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * step, row = i >> 3, c4 = (i & 7) * 4;
            const int rr = row < S ? row : S - 1;           // clamped: unconditional loads, selected below
            v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)rr * ld_src + c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * step, row = i >> 3, c4 = (i & 7) * 4;
            if (i < n) *reinterpret_cast<f32x4*>(dst + row * TA_LD + c4) = row < S ? v[u] * scale : f32x4{0, 0, 0, 0};
        }
    }
}
}  // namespace

// The three kernels share one shape: NW waves per workgroup, the clip-wide operand in LDS (rows padded to a multiple of 32 with
// zeros), a wave walks the 32-row K-blocks of that operand with a fixed, small register footprint (two score tiles live at a
// time), so that two waves per SIMD overlap one wave's softmax / split VALU work with the other's MFMAs.
// Waves per workgroup are chosen at launch (4 .. 8: one group of TA_G 16-row tiles per wave when the clip has that many). The
// loops are chains of dependent LDS read -> MFMA -> cross-lane max -> exp -> MFMA steps AND heavy in VALU work (operand
// conversion / three-way split, softmax, dropout hash: 4 cycles per wave64 instruction): two waves per SIMD overlap them.
constexpr int TA_MAX_THREADS = 512;
constexpr int TA_G = 2;         // 16-row tiles a wave works on at a time

// forward: online softmax over the key blocks (running max m and per-lane partial sums l; O^T rescaled by exp(m_old - m_new), a
// per-lane scalar because a lane's four accumulator rows belong to ONE query column)
template <int CM>
__global__ __launch_bounds__(TA_MAX_THREADS) void tiled_attn_fwd_kernel(TiledAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = p.S, SKP = (S + 31) & ~31, LDVT = SKP + 4;
    float* Ks = lds;                    // [SKP][TA_LD]
    float* Vt = lds + SKP * TA_LD;      // [32][LDVT] V^T, keys >= S zero
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;     // Q of the head; K at + 128, V at + 256
    const float vscale = p.drop_thresh ? p.drop_inv : 1.f;
    stage_rows(Ks, base + FD, 3 * FD, S, SKP);
    for (int i0 = tid; i0 < SKP * 8; i0 += 4 * blockDim.x) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x, row = i >> 3, c4 = (i & 7) * 4;
            v[u] = *reinterpret_cast<const f32x4*>(base + (size_t)(row < S ? row : S - 1) * (3 * FD) + 2 * FD + c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x, row = i >> 3, c4 = (i & 7) * 4;
            if (i < SKP * 8) {
                const f32x4 w = row < S ? v[u] * vscale : f32x4{0, 0, 0, 0};
                Vt[(c4 + 0) * LDVT + row] = w[0]; Vt[(c4 + 1) * LDVT + row] = w[1];
                Vt[(c4 + 2) * LDVT + row] = w[2]; Vt[(c4 + 3) * LDVT + row] = w[3];
            }
        }
    }
    __syncthreads();
    const uint64_t dkey = attn_key(p);
    const int nqt = (S + 15) >> 4, nkb = SKP >> 5, ngrp = (nqt + TA_G - 1) / TA_G;
    // a wave owns TA_G query tiles at a time: every K / V fragment (LDS read + operand conversion or three-way split) then serves
    // TA_G MFMA groups — the kernels are bound by that VALU work, not by the matrix pipe
    for (int g = blockIdx.y * nw + wave; g < ngrp; g += nw * gridDim.y) {
        int query[TA_G];
        Frag<CM> bq[TA_G];
        float m[TA_G], l[TA_G];
        f32x4 oc[TA_G][2];
#pragma unroll
        for (int t = 0; t < TA_G; ++t) {
            query[t] = (g * TA_G + t) * 16 + r;
            const int qrow = query[t] < S ? query[t] : S - 1;       // padded queries recompute the last row; never stored
            bq[t] = load_frag_scaled<CM>(base + (size_t)qrow * (3 * FD), q, TA_C2);
            m[t] = -INFINITY; l[t] = 0.f;
            oc[t][0] = f32x4{0, 0, 0, 0}; oc[t][1] = f32x4{0, 0, 0, 0};
        }
        for (int kb = 0; kb < nkb; ++kb) {
            f32x4 sc[TA_G][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const Frag<CM> ak = load_frag<CM>(Ks + (kb * 32 + j * 16 + r) * TA_LD, q);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) { sc[t][j] = f32x4{0, 0, 0, 0}; mma<CM>(sc[t][j], ak, bq[t]); }
            }
            if (kb == nkb - 1) {        // the only block with keys >= S
#pragma unroll
                for (int t = 0; t < TA_G; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (kb * 32 + j * 16 + 4 * q + e >= S) sc[t][j][e] = -INFINITY;
            }
            Frag<CM> bp[TA_G];
#pragma unroll
            for (int t = 0; t < TA_G; ++t) {
                float mb = fmaxf(fmaxf(fmaxf(sc[t][0][0], sc[t][0][1]), fmaxf(sc[t][0][2], sc[t][0][3])),
                                 fmaxf(fmaxf(sc[t][1][0], sc[t][1][1]), fmaxf(sc[t][1][2], sc[t][1][3])));
                mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
                mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
                const float mn = fmaxf(m[t], mb);               // finite: every block holds at least one key < S
                const float corr = __builtin_amdgcn_exp2f(m[t] - mn);
                m[t] = mn;
                l[t] *= corr;
                oc[t][0] *= corr; oc[t][1] *= corr;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pv = __builtin_amdgcn_exp2f(sc[t][j][e] - mn);
                        l[t] += pv;
                        if (p.drop_thresh) pv = drop_keep(dkey, (uint32_t)(bh * S + query[t]), (uint32_t)(kb * 32 + j * 16 + 4 * q + e), p.drop_thresh) ? pv : 0.f;
                        sc[t][j][e] = pv;
                    }
                bp[t] = chain_frag<CM>(sc[t][0], sc[t][1]);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const Frag<CM> av = load_frag<CM>(Vt + (ct * 16 + r) * LDVT + kb * 32, q);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) mma<CM>(oc[t][ct], av, bp[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < TA_G; ++t) {
            float lt = l[t];
            lt += __shfl_xor(lt, 16, 64);
            lt += __shfl_xor(lt, 32, 64);
            const float inv = 1.f / lt;
            if (query[t] < S) {
                if (q == 0) p.lse[(size_t)bh * S + query[t]] = m[t] + log2f(lt);     // log2 units (internal to these kernels)
                float* o = p.attn_o + ((size_t)b * S + query[t]) * FD + h * FDH + 4 * q;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4*>(o + ct * 16) = oc[t][ct] * inv;
            }
        }
    }
}

// backward, query side: dQ and delta
template <int CM>
__global__ __launch_bounds__(TA_MAX_THREADS) void tiled_attn_dq_kernel(TiledAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = p.S, SKP = (S + 31) & ~31;
    float* Ks = lds;
    float* Vs = lds + SKP * TA_LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;
    stage_rows(Ks, base + FD, 3 * FD, S, SKP);
    stage_rows(Vs, base + 2 * FD, 3 * FD, S, SKP, p.drop_thresh ? p.drop_inv : 1.f);
    __syncthreads();
    const uint64_t dkey = attn_key(p);
    const int nqt = (S + 15) >> 4, nkb = SKP >> 5, ngrp = (nqt + TA_G - 1) / TA_G;
    for (int g = blockIdx.y * nw + wave; g < ngrp; g += nw * gridDim.y) {
        int query[TA_G];
        Frag<CM> bq[TA_G], bdo[TA_G];
        float delta[TA_G], L[TA_G];
        f32x4 dq[TA_G][2];
#pragma unroll
        for (int t = 0; t < TA_G; ++t) {
            query[t] = (g * TA_G + t) * 16 + r;
            const int qrow = query[t] < S ? query[t] : S - 1;
            const size_t tok = (size_t)b * S + qrow;
            bq[t] = load_frag_scaled<CM>(base + (size_t)qrow * (3 * FD), q, TA_C2);
            // dO row and O row of this query: the lane's 8 of the head's 32 channels (the K-block positions of a fragment)
            const float* dop = p.d_o + tok * FD + h * FDH;
            const float* op = p.attn_o + tok * FD + h * FDH;
            const float4 d0 = *reinterpret_cast<const float4*>(dop + 4 * q), d1 = *reinterpret_cast<const float4*>(dop + 16 + 4 * q);
            const float4 o0 = *reinterpret_cast<const float4*>(op + 4 * q), o1 = *reinterpret_cast<const float4*>(op + 16 + 4 * q);
            float dl = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w;
            dl += __shfl_xor(dl, 16, 64);
            dl += __shfl_xor(dl, 32, 64);
            delta[t] = dl;
            bdo[t] = make_frag<CM>(d0, d1);
            L[t] = p.lse[(size_t)bh * S + qrow];
            if (q == 0 && query[t] < S) p.delta[(size_t)bh * S + query[t]] = dl;
            dq[t][0] = f32x4{0, 0, 0, 0}; dq[t][1] = f32x4{0, 0, 0, 0};
        }
        for (int kb = 0; kb < nkb; ++kb) {
            f32x4 ds[TA_G][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const Frag<CM> ak = load_frag<CM>(Ks + (kb * 32 + j * 16 + r) * TA_LD, q);
                const Frag<CM> av = load_frag<CM>(Vs + (kb * 32 + j * 16 + r) * TA_LD, q);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) {
                    f32x4 st = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
                    mma<CM>(st, ak, bq[t]);         // S^T = K Q^T
                    mma<CM>(dp, av, bdo[t]);        // dP^T = V dO^T
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = kb * 32 + j * 16 + 4 * q + e;
                        const float pv = __builtin_amdgcn_exp2f(st[e] - L[t]);
                        float dpe = dp[e];
                        if (p.drop_thresh) dpe = drop_keep(dkey, (uint32_t)(bh * S + query[t]), (uint32_t)key, p.drop_thresh) ? dpe : 0.f;
                        ds[t][j][e] = pv * (dpe - delta[t]);
                    }
                }
            }
            if (kb == nkb - 1) {        // padded keys: their K rows are zero, but dS must be finite for 0 * dS to vanish
#pragma unroll
                for (int t = 0; t < TA_G; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (kb * 32 + j * 16 + 4 * q + e >= S) ds[t][j][e] = 0.f;
            }
            // dQ^T[c][query] += sum_key K^T[c][key] dS^T[key][query]
            Frag<CM> bs[TA_G];
#pragma unroll
            for (int t = 0; t < TA_G; ++t) bs[t] = chain_frag<CM>(ds[t][0], ds[t][1]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const Frag<CM> akt = gather_frag<CM>(Ks, ct * 16 + r, kb * 32, q, SKP - 1, TA_LD);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) mma<CM>(dq[t][ct], akt, bs[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < TA_G; ++t)
            if (query[t] < S) {
                float* o = p.dqkv + ((size_t)b * S + query[t]) * (3 * FD) + h * FDH + 4 * q;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4*>(o + ct * 16) = dq[t][ct] * TA_SCALE;
            }
    }
}

// backward, key side: dK and dV (after the query side: reads delta)
template <int CM>
__global__ __launch_bounds__(TA_MAX_THREADS) void tiled_attn_dkv_kernel(TiledAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = p.S, SKP = (S + 31) & ~31;
    float* Qs = lds;
    float* Os = lds + SKP * TA_LD;          // dO rows
    float* Ls = Os + SKP * TA_LD;           // [SKP] lse
    float* Ds = Ls + SKP;                   // [SKP] delta
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;
    stage_rows(Qs, base, 3 * FD, S, SKP);
    stage_rows(Os, p.d_o + (size_t)b * S * FD + h * FDH, FD, S, SKP, p.drop_thresh ? p.drop_inv : 1.f);
    for (int i = tid; i < SKP; i += blockDim.x) {
        Ls[i] = i < S ? p.lse[(size_t)bh * S + i] : 0.f;
        Ds[i] = i < S ? p.delta[(size_t)bh * S + i] : 0.f;
    }
    __syncthreads();
    const uint64_t dkey = attn_key(p);
    const int nkt = (S + 15) >> 4, nqb = SKP >> 5, ngrp = (nkt + TA_G - 1) / TA_G;
    for (int g = blockIdx.y * nw + wave; g < ngrp; g += nw * gridDim.y) {
        int key[TA_G];
        Frag<CM> bk[TA_G], bv[TA_G];
        f32x4 dv[TA_G][2], dk[TA_G][2];
#pragma unroll
        for (int t = 0; t < TA_G; ++t) {
            key[t] = (g * TA_G + t) * 16 + r;
            const int krow = key[t] < S ? key[t] : S - 1;
            bk[t] = load_frag_scaled<CM>(base + (size_t)krow * (3 * FD) + FD, q, TA_C2);      // (only the scores use K here)
            bv[t] = load_frag<CM>(base + (size_t)krow * (3 * FD) + 2 * FD, q);
            dv[t][0] = dv[t][1] = dk[t][0] = dk[t][1] = f32x4{0, 0, 0, 0};
        }
        for (int qb = 0; qb < nqb; ++qb) {
            f32x4 pn[TA_G][2], dsn[TA_G][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int qt = 2 * qb + j;
                const Frag<CM> aq = load_frag<CM>(Qs + (qt * 16 + r) * TA_LD, q);
                const Frag<CM> ao = load_frag<CM>(Os + (qt * 16 + r) * TA_LD, q);
                const float4 l4 = *reinterpret_cast<const float4*>(Ls + qt * 16 + 4 * q);
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + qt * 16 + 4 * q);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int t = 0; t < TA_G; ++t) {
                    f32x4 sN = f32x4{0, 0, 0, 0}, dN = f32x4{0, 0, 0, 0};
                    mma<CM>(sN, aq, bk[t]);         // S = Q K^T
                    mma<CM>(dN, ao, bv[t]);         // dP = dO V^T
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int query = qt * 16 + 4 * q + e;
                        float pv = __builtin_amdgcn_exp2f(sN[e] - lq[e]);
                        if (qb == nqb - 1) pv = query < S ? pv : 0.f;       // padded queries (zero Q / dO rows, lse 0): P := 0
                        bool kp = true;
                        if (p.drop_thresh) kp = drop_keep(dkey, (uint32_t)(bh * S + query), (uint32_t)key[t], p.drop_thresh);
                        pn[t][j][e] = kp ? pv : 0.f;
                        dsn[t][j][e] = pv * ((kp ? dN[e] : 0.f) - dq4[e]);
                    }
                }
            }
            Frag<CM> bp[TA_G], bs[TA_G];
#pragma unroll
            for (int t = 0; t < TA_G; ++t) { bp[t] = chain_frag<CM>(pn[t][0], pn[t][1]); bs[t] = chain_frag<CM>(dsn[t][0], dsn[t][1]); }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const Frag<CM> aot = gather_frag<CM>(Os, ct * 16 + r, qb * 32, q, SKP - 1, TA_LD);
                const Frag<CM> aqt = gather_frag<CM>(Qs, ct * 16 + r, qb * 32, q, SKP - 1, TA_LD);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) {
                    mma<CM>(dv[t][ct], aot, bp[t]);     // dV^T = dO^T P
                    mma<CM>(dk[t][ct], aqt, bs[t]);     // dK^T = Q^T dS
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TA_G; ++t)
            if (key[t] < S) {
                float* o = p.dqkv + ((size_t)b * S + key[t]) * (3 * FD) + FD + h * FDH + 4 * q;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    *reinterpret_cast<f32x4*>(o + ct * 16) = dk[t][ct] * TA_SCALE;
                    *reinterpret_cast<f32x4*>(o + FD + ct * 16) = dv[t][ct];
                }
            }
    }
}

namespace {
// (clip, head, tile range) workgroups: ONE round on the chip whenever the batch allows it (a second, mostly empty round
// doubles the launch), more ranges when the batch is small (a validation batch is one clip)
int range_split(int B, int ntile, int* nw_out) {
    int s = 256 / (B * FH);
    s = s < 1 ? 1 : s;
    int per = (ntile + s - 1) / s;                          // tiles per workgroup
    int nw = per < 4 ? 4 : (per > TA_MAX_THREADS / 64 ? TA_MAX_THREADS / 64 : per);
    const int maxs = (ntile + nw - 1) / nw;                 // at least one tile per wave
    s = s > maxs ? maxs : s;
    *nw_out = nw;
    return s;
}

template <class K>
int set_lds(K kernel, size_t bytes) {
    EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
constexpr size_t TA_LDS_MAX = (size_t)(2 * TILED_MAX_S * TA_LD + 2 * TILED_MAX_S) * sizeof(float);      // 151.5 KB

template <int CM>
int dispatch(const TiledAttnParams& p, bool bwd, hipStream_t st) {
    EGX_CHECK(p.S >= 1 && p.S <= TILED_MAX_S, "tiled attention: S=%d (1 .. %d)", p.S, TILED_MAX_S);
    static bool once = false;
    if (!once) {
        if (set_lds(&tiled_attn_fwd_kernel<CM>, TA_LDS_MAX) || set_lds(&tiled_attn_dq_kernel<CM>, TA_LDS_MAX) ||
            set_lds(&tiled_attn_dkv_kernel<CM>, TA_LDS_MAX)) return 1;
        once = true;
    }
    const int SKP = (p.S + 31) & ~31;
    int nw = 8;
    dim3 grid(p.B * FH, range_split(p.B, ((p.S + 15) / 16 + TA_G - 1) / TA_G, &nw));
    const dim3 block(nw * 64);
    if (!bwd) {
        const size_t lds = (size_t)(SKP * TA_LD + FDH * (SKP + 4)) * sizeof(float);
        hipLaunchKernelGGL((tiled_attn_fwd_kernel<CM>), grid, block, lds, st, p);
        EGX_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds_a = (size_t)(2 * SKP * TA_LD) * sizeof(float), lds_b = lds_a + (size_t)2 * SKP * sizeof(float);
    hipLaunchKernelGGL((tiled_attn_dq_kernel<CM>), grid, block, lds_a, st, p);
    EGX_LAUNCH_CHECK();
    hipLaunchKernelGGL((tiled_attn_dkv_kernel<CM>), grid, block, lds_b, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}
}  // namespace

int tiled_attn_fwd(const TiledAttnParams& p, int compute, hipStream_t st) {
    EGX_CHECK(compute == CM_BF16 || compute == CM_SPLIT, "tiled attention: compute must be bf16 or f32s");
    timing_begin(TIMER_WIDE_ATTN_FWD, st);
    int rc = compute == CM_BF16 ? dispatch<CM_BF16>(p, false, st) : dispatch<CM_SPLIT>(p, false, st);
    timing_end(TIMER_WIDE_ATTN_FWD, st);
    return rc;
}
int tiled_attn_bwd(const TiledAttnParams& p, int compute, hipStream_t st) {
    EGX_CHECK(compute == CM_BF16 || compute == CM_SPLIT, "tiled attention: compute must be bf16 or f32s");
    timing_begin(TIMER_WIDE_ATTN_BWD, st);
    int rc = compute == CM_BF16 ? dispatch<CM_BF16>(p, true, st) : dispatch<CM_SPLIT>(p, true, st);
    timing_end(TIMER_WIDE_ATTN_BWD, st);
    return rc;
}

}  // namespace egx
