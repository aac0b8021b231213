// Attention of the tiled d = 128 path (48 < S <= 512: the reference's real TTM / ASD batches are 15 .. 150 frames per task,
// HHI/dataset/ttm/data_loader_2task.py:119,150-162; validation runs one clip of up to 3 x 150 tokens). Replaces the
// bmm / softmax / dropout / bmm chain inside F.multi_head_attention_forward as called by nn.TransformerEncoderLayer
// (HHI/models/ttm/model_taskspecific.py:211-215) for 4 heads of 32.
//
// One workgroup (4 .. 8 waves) per (clip, head, range of 16-row tiles); the other operand of the clip (K | V, or Q | dO) sits in LDS
// as bf16 PLANES (one in bf16 mode, the three exact parts of the f32s mode), converted / split once while it is staged; a wave owns ONE
// group of two 16-query (or 16-key) tiles and walks the 32-row blocks of the LDS operand with two score tiles live (forward: online
// softmax). Operand convention, compute modes and the feature-major chaining (S^T = K Q^T puts the probabilities where the next MFMA
// wants its B operand) are those of fused_dev.h.
//   forward      S^T = K Q^T, P = softmax, O^T = V^T P^T                        -> attn_o (Ntok, 128), lse (B, 4, S)
//   backward A   per query tile: dS^T = P (mask dP^T - delta), dQ^T = K^T dS^T  -> dqkv[:, 0:128], delta (B, 4, S)
//   backward B   per key tile:   dV^T = dO^T P, dK^T = Q^T dS                   -> dqkv[:, 128:384]
// delta = rowsum(dO . O) (equal to sum_k P mask dP). Dropout masks are regenerated from (seed, layer, SITE_ATTN) with the
// generic kernels' keying: row = (clip * 4 + head) * S + query, column = key.
#include "fused.h"
#include "fused_dev.h"

namespace egx {

namespace {
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
// keep decisions of the four columns 4 cq .. 4 cq + 3 of one row: ONE hash, spelled out. (A per-element drop_scale() on a column
// `c0 + kb * 32 + ... + e` with a RUN-TIME chunk offset c0 is four hashes: the compiler cannot know that c0 is a multiple of 4 and does
// not merge them — SQ_INSTS_VALU of the forward and dQ kernels doubled until the quad was made explicit.)
__device__ __forceinline__ void drop_keep4(uint64_t key, uint32_t row, uint32_t cq, uint32_t thresh, bool (&k)[4]) {
    const uint2 h = rand_quad(key, row, cq);
    k[0] = (h.x & 0xffffu) >= thresh; k[1] = (h.x >> 16) >= thresh;
    k[2] = (h.y & 0xffffu) >= thresh; k[3] = (h.y >> 16) >= thresh;
}

__device__ __forceinline__ uint64_t attn_key(const TiledAttnParams& p) {
    return p.seed_ptr ? site_key(*p.seed_ptr, (uint32_t)p.layer, SITE_ATTN) : p.drop_key;
}

}  // namespace

// The three kernels share one shape: NW waves per workgroup, the clip-wide operand in LDS (rows padded to a multiple of 32 with zeros),
// a wave walks the 32-row K-blocks of that operand with a small register footprint (two score tiles live at a time), two waves per
// SIMD: the loops are chains of dependent LDS read -> MFMA -> cross-lane max -> exp -> MFMA steps and heavy in VALU work (softmax,
// dropout hash, the probabilities' conversion / three-way split: 4 cycles per wave64 instruction) that the other wave's MFMAs overlap.
constexpr int TA_MAX_THREADS = 512;
constexpr int TA_G = 2;         // 16-row tiles a wave works on

// Round 5: the clip-wide operand as PRE-SPLIT planes. Rounds 3-4 kept it as fp32 rows and every wave converted (bf16) or three-way
// split (f32s: 44 VALU instructions per fragment) each fragment it read — the same rows again in each of the 8 waves — and gathered
// the transposed operands (K^T, dO^T, Q^T) by eight scalar LDS reads per lane: 43 + 52 + 74 us per step at T = 150
// (profiles/r04_bench_c2_t150_kernel_stats.csv), VALU-bound. Now a row fragment is two 8-byte reads per plane (fused_dev.h
// load_split_frag's shape) and a transposed fragment (keys or queries along K) two ds_read_b64_tr_b16 per plane, no VALU work in
// either: 35 + 39 + 53 us (f32s), 23 + 22 + 30 us against 25 + 27 + 37 (bf16); same products in the same order, bit-identical results.
// Three planes of two operands are 480 B per row: clips beyond 320 rows are walked in CHUNKS of the clip-wide operand (S = 450:
// 256 + 224 rows) while the wave's tiles and their running state stay in registers. Every wave owns ONE group (grid.y = ceil(groups /
// waves)), so all waves meet at the chunk barriers.
// LDS layout: ONE array of combined rows per plane, [rows][80 halfwords] = [operand 0: 32 channels | operand 1: 32 channels | 32 B pad].
// Banking (32 banks of 4 B, 16 consecutive lanes per pass for 8-byte reads): a row stride of 40 dwords = 8 (mod 32) puts the four rows of a
// transposed read's 16-lane group into four disjoint 8-dword windows; the row reads (16 rows, one 8-byte chunk each) would then hit
// only four windows — so the chunk index is XOR-swizzled with bits 2..3 of the row: rows r, r + 4, r + 8, r + 12 (same window) take
// four different chunks of it. (Measured: the same kernel times as separate 80-byte rows per operand without the swizzle — the
// kernels are not LDS-bound; kept because it costs nothing.) The low / high half of a row fragment stay 32 B apart (ds_read2_b64).
namespace {
constexpr int TP_LD = 80;
constexpr int TP_OP = 32;               // halfword offset of operand 1 inside a combined row
template <int CM> struct Npl { static constexpr int v = CM == CM_SPLIT ? 3 : 1; };
// halfword offset of 8-byte chunk `chunk` (0 .. 7: channels 4 chunk .. 4 chunk + 3) of combined row `row`
__device__ __forceinline__ int tp_off(int row, int chunk) { return row * TP_LD + 4 * ((chunk & 4) | ((chunk ^ (row >> 2)) & 3)); }
typedef short tp_s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tp_s4v tp_lds_s4v;

// rows [row0, row0 + rows) of a 32-wide column block (global rows >= S read as zero) -> planes[pl][local row][TP_LD]
template <int CM>
__device__ __forceinline__ void stage_planes(unsigned short* dst, int plane_stride, const float* src, int ld_src, int row0, int rows, int S,
                                             float scale = 1.f) {
    const int n = rows * 8, step = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * step, row = row0 + (i >> 3), c4 = (i & 7) * 4;
            const int rr = row < S ? row : S - 1;           // clamped: unconditional loads, selected below
            v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)rr * ld_src + c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * step, lrow = i >> 3, c4 = (i & 7) * 4;
            if (i < n) {
                const f32x4 w = row0 + lrow < S ? v[u] * scale : f32x4{0, 0, 0, 0};
                unsigned short* d = dst + tp_off(lrow, c4 >> 2);
                if constexpr (CM == CM_SPLIT) {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split_pair(w[0], w[1], h0, m0, l0);
                    split_pair(w[2], w[3], h1, m1, l1);
                    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(d + plane_stride) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(d + 2 * plane_stride) = make_uint2(l0, l1);
                } else {
                    *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]));
                }
            }
        }
    }
}
// A / B fragment of one staged row: channels {4q .. 4q+3, 16+4q .. 16+4q+3} (the K order of load_frag)
template <int CM>
__device__ __forceinline__ Frag<CM> plane_row_frag(const unsigned short* planes, int plane_stride, int row, int q) {
    Frag<CM> f;
#pragma unroll
    for (int pl = 0; pl < Npl<CM>::v; ++pl) {
        const unsigned short* b = planes + pl * plane_stride + tp_off(row, q);
        const uint2 lo = *reinterpret_cast<const uint2*>(b), hi = *reinterpret_cast<const uint2*>(b + 16);
        const bf16x8 w = __builtin_bit_cast(bf16x8, (u32x4){lo.x, lo.y, hi.x, hi.y});
        if constexpr (CM == CM_SPLIT) f.p[pl] = w; else f.v = w;
    }
    return f;
}
// transposed fragment: matrix row = channel c0 + (lane & 15), K = the 32 staged rows row0 + {4q .. 4q+3, 16+4q .. 16+4q+3}
template <int CM>
__device__ __forceinline__ Frag<CM> plane_tr_frag(const unsigned short* planes, int plane_stride, int row0, int c0, int lane) {
    const int i16 = lane & 15, q = lane >> 4;
    const unsigned short* b = planes + tp_off(row0 + 4 * q + (i16 >> 2), (c0 >> 2) + (i16 & 3));      // (row + 16 has the same swizzle)
    Frag<CM> f;
#pragma unroll
    for (int pl = 0; pl < Npl<CM>::v; ++pl) {
        tp_s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tp_lds_s4v*)(b + pl * plane_stride));
        tp_s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tp_lds_s4v*)(b + pl * plane_stride + 16 * TP_LD));
        const bf16x8 w = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        if constexpr (CM == CM_SPLIT) f.p[pl] = w; else f.v = w;
    }
    return f;
}
}  // namespace

template <int CM>
__global__ __launch_bounds__(TA_MAX_THREADS) void tiled_attn_fwd_planes_kernel(TiledAttnParams p, int CH) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = p.S, SKP = (S + 31) & ~31, PS = CH * TP_LD;
    unsigned short* Kp = reinterpret_cast<unsigned short*>(lds);      // [NPL][CH][TP_LD]: K at halfword 0 of a row, V at TP_OP
    unsigned short* Vp = Kp + TP_OP;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;     // Q of the head; K at + 128, V at + 256
    const float vscale = p.drop_thresh ? p.drop_inv : 1.f;
    const uint64_t dkey = attn_key(p);
    const int nqt = (S + 15) >> 4, ngrp = (nqt + TA_G - 1) / TA_G;
    const int g = blockIdx.y * nw + wave;
    const bool active = g < ngrp;
    int query[TA_G];
    Frag<CM> bq[TA_G];
    float m[TA_G], l[TA_G];
    f32x4 oc[TA_G][2];
#pragma unroll
    for (int t = 0; t < TA_G; ++t) {
        query[t] = (g * TA_G + t) * 16 + r;
        const int qrow = query[t] < S ? query[t] : S - 1;       // padded queries (and idle waves) recompute the last row; never stored
        bq[t] = load_frag_scaled<CM>(base + (size_t)qrow * (3 * FD), q, TA_C2);
        m[t] = -INFINITY; l[t] = 0.f;
        oc[t][0] = f32x4{0, 0, 0, 0}; oc[t][1] = f32x4{0, 0, 0, 0};
    }
    for (int c0 = 0; c0 < SKP; c0 += CH) {
        const int rows = min(CH, SKP - c0);
        if (c0) __syncthreads();            // every wave is done with the previous chunk
        stage_planes<CM>(Kp, PS, base + FD, 3 * FD, c0, rows, S);
        stage_planes<CM>(Vp, PS, base + 2 * FD, 3 * FD, c0, rows, S, vscale);
        __syncthreads();
        if (!active) continue;
        const int nkb = rows >> 5;
        for (int kb = 0; kb < nkb; ++kb) {
            const int key0 = c0 + kb * 32;
            f32x4 sc[TA_G][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const Frag<CM> ak = plane_row_frag<CM>(Kp, PS, kb * 32 + j * 16 + r, q);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) { sc[t][j] = f32x4{0, 0, 0, 0}; mma<CM>(sc[t][j], ak, bq[t]); }
            }
            if (key0 + 32 > S) {        // the only block with keys >= S
#pragma unroll
                for (int t = 0; t < TA_G; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (key0 + j * 16 + 4 * q + e >= S) sc[t][j][e] = -INFINITY;
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
                for (int j = 0; j < 2; ++j) {
                    bool kp[4] = {true, true, true, true};
                    if (p.drop_thresh) drop_keep4(dkey, (uint32_t)(bh * S + query[t]), (uint32_t)((key0 >> 2) + j * 4 + q), p.drop_thresh, kp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pv = __builtin_amdgcn_exp2f(sc[t][j][e] - mn);
                        l[t] += pv;
                        sc[t][j][e] = kp[e] ? pv : 0.f;
                    }
                }
                bp[t] = chain_frag<CM>(sc[t][0], sc[t][1]);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const Frag<CM> av = plane_tr_frag<CM>(Vp, PS, kb * 32, ct * 16, lane);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) mma<CM>(oc[t][ct], av, bp[t]);
            }
        }
    }
    if (!active) return;
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

template <int CM>
__global__ __launch_bounds__(TA_MAX_THREADS) void tiled_attn_dq_planes_kernel(TiledAttnParams p, int CH) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = p.S, SKP = (S + 31) & ~31, PS = CH * TP_LD;
    unsigned short* Kp = reinterpret_cast<unsigned short*>(lds);
    unsigned short* Vp = Kp + TP_OP;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;
    const float vscale = p.drop_thresh ? p.drop_inv : 1.f;
    const uint64_t dkey = attn_key(p);
    const int nqt = (S + 15) >> 4, ngrp = (nqt + TA_G - 1) / TA_G;
    const int g = blockIdx.y * nw + wave;
    const bool active = g < ngrp;
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
        if (active && q == 0 && query[t] < S) p.delta[(size_t)bh * S + query[t]] = dl;
        dq[t][0] = f32x4{0, 0, 0, 0}; dq[t][1] = f32x4{0, 0, 0, 0};
    }
    for (int c0 = 0; c0 < SKP; c0 += CH) {
        const int rows = min(CH, SKP - c0);
        if (c0) __syncthreads();
        stage_planes<CM>(Kp, PS, base + FD, 3 * FD, c0, rows, S);
        stage_planes<CM>(Vp, PS, base + 2 * FD, 3 * FD, c0, rows, S, vscale);
        __syncthreads();
        if (!active) continue;
        const int nkb = rows >> 5;
        for (int kb = 0; kb < nkb; ++kb) {
            const int key0 = c0 + kb * 32;
            f32x4 ds[TA_G][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const Frag<CM> ak = plane_row_frag<CM>(Kp, PS, kb * 32 + j * 16 + r, q);
                const Frag<CM> av = plane_row_frag<CM>(Vp, PS, kb * 32 + j * 16 + r, q);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) {
                    f32x4 st = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
                    mma<CM>(st, ak, bq[t]);         // S^T = K Q^T
                    mma<CM>(dp, av, bdo[t]);        // dP^T = V dO^T
                    bool kp[4] = {true, true, true, true};
                    if (p.drop_thresh) drop_keep4(dkey, (uint32_t)(bh * S + query[t]), (uint32_t)((key0 >> 2) + j * 4 + q), p.drop_thresh, kp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pv = __builtin_amdgcn_exp2f(st[e] - L[t]);
                        ds[t][j][e] = pv * ((kp[e] ? dp[e] : 0.f) - delta[t]);
                    }
                }
            }
            if (key0 + 32 > S) {        // padded keys: their K rows are zero, but dS must be finite for 0 * dS to vanish
#pragma unroll
                for (int t = 0; t < TA_G; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (key0 + j * 16 + 4 * q + e >= S) ds[t][j][e] = 0.f;
            }
            Frag<CM> bs[TA_G];
#pragma unroll
            for (int t = 0; t < TA_G; ++t) bs[t] = chain_frag<CM>(ds[t][0], ds[t][1]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const Frag<CM> akt = plane_tr_frag<CM>(Kp, PS, kb * 32, ct * 16, lane);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) mma<CM>(dq[t][ct], akt, bs[t]);
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int t = 0; t < TA_G; ++t)
        if (query[t] < S) {
            float* o = p.dqkv + ((size_t)b * S + query[t]) * (3 * FD) + h * FDH + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4*>(o + ct * 16) = dq[t][ct] * TA_SCALE;
        }
}

template <int CM>
__global__ __launch_bounds__(TA_MAX_THREADS) void tiled_attn_dkv_planes_kernel(TiledAttnParams p, int CH) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = p.S, SKP = (S + 31) & ~31, PS = CH * TP_LD;
    unsigned short* Qp = reinterpret_cast<unsigned short*>(lds);
    unsigned short* Op = Qp + TP_OP;                    // dO rows (x 1 / (1 - p))
    float* Ls = reinterpret_cast<float*>(Qp + Npl<CM>::v * PS);       // [CH] lse
    float* Ds = Ls + CH;                                              // [CH] delta
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, nw = blockDim.x >> 6;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;
    const float oscale = p.drop_thresh ? p.drop_inv : 1.f;
    const uint64_t dkey = attn_key(p);
    const int nkt = (S + 15) >> 4, ngrp = (nkt + TA_G - 1) / TA_G;
    const int g = blockIdx.y * nw + wave;
    const bool active = g < ngrp;
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
    for (int c0 = 0; c0 < SKP; c0 += CH) {
        const int rows = min(CH, SKP - c0);
        if (c0) __syncthreads();
        stage_planes<CM>(Qp, PS, base, 3 * FD, c0, rows, S);
        stage_planes<CM>(Op, PS, p.d_o + (size_t)b * S * FD + h * FDH, FD, c0, rows, S, oscale);
        for (int i = tid; i < rows; i += blockDim.x) {
            Ls[i] = c0 + i < S ? p.lse[(size_t)bh * S + c0 + i] : 0.f;
            Ds[i] = c0 + i < S ? p.delta[(size_t)bh * S + c0 + i] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        const int nqb = rows >> 5;
        for (int qb = 0; qb < nqb; ++qb) {
            f32x4 pn[TA_G][2], dsn[TA_G][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int qt = 2 * qb + j;
                const Frag<CM> aq = plane_row_frag<CM>(Qp, PS, qt * 16 + r, q);
                const Frag<CM> ao = plane_row_frag<CM>(Op, PS, qt * 16 + r, q);
                const float4 l4 = *reinterpret_cast<const float4*>(Ls + qt * 16 + 4 * q);
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + qt * 16 + 4 * q);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int t = 0; t < TA_G; ++t) {
                    f32x4 sN = f32x4{0, 0, 0, 0}, dN = f32x4{0, 0, 0, 0};
                    mma<CM>(sN, aq, bk[t]);         // S = Q K^T
                    mma<CM>(dN, ao, bv[t]);         // dP = dO V^T
                    // dropout mask: the lane's four elements are four QUERIES (mask rows) of one key: one hash per lane and tile, exchanged
                    // inside the quad (common.h tile_keep_rows) instead of four
                    uint32_t m4[4] = {0xFu, 0xFu, 0xFu, 0xFu};
                    if (p.drop_thresh) tile_keep_rows(dkey, (uint32_t)(bh * S + c0 + qt * 16), (uint32_t)((g * TA_G + t) * 4), r, q, p.drop_thresh, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int query = c0 + qt * 16 + 4 * q + e;
                        float pv = __builtin_amdgcn_exp2f(sN[e] - lq[e]);
                        if (c0 + qb * 32 + 32 > S) pv = query < S ? pv : 0.f;       // padded queries (zero Q / dO rows, lse 0): P := 0
                        const bool kp = (m4[e] >> (r & 3)) & 1u;
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
                const Frag<CM> aot = plane_tr_frag<CM>(Op, PS, qb * 32, ct * 16, lane);
                const Frag<CM> aqt = plane_tr_frag<CM>(Qp, PS, qb * 32, ct * 16, lane);
#pragma unroll
                for (int t = 0; t < TA_G; ++t) {
                    mma<CM>(dv[t][ct], aot, bp[t]);     // dV^T = dO^T P
                    mma<CM>(dk[t][ct], aqt, bs[t]);     // dK^T = Q^T dS
                }
            }
        }
    }
    if (!active) return;
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

// rows of the clip-wide operand per chunk: NPL planes x 160 B per combined row (+ lse / delta) within 157 KB
template <int CM>
int planes_chunk_rows(int SKP) {
    const int row_bytes = Npl<CM>::v * TP_LD * 2 + 8;
    const int max_rows = (157 * 1024 / row_bytes) & ~31;       // f32s: 320 rows, bf16: the whole clip
    const int nchunks = (SKP + max_rows - 1) / max_rows;
    return (((SKP + nchunks - 1) / nchunks) + 31) & ~31;
}
template <int CM>
int dispatch_planes(const TiledAttnParams& p, bool bwd, hipStream_t st) {
    EGX_CHECK(p.S >= 1 && p.S <= TILED_MAX_S, "tiled attention: S=%d (1 .. %d)", p.S, TILED_MAX_S);
    static bool once = false;
    if (!once) {
        const size_t mx = 158 * 1024;
        if (set_lds(&tiled_attn_fwd_planes_kernel<CM>, mx) || set_lds(&tiled_attn_dq_planes_kernel<CM>, mx) ||
            set_lds(&tiled_attn_dkv_planes_kernel<CM>, mx)) return 1;
        once = true;
    }
    const int SKP = (p.S + 31) & ~31, CH = planes_chunk_rows<CM>(SKP);
    const int ngrp = ((p.S + 15) / 16 + TA_G - 1) / TA_G;
    // every wave owns ONE group: waves per workgroup as range_split() picks them, ranges = what covers the groups
    int nw = 8;
    (void)range_split(p.B, ngrp, &nw);
    const dim3 grid(p.B * FH, (ngrp + nw - 1) / nw), block(nw * 64);
    const size_t lds_a = (size_t)Npl<CM>::v * CH * TP_LD * 2, lds_b = lds_a + (size_t)2 * CH * sizeof(float);
    if (!bwd) {
        hipLaunchKernelGGL((tiled_attn_fwd_planes_kernel<CM>), grid, block, lds_a, st, p, CH);
        EGX_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL((tiled_attn_dq_planes_kernel<CM>), grid, block, lds_a, st, p, CH);
    EGX_LAUNCH_CHECK();
    hipLaunchKernelGGL((tiled_attn_dkv_planes_kernel<CM>), grid, block, lds_b, st, p, CH);
    EGX_LAUNCH_CHECK();
    return 0;
}
}  // namespace

int tiled_attn_fwd(const TiledAttnParams& p, int compute, hipStream_t st) {
    EGX_CHECK(compute == CM_BF16 || compute == CM_SPLIT, "tiled attention: compute must be bf16 or f32s");
    timing_begin(TIMER_WIDE_ATTN_FWD, st);
    int rc = compute == CM_BF16 ? dispatch_planes<CM_BF16>(p, false, st) : dispatch_planes<CM_SPLIT>(p, false, st);
    timing_end(TIMER_WIDE_ATTN_FWD, st);
    return rc;
}
int tiled_attn_bwd(const TiledAttnParams& p, int compute, hipStream_t st) {
    EGX_CHECK(compute == CM_BF16 || compute == CM_SPLIT, "tiled attention: compute must be bf16 or f32s");
    timing_begin(TIMER_WIDE_ATTN_BWD, st);
    int rc = compute == CM_BF16 ? dispatch_planes<CM_BF16>(p, true, st) : dispatch_planes<CM_SPLIT>(p, true, st);
    timing_end(TIMER_WIDE_ATTN_BWD, st);
    return rc;
}

}  // namespace egx
