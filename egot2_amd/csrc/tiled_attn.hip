// Attention of the tiled d = 128 path (48 < S <= 512: the reference's real TTM / ASD batches are 15 .. 150 frames per task,
// HHI/dataset/ttm/data_loader_2task.py:119,150-162; validation runs one clip of up to 3 x 150 tokens). Replaces the
// bmm / softmax / dropout / bmm chain inside F.multi_head_attention_forward as called by nn.TransformerEncoderLayer
// (HHI/models/ttm/model_taskspecific.py:211-215) for 4 heads of 32.
//
// One workgroup per (clip, head, range of 16-row tiles); the other operand of the whole clip (K | V, or Q | dO) sits in LDS as
// fp32 rows; a wave owns one 16-query (or 16-key) tile at a time and keeps EVERY score of that tile in accumulator registers
// (S <= 512 -> at most 32 tiles), so the softmax is exact two-pass arithmetic on registers as in the per-clip kernels
// (fused.hip): no online rescaling. Operand convention, compute modes (bf16 / split-bf16 "f32s") and the feature-major chaining
// (S^T = K Q^T puts the probabilities where the next MFMA wants its B operand) are those of fused_dev.h.
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

__device__ __forceinline__ uint64_t attn_key(const TiledAttnParams& p) {
    return p.seed_ptr ? site_key(*p.seed_ptr, (uint32_t)p.layer, SITE_ATTN) : p.drop_key;
}

// rows [0, S) of one 32-wide column block of the clip's Q | K | V grid -> token-major LDS rows, rows >= S zero
template <int SKP>
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int ld_src, int S) {
    for (int i = threadIdx.x; i < SKP * 8; i += 256) {
        const int row = i >> 3, c4 = (i & 7) * 4;
        f32x4 v = f32x4{0, 0, 0, 0};
        if (row < S) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * ld_src + c4);
        *reinterpret_cast<f32x4*>(dst + row * TA_LD + c4) = v;
    }
}
}  // namespace

template <int CM, int NKT>
__global__ __launch_bounds__(256, 1) void tiled_attn_fwd_kernel(TiledAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SKP = NKT * 16, LDVT = SKP + 4;
    float* Ks = lds;                    // [SKP][TA_LD]
    float* Vt = lds + SKP * TA_LD;      // [32][LDVT] V^T, keys >= S zero
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const int S = p.S;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;     // Q of the head; K at + 128, V at + 256
    stage_rows<SKP>(Ks, base + FD, 3 * FD, S);
    for (int i = tid; i < SKP * 8; i += 256) {
        const int row = i >> 3, c4 = (i & 7) * 4;
        f32x4 v = f32x4{0, 0, 0, 0};
        if (row < S) v = *reinterpret_cast<const f32x4*>(base + (size_t)row * (3 * FD) + 2 * FD + c4);
        Vt[(c4 + 0) * LDVT + row] = v[0]; Vt[(c4 + 1) * LDVT + row] = v[1];
        Vt[(c4 + 2) * LDVT + row] = v[2]; Vt[(c4 + 3) * LDVT + row] = v[3];
    }
    __syncthreads();
    const uint64_t dkey = attn_key(p);
    const int nqt = (S + 15) >> 4;
    const int nkb = (nqt + 1) >> 1;             // 32-key K-blocks in use
    for (int qt = blockIdx.y * 4 + wave; qt < nqt; qt += 4 * gridDim.y) {
        const int query = qt * 16 + r;
        const int qrow = query < S ? query : S - 1;         // padded queries recompute the last row; never stored
        const Frag<CM> bq = load_frag<CM>(base + (size_t)qrow * (3 * FD), q);
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            sc[kt] = f32x4{0, 0, 0, 0};
            if (kt < 2 * nkb) mma<CM>(sc[kt], load_frag<CM>(Ks + (kt * 16 + r) * TA_LD, q), bq);
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kt * 16 + 4 * q + e;
                const float s = key < S ? sc[kt][e] * TA_SCALE : -INFINITY;
                sc[kt][e] = s;
                m = fmaxf(m, s);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float pv = __expf(sc[kt][e] - m); sc[kt][e] = pv; sum += pv; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (q == 0 && query < S) p.lse[(size_t)bh * S + query] = m + __logf(sum);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pv = sc[kt][e] * inv;
                if (p.drop_thresh) pv *= drop_scale(dkey, (uint32_t)(bh * S + query), (uint32_t)(kt * 16 + 4 * q + e), p.drop_thresh, p.drop_inv);
                sc[kt][e] = pv;
            }
        f32x4 oc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int kb = 0; kb < NKT / 2; ++kb) {
            if (kb < nkb) {
                const Frag<CM> bp = chain_frag<CM>(sc[2 * kb], sc[2 * kb + 1]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) mma<CM>(oc[ct], load_frag<CM>(Vt + (ct * 16 + r) * LDVT + kb * 32, q), bp);
            }
        }
        if (query < S) {
            float* o = p.attn_o + ((size_t)b * S + query) * FD + h * FDH + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4*>(o + ct * 16) = oc[ct];
        }
    }
}

// backward, query side: dQ and delta
template <int CM, int NKT>
__global__ __launch_bounds__(256, 1) void tiled_attn_dq_kernel(TiledAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SKP = NKT * 16;
    float* Ks = lds;
    float* Vs = lds + SKP * TA_LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const int S = p.S;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;
    stage_rows<SKP>(Ks, base + FD, 3 * FD, S);
    stage_rows<SKP>(Vs, base + 2 * FD, 3 * FD, S);
    __syncthreads();
    const uint64_t dkey = attn_key(p);
    const int nqt = (S + 15) >> 4;
    const int nkb = (nqt + 1) >> 1;
    for (int qt = blockIdx.y * 4 + wave; qt < nqt; qt += 4 * gridDim.y) {
        const int query = qt * 16 + r;
        const int qrow = query < S ? query : S - 1;
        const size_t tok = (size_t)b * S + qrow;
        const Frag<CM> bq = load_frag<CM>(base + (size_t)qrow * (3 * FD), q);
        // dO row and O row of this query: the lane's 8 of the head's 32 channels (the K-block positions of a fragment)
        const float* dop = p.d_o + tok * FD + h * FDH;
        const float* op = p.attn_o + tok * FD + h * FDH;
        const float4 d0 = *reinterpret_cast<const float4*>(dop + 4 * q), d1 = *reinterpret_cast<const float4*>(dop + 16 + 4 * q);
        const float4 o0 = *reinterpret_cast<const float4*>(op + 4 * q), o1 = *reinterpret_cast<const float4*>(op + 16 + 4 * q);
        float delta = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w;
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        const Frag<CM> bdo = make_frag<CM>(d0, d1);
        const float L = p.lse[(size_t)bh * S + qrow];
        if (q == 0 && query < S) p.delta[(size_t)bh * S + query] = delta;
        f32x4 ds[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            ds[kt] = f32x4{0, 0, 0, 0};
            if (kt < 2 * nkb) {
                f32x4 st = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
                mma<CM>(st, load_frag<CM>(Ks + (kt * 16 + r) * TA_LD, q), bq);      // S^T = K Q^T
                mma<CM>(dp, load_frag<CM>(Vs + (kt * 16 + r) * TA_LD, q), bdo);     // dP^T = V dO^T
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kt * 16 + 4 * q + e;
                    const float pv = key < S ? __expf(st[e] * TA_SCALE - L) : 0.f;
                    const float ks = p.drop_thresh ? drop_scale(dkey, (uint32_t)(bh * S + query), (uint32_t)key, p.drop_thresh, p.drop_inv) : 1.f;
                    ds[kt][e] = pv * (ks * dp[e] - delta) * TA_SCALE;
                }
            }
        }
        // dQ^T[c][query] = sum_key K^T[c][key] dS^T[key][query]
        f32x4 dq[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int kb = 0; kb < NKT / 2; ++kb) {
            if (kb < nkb) {
                const Frag<CM> bs = chain_frag<CM>(ds[2 * kb], ds[2 * kb + 1]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) mma<CM>(dq[ct], gather_frag<CM>(Ks, ct * 16 + r, kb * 32, q, SKP - 1, TA_LD), bs);
            }
        }
        if (query < S) {
            float* o = p.dqkv + ((size_t)b * S + query) * (3 * FD) + h * FDH + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4*>(o + ct * 16) = dq[ct];
        }
    }
}

// backward, key side: dK and dV (after the query side: reads delta)
template <int CM, int NKT>
__global__ __launch_bounds__(256, 1) void tiled_attn_dkv_kernel(TiledAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SKP = NKT * 16;
    float* Qs = lds;
    float* Os = lds + SKP * TA_LD;          // dO rows
    float* Ls = Os + SKP * TA_LD;           // [SKP] lse
    float* Ds = Ls + SKP;                   // [SKP] delta
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const int S = p.S;
    const float* base = p.qkv + (size_t)b * p.tpc * 48 * (3 * FD) + h * FDH;
    stage_rows<SKP>(Qs, base, 3 * FD, S);
    stage_rows<SKP>(Os, p.d_o + (size_t)b * S * FD + h * FDH, FD, S);
    for (int i = tid; i < SKP; i += 256) {
        Ls[i] = i < S ? p.lse[(size_t)bh * S + i] : 0.f;
        Ds[i] = i < S ? p.delta[(size_t)bh * S + i] : 0.f;
    }
    __syncthreads();
    const uint64_t dkey = attn_key(p);
    const int nkt = (S + 15) >> 4;
    const int nqb = (nkt + 1) >> 1;             // 32-query K-blocks in use
    for (int kt = blockIdx.y * 4 + wave; kt < nkt; kt += 4 * gridDim.y) {
        const int key = kt * 16 + r;
        const int krow = key < S ? key : S - 1;
        const Frag<CM> bk = load_frag<CM>(base + (size_t)krow * (3 * FD) + FD, q);
        const Frag<CM> bv = load_frag<CM>(base + (size_t)krow * (3 * FD) + 2 * FD, q);
        f32x4 dv[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, dk[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        for (int qb = 0; qb < nqb; ++qb) {
            f32x4 pn[2], dsn[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int qt = 2 * qb + j;
                f32x4 sN = f32x4{0, 0, 0, 0}, dN = f32x4{0, 0, 0, 0};
                mma<CM>(sN, load_frag<CM>(Qs + (qt * 16 + r) * TA_LD, q), bk);     // S = Q K^T
                mma<CM>(dN, load_frag<CM>(Os + (qt * 16 + r) * TA_LD, q), bv);     // dP = dO V^T
                const float4 l4 = *reinterpret_cast<const float4*>(Ls + qt * 16 + 4 * q);
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + qt * 16 + 4 * q);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int query = qt * 16 + 4 * q + e;
                    const bool ok = key < S && query < S;
                    const float pv = ok ? __expf(sN[e] * TA_SCALE - lq[e]) : 0.f;
                    const float ks = p.drop_thresh ? drop_scale(dkey, (uint32_t)(bh * S + query), (uint32_t)key, p.drop_thresh, p.drop_inv) : 1.f;
                    pn[j][e] = pv * ks;
                    dsn[j][e] = pv * (ks * dN[e] - dq4[e]) * TA_SCALE;
                }
            }
            const Frag<CM> bp = chain_frag<CM>(pn[0], pn[1]), bs = chain_frag<CM>(dsn[0], dsn[1]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                mma<CM>(dv[ct], gather_frag<CM>(Os, ct * 16 + r, qb * 32, q, SKP - 1, TA_LD), bp);     // dV^T = dO^T P
                mma<CM>(dk[ct], gather_frag<CM>(Qs, ct * 16 + r, qb * 32, q, SKP - 1, TA_LD), bs);     // dK^T = Q^T dS
            }
        }
        if (key < S) {
            float* o = p.dqkv + ((size_t)b * S + key) * (3 * FD) + FD + h * FDH + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                *reinterpret_cast<f32x4*>(o + ct * 16) = dk[ct];
                *reinterpret_cast<f32x4*>(o + FD + ct * 16) = dv[ct];
            }
        }
    }
}

namespace {
// enough (clip, head, range) workgroups to cover the chip when the batch is small (a validation batch is ONE clip)
int range_split(int B, int ntile) {
    const int per_wg = 4;                                   // tiles a workgroup's four waves take per round
    int want = (256 + B * 4 - 1) / (B * 4);
    int maxs = (ntile + per_wg - 1) / per_wg;
    return want < 1 ? 1 : (want > maxs ? maxs : want);
}

template <class K>
int set_lds(K kernel, size_t bytes) {
    EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

template <int CM, int NKT>
int launch_fwd(const TiledAttnParams& p, hipStream_t st) {
    constexpr int SKP = NKT * 16;
    const size_t lds = (size_t)(SKP * TA_LD + FDH * (SKP + 4)) * sizeof(float);
    static bool once = false;
    if (!once) { if (set_lds(&tiled_attn_fwd_kernel<CM, NKT>, lds)) return 1; once = true; }
    dim3 grid(p.B * FH, range_split(p.B, (p.S + 15) / 16));
    hipLaunchKernelGGL((tiled_attn_fwd_kernel<CM, NKT>), grid, dim3(256), lds, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}
template <int CM, int NKT>
int launch_bwd(const TiledAttnParams& p, hipStream_t st) {
    constexpr int SKP = NKT * 16;
    const size_t lds_a = (size_t)(2 * SKP * TA_LD) * sizeof(float), lds_b = lds_a + (size_t)2 * SKP * sizeof(float);
    static bool once = false;
    if (!once) {
        if (set_lds(&tiled_attn_dq_kernel<CM, NKT>, lds_a) || set_lds(&tiled_attn_dkv_kernel<CM, NKT>, lds_b)) return 1;
        once = true;
    }
    dim3 grid(p.B * FH, range_split(p.B, (p.S + 15) / 16));
    hipLaunchKernelGGL((tiled_attn_dq_kernel<CM, NKT>), grid, dim3(256), lds_a, st, p);
    EGX_LAUNCH_CHECK();
    hipLaunchKernelGGL((tiled_attn_dkv_kernel<CM, NKT>), grid, dim3(256), lds_b, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}
template <int CM>
int dispatch(const TiledAttnParams& p, bool bwd, hipStream_t st) {
    EGX_CHECK(p.S >= 1 && p.S <= TILED_MAX_S, "tiled attention: S=%d (1 .. %d)", p.S, TILED_MAX_S);
    if (p.S <= 128) return bwd ? launch_bwd<CM, 8>(p, st) : launch_fwd<CM, 8>(p, st);
    if (p.S <= 256) return bwd ? launch_bwd<CM, 16>(p, st) : launch_fwd<CM, 16>(p, st);
    return bwd ? launch_bwd<CM, 32>(p, st) : launch_fwd<CM, 32>(p, st);
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
