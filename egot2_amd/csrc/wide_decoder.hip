// The EgoT2-g sequence decoder + vocabulary head as ONE call per direction:
// decode() of HHI/models/multitask/task_prompt_model.py:260-269 and HOI/models/multitask/video_model_builder.py:150-159 —
// `embedding(y) * sqrt(d)` + positional encoding -> nn.TransformerDecoder of CustomDecoderLayer (post-LN: causal
// self-attention over the 2..8 target tokens, cross-attention onto the S <= 1024 memory tokens of the clip, ReLU FFN) -> `fc`.
//
// Round 2 composed it from ~40 autograd functions per layer (60 small fp32 GEMMs, 75 zero-fills and 41 copies per step:
// launch-bound, 60 % of the EgoT2-g step). Here the whole stack is orchestrated in C++ like the wide encoder
// (wide_host.hip): all B * sy target rows go through the bf16 MFMA GEMMs with fused epilogues (bias, ReLU, dropout,
// residual, column sums), LayerNorms through the row kernels with bf16 side outputs, the two tiny attentions through a
// register-resident kernel (one wave per (clip, head)), every gradient accumulates into ONE caller-provided flat buffer
// that the first memset zeroes, and the split-K slabs of all weight gradients are summed by one launch at the end.
// Supported: compute = bf16, d_model a multiple of 128 in [256, 1024], head dim 32 or 64, sy <= 8, S <= 1024 (one wave per (clip, head) up to 64 memory tokens, a chunked four-wave kernel beyond); anything else
// stays on the composed path (egot2_amd/decoder.py).
#include <string.h>
#include <mutex>

#include "../../include/egot2x.h"
#include "common.h"
#include "kernels.h"
#include "wide.h"

namespace egx {

namespace {

constexpr int DA_MAXQ = 8, DA_MAXK = 64;

__device__ __forceinline__ float wmax64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wsum64d(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float bf1(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

struct DecAttnParams {
    const void* q; const void* k; const void* v; int ldq, ldk, ldv;           // rows b * Sq + i / b * Sk + j; head h at columns h * DH
    bf16_t* o; int ldo;                                                        // (q / k / v and their gradients: bf16, or fp32 with F32 = true)
    const bf16_t* d_o; void* dq; void* dk; void* dv;                           // backward (same strides as o / q / k / v)
    int B, H, Sq, Sk, causal;
    float scale;
    uint64_t drop_key; uint32_t drop_thresh; float drop_inv;
};

// One wave per (clip, head), four per workgroup. Lane j holds key row j (and, in the backward, value row j) in registers,
// lane c holds column c of V (forward) / of K and dO (backward); the Sq <= 8 query rows and the probabilities go through a
// per-wave LDS slice as broadcast reads. Probabilities are recomputed in the backward (nothing saved).
// 8 consecutive elements of a q / k / v row as fp32
template <bool F32>
__device__ __forceinline__ void load8(const void* base, size_t idx, float (&o)[8]) {
    if constexpr (F32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + idx);
        o[0] = bf_lo(u.x); o[1] = bf_hi(u.x); o[2] = bf_lo(u.y); o[3] = bf_hi(u.y);
        o[4] = bf_lo(u.z); o[5] = bf_hi(u.z); o[6] = bf_lo(u.w); o[7] = bf_hi(u.w);
    }
}
template <bool F32>
__device__ __forceinline__ float load1(const void* base, size_t idx) {
    if constexpr (F32) return reinterpret_cast<const float*>(base)[idx];
    else return bf1(reinterpret_cast<const bf16_t*>(base)[idx]);
}
template <bool F32>
__device__ __forceinline__ void store1(void* base, size_t idx, float v) {
    if constexpr (F32) reinterpret_cast<float*>(base)[idx] = v;
    else reinterpret_cast<bf16_t*>(base)[idx] = f2bf(v);
}
template <bool F32>
__device__ __forceinline__ void store8(void* base, size_t idx, const float (&a)[8]) {
    if constexpr (F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx + 4) = make_float4(a[4], a[5], a[6], a[7]);
    } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(base) + idx) =
            make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7]));
    }
}

template <int DH, bool BWD, bool F32>
__global__ __launch_bounds__(256) void dec_attn_kernel(DecAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    __shared__ float sQ[4][DA_MAXQ][DH];        // query rows (fp32)
    __shared__ float sG[4][DA_MAXQ][DH];        // dO rows (backward)
    __shared__ float sP[4][DA_MAXQ][64];        // probabilities after dropout
    __shared__ float sD[4][DA_MAXQ][64];        // dS (backward)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= p.B * p.H) return;                // (no barrier below: every wave works alone)
    const int b = bh / p.H, h = bh % p.H;
    const int Sq = p.Sq, Sk = p.Sk;
    const bool kv_lane = lane < Sk;
    // key row of this lane
    float kr[DH], vr[BWD ? DH : 1];
    {
        const size_t krow = ((size_t)b * Sk + (kv_lane ? lane : 0)) * p.ldk + h * DH;
#pragma unroll
        for (int c = 0; c < DH; c += 8) {
            float t8[8];
            load8<F32>(p.k, krow + c, t8);
#pragma unroll
            for (int e = 0; e < 8; ++e) kr[c + e] = t8[e];
        }
        if constexpr (BWD) {
            const size_t vrow = ((size_t)b * Sk + (kv_lane ? lane : 0)) * p.ldv + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 8) {
                float t8[8];
                load8<F32>(p.v, vrow + c, t8);
#pragma unroll
                for (int e = 0; e < 8; ++e) vr[c + e] = t8[e];
            }
        }
    }
    // query (and dO) rows -> LDS
    for (int i = lane; i < Sq * DH; i += 64) {
        const int r = i / DH, c = i - r * DH;
        sQ[wave][r][c] = load1<F32>(p.q, ((size_t)b * Sq + r) * p.ldq + h * DH + c);
        if constexpr (BWD) sG[wave][r][c] = bf1(p.d_o[((size_t)b * Sq + r) * p.ldo + h * DH + c]);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float ds_reg[BWD ? DA_MAXQ : 1];            // dS[i][lane]
#pragma unroll
    for (int i = 0; i < DA_MAXQ; ++i) {         // (static indices: a runtime-indexed register array would live in scratch)
        if (i >= Sq) break;
        const bool live = kv_lane && !(p.causal && lane > i);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DH; c += 4) {
            const float4 qv = *reinterpret_cast<const float4*>(&sQ[wave][i][c]);
            s += (qv.x * kr[c] + qv.y * kr[c + 1]) + (qv.z * kr[c + 2] + qv.w * kr[c + 3]);
        }
        s = live ? s * p.scale : -INFINITY;
        const float m = wmax64(s);
        const float e = live ? __expf(s - m) : 0.f;
        const float prob = e / wsum64d(e);
        float mask = 1.f;
        if (p.drop_thresh) mask = drop_scale(dkey, (uint32_t)(bh * DA_MAXQ + i), (uint32_t)lane, p.drop_thresh, p.drop_inv);
        sP[wave][i][lane] = prob * mask;
        if constexpr (BWD) {
            float dp = 0.f;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const float4 gv = *reinterpret_cast<const float4*>(&sG[wave][i][c]);
                dp += (gv.x * vr[c] + gv.y * vr[c + 1]) + (gv.z * vr[c + 2] + gv.w * vr[c + 3]);
            }
            dp = live ? dp * mask : 0.f;
            const float delta = wsum64d(prob * dp);
            const float dsv = live ? prob * (dp - delta) * p.scale : 0.f;
            ds_reg[i] = dsv;
            sD[wave][i][lane] = dsv;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!BWD) {
        // lane c: O[i][c] = sum_j P[i][j] V[j][c]
        if (lane < DH) {
            float acc[DA_MAXQ];
#pragma unroll
            for (int i = 0; i < DA_MAXQ; ++i) acc[i] = 0.f;
            const size_t v0 = (size_t)b * Sk * p.ldv + h * DH + lane;
            for (int j = 0; j < Sk; ++j) {
                const float vv = load1<F32>(p.v, v0 + (size_t)j * p.ldv);
#pragma unroll
                for (int i = 0; i < DA_MAXQ; ++i)
                    if (i < Sq) acc[i] += sP[wave][i][j] * vv;
            }
#pragma unroll
            for (int i = 0; i < DA_MAXQ; ++i)
                if (i < Sq) p.o[((size_t)b * Sq + i) * p.ldo + h * DH + lane] = f2bf(acc[i]);
        }
    } else {
        // lane j: dK[j][:] = sum_i dS[i][j] Q[i][:]  (row store)
        if (kv_lane) {
            const size_t dk0 = ((size_t)b * Sk + lane) * p.ldk + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 8) {
                float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < DA_MAXQ; ++i) {
                    if (i >= Sq) break;
                    const float4 q0 = *reinterpret_cast<const float4*>(&sQ[wave][i][c]);
                    const float4 q1 = *reinterpret_cast<const float4*>(&sQ[wave][i][c + 4]);
                    const float dsv = ds_reg[i];
                    a[0] += dsv * q0.x; a[1] += dsv * q0.y; a[2] += dsv * q0.z; a[3] += dsv * q0.w;
                    a[4] += dsv * q1.x; a[5] += dsv * q1.y; a[6] += dsv * q1.z; a[7] += dsv * q1.w;
                }
                store8<F32>(p.dk, dk0 + c, a);
            }
        }
        // lane c: dQ[i][c] = sum_j dS[i][j] K[j][c];  dV[j][c] = sum_i P[i][j] dO[i][c]
        if (lane < DH) {
            float accq[DA_MAXQ], go[DA_MAXQ];
#pragma unroll
            for (int i = 0; i < DA_MAXQ; ++i) { accq[i] = 0.f; go[i] = i < Sq ? sG[wave][i][lane] : 0.f; }
            const size_t k0 = (size_t)b * Sk * p.ldk + h * DH + lane, dv0 = (size_t)b * Sk * p.ldv + h * DH + lane;
            for (int j = 0; j < Sk; ++j) {
                const float kk = load1<F32>(p.k, k0 + (size_t)j * p.ldk);
                float av = 0.f;
#pragma unroll
                for (int i = 0; i < DA_MAXQ; ++i)
                    if (i < Sq) { accq[i] += sD[wave][i][j] * kk; av += sP[wave][i][j] * go[i]; }
                store1<F32>(p.dv, dv0 + (size_t)j * p.ldv, av);
            }
#pragma unroll
            for (int i = 0; i < DA_MAXQ; ++i)
                if (i < Sq) store1<F32>(p.dq, ((size_t)b * Sq + i) * p.ldq + h * DH + lane, accq[i]);
        }
    }
}

// Cross-attention onto a memory of more than 64 tokens (EgoT2-g HHI on real-length TTM / ASD sequences: up to 3 x 150): one
// workgroup of four waves per (clip, head); K and V stream through ONE 64-row LDS buffer (fp32, padded rows) chunk by chunk,
// all Sq x Sk probabilities stay in LDS. bf16 operands and gradients; same dropout keying as dec_attn_kernel.
constexpr int DAL_MAXK = 1024;
template <int DH, bool BWD>
__global__ __launch_bounds__(256) void dec_attn_long_kernel(DecAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    constexpr int LDK = DH + 1;
    extern __shared__ float dal_sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int Sq = p.Sq, Sk = p.Sk, SKP = (Sk + 63) & ~63;
    float* Cs = dal_sm;                     // K or V chunk [64][DH + 1]
    float* Qs = Cs + 64 * LDK;
    float* Gs = Qs + DA_MAXQ * DH;
    float* Ps = Gs + DA_MAXQ * DH;          // [Sq][SKP]
    float* Ds = Ps + DA_MAXQ * SKP;         // backward only
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * Sk * p.ldk + h * DH;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * Sk * p.ldv + h * DH;
    auto stage = [&](const bf16_t* src, int ld, int j0) {
        __syncthreads();
        for (int i = tid; i < 64 * (DH / 8); i += 256) {
            const int j = i / (DH / 8), c = (i - j * (DH / 8)) * 8;
            float t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (j0 + j < Sk) load8<false>(src, (size_t)(j0 + j) * ld + c, t8);
#pragma unroll
            for (int e = 0; e < 8; ++e) Cs[j * LDK + c + e] = t8[e];
        }
        __syncthreads();
    };
    for (int i = tid; i < Sq * DH; i += 256) {
        const int r = i / DH, c = i - r * DH;
        Qs[i] = load1<false>(p.q, ((size_t)b * Sq + r) * p.ldq + h * DH + c);
        if constexpr (BWD) Gs[i] = bf1(p.d_o[((size_t)b * Sq + r) * p.ldo + h * DH + c]);
    }
    for (int j0 = 0; j0 < Sk; j0 += 64) {          // scores: wave w owns queries w, w + 4; lane = key within the chunk
        stage(kb, p.ldk, j0);
        for (int i = wave; i < Sq; i += 4) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) sc += Qs[i * DH + c] * Cs[lane * LDK + c];
            Ps[i * SKP + j0 + lane] = j0 + lane < Sk ? sc * p.scale : -INFINITY;
        }
    }
    __syncthreads();
    for (int i = wave; i < Sq; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < SKP; j += 64) m = fmaxf(m, Ps[i * SKP + j]);
        m = wmax64(m);
        float sum = 0.f;
        for (int j = lane; j < SKP; j += 64) { const float e = __expf(Ps[i * SKP + j] - m); Ps[i * SKP + j] = e; sum += e; }
        sum = 1.f / wsum64d(sum);
        for (int j = lane; j < SKP; j += 64) Ps[i * SKP + j] *= sum;
    }
    auto keep = [&](int i, int j) { return p.drop_thresh ? drop_scale(dkey, (uint32_t)(bh * DA_MAXQ + i), (uint32_t)j, p.drop_thresh, p.drop_inv) : 1.f; };
    constexpr int NE = DA_MAXQ * DH / 256;          // (query, column) accumulators per thread: 1 (DH = 32) or 2 (DH = 64)
    float acc[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) acc[u] = 0.f;
    if constexpr (BWD) {
        for (int j0 = 0; j0 < Sk; j0 += 64) {      // dP = dO V^T
            stage(vb, p.ldv, j0);
            for (int i = wave; i < Sq; i += 4) {
                float dp = 0.f;
#pragma unroll
                for (int c = 0; c < DH; ++c) dp += Gs[i * DH + c] * Cs[lane * LDK + c];
                Ds[i * SKP + j0 + lane] = dp;
            }
        }
        __syncthreads();
        for (int i = wave; i < Sq; i += 4) {
            float delta = 0.f;
            for (int j = lane; j < SKP; j += 64) {
                const float dp = Ds[i * SKP + j] * keep(i, j);
                Ds[i * SKP + j] = dp;
                delta += Ps[i * SKP + j] * dp;
            }
            delta = wsum64d(delta);
            for (int j = lane; j < SKP; j += 64) {
                const float pr = Ps[i * SKP + j];
                Ds[i * SKP + j] = pr * (Ds[i * SKP + j] - delta) * p.scale;
                Ps[i * SKP + j] = pr * keep(i, j);
            }
        }
        bf16_t* dkb = reinterpret_cast<bf16_t*>(p.dk) + (size_t)b * Sk * p.ldk + h * DH;
        bf16_t* dvb = reinterpret_cast<bf16_t*>(p.dv) + (size_t)b * Sk * p.ldv + h * DH;
        for (int j0 = 0; j0 < Sk; j0 += 64) {      // dQ += dS K; dK = dS^T Q, dV = P^T dO for the chunk's keys
            stage(kb, p.ldk, j0);
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int e = tid + u * 256, i = e / DH, c = e - i * DH;
                if (i < Sq) {
                    float a = 0.f;
                    for (int j = 0; j < 64; ++j) a += Ds[i * SKP + j0 + j] * Cs[j * LDK + c];
                    acc[u] += a;
                }
            }
            for (int e = tid; e < 64 * (DH / 8); e += 256) {
                const int j = e / (DH / 8), c = (e - j * (DH / 8)) * 8;
                if (j0 + j < Sk) {
                    float ak[8] = {0, 0, 0, 0, 0, 0, 0, 0}, av[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int i = 0; i < Sq; ++i) {
                        const float dsv = Ds[i * SKP + j0 + j], pv = Ps[i * SKP + j0 + j];
#pragma unroll
                        for (int x = 0; x < 8; ++x) { ak[x] += dsv * Qs[i * DH + c + x]; av[x] += pv * Gs[i * DH + c + x]; }
                    }
                    store8<false>(dkb, (size_t)(j0 + j) * p.ldk + c, ak);
                    store8<false>(dvb, (size_t)(j0 + j) * p.ldv + c, av);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + u * 256, i = e / DH, c = e - i * DH;
            if (i < Sq) store1<false>(p.dq, ((size_t)b * Sq + i) * p.ldq + h * DH + c, acc[u]);
        }
    } else {
        if (p.drop_thresh) {
            for (int i = wave; i < Sq; i += 4)
                for (int j = lane; j < SKP; j += 64) Ps[i * SKP + j] *= keep(i, j);
        }
        for (int j0 = 0; j0 < Sk; j0 += 64) {
            stage(vb, p.ldv, j0);
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int e = tid + u * 256, i = e / DH, c = e - i * DH;
                if (i < Sq) {
                    float a = 0.f;
                    for (int j = 0; j < 64; ++j) a += Ps[i * SKP + j0 + j] * Cs[j * LDK + c];
                    acc[u] += a;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + u * 256, i = e / DH, c = e - i * DH;
            if (i < Sq) p.o[((size_t)b * Sq + i) * p.ldo + h * DH + c] = f2bf(acc[u]);
        }
    }
}
template <int DH, bool BWD>
int dec_attn_long(const DecAttnParams& p, hipStream_t st) {
    auto lds = [](int Sk) { return ((size_t)64 * (DH + 1) + (size_t)2 * DA_MAXQ * DH + (size_t)2 * DA_MAXQ * ((Sk + 63) & ~63)) * sizeof(float); };
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_attn_long_kernel<DH, BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(DAL_MAXK)));
        attr = true;
    }
    hipLaunchKernelGGL((dec_attn_long_kernel<DH, BWD>), dim3(p.B * p.H), dim3(256), lds(p.Sk), st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

template <bool BWD>
int dec_attn(DecAttnParams p, int dh, bool f32, hipStream_t st) {
    p.scale = 1.f / sqrtf((float)dh);
    if (p.Sk > DA_MAXK) {
        EGX_CHECK(!f32 && !p.causal && (dh == 32 || dh == 64), "decoder attention: a memory of %d tokens needs bf16 operands and head dim 32 / 64", p.Sk);
        return dh == 64 ? dec_attn_long<64, BWD>(p, st) : dec_attn_long<32, BWD>(p, st);
    }
    const dim3 grid(cdiv(p.B * p.H, 4)), block(256);
    if (dh == 64 && !f32) hipLaunchKernelGGL((dec_attn_kernel<64, BWD, false>), grid, block, 0, st, p);
    else if (dh == 32 && !f32) hipLaunchKernelGGL((dec_attn_kernel<32, BWD, false>), grid, block, 0, st, p);
    else if (dh == 64) hipLaunchKernelGGL((dec_attn_kernel<64, BWD, true>), grid, block, 0, st, p);
    else if (dh == 32) hipLaunchKernelGGL((dec_attn_kernel<32, BWD, true>), grid, block, 0, st, p);
    else EGX_CHECK(false, "decoder attention: head dim %d (32 or 64)", dh);
    EGX_LAUNCH_CHECK();
    return 0;
}

// x32 / x16 [row] = dropout(emb[tok[row]] * scale + pe[row % sy])
__global__ __launch_bounds__(256) void dec_embed_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pe,
                                                        int pe_stride, float scale, float* __restrict__ x32, bf16_t* __restrict__ x16, int rows,
                                                        int sy, int d, int V, uint64_t key, uint32_t thresh, float inv) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one float4 each
    if (i >= (size_t)rows * (d / 4)) return;
    const int row = (int)(i / (d / 4)), c = (int)(i % (d / 4)) * 4;
    const int64_t t = tok[row];
    float4 e = make_float4(0, 0, 0, 0);
    if (t >= 0 && t < V) e = *reinterpret_cast<const float4*>(emb + (size_t)t * d + c);
    const float4 pp = *reinterpret_cast<const float4*>(pe + (size_t)(row % sy) * pe_stride + c);
    float o[4] = {e.x * scale + pp.x, e.y * scale + pp.y, e.z * scale + pp.z, e.w * scale + pp.w};
    if (thresh) {
        float ds[4];
        drop_scale4(resolve_key(key), (uint32_t)row, (uint32_t)c, thresh, inv, ds);
        o[0] *= ds[0]; o[1] *= ds[1]; o[2] *= ds[2]; o[3] *= ds[3];
    }
    *reinterpret_cast<float4*>(x32 + (size_t)row * d + c) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint2*>(x16 + (size_t)row * d + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
}
// d_emb[v][c] += scale * sum over the rows whose token is v of mask * dy[row][c]: one workgroup per (v, 64 columns), four
// row groups summed through LDS in a fixed order (deterministic; replaces one atomic add per element onto a handful of
// vocabulary rows)
__global__ __launch_bounds__(256) void dec_embed_grad_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dy, float* __restrict__ d_emb,
                                                             float scale, int rows, int d, int V, uint64_t key, uint32_t thresh, float inv) {
    __shared__ float part[4][64];
    const int v = blockIdx.y, lane = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (c < d) {
        for (int row = grp; row < rows; row += 4) {
            if (tok[row] != v) continue;                                 // wave-uniform
            float g = dy[(size_t)row * d + c];
            if (thresh) g *= drop_scale(resolve_key(key), (uint32_t)row, (uint32_t)c, thresh, inv);
            acc += g;
        }
    }
    part[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && c < d) d_emb[(size_t)v * d + c] += ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * scale;
}

// Small vocabularies (V <= 64: the EgoT2-g task vocabularies): one workgroup per 64 columns walks ALL rows once, every row's
// gradient is loaded unconditionally (eight rows in flight per wave) and added to the LDS accumulator row of its token; four
// up to 16 waves take a slice of the rows each and are summed in wave order (deterministic). The per-(token, column) kernel above
// scans the token list once per vocabulary entry with a dependent load per match: 43-61 us at B * sy = 512 against 6 here.
constexpr int EMB_VMAX = 64;
__global__ __launch_bounds__(1024) void dec_embed_grad_small_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dy, float* __restrict__ d_emb,
                                                                    float scale, int rows, int d, int V, uint64_t key, uint32_t thresh, float inv) {
    extern __shared__ float acc[];          // [G row groups][V][64], G = blockDim.x / 64 (as many as 64 KB hold, at most 16)
    const int G = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    float* mine = acc + (size_t)grp * V * 64;
    for (int v = 0; v < V; ++v) mine[v * 64 + lane] = 0.f;
    const int cc = c < d ? c : d - 1;
    const int per = (rows + G - 1) / G, r0 = grp * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int row = r0; row < r1; row += 8) {
        float g[8];
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = row + u < r1 ? row + u : r1 - 1;
            g[u] = dy[(size_t)rr * d + cc];
            t[u] = (int)tok[rr];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (row + u >= r1) break;
            float gv = g[u];
            if (thresh) gv *= drop_scale(resolve_key(key), (uint32_t)(row + u), (uint32_t)cc, thresh, inv);
            if (t[u] >= 0 && t[u] < V) mine[t[u] * 64 + lane] += gv;      // lane-private column: no conflict, row order kept
        }
    }
    __syncthreads();
    if (c < d)
        for (int v = grp; v < V; v += G) {
            float sum = 0.f;
            for (int k = 0; k < G; ++k) sum += acc[((size_t)k * V + v) * 64 + lane];     // group order
            d_emb[(size_t)v * d + c] += sum * scale;
        }
}

struct DLayer {
    size_t w_sa_in, w_sa_in_t, w_sa_o, w_sa_o_t, w_q, w_q_t, w_kv, w_kv_t, w_ca_o, w_ca_o_t, w1, w1_t, w2, w2_t;   // bf16 weights
    size_t x32, x16, qkv, sa, res1, st1, x1_32, x1_16, q, kv, ca, res2, st2, x2_32, x2_16, hid, res3, st3;
};
struct DPlan {
    int B, sy, S, d, H, dff, L, V;
    size_t Md, Nm;
    size_t zero, mem16, xL32, qkv32, keys;
    DLayer layer[16];
    size_t saved_bytes;
    size_t gA, gB, dres, dattn16, dqkv32, slab_all, slab_all_bytes, lnpart, cspart, cspart_side, rowpart, rowpart_bytes, fcslab, fcslab_bytes, scratch_bytes;
    // operands of the weight-gradient GEMMs: one buffer per (layer, use) — those GEMMs run on a side stream beside the input-gradient
    // chain (decoder_bwd), so the chain must not overwrite an operand while its weight gradient may still be reading it
    struct { size_t dy16[3], dhid16, dqkv16, dq16, dkv16; } g[16];
};
size_t dtake(size_t& cur, size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; }
size_t dmax(size_t a, size_t b) { return a > b ? a : b; }
template <class T> T* at(void* base, size_t off) { return reinterpret_cast<T*>((char*)base + off); }
template <class T> const T* cat(const void* base, size_t off) { return reinterpret_cast<const T*>((const char*)base + off); }

int make_dplan(const egx_dec_config* c, int B, DPlan& pl) {
    EGX_CHECK(c, "null decoder config");
    EGX_CHECK(c->compute == EGX_BF16, "the fused decoder runs compute = bf16 (the composed path serves the other modes)");
    EGX_CHECK(c->d_model >= 256 && c->d_model <= 1024 && c->d_model % 128 == 0, "fused decoder: d_model = %d (multiples of 128 in [256, 1024])", c->d_model);
    EGX_CHECK(c->n_heads > 0 && c->d_model % c->n_heads == 0 && (c->d_model / c->n_heads == 32 || c->d_model / c->n_heads == 64),
              "fused decoder: head dim %d (32 or 64)", c->n_heads > 0 ? c->d_model / c->n_heads : 0);
    EGX_CHECK(c->d_ff >= 128 && c->d_ff % 128 == 0, "fused decoder: d_ff = %d (multiples of 128)", c->d_ff);
    EGX_CHECK(c->n_layers >= 1 && c->n_layers <= 16, "fused decoder: %d layers (1..16)", c->n_layers);
    EGX_CHECK(c->sy >= 1 && c->sy <= DA_MAXQ && c->S >= 1 && c->S <= DAL_MAXK, "fused decoder: sy = %d (1..%d), S = %d (1..%d)", c->sy, DA_MAXQ, c->S, DAL_MAXK);
    EGX_CHECK(c->vocab >= 1 && B >= 1, "fused decoder: vocab = %d, B = %d", c->vocab, B);
    memset(&pl, 0, sizeof(pl));
    pl.B = B; pl.sy = c->sy; pl.S = c->S; pl.d = c->d_model; pl.H = c->n_heads; pl.dff = c->d_ff; pl.L = c->n_layers; pl.V = c->vocab;
    pl.Md = (size_t)B * c->sy; pl.Nm = (size_t)B * c->S;
    const size_t d = pl.d, dff = pl.dff, Md = pl.Md, Nm = pl.Nm;
    size_t cur = 0;
    pl.zero = dtake(cur, 1024);
    pl.keys = dtake(cur, (size_t)16 * DROP_KEY_SLOTS * sizeof(uint64_t));
    pl.mem16 = dtake(cur, Nm * d * 2);
    for (int l = 0; l < pl.L; ++l) {
        DLayer& o = pl.layer[l];
        o.w_sa_in = dtake(cur, 3 * d * d * 2); o.w_sa_in_t = dtake(cur, 3 * d * d * 2);
        o.w_sa_o = dtake(cur, d * d * 2); o.w_sa_o_t = dtake(cur, d * d * 2);
        o.w_q = dtake(cur, d * d * 2); o.w_q_t = dtake(cur, d * d * 2);
        o.w_kv = dtake(cur, 2 * d * d * 2); o.w_kv_t = dtake(cur, 2 * d * d * 2);
        o.w_ca_o = dtake(cur, d * d * 2); o.w_ca_o_t = dtake(cur, d * d * 2);
        o.w1 = dtake(cur, dff * d * 2); o.w1_t = dtake(cur, dff * d * 2);
        o.w2 = dtake(cur, dff * d * 2); o.w2_t = dtake(cur, dff * d * 2);
        o.x32 = dtake(cur, Md * d * 4); o.x16 = dtake(cur, Md * d * 2);
        o.qkv = dtake(cur, Md * 3 * d * 2); o.sa = dtake(cur, Md * d * 2);
        o.res1 = dtake(cur, Md * d * 4); o.st1 = dtake(cur, Md * 8);
        o.x1_32 = dtake(cur, Md * d * 4); o.x1_16 = dtake(cur, Md * d * 2);
        o.q = dtake(cur, Md * d * 2); o.kv = dtake(cur, Nm * 2 * d * 2); o.ca = dtake(cur, Md * d * 2);
        o.res2 = dtake(cur, Md * d * 4); o.st2 = dtake(cur, Md * 8);
        o.x2_32 = dtake(cur, Md * d * 4); o.x2_16 = dtake(cur, Md * d * 2);
        o.hid = dtake(cur, Md * dff * 2);
        o.res3 = dtake(cur, Md * d * 4); o.st3 = dtake(cur, Md * 8);
    }
    pl.xL32 = dtake(cur, Md * d * 4);
    pl.qkv32 = dtake(cur, Md * 3 * d * 4);
    pl.saved_bytes = cur;

    size_t sc = 0;
    pl.gA = dtake(sc, Md * d * 4); pl.gB = dtake(sc, Md * d * 4); pl.dres = dtake(sc, Md * d * 4);
    pl.dattn16 = dtake(sc, Md * d * 2); pl.dqkv32 = dtake(sc, Md * 3 * d * 4);
    for (int l = 0; l < pl.L; ++l) {
        for (int u = 0; u < 3; ++u) pl.g[l].dy16[u] = dtake(sc, Md * d * 2);
        pl.g[l].dhid16 = dtake(sc, Md * dff * 2); pl.g[l].dqkv16 = dtake(sc, Md * 3 * d * 2);
        pl.g[l].dq16 = dtake(sc, Md * d * 2); pl.g[l].dkv16 = dtake(sc, Nm * 2 * d * 2);
    }
    size_t all = 0;
    auto add = [&](int M, int Nn, size_t K) { all += align_up(wide_gemm_tn_scratch(M, Nn, (int)K), 256); };
    for (int l = 0; l < pl.L; ++l) {
        add(pl.d, pl.dff, Md); add(pl.dff, pl.d, Md); add(pl.d, pl.d, Md); add(pl.d, pl.d, Md); add(pl.d, pl.d, Md);
        add(3 * pl.d, pl.d, Md); add(2 * pl.d, pl.d, Nm);
    }
    pl.slab_all = dtake(sc, all); pl.slab_all_bytes = all;
    pl.lnpart = dtake(sc, wide_ln_bwd_scratch((int)Md, pl.d));
    size_t cs = dmax(wide_colsum_scratch((int)Md, 3 * pl.d), wide_colsum_scratch((int)Nm, 2 * pl.d));
    cs = dmax(cs, (size_t)(4 * cdiv((int)Md, 256) + 4) * pl.dff * 4);
    pl.cspart = dtake(sc, cs);
    pl.cspart_side = dtake(sc, cs);
    // a partial buffer per deferred row reduction of the backward (wide.h WideRowReduceBatch): per layer three LayerNorm backwards,
    // the lin1 bias sums and three bf16 column sums (self-attention in-projection, cross-attention q and k | v)
    pl.rowpart_bytes = (size_t)pl.L * (3 * align_up(wide_ln_bwd_scratch((int)Md, pl.d), 256) + 4 * align_up(cs, 256));
    pl.rowpart = dtake(sc, pl.rowpart_bytes);
    pl.fcslab_bytes = dmax(gemm_scratch_bytes(2, pl.V, pl.d, (int)Md), gemm_scratch_bytes(1, (int)Md, pl.d, pl.V));
    pl.fcslab_bytes = dmax(pl.fcslab_bytes, dmax(gemm_scratch_bytes(2, 3 * pl.d, pl.d, (int)Md), gemm_scratch_bytes(1, (int)Md, pl.d, 3 * pl.d)));
    pl.fcslab = dtake(sc, pl.fcslab_bytes);
    pl.scratch_bytes = sc;
    return 0;
}

struct DDrop { uint64_t key = 0; uint32_t thresh = 0; float inv = 1.f; };
thread_local const uint64_t* g_dkeys = nullptr;     // device-resident seed: table of this call's keys (layers 0x40 + l), see derive_keys
DDrop ddrop(int training, float p, uint64_t seed, uint32_t layer, uint32_t site) {
    DDrop dr;
    if (training && p > 0.f) {
        dr.key = g_dkeys ? key_slot(g_dkeys, layer, site) : site_key(seed, 0x40u + layer, site);
        dr.thresh = drop_threshold(p); dr.inv = p < 1.f ? 1.f / (1.f - p) : 0.f;
    }
    return dr;
}
struct DKeyScope {
    DKeyScope(const uint64_t* t) { g_dkeys = t; }
    ~DKeyScope() { g_dkeys = nullptr; }
};
enum { DS_SELF = 1, DS_SA_OUT = 2, DS_CROSS = 3, DS_CA_OUT = 4, DS_FFN = 5, DS_FFN_OUT = 6, DS_EMBED = 7 };

// The decoder's weight gradients (small TN GEMMs over B * sy target rows, bias column sums) are off the critical path of its
// backward: the input-gradient chain is ~16 dependent launches per layer that each occupy 16-64 CUs. They run on a library-owned
// side stream, forked off the caller's stream by an event after the kernel that produces their operand and joined before the
// slab reduction (works the same under stream capture: the side stream's launches become a parallel branch of the graph).
// EGX_DEC_SIDE=0 keeps everything on the caller's stream. Under stream CAPTURE the fork is not taken either (side_wanted()): as
// branches of a hipGraph the side launches cost more than they hide — measured on one box, C5 HHI step as one graph 2.71 ms with
// the branches, 2.42 ms without (eager: 2.55 with the side stream, 2.50 without); C5 HOI 4.14 vs 3.88 (eager 3.82 / 3.98).
// EGX_DEC_SIDE=2 forks under capture as well.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[8] = {};
    hipEvent_t kv_ev[16] = {};      // forward: layer l's K | V projection of the memory is done
    int next = 0;
    bool on = false, on_captured = false;
    std::mutex mu;                  // forward and (autograd-thread) backward share the event ring
    // `to` continues behind everything enqueued on `from` so far. Record + wait of one ring event under the lock: two threads
    // picking the same event between the record and the wait would wait for each other's position.
    int order(hipStream_t from, hipStream_t to) {
        std::lock_guard<std::mutex> lk(mu);
        hipEvent_t e = ev[next]; next = (next + 1) % 8;
        EGX_HIP(hipEventRecord(e, from));
        EGX_HIP(hipStreamWaitEvent(to, e, 0));
        return 0;
    }
};
// one side stream per DEVICE, created on that device the first time a decoder call runs there (a process-global stream made
// on whichever device was current first would be the wrong device's stream for every other one)
SideStream& side_stream() {
    enum { MAXDEV = 16 };
    static SideStream S[MAXDEV];
    static std::once_flag once[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) dev = 0;
    std::call_once(once[dev], [dev]() {
        SideStream& T = S[dev];
        const char* e = getenv("EGX_DEC_SIDE");
        if (!(e && e[0] == '0') && hipStreamCreateWithFlags(&T.s, hipStreamNonBlocking) == hipSuccess) {
            T.on = true; T.on_captured = e && e[0] == '2';
            for (auto& v : T.ev) if (hipEventCreateWithFlags(&v, hipEventDisableTiming) != hipSuccess) T.on = false;
            for (auto& v : T.kv_ev) if (hipEventCreateWithFlags(&v, hipEventDisableTiming) != hipSuccess) T.on = false;
        }
    });
    return S[dev];
}
// does this call fork? (never while `st` is being captured, unless EGX_DEC_SIDE=2)
bool side_wanted(const SideStream& SS, hipStream_t st) {
    if (!SS.on || SS.on_captured) return SS.on;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }
    return cs == hipStreamCaptureStatusNone;
}
// joins the side stream back into the caller's stream when a call leaves — on the error paths too: an un-joined fork
// would invalidate an active stream capture (and leave work running that the caller's next launch may overwrite)
struct SideJoin {
    SideStream& SS; hipStream_t st; bool forked = false;
    SideJoin(SideStream& ss, hipStream_t s) : SS(ss), st(s) {}
    int join() { if (!forked) return 0; forked = false; return SS.order(SS.s, st); }
    ~SideJoin() { if (forked) (void)SS.order(SS.s, st); }
};
// a host seed is baked into a captured graph: every replay would redraw the SAME masks. The encoder's fused kernels have the
// device-resident seed (egx_config.seed_ptr) for that; the decoder has not, so training-mode dropout under capture is refused.
int refuse_captured_dropout(const egx_dec_config* cfg, int training, hipStream_t st) {
    if (!training || !(cfg->p_drop > 0.f || cfg->p_pos > 0.f) || cfg->seed_ptr) return 0;      // with seed_ptr every replay draws fresh masks
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
    EGX_CHECK(cs == hipStreamCaptureStatusNone, "egx_decoder: training-mode dropout (p > 0) cannot be captured in a hipGraph: the host seed "
              "would be baked in and every replay would repeat the same masks; launch eagerly, capture with p = 0, or pass egx_dec_config.seed_ptr");
    return 0;
}

}  // namespace

}  // namespace egx

using namespace egx;

extern "C" {

int egx_decoder_workspace(const egx_dec_config* cfg, int B, size_t* saved_bytes, size_t* scratch_bytes) {
    DPlan pl;
    if (make_dplan(cfg, B, pl)) return 1;
    if (saved_bytes) *saved_bytes = pl.saved_bytes;
    if (scratch_bytes) *scratch_bytes = pl.scratch_bytes;
    return 0;
}

int egx_decoder_fwd(const egx_dec_config* cfg, const int64_t* tokens, const float* memory, const float* emb, const float* pe, int pe_stride,
                    const egx_dec_layer* layers, const float* fc_w, const float* fc_b, int B, float* logits, void* saved, void* scratch,
                    int training, uint64_t seed, void* stream) {
    DPlan pl;
    if (make_dplan(cfg, B, pl)) return 1;
    EGX_CHECK(tokens && memory && emb && pe && layers && fc_w && logits && saved && scratch, "egx_decoder_fwd: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    if (refuse_captured_dropout(cfg, training, st)) return 1;
    const int d = pl.d, dff = pl.dff, Md = (int)pl.Md, Nm = (int)pl.Nm, dh = d / pl.H;
    // device-resident seed (the encoder's forward of this step has advanced it): this call's keys, derived on the stream
    const bool dev_keys = cfg->seed_ptr && training && (cfg->p_drop > 0.f || cfg->p_pos > 0.f);
    if (dev_keys && derive_keys(const_cast<uint64_t*>(cfg->seed_ptr), at<uint64_t>(saved, pl.keys), 0x40u, 16, 0, st)) return 1;
    DKeyScope key_scope(dev_keys ? cat<uint64_t>(saved, pl.keys) : nullptr);
    EGX_HIP(hipMemsetAsync(at<char>(saved, pl.zero), 0, 1024, st));
    const void* zero = at<char>(saved, pl.zero);
    bf16_t* mem16 = at<bf16_t>(saved, pl.mem16);
    if (wide_cast(memory, Nm, d, d, mem16, nullptr, st)) return 1;
    {   // every weight -> bf16 (W and W^T), one launch
        WideCastBatch cb;
        for (int l = 0; l < pl.L; ++l) {
            const DLayer& o = pl.layer[l];
            const egx_dec_layer& w = layers[l];
            if (wide_cast_add(cb, w.sa_in_w, 3 * d, d, d, at<bf16_t>(saved, o.w_sa_in), at<bf16_t>(saved, o.w_sa_in_t), st)) return 1;
            if (wide_cast_add(cb, w.sa_out_w, d, d, d, at<bf16_t>(saved, o.w_sa_o), at<bf16_t>(saved, o.w_sa_o_t), st)) return 1;
            if (wide_cast_add(cb, w.ca_in_w, d, d, d, at<bf16_t>(saved, o.w_q), at<bf16_t>(saved, o.w_q_t), st)) return 1;
            if (wide_cast_add(cb, w.ca_in_w + (size_t)d * d, 2 * d, d, d, at<bf16_t>(saved, o.w_kv), at<bf16_t>(saved, o.w_kv_t), st)) return 1;
            if (wide_cast_add(cb, w.ca_out_w, d, d, d, at<bf16_t>(saved, o.w_ca_o), at<bf16_t>(saved, o.w_ca_o_t), st)) return 1;
            if (wide_cast_add(cb, w.lin1_w, dff, d, d, at<bf16_t>(saved, o.w1), at<bf16_t>(saved, o.w1_t), st)) return 1;
            if (wide_cast_add(cb, w.lin2_w, d, dff, dff, at<bf16_t>(saved, o.w2), at<bf16_t>(saved, o.w2_t), st)) return 1;
        }
        if (wide_cast_flush(cb, st)) return 1;
    }
    {   // embedding * sqrt(d) + positional encoding (+ dropout)
        DDrop de = ddrop(training, cfg->p_pos, seed, 0, DS_EMBED);
        const size_t n4 = pl.Md * (d / 4);
        hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, tokens, emb, pe, pe_stride, sqrtf((float)d),
                           at<float>(saved, pl.layer[0].x32), at<bf16_t>(saved, pl.layer[0].x16), Md, pl.sy, d, pl.V, de.key, de.thresh, de.inv);
        EGX_LAUNCH_CHECK();
    }
    auto nt_on = [&](hipStream_t s_, const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const float* bias, float* Cf, bf16_t* Cb, int relu,
                     const DDrop& dr, const float* residual) -> int {
        WideGemmParams g;
        g.A = A; g.B = W; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.Cf = Cf; g.Cb = Cb; g.ldc = N; g.bias = bias; g.relu = relu;
        g.drop_key = dr.key; g.drop_thresh = dr.thresh; g.drop_inv = dr.inv; g.residual = residual; g.ldr = N; g.zero_page = zero;
        return wide_gemm_nt(g, s_);
    };
    auto nt = [&](const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const float* bias, float* Cf, bf16_t* Cb, int relu,
                  const DDrop& dr, const float* residual) -> int { return nt_on(st, A, lda, W, M, N, K, bias, Cf, Cb, relu, dr, residual); };
    auto ln = [&](const float* x, const float* w, const float* b, float* stats, float* y32, bf16_t* y16) -> int {
        WideLnFwdParams lp;
        lp.x = x; lp.w = w; lp.b = b; lp.eps = cfg->ln_eps; lp.stats = stats; lp.y32 = y32; lp.y16 = y16; lp.rows = Md; lp.d = d;
        return wide_ln_fwd(lp, st);
    };
    const DDrop none;
    // the K | V projections of the memory (the one large GEMM of a layer) depend on nothing the target-token chain computes: all
    // of them go to the side stream now, beside the chain's 16-WG launches; a layer's cross-attention waits for its own
    SideStream& SS = side_stream();
    SideJoin sj(SS, st);        // error paths: join whatever was forked
    const bool side = side_wanted(SS, st);
    if (side) {
        if (SS.order(st, SS.s)) return 1;
        sj.forked = true;
        for (int l = 0; l < pl.L; ++l) {
            const DLayer& o = pl.layer[l];
            if (nt_on(SS.s, mem16, d, cat<bf16_t>(saved, o.w_kv), Nm, 2 * d, d, layers[l].ca_in_b + d, nullptr, at<bf16_t>(saved, o.kv), 0, none, nullptr)) return 1;
            EGX_HIP(hipEventRecord(SS.kv_ev[l], SS.s));
        }
    }
    for (int l = 0; l < pl.L; ++l) {
        const DLayer& o = pl.layer[l];
        const egx_dec_layer& w = layers[l];
        const bool last = l + 1 == pl.L;
        float* xo32 = last ? at<float>(saved, pl.xL32) : at<float>(saved, pl.layer[l + 1].x32);
        bf16_t* xo16 = last ? nullptr : at<bf16_t>(saved, pl.layer[l + 1].x16);
        // causal self-attention over the target tokens. Layer 0 sees the embeddings scaled by sqrt(d): its q / k are an order
        // of magnitude larger than a LayerNorm output's and bf16-rounded scores would move the (near-saturated) softmax, so
        // its in-projection runs on the exact fp32 MFMA GEMM and its attention reads fp32 q / k / v (three small GEMMs per step)
        const bool f32_self = l == 0;
        if (f32_self) {
            GemmParams g;
            g.A = cat<float>(saved, o.x32); g.B = w.sa_in_w; g.C = at<float>(saved, pl.qkv32); g.M = Md; g.N = 3 * d; g.K = d;
            g.lda = d; g.ldb = d; g.ldc = 3 * d; g.bias = w.sa_in_b;
            if (gemm(0, g, 0, 0, nullptr, 0, st)) return 1;
        } else if (nt(cat<bf16_t>(saved, o.x16), d, cat<bf16_t>(saved, o.w_sa_in), Md, 3 * d, d, w.sa_in_b, nullptr, at<bf16_t>(saved, o.qkv), 0, none, nullptr)) return 1;
        {
            DecAttnParams a;
            memset(&a, 0, sizeof(a));
            if (f32_self) { const float* qkv = cat<float>(saved, pl.qkv32); a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d; }
            else { const bf16_t* qkv = cat<bf16_t>(saved, o.qkv); a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d; }
            a.ldq = a.ldk = a.ldv = 3 * d; a.o = at<bf16_t>(saved, o.sa); a.ldo = d;
            a.B = B; a.H = pl.H; a.Sq = pl.sy; a.Sk = pl.sy; a.causal = 1;
            DDrop da = ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_SELF);
            a.drop_key = da.key; a.drop_thresh = da.thresh; a.drop_inv = da.inv;
            if (dec_attn<false>(a, dh, f32_self, st)) return 1;
        }
        if (nt(cat<bf16_t>(saved, o.sa), d, cat<bf16_t>(saved, o.w_sa_o), Md, d, d, w.sa_out_b, at<float>(saved, o.res1), nullptr, 0,
               ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_SA_OUT), cat<float>(saved, o.x32))) return 1;
        if (ln(cat<float>(saved, o.res1), w.norm1_w, w.norm1_b, at<float>(saved, o.st1), at<float>(saved, o.x1_32), at<bf16_t>(saved, o.x1_16))) return 1;
        // cross-attention onto the memory
        if (nt(cat<bf16_t>(saved, o.x1_16), d, cat<bf16_t>(saved, o.w_q), Md, d, d, w.ca_in_b, nullptr, at<bf16_t>(saved, o.q), 0, none, nullptr)) return 1;
        if (side) EGX_HIP(hipStreamWaitEvent(st, SS.kv_ev[l], 0));
        else if (nt(mem16, d, cat<bf16_t>(saved, o.w_kv), Nm, 2 * d, d, w.ca_in_b + d, nullptr, at<bf16_t>(saved, o.kv), 0, none, nullptr)) return 1;
        {
            DecAttnParams a;
            memset(&a, 0, sizeof(a));
            const bf16_t* kv = cat<bf16_t>(saved, o.kv);
            a.q = cat<bf16_t>(saved, o.q); a.ldq = d; a.k = kv; a.v = kv + d; a.ldk = a.ldv = 2 * d; a.o = at<bf16_t>(saved, o.ca); a.ldo = d;
            a.B = B; a.H = pl.H; a.Sq = pl.sy; a.Sk = pl.S; a.causal = 0;
            DDrop da = ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_CROSS);
            a.drop_key = da.key; a.drop_thresh = da.thresh; a.drop_inv = da.inv;
            if (dec_attn<false>(a, dh, false, st)) return 1;
        }
        if (nt(cat<bf16_t>(saved, o.ca), d, cat<bf16_t>(saved, o.w_ca_o), Md, d, d, w.ca_out_b, at<float>(saved, o.res2), nullptr, 0,
               ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_CA_OUT), cat<float>(saved, o.x1_32))) return 1;
        if (ln(cat<float>(saved, o.res2), w.norm2_w, w.norm2_b, at<float>(saved, o.st2), at<float>(saved, o.x2_32), at<bf16_t>(saved, o.x2_16))) return 1;
        // FFN
        if (nt(cat<bf16_t>(saved, o.x2_16), d, cat<bf16_t>(saved, o.w1), Md, dff, d, w.lin1_b, nullptr, at<bf16_t>(saved, o.hid), 1,
               ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_FFN), nullptr)) return 1;
        if (nt(cat<bf16_t>(saved, o.hid), dff, cat<bf16_t>(saved, o.w2), Md, d, dff, w.lin2_b, at<float>(saved, o.res3), nullptr, 0,
               ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_FFN_OUT), cat<float>(saved, o.x2_32))) return 1;
        if (ln(cat<float>(saved, o.res3), w.norm3_w, w.norm3_b, at<float>(saved, o.st3), xo32, xo16)) return 1;
    }
    {   // vocabulary head (|V| is 7..12: the shape-generic GEMM, fp32 rows in, fp32 logits out)
        GemmParams g;
        g.A = cat<float>(saved, pl.xL32); g.B = fc_w; g.C = logits; g.M = Md; g.N = pl.V; g.K = d; g.lda = d; g.ldb = d; g.ldc = pl.V; g.bias = fc_b;
        if (gemm(0, g, 0, 0, nullptr, 0, st)) return 1;
    }
    sj.forked = false;      // joined already: every layer's cross-attention waited for its kv_ev, the last side-stream operation
    return 0;
}

int egx_decoder_bwd(const egx_dec_config* cfg, const int64_t* tokens, const egx_dec_layer* layers, const float* fc_w, int B, const float* d_logits,
                    const void* saved, void* scratch, float* d_memory, float* d_emb, const egx_dec_layer_grads* grads, float* d_fc_w,
                    float* d_fc_b, void* zero_buf, size_t zero_bytes, int training, uint64_t seed, void* stream) {
    DPlan pl;
    if (make_dplan(cfg, B, pl)) return 1;
    EGX_CHECK(tokens && layers && fc_w && d_logits && saved && scratch && grads, "egx_decoder_bwd: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    if (refuse_captured_dropout(cfg, training, st)) return 1;
    const int d = pl.d, dff = pl.dff, Md = (int)pl.Md, Nm = (int)pl.Nm, dh = d / pl.H;
    const bool dev_keys = cfg->seed_ptr && training && (cfg->p_drop > 0.f || cfg->p_pos > 0.f);
    DKeyScope key_scope(dev_keys ? cat<uint64_t>(saved, pl.keys) : nullptr);
    const void* zero = cat<char>(saved, pl.zero);
    if (zero_buf && zero_bytes) EGX_HIP(hipMemsetAsync(zero_buf, 0, zero_bytes, st));
    float* gA = at<float>(scratch, pl.gA);
    float* gB = at<float>(scratch, pl.gB);
    float* dres = at<float>(scratch, pl.dres);
    bf16_t* dattn16 = at<bf16_t>(scratch, pl.dattn16);
    void* lnpart = at<char>(scratch, pl.lnpart);
    float* cspart = at<float>(scratch, pl.cspart);
    float* cspart_side = at<float>(scratch, pl.cspart_side);
    // side stream for the weight gradients: fork() orders it behind everything enqueued on `st` so far
    SideStream& SS = side_stream();
    const bool side = side_wanted(SS, st);
    hipStream_t sd = side ? SS.s : st;
    SideJoin sj(SS, st);
    auto fork = [&]() -> int {
        if (!side) return 0;
        if (SS.order(st, sd)) return 1;
        sj.forked = true;
        return 0;
    };
    const bf16_t* mem16 = cat<bf16_t>(saved, pl.mem16);

    WideReduceBatch rb;
    size_t slab_cur = 0;
    // second stages of the two-stage column sums (LayerNorm / bias gradients): queued with a partial buffer of their own each, summed
    // by ONE launch at the end (every target is a different parameter: concurrent sums never meet)
    WideRowReduceBatch rrb;
    size_t row_cur = 0;
    static int row_env = -2;
    if (row_env == -2) { const char* e = getenv("EGX_ROW_DEFER"); row_env = e ? atoi(e) : 1; }
    auto row_region = [&](size_t need) -> void* {
        need = align_up(need, 256);
        if (row_env == 0 || row_cur + need > pl.rowpart_bytes) return nullptr;
        void* r = at<char>(scratch, pl.rowpart) + row_cur;
        row_cur += need;
        return r;
    };
    // Round 5: the weight gradients are QUEUED and leave as one grouped launch per tile variant at the end of the call (wide_gemm.hip
    // wide_tn_queue_*): 35 launches of 14 us each over 512 target rows become one or two grids that fill the chip. Their operands sit in
    // per-(layer, use) buffers until then. Used where the side stream is not (a captured step, EGX_DEC_SIDE=0): with eager launches the
    // side stream already hides these GEMMs beside the target-token chain and grouping them at the end measured 1.4 % slower (C5 HOI
    // 3.80 -> 3.86 ms); captured, C5 HHI 2.42 -> 2.26 ms (profiles/r05_dec_group.txt). EGX_DEC_GROUP=0 / 1 forces either.
    WideTnQueue tq;
    static int group_env = -2;
    if (group_env == -2) { const char* e = getenv("EGX_DEC_GROUP"); group_env = e ? atoi(e) : -1; }
    const bool grouped = group_env >= 0 ? group_env != 0 : !side;
    // (ungrouped: enqueued on the side stream; the caller forks first)
    auto dw_tn = [&](const bf16_t* dy, int ldy, const bf16_t* x, int ldx, float* dW, int n_out, int k_in, int tokens_k) -> int {
        if (!dW) return 0;
        WideGemmParams t;
        t.A = dy; t.B = x; t.M = n_out; t.N = k_in; t.K = tokens_k; t.lda = ldy; t.ldb = ldx;
        t.Cf = dW; t.ldc = k_in; t.zero_page = zero;
        // EGX_DEC_DIRECT=1 (tuning aid, off): over <= 1024 target rows ONE split
        // straight into dW (no slab, no reduction), at most eight over the B * S memory rows: removes the batched slab reduction
        // (55-73 us, 230 MB per step) but lengthens the side stream - same-box A/B: C5 HOI 3.95 = 3.95 ms, C5 HHI 2.46 -> 2.62 ms
        static int direct = -1;
        if (direct < 0) { const char* e = getenv("EGX_DEC_DIRECT"); direct = e ? atoi(e) : 0; }
        // a target inside the buffer this call has just zero-filled is overwritten (saves the read of the += ), anything else keeps
        // the += contract of the header
        const char* zb = (const char*)zero_buf;
        const bool zeroed = zb && (const char*)dW >= zb && (const char*)(dW + (size_t)n_out * k_in) <= zb + zero_bytes;
        t.accumulate = zeroed ? 0 : 1;
        t.tn_max_splits = !(direct && zeroed) ? 0 : (tokens_k <= 1024 ? 1 : (tokens_k <= 4096 ? 0 : 8));
        const size_t need = align_up(wide_gemm_tn_scratch(n_out, k_in, tokens_k), 256);
        EGX_CHECK(slab_cur + need <= pl.slab_all_bytes, "decoder backward: slab region exhausted");
        void* region = at<char>(scratch, pl.slab_all) + slab_cur;
        slab_cur += need;
        if (grouped && !t.tn_max_splits) return wide_tn_queue_add(tq, t, region, st, &rb);
        return wide_gemm_tn(t, region, sd, &rb);
    };
    auto nt = [&](const bf16_t* A, int lda, const bf16_t* Wt, int M, int N, int K, float* Cf, bf16_t* Cb, const float* residual,
                  const bf16_t* mask, float mask_scale, float* colsum) -> int {
        WideGemmParams g;
        g.A = A; g.B = Wt; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.Cf = Cf; g.Cb = Cb; g.ldc = N;
        g.residual = residual; g.ldr = N; g.mask = mask; g.ldm = N; g.mask_scale = mask_scale; g.colsum = colsum; g.zero_page = zero;
        return wide_gemm_nt(g, st);
    };
    auto ln_bwd = [&](const float* dy, const float* pre, const float* stats, const float* w, const DDrop& outm, float* dw, float* db, float* dbias, bf16_t* dy16) -> int {
        WideLnBwdParams b;
        b.dy = dy; b.pre = pre; b.stats = stats; b.w = w; b.dx32 = dres; b.dx16 = dy16; b.rows = Md; b.d = d;
        b.out_key = outm.key; b.out_thresh = outm.thresh; b.out_inv = outm.inv;
        b.dw = dw; b.db = db; b.dbias = dbias;
        void* reg = row_region(wide_ln_bwd_scratch(b.rows, b.d));
        return reg ? wide_ln_bwd(b, reg, st, &rrb) : wide_ln_bwd(b, lnpart, st);
    };

    // vocabulary head: d(fc_w) += d_logits^T x, d(fc_b) += colsum, g = d_logits fc_w
    float* g = gA;
    {
        const float* xL = cat<float>(saved, pl.xL32);
        if (d_fc_w) {
            GemmParams t;
            t.A = d_logits; t.B = xL; t.C = d_fc_w; t.M = pl.V; t.N = d; t.K = Md; t.lda = pl.V; t.ldb = d; t.ldc = d;
            if (gemm(2, t, 0, 1, at<char>(scratch, pl.fcslab), pl.fcslab_bytes, st)) return 1;
        }
        if (d_fc_b && colsum_accum(d_logits, Md, pl.V, pl.V, d_fc_b, st)) return 1;
        GemmParams q;
        q.A = d_logits; q.B = fc_w; q.C = g; q.M = Md; q.N = d; q.K = pl.V; q.lda = pl.V; q.ldb = d; q.ldc = d;
        if (gemm(1, q, 0, 0, at<char>(scratch, pl.fcslab), pl.fcslab_bytes, st)) return 1;
    }
    bool mem_started = false;
    for (int l = pl.L - 1; l >= 0; --l) {
        const DLayer& o = pl.layer[l];
        const egx_dec_layer& w = layers[l];
        const egx_dec_layer_grads& gw = grads[l];
        bf16_t* dy16a = at<bf16_t>(scratch, pl.g[l].dy16[0]);
        bf16_t* dy16b = at<bf16_t>(scratch, pl.g[l].dy16[1]);
        bf16_t* dy16c = at<bf16_t>(scratch, pl.g[l].dy16[2]);
        bf16_t* dhid16 = at<bf16_t>(scratch, pl.g[l].dhid16);
        bf16_t* dqkv16 = at<bf16_t>(scratch, pl.g[l].dqkv16);
        bf16_t* dq16 = at<bf16_t>(scratch, pl.g[l].dq16);
        bf16_t* dkv16 = at<bf16_t>(scratch, pl.g[l].dkv16);
        // FFN
        if (ln_bwd(g, cat<float>(saved, o.res3), cat<float>(saved, o.st3), w.norm3_w, ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_FFN_OUT),
                   gw.norm3_w, gw.norm3_b, gw.lin2_b, dy16a)) return 1;
        if (fork() || dw_tn(dy16a, d, cat<bf16_t>(saved, o.hid), dff, gw.lin2_w, d, dff, Md)) return 1;
        float* cs_l1 = cspart;
        if (gw.lin1_b) { void* rg = row_region((size_t)wide_gemm_nt_colsum_rows(Md, dff) * dff * 4); if (rg) cs_l1 = (float*)rg; }
        if (nt(dy16a, d, cat<bf16_t>(saved, o.w2_t), Md, dff, d, nullptr, dhid16, nullptr, cat<bf16_t>(saved, o.hid),
               ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_FFN).inv, gw.lin1_b ? cs_l1 : nullptr)) return 1;
        if (gw.lin1_b && wide_reduce_rows(cs_l1, wide_gemm_nt_colsum_rows(Md, dff), dff, gw.lin1_b, st, cs_l1 != cspart ? &rrb : nullptr)) return 1;
        if (fork() || dw_tn(dhid16, dff, cat<bf16_t>(saved, o.x2_16), d, gw.lin1_w, dff, d, Md)) return 1;
        float* g1 = (g == gA) ? gB : gA;
        if (nt(dhid16, dff, cat<bf16_t>(saved, o.w1_t), Md, d, dff, g1, nullptr, dres, nullptr, 1.f, nullptr)) return 1;
        // cross-attention
        if (ln_bwd(g1, cat<float>(saved, o.res2), cat<float>(saved, o.st2), w.norm2_w, ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_CA_OUT),
                   gw.norm2_w, gw.norm2_b, gw.ca_out_b, dy16b)) return 1;
        if (fork() || dw_tn(dy16b, d, cat<bf16_t>(saved, o.ca), d, gw.ca_out_w, d, d, Md)) return 1;
        if (nt(dy16b, d, cat<bf16_t>(saved, o.w_ca_o_t), Md, d, d, nullptr, dattn16, nullptr, nullptr, 1.f, nullptr)) return 1;
        {
            DecAttnParams a;
            memset(&a, 0, sizeof(a));
            const bf16_t* kv = cat<bf16_t>(saved, o.kv);
            a.q = cat<bf16_t>(saved, o.q); a.ldq = d; a.k = kv; a.v = kv + d; a.ldk = a.ldv = 2 * d; a.ldo = d;
            a.d_o = dattn16; a.dq = dq16; a.dk = dkv16; a.dv = dkv16 + d;
            a.B = B; a.H = pl.H; a.Sq = pl.sy; a.Sk = pl.S; a.causal = 0;
            DDrop da = ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_CROSS);
            a.drop_key = da.key; a.drop_thresh = da.thresh; a.drop_inv = da.inv;
            if (dec_attn<true>(a, dh, false, st)) return 1;
        }
        if (fork()) return 1;
        if (gw.ca_in_b) {
            void* r1 = row_region(wide_colsum_scratch(Md, d));
            void* r2 = row_region(wide_colsum_scratch(Nm, 2 * d));
            if (wide_colsum_bf16(dq16, Md, d, d, gw.ca_in_b, r1 ? r1 : (void*)cspart_side, sd, r1 ? &rrb : nullptr)) return 1;
            if (wide_colsum_bf16(dkv16, Nm, 2 * d, 2 * d, gw.ca_in_b + d, r2 ? r2 : (void*)cspart_side, sd, r2 ? &rrb : nullptr)) return 1;
        }
        if (dw_tn(dq16, d, cat<bf16_t>(saved, o.x1_16), d, gw.ca_in_w, d, d, Md)) return 1;
        if (dw_tn(dkv16, 2 * d, mem16, d, gw.ca_in_w ? gw.ca_in_w + (size_t)d * d : nullptr, 2 * d, d, Nm)) return 1;
        if (d_memory) {     // d(memory) (+)= d(kv) W_kv, summed over the layers
            if (nt(dkv16, 2 * d, cat<bf16_t>(saved, o.w_kv_t), Nm, d, 2 * d, d_memory, nullptr, mem_started ? d_memory : nullptr, nullptr, 1.f, nullptr)) return 1;
            mem_started = true;
        }
        float* g2 = (g1 == gA) ? gB : gA;
        if (nt(dq16, d, cat<bf16_t>(saved, o.w_q_t), Md, d, d, g2, nullptr, dres, nullptr, 1.f, nullptr)) return 1;
        // self-attention
        if (ln_bwd(g2, cat<float>(saved, o.res1), cat<float>(saved, o.st1), w.norm1_w, ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_SA_OUT),
                   gw.norm1_w, gw.norm1_b, gw.sa_out_b, dy16c)) return 1;
        if (fork() || dw_tn(dy16c, d, cat<bf16_t>(saved, o.sa), d, gw.sa_out_w, d, d, Md)) return 1;
        if (nt(dy16c, d, cat<bf16_t>(saved, o.w_sa_o_t), Md, d, d, nullptr, dattn16, nullptr, nullptr, 1.f, nullptr)) return 1;
        const bool f32_self = l == 0;
        float* dqkv32 = at<float>(scratch, pl.dqkv32);
        {
            DecAttnParams a;
            memset(&a, 0, sizeof(a));
            if (f32_self) {
                const float* qkv = cat<float>(saved, pl.qkv32);
                a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d; a.dq = dqkv32; a.dk = dqkv32 + d; a.dv = dqkv32 + 2 * d;
            } else {
                const bf16_t* qkv = cat<bf16_t>(saved, o.qkv);
                a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d; a.dq = dqkv16; a.dk = dqkv16 + d; a.dv = dqkv16 + 2 * d;
            }
            a.ldq = a.ldk = a.ldv = 3 * d; a.ldo = d; a.d_o = dattn16;
            a.B = B; a.H = pl.H; a.Sq = pl.sy; a.Sk = pl.sy; a.causal = 1;
            DDrop da = ddrop(training, cfg->p_drop, seed, (uint32_t)l, DS_SELF);
            a.drop_key = da.key; a.drop_thresh = da.thresh; a.drop_inv = da.inv;
            if (dec_attn<true>(a, dh, f32_self, st)) return 1;
        }
        float* g0 = (g2 == gA) ? gB : gA;
        if (f32_self) {     // exact fp32 in-projection gradients (see the forward); these stay on the caller's stream (they share
                            // the fp32 slab with the vocabulary head's GEMMs)
            if (gw.sa_in_b && colsum_accum(dqkv32, Md, 3 * d, 3 * d, gw.sa_in_b, st)) return 1;
            if (gw.sa_in_w) {
                GemmParams t;
                t.A = dqkv32; t.B = cat<float>(saved, o.x32); t.C = gw.sa_in_w; t.M = 3 * d; t.N = d; t.K = Md; t.lda = 3 * d; t.ldb = d; t.ldc = d;
                if (gemm(2, t, 0, 1, at<char>(scratch, pl.fcslab), pl.fcslab_bytes, st)) return 1;
            }
            // d(layer input): only the embedding gradient reads it -> the bf16 GEMM like the other layers, on a bf16 copy of dqkv
            if (wide_cast(dqkv32, Md, 3 * d, 3 * d, dqkv16, nullptr, st)) return 1;
            if (nt(dqkv16, 3 * d, cat<bf16_t>(saved, o.w_sa_in_t), Md, d, 3 * d, g0, nullptr, dres, nullptr, 1.f, nullptr)) return 1;
        } else {
            if (fork()) return 1;
            if (gw.sa_in_b) {
                void* r3 = row_region(wide_colsum_scratch(Md, 3 * d));
                if (wide_colsum_bf16(dqkv16, Md, 3 * d, 3 * d, gw.sa_in_b, r3 ? r3 : (void*)cspart_side, sd, r3 ? &rrb : nullptr)) return 1;
            }
            if (dw_tn(dqkv16, 3 * d, cat<bf16_t>(saved, o.x16), d, gw.sa_in_w, 3 * d, d, Md)) return 1;
            if (nt(dqkv16, 3 * d, cat<bf16_t>(saved, o.w_sa_in_t), Md, d, 3 * d, g0, nullptr, dres, nullptr, 1.f, nullptr)) return 1;
        }
        g = g0;
    }
    if (sj.join()) return 1;    // the slab reduction below (and whatever the caller enqueues next) comes after the side stream's work
    if (d_memory && !mem_started) EGX_HIP(hipMemsetAsync(d_memory, 0, pl.Nm * d * sizeof(float), st));
    if (d_emb) {
        DDrop de = ddrop(training, cfg->p_pos, seed, 0, DS_EMBED);
        if (pl.V <= EMB_VMAX)
        {
            int G = 65536 / (pl.V * 256);
            G = G > 16 ? 16 : (G < 1 ? 1 : G);
            hipLaunchKernelGGL(dec_embed_grad_small_kernel, dim3(cdiv(d, 64)), dim3(64 * G), (size_t)G * pl.V * 256, st, tokens, g, d_emb,
                               sqrtf((float)d), Md, d, pl.V, de.key, de.thresh, de.inv);
        }
        else
            hipLaunchKernelGGL(dec_embed_grad_kernel, dim3(cdiv(d, 64), pl.V), dim3(256), 0, st, tokens, g, d_emb, sqrtf((float)d), Md, d, pl.V,
                               de.key, de.thresh, de.inv);
        EGX_LAUNCH_CHECK();
    }
    if (wide_row_reduce_flush(rrb, st)) return 1;   // every queued LayerNorm / bias column sum
    if (wide_tn_queue_flush(tq, st)) return 1;      // every queued weight gradient: one grid per tile variant, then their slab sums
    return wide_reduce_flush(rb, st);
}

}  // extern "C"
