// MFMA attention of the wide path: softmax(Q K^T / sqrt(dh)) V per (clip, head) over bf16 packed qkv rows, S <= 128.
// Replaces the core of F.multi_head_attention_forward (torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer,
// HOI/models/lta/lta_models_lta_transfer.py:272-275, HOI/models/multitask/video_model_builder.py:70-77) and its backward.
//
// One 256-thread workgroup per (clip, head). K and V (backward: Q, K, V, dO) of the head sit in LDS as token-major
// images with padded rows (2 dh + 32 bytes: row reads by ds_read_b128 are at most 2-way conflicted, transposed reads by
// ds_read_b64_tr_b16 conflict-free). Scores are computed TRANSPOSED (S^T = K Q^T: key on the accumulator rows, query on
// the lane), so the softmax reductions run over registers plus two cross-lane steps, and the probability tiles are
// already the B operand of the next product (O^T = V^T P^T) - no LDS round trip for P. A K-block of that product is two
// 16-key tiles, i.e. lane group g holds keys {4g..4g+3} and {16+4g..16+4g+3} of the block; the V^T operand is read with
// the same key order by two transposed reads.
// Backward: pass T (a wave owns query tiles; key-on-rows orientation) yields delta = rowsum(P .* dP) and dQ; pass N
// (a wave owns key tiles; query-on-rows orientation) recomputes P and dS in the other orientation and yields dK, dV.
// Every output element is produced by exactly one wave: no atomics, bitwise reproducible.
#include "common.h"
#include "wide.h"
#include "fused.h"

namespace egx {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 rd128(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 rd_tr2(const unsigned char* p0, const unsigned char* p1) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    bf16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
__device__ __forceinline__ uint32_t pk(float a, float b) { return pack_bf16x2(a, b); }
// two 16-row accumulator tiles (rows 4g + e on the registers) -> one 32-deep B operand
__device__ __forceinline__ bf16x8 chain(const f32x4& t0, const f32x4& t1) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 u = {pk(t0[0], t0[1]), pk(t0[2], t0[3]), pk(t1[0], t1[1]), pk(t1[2], t1[3])};
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// copy `rows` token rows of one head (dh bf16 each, global row stride ld elements) into an LDS image with row stride RS
// bytes; rows [rows, SP) are zero-filled. All of a thread's loads are issued before its first LDS store (one exposed
// memory latency per image set instead of one per 16-byte chunk).
template <int DH, int SP, int NTH = 256>
struct ImageRegs { uint4 v[(SP * (DH / 8) + NTH - 1) / NTH]; };
template <int DH, int SP, int NTH = 256>
__device__ __forceinline__ void image_fetch(ImageRegs<DH, SP, NTH>& R, const bf16_t* src, int ld, int rows) {
    constexpr int CH = DH / 8, NIT = (SP * CH + NTH - 1) / NTH;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * NTH;
        const int row = i / CH, c = i % CH;
        R.v[it] = make_uint4(0, 0, 0, 0);
        if (i < SP * CH && row < rows) R.v[it] = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c * 8);
    }
}
template <int DH, int SP, int NTH = 256>
__device__ __forceinline__ void image_store(const ImageRegs<DH, SP, NTH>& R, unsigned char* img) {
    constexpr int RS = DH * 2 + 32, CH = DH / 8, NIT = (SP * CH + NTH - 1) / NTH;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * NTH;
        const int row = i / CH, c = i % CH;
        if (i < SP * CH) *reinterpret_cast<uint4*>(img + row * RS + c * 16) = R.v[it];
    }
}

}  // namespace

// ---- forward ----------------------------------------------------------------------------------------------------
template <int DH, int NKT>
__global__ __launch_bounds__(256) void wide_attn_fwd_kernel(WideAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    constexpr int RS = DH * 2 + 32, SP = NKT * 16, NKB = DH / 32, NCT = DH / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kimg = smem;
    unsigned char* Vimg = smem + SP * RS;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int S = p.S, d = p.d, ld = 3 * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + (size_t)b * S * ld + h * DH;
    {
        ImageRegs<DH, SP> rk, rv;
        image_fetch<DH, SP>(rk, base + d, ld, S);
        image_fetch<DH, SP>(rv, base + 2 * d, ld, S);
        image_store<DH, SP>(rk, Kimg);
        image_store<DH, SP>(rv, Vimg);
    }
    __syncthreads();
    const float scale = rsqrtf((float)DH);
    const int nqt = (S + 15) / 16;
    for (int qt = wave; qt < nqt; qt += 4) {
        const int query = qt * 16 + r;
        const int qrow = query < S ? query : S - 1;
        bf16x8 qf[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) qf[kb] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + kb * 32 + 8 * g);
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) a = mfma(rd128(Kimg + (kt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), qf[kb], a);
            sc[kt] = a;
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int key = kt * 16 + 4 * g + e;
                float s = key < S ? sc[kt][e] * scale : -INFINITY;
                sc[kt][e] = s;
                m = fmaxf(m, s);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) { float pv = __expf(sc[kt][e] - m); sc[kt][e] = pv; sum += pv; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (g == 0 && query < S) p.lse[(size_t)bh * S + query] = m + __logf(sum);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pv = sc[kt][e] * inv;
                if (p.drop_thresh) pv *= drop_scale(dkey, (uint32_t)(bh * 128 + query), (uint32_t)(kt * 16 + 4 * g + e), p.drop_thresh, p.drop_inv);
                sc[kt][e] = pv;
            }
        f32x4 oc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) oc[ct] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kb2 = 0; kb2 < NKT / 2; ++kb2) {
            const bf16x8 pf = chain(sc[2 * kb2], sc[2 * kb2 + 1]);
            const unsigned char* v0 = Vimg + (kb2 * 32 + 4 * g + (r >> 2)) * RS + 8 * (r & 3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) oc[ct] = mfma(rd_tr2(v0 + ct * 32, v0 + 16 * RS + ct * 32), pf, oc[ct]);
        }
        if (query < S) {
            bf16_t* o = p.out + ((size_t)b * S + query) * d + h * DH + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                *reinterpret_cast<uint2*>(o + ct * 16) = make_uint2(pk(oc[ct][0], oc[ct][1]), pk(oc[ct][2], oc[ct][3]));
        }
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------
// NKT = 8 (S <= 128): 512 threads, one query tile (pass T) and one key tile (pass N) per wave, two waves per SIMD.
template <int DH, int NKT>
__global__ __launch_bounds__(NKT * 64) void wide_attn_bwd_kernel(WideAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    constexpr int RS = DH * 2 + 32, SP = NKT * 16, NKB = DH / 32, NCT = DH / 16, NW = NKT, NTH = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qimg = smem;
    unsigned char* Kimg = Qimg + SP * RS;
    unsigned char* Vimg = Kimg + SP * RS;
    unsigned char* Dimg = Vimg + SP * RS;                    // dO
    float* lse_s = reinterpret_cast<float*>(Dimg + SP * RS); // [SP]
    float* delta_s = lse_s + SP;                             // [SP]
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int S = p.S, d = p.d, ld = 3 * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + (size_t)b * S * ld + h * DH;
    {
        ImageRegs<DH, SP, NTH> rq, rk, rv, rd;
        image_fetch<DH, SP, NTH>(rq, base, ld, S);
        image_fetch<DH, SP, NTH>(rk, base + d, ld, S);
        image_fetch<DH, SP, NTH>(rv, base + 2 * d, ld, S);
        image_fetch<DH, SP, NTH>(rd, p.d_out + (size_t)b * S * d + h * DH, d, S);
        image_store<DH, SP, NTH>(rq, Qimg);
        image_store<DH, SP, NTH>(rk, Kimg);
        image_store<DH, SP, NTH>(rv, Vimg);
        image_store<DH, SP, NTH>(rd, Dimg);
    }
    for (int i = threadIdx.x; i < SP; i += NTH) lse_s[i] = i < S ? p.lse[(size_t)bh * S + i] : 0.f;
    __syncthreads();
    const float scale = rsqrtf((float)DH);
    const int nt = (S + 15) / 16;
    bf16_t* gq = p.d_qkv + (size_t)b * S * ld + h * DH;

    // ---- pass T: rows = key, cols = query; this wave's query tiles
    for (int qt = wave; qt < nt; qt += NW) {
        const int query = qt * 16 + r;
        bf16x8 qf[NKB], df[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            qf[kb] = rd128(Qimg + query * RS + (kb * 32 + 8 * g) * 2);
            df[kb] = rd128(Dimg + query * RS + (kb * 32 + 8 * g) * 2);
        }
        const float lq = lse_s[query];
        f32x4 pt[NKT], dpt[NKT];
        float dl = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = f32x4{0, 0, 0, 0}, c = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                a = mfma(rd128(Kimg + (kt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), qf[kb], a);
                c = mfma(rd128(Vimg + (kt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), df[kb], c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int key = kt * 16 + 4 * g + e;
                float pv = (key < S && query < S) ? __expf(a[e] * scale - lq) : 0.f;
                float ks = p.drop_thresh ? drop_scale(dkey, (uint32_t)(bh * 128 + query), (uint32_t)key, p.drop_thresh, p.drop_inv) : 1.f;
                float dm = c[e] * ks;
                dl += pv * dm;
                a[e] = pv; c[e] = dm;
            }
            pt[kt] = a; dpt[kt] = c;
            __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads of tile kt + 1 behind this tile (register budget)
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (g == 0) delta_s[query] = dl;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) dpt[kt][e] = pt[kt][e] * (dpt[kt][e] - dl) * scale;       // dS^T
        f32x4 dq[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) dq[ct] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kb2 = 0; kb2 < NKT / 2; ++kb2) {
            const bf16x8 sf = chain(dpt[2 * kb2], dpt[2 * kb2 + 1]);
            const unsigned char* k0 = Kimg + (kb2 * 32 + 4 * g + (r >> 2)) * RS + 8 * (r & 3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) dq[ct] = mfma(rd_tr2(k0 + ct * 32, k0 + 16 * RS + ct * 32), sf, dq[ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (query < S) {
            bf16_t* o = gq + (size_t)query * ld + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                *reinterpret_cast<uint2*>(o + ct * 16) = make_uint2(pk(dq[ct][0], dq[ct][1]), pk(dq[ct][2], dq[ct][3]));
        }
    }
    __syncthreads();

    // ---- pass N: rows = query, cols = key; this wave's key tiles
    for (int kt = wave; kt < nt; kt += NW) {
        const int key = kt * 16 + r;
        bf16x8 kf[NKB], vf[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            kf[kb] = rd128(Kimg + key * RS + (kb * 32 + 8 * g) * 2);
            vf[kb] = rd128(Vimg + key * RS + (kb * 32 + 8 * g) * 2);
        }
        f32x4 pn[NKT], dsn[NKT];
#pragma unroll
        for (int qt = 0; qt < NKT; ++qt) {
            f32x4 a = f32x4{0, 0, 0, 0}, c = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                a = mfma(rd128(Qimg + (qt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), kf[kb], a);
                c = mfma(rd128(Dimg + (qt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), vf[kb], c);
            }
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qt * 16 + 4 * g);
            const float4 d4 = *reinterpret_cast<const float4*>(delta_s + qt * 16 + 4 * g);
            const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
            // the lane's four elements are four mask ROWS (queries) of one key: one hash per lane and tile (common.h tile_keep_rows)
            uint32_t m4[4] = {0xFu, 0xFu, 0xFu, 0xFu};
            if (p.drop_thresh) tile_keep_rows(dkey, (uint32_t)(bh * 128 + qt * 16), (uint32_t)(kt * 4), r, g, p.drop_thresh, m4);
            const float kinv = p.drop_thresh ? p.drop_inv : 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int query = qt * 16 + 4 * g + e;
                float pv = (key < S && query < S) ? __expf(a[e] * scale - lq[e]) : 0.f;
                float ks = ((m4[e] >> (r & 3)) & 1u) ? kinv : 0.f;
                a[e] = pv * ks;                                   // dropped P
                c[e] = pv * (ks * c[e] - dq4[e]) * scale;         // dS
            }
            pn[qt] = a; dsn[qt] = c;
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 dk[NCT], dv[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { dk[ct] = f32x4{0, 0, 0, 0}; dv[ct] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int qb2 = 0; qb2 < NKT / 2; ++qb2) {
            const bf16x8 pf = chain(pn[2 * qb2], pn[2 * qb2 + 1]);
            const bf16x8 sf = chain(dsn[2 * qb2], dsn[2 * qb2 + 1]);
            const int roff = (qb2 * 32 + 4 * g + (r >> 2)) * RS + 8 * (r & 3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                dv[ct] = mfma(rd_tr2(Dimg + roff + ct * 32, Dimg + roff + 16 * RS + ct * 32), pf, dv[ct]);
                dk[ct] = mfma(rd_tr2(Qimg + roff + ct * 32, Qimg + roff + 16 * RS + ct * 32), sf, dk[ct]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (key < S) {
            bf16_t* o = gq + (size_t)key * ld + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                *reinterpret_cast<uint2*>(o + d + ct * 16) = make_uint2(pk(dk[ct][0], dk[ct][1]), pk(dk[ct][2], dk[ct][3]));
                *reinterpret_cast<uint2*>(o + 2 * d + ct * 16) = make_uint2(pk(dv[ct][0], dv[ct][1]), pk(dv[ct][2], dv[ct][3]));
            }
        }
    }
}


// ---- long sequences (128 < S <= WIDE_ATTN_LONG_MAX_S: the EgoT2-g HHI encoder on real TTM / ASD batches of up to 3 x 150 tokens) ----
// Same images, fragment order and transposed reads; what changes: the image height is a run-time SP (rows padded to 32), a wave
// walks the 32-row blocks of the image with two score tiles live (forward: online softmax, O^T rescaled by a per-lane scalar),
// and the backward is two launches (query side: K | V images, dQ and delta = rowsum(dO . O) from the saved attention output;
// key side: Q | dO images, dK and dV) because four images of 464 x 160 bytes do not fit the LDS. Eight waves per workgroup.
// Dropout keying: row = bh * 512 + query.
constexpr int WAL_NW = 8, WAL_NTH = WAL_NW * 64;

template <int DH>
__device__ __forceinline__ void image_stage(unsigned char* img, const bf16_t* src, int ld, int rows, int SP) {
    constexpr int RS = DH * 2 + 32, CH = DH / 8;
    for (int i0 = threadIdx.x; i0 < SP * CH; i0 += 4 * WAL_NTH) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * WAL_NTH, row = i / CH, c = i % CH;
            v[u] = *reinterpret_cast<const uint4*>(src + (size_t)(row < rows ? row : rows - 1) * ld + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * WAL_NTH, row = i / CH, c = i % CH;
            if (i < SP * CH) *reinterpret_cast<uint4*>(img + row * RS + c * 16) = row < rows ? v[u] : make_uint4(0, 0, 0, 0);
        }
    }
}

template <int DH>
__global__ __launch_bounds__(WAL_NTH) void wide_attn_long_fwd_kernel(WideAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    constexpr int RS = DH * 2 + 32, NKB = DH / 32, NCT = DH / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = p.S, SP = (S + 31) & ~31, d = p.d, ld = 3 * d;
    unsigned char* Kimg = smem;
    unsigned char* Vimg = smem + SP * RS;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + (size_t)b * S * ld + h * DH;
    image_stage<DH>(Kimg, base + d, ld, S, SP);
    image_stage<DH>(Vimg, base + 2 * d, ld, S, SP);
    __syncthreads();
    const float c2 = rsqrtf((float)DH) * 1.4426950408889634f;       // scores in log2 units
    const float vinv = p.drop_thresh ? p.drop_inv : 1.f;
    const int nqt = (S + 15) / 16, nkb2 = SP / 32;
    for (int qt = blockIdx.y * WAL_NW + wave; qt < nqt; qt += WAL_NW * gridDim.y) {
        const int query = qt * 16 + r;
        const int qrow = query < S ? query : S - 1;
        bf16x8 qf[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) qf[kb] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + kb * 32 + 8 * g);
        float m = -INFINITY, l = 0.f;
        f32x4 oc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) oc[ct] = f32x4{0, 0, 0, 0};
        for (int kb2 = 0; kb2 < nkb2; ++kb2) {
            f32x4 sc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 a = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) a = mfma(rd128(Kimg + (kb2 * 32 + j * 16 + r) * RS + (kb * 32 + 8 * g) * 2), qf[kb], a);
                sc[j] = a * c2;
            }
            if (kb2 == nkb2 - 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kb2 * 32 + j * 16 + 4 * g + e >= S) sc[j][e] = -INFINITY;
            }
            float mb = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
            mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
            mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
            const float mn = fmaxf(m, mb);
            const float corr = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= corr;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) oc[ct] *= corr;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pv = __builtin_amdgcn_exp2f(sc[j][e] - mn);
                    l += pv;
                    if (p.drop_thresh) pv *= drop_scale(dkey, (uint32_t)(bh * 512 + query), (uint32_t)(kb2 * 32 + j * 16 + 4 * g + e), p.drop_thresh, vinv);
                    sc[j][e] = pv;
                }
            const bf16x8 pf = chain(sc[0], sc[1]);
            const unsigned char* v0 = Vimg + (kb2 * 32 + 4 * g + (r >> 2)) * RS + 8 * (r & 3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) oc[ct] = mfma(rd_tr2(v0 + ct * 32, v0 + 16 * RS + ct * 32), pf, oc[ct]);
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        if (query < S) {
            if (g == 0) p.lse[(size_t)bh * S + query] = (m + log2f(l)) * 0.6931471805599453f;
            bf16_t* o = p.out + ((size_t)b * S + query) * d + h * DH + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                *reinterpret_cast<uint2*>(o + ct * 16) = make_uint2(pk(oc[ct][0] * inv, oc[ct][1] * inv), pk(oc[ct][2] * inv, oc[ct][3] * inv));
        }
    }
}

// query side of the long backward: dQ, delta
template <int DH>
__global__ __launch_bounds__(WAL_NTH) void wide_attn_long_dq_kernel(WideAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    constexpr int RS = DH * 2 + 32, NKB = DH / 32, NCT = DH / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = p.S, SP = (S + 31) & ~31, d = p.d, ld = 3 * d;
    unsigned char* Kimg = smem;
    unsigned char* Vimg = smem + SP * RS;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + (size_t)b * S * ld + h * DH;
    image_stage<DH>(Kimg, base + d, ld, S, SP);
    image_stage<DH>(Vimg, base + 2 * d, ld, S, SP);
    __syncthreads();
    const float scale = rsqrtf((float)DH), c2 = scale * 1.4426950408889634f;
    const int nqt = (S + 15) / 16, nkb2 = SP / 32;
    bf16_t* gq = p.d_qkv + (size_t)b * S * ld + h * DH;
    for (int qt = blockIdx.y * WAL_NW + wave; qt < nqt; qt += WAL_NW * gridDim.y) {
        const int query = qt * 16 + r;
        const int qrow = query < S ? query : S - 1;
        bf16x8 qf[NKB], df[NKB];
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            qf[kb] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + kb * 32 + 8 * g);
            df[kb] = *reinterpret_cast<const bf16x8*>(p.d_out + ((size_t)b * S + qrow) * d + h * DH + kb * 32 + 8 * g);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(p.out + ((size_t)b * S + qrow) * d + h * DH + kb * 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                delta += __uint_as_float((uint32_t)(unsigned short)df[kb][e] << 16) * __uint_as_float((uint32_t)(unsigned short)of[e] << 16);
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        if (g == 0 && query < S) p.delta[(size_t)bh * S + query] = delta;
        const float lq = p.lse[(size_t)bh * S + qrow] * 1.4426950408889634f;
        f32x4 dq[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) dq[ct] = f32x4{0, 0, 0, 0};
        for (int kb2 = 0; kb2 < nkb2; ++kb2) {
            f32x4 ds[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 a = f32x4{0, 0, 0, 0}, c = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    a = mfma(rd128(Kimg + (kb2 * 32 + j * 16 + r) * RS + (kb * 32 + 8 * g) * 2), qf[kb], a);
                    c = mfma(rd128(Vimg + (kb2 * 32 + j * 16 + r) * RS + (kb * 32 + 8 * g) * 2), df[kb], c);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kb2 * 32 + j * 16 + 4 * g + e;
                    const float pv = key < S ? __builtin_amdgcn_exp2f(a[e] * c2 - lq) : 0.f;
                    const float ks = p.drop_thresh ? drop_scale(dkey, (uint32_t)(bh * 512 + query), (uint32_t)key, p.drop_thresh, p.drop_inv) : 1.f;
                    ds[j][e] = pv * (ks * c[e] - delta) * scale;
                }
            }
            const bf16x8 sf = chain(ds[0], ds[1]);
            const unsigned char* k0 = Kimg + (kb2 * 32 + 4 * g + (r >> 2)) * RS + 8 * (r & 3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) dq[ct] = mfma(rd_tr2(k0 + ct * 32, k0 + 16 * RS + ct * 32), sf, dq[ct]);
        }
        if (query < S) {
            bf16_t* o = gq + (size_t)query * ld + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                *reinterpret_cast<uint2*>(o + ct * 16) = make_uint2(pk(dq[ct][0], dq[ct][1]), pk(dq[ct][2], dq[ct][3]));
        }
    }
}

// key side of the long backward: dK, dV
template <int DH>
__global__ __launch_bounds__(WAL_NTH) void wide_attn_long_dkv_kernel(WideAttnParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    constexpr int RS = DH * 2 + 32, NKB = DH / 32, NCT = DH / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = p.S, SP = (S + 31) & ~31, d = p.d, ld = 3 * d;
    unsigned char* Qimg = smem;
    unsigned char* Dimg = smem + SP * RS;
    float* lse_s = reinterpret_cast<float*>(Dimg + SP * RS);
    float* delta_s = lse_s + SP;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + (size_t)b * S * ld + h * DH;
    image_stage<DH>(Qimg, base, ld, S, SP);
    image_stage<DH>(Dimg, p.d_out + (size_t)b * S * d + h * DH, d, S, SP);
    for (int i = threadIdx.x; i < SP; i += WAL_NTH) {
        lse_s[i] = i < S ? p.lse[(size_t)bh * S + i] * 1.4426950408889634f : 0.f;
        delta_s[i] = i < S ? p.delta[(size_t)bh * S + i] : 0.f;
    }
    __syncthreads();
    const float scale = rsqrtf((float)DH), c2 = scale * 1.4426950408889634f;
    const int nkt = (S + 15) / 16, nqb2 = SP / 32;
    bf16_t* gq = p.d_qkv + (size_t)b * S * ld + h * DH;
    for (int kt = blockIdx.y * WAL_NW + wave; kt < nkt; kt += WAL_NW * gridDim.y) {
        const int key = kt * 16 + r;
        const int krow = key < S ? key : S - 1;
        bf16x8 kf[NKB], vf[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            kf[kb] = *reinterpret_cast<const bf16x8*>(base + (size_t)krow * ld + d + kb * 32 + 8 * g);
            vf[kb] = *reinterpret_cast<const bf16x8*>(base + (size_t)krow * ld + 2 * d + kb * 32 + 8 * g);
        }
        f32x4 dk[NCT], dv[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { dk[ct] = f32x4{0, 0, 0, 0}; dv[ct] = f32x4{0, 0, 0, 0}; }
        for (int qb2 = 0; qb2 < nqb2; ++qb2) {
            f32x4 pn[2], dsn[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int qt = 2 * qb2 + j;
                f32x4 a = f32x4{0, 0, 0, 0}, c = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    a = mfma(rd128(Qimg + (qt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), kf[kb], a);
                    c = mfma(rd128(Dimg + (qt * 16 + r) * RS + (kb * 32 + 8 * g) * 2), vf[kb], c);
                }
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qt * 16 + 4 * g);
                const float4 d4 = *reinterpret_cast<const float4*>(delta_s + qt * 16 + 4 * g);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
                uint32_t m4[4] = {0xFu, 0xFu, 0xFu, 0xFu};      // (common.h tile_keep_rows: one hash per lane and tile)
                if (p.drop_thresh) tile_keep_rows(dkey, (uint32_t)(bh * 512 + qt * 16), (uint32_t)(kt * 4), r, g, p.drop_thresh, m4);
                const float kinv = p.drop_thresh ? p.drop_inv : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int query = qt * 16 + 4 * g + e;
                    const float pv = (key < S && query < S) ? __builtin_amdgcn_exp2f(a[e] * c2 - lq[e]) : 0.f;
                    const float ks = ((m4[e] >> (r & 3)) & 1u) ? kinv : 0.f;
                    pn[j][e] = pv * ks;
                    dsn[j][e] = pv * (ks * c[e] - dq4[e]) * scale;
                }
            }
            const bf16x8 pf = chain(pn[0], pn[1]), sf = chain(dsn[0], dsn[1]);
            const int roff = (qb2 * 32 + 4 * g + (r >> 2)) * RS + 8 * (r & 3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                dv[ct] = mfma(rd_tr2(Dimg + roff + ct * 32, Dimg + roff + 16 * RS + ct * 32), pf, dv[ct]);
                dk[ct] = mfma(rd_tr2(Qimg + roff + ct * 32, Qimg + roff + 16 * RS + ct * 32), sf, dk[ct]);
            }
        }
        if (key < S) {
            bf16_t* o = gq + (size_t)key * ld + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                *reinterpret_cast<uint2*>(o + d + ct * 16) = make_uint2(pk(dk[ct][0], dk[ct][1]), pk(dk[ct][2], dk[ct][3]));
                *reinterpret_cast<uint2*>(o + 2 * d + ct * 16) = make_uint2(pk(dv[ct][0], dv[ct][1]), pk(dv[ct][2], dv[ct][3]));
            }
        }
    }
}

bool wide_attn_long_supported(int S, int dh) {
    // two images of ((S + 31) & ~31) x (2 dh + 32) bytes (+ the per-query statistics of the key side) in 160 KB of LDS
    if (S <= 128 || S > 512 || (dh != 32 && dh != 64)) return false;      // 512: the dropout row key is bh * 512 + query
    const size_t SP = (size_t)((S + 31) & ~31);
    return 2 * SP * (2 * dh + 32) + 2 * SP * sizeof(float) <= 160 * 1024;
}

template <int DH>
static int launch_attn_long(const WideAttnParams& p, bool bwd, hipStream_t st) {
    constexpr int RS = DH * 2 + 32;
    const size_t SP = (size_t)((p.S + 31) & ~31);
    const size_t lds2 = 2 * SP * RS, lds_b = lds2 + 2 * SP * sizeof(float);
    static bool once = false;
    if (!once) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_attn_long_fwd_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_attn_long_dq_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_attn_long_dkv_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once = true;
    }
    const int ntile = (p.S + 15) / 16;
    int split = 256 / (p.B * p.H);
    const int maxs = (ntile + WAL_NW - 1) / WAL_NW;
    split = split < 1 ? 1 : (split > maxs ? maxs : split);
    dim3 grid(p.B * p.H, split);
    timing_begin(bwd ? TIMER_WIDE_ATTN_BWD : TIMER_WIDE_ATTN_FWD, st);
    if (!bwd) {
        hipLaunchKernelGGL((wide_attn_long_fwd_kernel<DH>), grid, dim3(WAL_NTH), lds2, st, p);
    } else {
        hipLaunchKernelGGL((wide_attn_long_dq_kernel<DH>), grid, dim3(WAL_NTH), lds2, st, p);
        count_launch();
        hipLaunchKernelGGL((wide_attn_long_dkv_kernel<DH>), grid, dim3(WAL_NTH), lds_b, st, p);
    }
    timing_end(bwd ? TIMER_WIDE_ATTN_BWD : TIMER_WIDE_ATTN_FWD, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

bool wide_attn_supported(int S, int dh) {
    return (S >= 1 && S <= 128 && (dh == 32 || dh == 64 || dh == 96 || dh == 128)) || wide_attn_long_supported(S, dh);
}

template <int DH, int NKT>
static int launch_attn(const WideAttnParams& p, bool bwd, hipStream_t st) {
    constexpr int RS = DH * 2 + 32, SP = NKT * 16;
    const size_t lds = bwd ? (size_t)4 * SP * RS + 2 * SP * sizeof(float) : (size_t)2 * SP * RS;
    const void* fn = bwd ? reinterpret_cast<const void*>(&wide_attn_bwd_kernel<DH, NKT>) : reinterpret_cast<const void*>(&wide_attn_fwd_kernel<DH, NKT>);
    static bool attr[2] = {false, false};
    if (!attr[bwd]) {
        EGX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[bwd] = true;
    }
    timing_begin(bwd ? TIMER_WIDE_ATTN_BWD : TIMER_WIDE_ATTN_FWD, st);
    if (bwd) hipLaunchKernelGGL((wide_attn_bwd_kernel<DH, NKT>), dim3(p.B * p.H), dim3(NKT * 64), lds, st, p);
    else hipLaunchKernelGGL((wide_attn_fwd_kernel<DH, NKT>), dim3(p.B * p.H), dim3(256), lds, st, p);
    timing_end(bwd ? TIMER_WIDE_ATTN_BWD : TIMER_WIDE_ATTN_FWD, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

static int dispatch_attn(const WideAttnParams& p, bool bwd, hipStream_t st) {
    EGX_CHECK(p.qkv && p.lse && p.B > 0 && p.H > 0 && p.d % p.H == 0, "wide attention: bad arguments");
    const int dh = p.d / p.H;
    EGX_CHECK(wide_attn_supported(p.S, dh), "wide attention: S=%d head dim %d unsupported (S <= 128 with head dim 32 / 64 / 96 / 128; longer sequences: head dim 32 / 64 while two images fit the LDS)", p.S, dh);
    EGX_CHECK(p.d % 8 == 0, "wide attention: d_model %% 8 != 0");
    if (p.S > 128) {
        EGX_CHECK(!bwd || (p.delta && p.out), "wide attention (S > 128): the backward needs the saved attention output and a delta buffer");
        return dh == 32 ? launch_attn_long<32>(p, bwd, st) : launch_attn_long<64>(p, bwd, st);
    }
    const bool small = p.S <= 64;
#define EGX_ATTN_CASE(D)                                                                   \
    case D: return small ? launch_attn<D, 4>(p, bwd, st) : launch_attn<D, 8>(p, bwd, st);
    switch (dh) {
        EGX_ATTN_CASE(32)
        EGX_ATTN_CASE(64)
        EGX_ATTN_CASE(96)
        EGX_ATTN_CASE(128)
    }
#undef EGX_ATTN_CASE
    return 1;
}

size_t wide_attn_delta_bytes(int B, int H, int S) { return S > 128 ? (size_t)B * H * S * sizeof(float) : 0; }

int wide_attn_fwd(const WideAttnParams& p, hipStream_t st) {
    EGX_CHECK(p.out, "wide_attn_fwd: null output");
    return dispatch_attn(p, false, st);
}
int wide_attn_bwd(const WideAttnParams& p, hipStream_t st) {
    EGX_CHECK(p.d_out && p.d_qkv, "wide_attn_bwd: null gradient pointer");
    // (S > 128 also reads p.out = the forward's attention output and writes p.delta (B, H, S))
    return dispatch_attn(p, true, st);
}

}  // namespace egx
