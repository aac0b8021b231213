// bf16 MFMA GEMMs of the wide path (d_model >= 256: BASELINE.json configs[3], configs[4]).
//
//   wide_gemm_nt   C[m][n] = sum_k X[m][k] W[n][k]   every forward projection (x W^T) and, with the transposed bf16 weight
//                  copy, every input gradient (dY W); replaces the addmm calls behind nn.Linear / nn.MultiheadAttention /
//                  TransformerEncoderLayer._ff_block (HOI/models/lta/lta_models_lta_transfer.py:268-275,355-361).
//   wide_gemm_tn   C[m][n] += sum_t dY[t][m] X[t][n]  every weight gradient (K = all B*S tokens), split over tokens into
//                  fp32 slabs that are summed in fixed order (deterministic, no atomics).
//
// Both: 128 x 128 output tile per 256-thread workgroup (4 waves, 64 x 64 each as 4 x 4 v_mfma_f32_16x16x32_bf16 tiles),
// 64-deep K steps, operand tiles copied global -> LDS by global_load_lds_dwordx4 (no registers, no ds_write) into two
// stages, one barrier per K step, two workgroups per CU (64 KB of LDS each). Operands stay bf16 in HBM and LDS.
// The MFMA "A" operand (rows -> accumulator registers) is the operand whose index is contiguous in the OUTPUT (n), so a
// lane ends up with 4 consecutive output columns of one row: 16-byte fp32 / 8-byte bf16 stores, and epilogues
// (bias, ReLU, dropout, mask, residual) work on adjacent elements.
// LDS images are written linearly by the DMA (wave base + lane * 16 B); bank conflicts are removed by permuting the
// per-lane SOURCE address and applying the same XOR on the read side:
//   NT: 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7); fragments by ds_read_b128.
//   TN: 256-byte token rows, 32-byte chunk c of row t at chunk c ^ ((t & 3) | ((t >> 3) & 1) << 2); the K axis is the
//       row axis of the image, so fragments come from ds_read_b64_tr_b16 (two per 8-deep fragment).
// Workgroups are numbered so that the blocks sharing an XCD (id % 8) walk consecutive tiles of the same activation rows.
#include "common.h"
#include "wide.h"
#include "fused.h"
#include <stdlib.h>

namespace egx {

namespace {

constexpr int TBM = 128, TBN = 128, TBK = 64;        // TBM: rows per 2x2-wave unit; the big variant stacks two units (256 rows)
constexpr int STAGE_BYTES = (TBM + TBN) * TBK * 2;     // 32 KB (BM = 128); 48 KB for BM = 256

#define EGX_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// workgroup barrier that does NOT drain the vector-memory counter (LDS-DMA stays in flight across it); LDS reads of the
// previous phase are retired first
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t*)g, (lds_ptr_t*)l, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 lds_read128(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds_read_tr2(const unsigned char* p0, const unsigned char* p1) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    bf16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }

// blocks sharing an XCD (id % 8) get a contiguous range of tiles (bijective for any total)
__device__ __forceinline__ int xcd_tile(int id, int total) {
    int xcd = id & 7, j = id >> 3, q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

}  // namespace

// ---- NT ---------------------------------------------------------------------------------------------------------------
// BM x BN output tile, (BM / WTM) x (BN / 64) waves of WTM x 64 each. D LDS stages form a ring: stage kt + D - 1 is issued
// while stage kt is consumed, a counted s_waitcnt leaves the newer stages in flight across the barrier (the L2 / HBM latency
// of a tile is several times the MFMAs a wave runs per K step, so a single prefetched stage leaves the matrix pipe waiting).
// Variants: 256 x 128 (wave 64 x 64, 3 stages) and 128 x 128 (2 stages, two workgroups per CU) fetch 11.4 B per kFLOP
// through L2 -> LDS, which caps them near 700 TFLOP/s (measured: the L2 -> CU path delivers ~6.5 TB/s to 256 CUs running
// them); 256 x 256 (wave 128 x 64, 2 stages of 64 KB) fetches 7.6 B per kFLOP.
// Epilogue of the NT kernels: lane (r, g) of accumulator tile (i, j) holds C[mb + j*16 + r][nb + i*16 + 4g .. +3] (mb, nb: first
// row / column of the wave's WTM x 64 sub-tile; prow0: its first row of the column-sum partial buffer, one per 64 output rows).
template <int TJ>
__device__ __forceinline__ void nt_epilogue(const WideGemmParams& p, f32x4 (&acc)[4][TJ], int mb, int nb, int prow0, int r, int g) {
    constexpr int JG = TJ >= 4 ? TJ / 4 : 1;         // 64-row groups of a wave: one column-sum partial row each (TJ < 4: no column sums, see wide_gemm_nt)
    float cs_part[JG][4][4];
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
#pragma unroll
    for (int jg = 0; jg < JG; ++jg)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) cs_part[jg][i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int m = mb + j * 16 + r;
        const bool mv = m < p.M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nb + i * 16 + 4 * g;
            if (!mv || n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
                float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.drop_thresh) {
                float ds[4];
                drop_scale4(dkey, (uint32_t)m, (uint32_t)n, p.drop_thresh, p.drop_inv, ds);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= ds[e];
            }
            if (p.mask) {
                uint2 mk = *reinterpret_cast<const uint2*>(p.mask + (size_t)m * p.ldm + n);
                v[0] = (mk.x & 0xffffu) ? v[0] * p.mask_scale : 0.f;
                v[1] = (mk.x >> 16) ? v[1] * p.mask_scale : 0.f;
                v[2] = (mk.y & 0xffffu) ? v[2] * p.mask_scale : 0.f;
                v[3] = (mk.y >> 16) ? v[3] * p.mask_scale : 0.f;
            }
            if (p.residual) {
                float4 rs = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldr + n);
                v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
            }
            if (p.Cf) *reinterpret_cast<float4*>(p.Cf + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (p.Cb) {
                uint2 o = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                *reinterpret_cast<uint2*>(p.Cb + (size_t)m * p.ldc + n) = o;
                if (p.colsum) {     // sums of the values as stored (bf16-rounded), so that db == colsum(stored dY) exactly
                    cs_part[j / 4][i][0] += bf2f((bf16_t)(o.x & 0xffffu)); cs_part[j / 4][i][1] += bf2f((bf16_t)(o.x >> 16));
                    cs_part[j / 4][i][2] += bf2f((bf16_t)(o.y & 0xffffu)); cs_part[j / 4][i][3] += bf2f((bf16_t)(o.y >> 16));
                }
            } else if (p.colsum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) cs_part[j / 4][i][e] += v[e];
            }
        }
    }
    if (p.colsum) {      // one partial row per 64 output rows: [ceil(M / 64)][N]
#pragma unroll
        for (int jg = 0; jg < JG; ++jg) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sv = cs_part[jg][i][e];
                    sv += __shfl_xor(sv, 1, 64); sv += __shfl_xor(sv, 2, 64); sv += __shfl_xor(sv, 4, 64); sv += __shfl_xor(sv, 8, 64);
                    cs_part[jg][i][e] = sv;
                }
            if (r == 0) {
                const int prow = prow0 + jg;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = nb + i * 16 + 4 * g;
                    if (n < p.N) *reinterpret_cast<float4*>(p.colsum + (size_t)prow * p.N + n) =
                        make_float4(cs_part[jg][i][0], cs_part[jg][i][1], cs_part[jg][i][2], cs_part[jg][i][3]);
                }
            }
        }
    }
}

// Epilogue through LDS (launches without mask / column sums and with N % 8 == 0): the accumulator layout gives a store
// instruction 16 rows x 32 B (bf16) or 64 B (fp32) - sixteen partial cache lines per instruction, and the CU's one address
// unit takes ~480 cycles for each: 15.5k of a 256 x 256 x 768 tile's 53.7k cycles were the ISSUE of its 32 stores per wave
// (stamps; the stores themselves drain in 1.2k). Here bias / ReLU / dropout are applied in the accumulator layout, the wave
// writes its (TJ * 16) x 64 sub-tile into a private LDS region (the K loop's stages, dead by now) and reads it back row-major:
// a store instruction then covers 8 full 128-byte lines (bf16) or 4 rows x 256 B (fp32; residual rows are read the same way).
// (NI = W tiles of 16 columns per wave: 4, or 3 for the 256 x 192 tile)
template <int NI> __host__ __device__ __forceinline__ constexpr int epi_ldb() { return NI * 32 + 16; }     // bf16 row + pad: 16-byte aligned, rows spread over the banks
template <int NI> __host__ __device__ __forceinline__ constexpr int epi_ldf() { return NI * 64 + 16; }     // fp32 row + pad
template <int TJ, int NI = 4>
__host__ __device__ __forceinline__ constexpr int epi_lds_f32_bytes() { return (TJ >= 4 ? TJ * 4 : 16) * epi_ldf<NI>(); }      // fp32 rounds only (TN slabs)
template <int TJ, int NI = 4>
__host__ __device__ __forceinline__ constexpr int epi_lds_bytes() { return (TJ * 16) * epi_ldb<NI>() > epi_lds_f32_bytes<TJ, NI>() ? (TJ * 16) * epi_ldb<NI>() : epi_lds_f32_bytes<TJ, NI>(); }
// R16: rounds of one 16-row tile through a region of 16 rows (the persistent kernel's epilogue beside its K-loop stages)
template <int TJ, int NI = 4, bool R16 = false>
__device__ __forceinline__ void nt_epilogue_lds(const WideGemmParams& p, f32x4 (&acc)[NI][TJ], int mb, int nb, int prow0, int r, int g, int lane, unsigned char* region) {
    constexpr int EPI_LDB = epi_ldb<NI>(), EPI_LDF = epi_ldf<NI>();
    // phase A (accumulator layout): bias, ReLU, dropout
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    float4 bb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = nb + i * 16 + 4 * g;
        bb[i] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
    }
    auto value = [&](int i, int j, float (&v)[4]) {
        v[0] = acc[i][j][0] + bb[i].x; v[1] = acc[i][j][1] + bb[i].y; v[2] = acc[i][j][2] + bb[i].z; v[3] = acc[i][j][3] + bb[i].w;
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.drop_thresh) {
            float ds[4];
            drop_scale4(dkey, (uint32_t)(mb + j * 16 + r), (uint32_t)(nb + i * 16 + 4 * g), p.drop_thresh, p.drop_inv, ds);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= ds[e];
        }
    };
    if (!p.Cf && !p.residual && !p.mask && !p.colsum) {
        // bf16 out only: the whole sub-tile at once
        // a row is NI * 2 pieces of 16 bytes: RPI rows per instruction (NI = 3: 10 rows, four lanes idle)
        constexpr int LPR = NI * 2, RPI = 64 / LPR, JR = R16 ? 1 : TJ, WR = JR * 16;      // JR row tiles per round
        const int rr = lane / LPR, ch = lane - rr * LPR, n = nb + ch * 8;
#pragma unroll
        for (int h = 0; h < TJ / JR; ++h) {
#pragma unroll
            for (int jj = 0; jj < JR; ++jj)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float v[4];
                    value(i, h * JR + jj, v);
                    *reinterpret_cast<uint2*>(region + (jj * 16 + r) * EPI_LDB + (i * 16 + 4 * g) * 2) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                }
#pragma unroll
            for (int k = 0; k < (WR + RPI - 1) / RPI; ++k) {
                const int row = k * RPI + rr, m = mb + h * WR + row;
                if (rr < RPI && row < WR) {
                    const uint4 q = *reinterpret_cast<const uint4*>(region + row * EPI_LDB + ch * 16);
                    if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = q;
                }
            }
        }
        return;
    }
    // fp32 (and optionally bf16) out, optional bf16 mask, fp32 residual, column sums: groups of RG row tiles through the region
    constexpr int RG = 1;                              // row tiles per round (two per round cost the 256-register ping-pong kernel 140 B of scratch)
    constexpr int LPR = NI * 4, RPI = 64 / LPR;        // lanes per row (16-byte pieces), rows per instruction (NI = 3: 5 rows, four lanes idle)
    static_assert(NI == 4 || NI == 3, "column sums below assume four lane rows (NI = 4)");
    const int rr = lane / LPR, ch = lane - rr * LPR, n = nb + ch * 4;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};                // column sums of the lane's four columns over the current 64-row group
    auto cs_flush = [&](int jg) {
        if (!p.colsum) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[e] += __shfl_xor(cs[e], 16, 64); cs[e] += __shfl_xor(cs[e], 32, 64); }
        if (rr == 0 && n < p.N) *reinterpret_cast<float4*>(p.colsum + (size_t)(prow0 + jg) * p.N + n) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        cs[0] = cs[1] = cs[2] = cs[3] = 0.f;
    };
    constexpr int KI = (RG * 16 + RPI - 1) / RPI;
    // Residual / mask rows of round h + 1 are requested BEFORE the stores of round h, unconditionally (clamped addresses): a load
    // that is used while older stores are in flight waits for those stores, and a load under a branch is waited for at once -
    // with the loads inside the store loop every one of the 32 row groups of a 256 x 256 tile paid a store round trip (the
    // fp32 + residual epilogue cost 55k cycles per tile: FFN2 forward 152 us against 99 us for the plain GEMM).
    f32x4 rs[KI];            // (one set: round h + 1 is requested behind the last use of round h)
    uint2 mk[KI];
    // Fast path for the commonest fp32 epilogue - C = acc (+ bias ...) + residual into fp32, nothing else, the wave's sub-tile fully
    // inside the output: every load and store below is unconditional, so hipcc counts them and the wait for round h + 1's
    // residual rows is vmcnt(stores of round h) instead of vmcnt(0). Idle lanes (NI = 3) mirror lane row 0 / row 15.
    if (p.Cf && p.residual && !p.Cb && !p.mask && !p.colsum && mb + TJ * 16 <= p.M && nb + NI * 16 <= p.N) {
        int rowl[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) { int rw = k * RPI + (rr < RPI ? rr : 0); rowl[k] = rw < 16 ? rw : 15; }
        f32x4 ra[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) ra[k] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(mb + rowl[k]) * p.ldr + n);
#pragma unroll
        for (int h = 0; h < TJ; ++h) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float v[4];
                value(i, h, v);
                *reinterpret_cast<float4*>(region + r * EPI_LDF + (i * 16 + 4 * g) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
            f32x4 qv[KI];
#pragma unroll
            for (int k = 0; k < KI; ++k) qv[k] = *reinterpret_cast<const f32x4*>(region + rowl[k] * EPI_LDF + ch * 16) + ra[k];
            if (h + 1 < TJ) {
#pragma unroll
                for (int k = 0; k < KI; ++k) ra[k] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(mb + (h + 1) * 16 + rowl[k]) * p.ldr + n);
            }
#pragma unroll
            for (int k = 0; k < KI; ++k) *reinterpret_cast<f32x4*>(p.Cf + (size_t)(mb + h * 16 + rowl[k]) * p.ldc + n) = qv[k];
        }
        return;
    }
    const bool side = p.residual || p.mask;
    auto fetch = [&](int h) {
        if (!side) return;
        const int nn = n < p.N ? n : 0;
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            int m = mb + h * RG * 16 + k * RPI + rr;
            m = m < p.M ? m : p.M - 1;
            if (p.residual) rs[k] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.ldr + nn);
            if (p.mask) mk[k] = *reinterpret_cast<const uint2*>(p.mask + (size_t)m * p.ldm + nn);
        }
    };
    fetch(0);
#pragma unroll
    for (int h = 0; h < TJ / RG; ++h) {
#pragma unroll
        for (int jj = 0; jj < RG; ++jj)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float v[4];
                value(i, h * RG + jj, v);
                *reinterpret_cast<float4*>(region + (jj * 16 + r) * EPI_LDF + (i * 16 + 4 * g) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        f32x4 qq[KI];           // (vector values: struct arrays end up in scratch memory)
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int row = k * RPI + rr;
            const int rowc = (rr < RPI && row < RG * 16) ? row : 0;            // idle lanes re-read row 0 (never stored)
            f32x4 q = *reinterpret_cast<const f32x4*>(region + rowc * EPI_LDF + ch * 16);
            if (p.mask) {
                const uint2 mv = mk[k];
                q[0] = (mv.x & 0xffffu) ? q[0] * p.mask_scale : 0.f;
                q[1] = (mv.x >> 16) ? q[1] * p.mask_scale : 0.f;
                q[2] = (mv.y & 0xffffu) ? q[2] * p.mask_scale : 0.f;
                q[3] = (mv.y >> 16) ? q[3] * p.mask_scale : 0.f;
            }
            if (p.residual) q += rs[k];
            qq[k] = q;
        }
        if (h + 1 < TJ / RG) fetch(h + 1);
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int row = k * RPI + rr, m = mb + h * RG * 16 + row;
            if (!(rr < RPI && row < RG * 16)) continue;
            const f32x4 q = qq[k];
            if (m < p.M && n < p.N) {
                if (p.Cf) *reinterpret_cast<f32x4*>(p.Cf + (size_t)m * p.ldc + n) = q;
                if (p.Cb) {
                    const uint2 o = make_uint2(pack2(q[0], q[1]), pack2(q[2], q[3]));
                    *reinterpret_cast<uint2*>(p.Cb + (size_t)m * p.ldc + n) = o;
                    if (p.colsum) {     // sums of the values as stored (bf16-rounded), so that db == colsum(stored dY) exactly
                        cs[0] += bf2f((bf16_t)(o.x & 0xffffu)); cs[1] += bf2f((bf16_t)(o.x >> 16));
                        cs[2] += bf2f((bf16_t)(o.y & 0xffffu)); cs[3] += bf2f((bf16_t)(o.y >> 16));
                    }
                } else if (p.colsum) {
                    cs[0] += q[0]; cs[1] += q[1]; cs[2] += q[2]; cs[3] += q[3];
                }
            }
        }
        // one partial row per 64 output rows: the rounds cover RG * 16 rows each
        if (((h + 1) * RG * 16) % 64 == 0 || h + 1 == TJ / RG) cs_flush((h * RG * 16) / 64);
    }
}
// (TJ < 4: 16-row waves cannot produce the one-partial-row-per-64-rows column sums)
template <int TJ, int NI = 4>
__device__ __forceinline__ bool nt_epilogue_simple(const WideGemmParams& p) {
    return (p.N & 7) == 0 && (p.epi_lds == 2 ? ((TJ >= 4 && NI == 4) || !p.colsum) : (p.epi_lds == 1 && !p.mask && !p.colsum));
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) EGX_WAIT_VM(0);
    else if constexpr (N == 4) EGX_WAIT_VM(4);
    else if constexpr (N == 6) EGX_WAIT_VM(6);
    else if constexpr (N == 8) EGX_WAIT_VM(8);
    else if constexpr (N == 10) EGX_WAIT_VM(10);
    else if constexpr (N == 12) EGX_WAIT_VM(12);
    else if constexpr (N == 16) EGX_WAIT_VM(16);
    else static_assert(N == 0, "add the s_waitcnt immediate");
}
template <int BM, int BN, int WTM, int D>
__global__ __launch_bounds__((BM / WTM) * (BN / 64) * 64, 1) void wide_gemm_nt_kernel(WideGemmParams p, int ntM, int ntN) {
    constexpr int WM = BM / WTM;                     // waves along m
    constexpr int NW = WM * (BN / 64);               // waves
    constexpr int TJ = WTM / 16;                     // 16-row tiles of X per wave (4 W tiles of 16 columns)
    constexpr int SB = (BM + BN) * TBK * 2;          // stage bytes
    constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;   // staging instructions per wave (8 rows = 1 KiB each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_tile(blockIdx.x, ntM * ntN);
    const int m0 = (t / ntN) * BM, n0 = (t % ntN) * BN;

    // staging: wave w copies rows [8 NA w, + 8 NA) of the X tile and [8 NB w, + 8 NB) of the W tile
    const int srow = lane >> 3;                       // row within the 8-row group == (row & 7)
    const int lch = (lane & 7) ^ srow;                // logical 16-byte chunk this lane fetches
    const bf16_t* srcA[NA];
    const bf16_t* srcB[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int gm = m0 + (wave * NA + j) * 8 + srow; gm = gm < p.M ? gm : p.M - 1;
        srcA[j] = p.A + (size_t)gm * p.lda + lch * 8;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int gn = n0 + (wave * NB + j) * 8 + srow; gn = gn < p.N ? gn : p.N - 1;
        srcB[j] = p.B + (size_t)gn * p.ldb + lch * 8;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * SB + wave * NA * 1024;
        unsigned char* sb = smem + buf * SB + BM * TBK * 2 + wave * NB * 1024;
        const int k0 = kt * TBK;
#pragma unroll
        for (int j = 0; j < NA; ++j) glds16(srcA[j] + k0, sa + j * 1024);
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16(srcB[j] + k0, sb + j * 1024);
    };

    const int r = lane & 15, g = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    // fragment byte offsets inside a stage: rows of X (m) / W (n); chunk (s * 4 + g) ^ (row & 7), row & 7 == r & 7
    const int offX = (wm * WTM + r) * 128, offW = BM * TBK * 2 + (wn * 64 + r) * 128;
    const int c0 = ((0 * 4 + g) ^ (r & 7)) * 16, c1 = ((1 * 4 + g) ^ (r & 7)) * 16;

    f32x4 acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int nk = p.K / TBK;
#pragma unroll
    for (int s0 = 0; s0 < D - 1; ++s0)
        if (s0 < nk) stage(s0, s0);
    int buf = 0, nbuf = D - 1;                       // buffer of stage kt / of stage kt + D - 1
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt must have landed; the (up to D - 2) stages issued after it may stay in flight
        if (D >= 4 && kt + 2 < nk) wait_vm<2 * (NA + NB)>();
        else if (D >= 3 && kt + 1 < nk) wait_vm<NA + NB>();
        else wait_vm<0>();
        ring_barrier();
        if (kt + D - 1 < nk) stage(kt + D - 1, nbuf);
        const unsigned char* st = smem + buf * SB;
        buf = buf + 1 == D ? 0 : buf + 1;
        nbuf = nbuf + 1 == D ? 0 : nbuf + 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cs = s ? c1 : c0;
            bf16x8 fw[4], fx[TJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = lds_read128(st + offW + i * 2048 + cs);
#pragma unroll
            for (int j = 0; j < TJ; ++j) fx[j] = lds_read128(st + offX + j * 2048 + cs);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
    }

    if constexpr (NW * epi_lds_bytes<TJ>() <= D * SB) {
        if (nt_epilogue_simple<TJ>(p)) {
            __syncthreads();        // the last stage's fragment reads are over in every wave: the stages become the wave regions
            nt_epilogue_lds<TJ>(p, acc, m0 + wm * WTM, n0 + wn * 64, m0 / 64 + wm * (WTM / 64), r, g, lane, smem + wave * epi_lds_bytes<TJ>());
            return;
        }
    }
    nt_epilogue<TJ>(p, acc, m0 + wm * WTM, n0 + wn * 64, m0 / 64 + wm * (WTM / 64), r, g);
}

template <int BM, int BN, int WTM, int D>
static int launch_nt(const WideGemmParams& p, hipStream_t st) {
    constexpr int LDS = D * (BM + BN) * TBK * 2;
    constexpr int THREADS = (BM / WTM) * (BN / 64) * 64;
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_nt_kernel<BM, BN, WTM, D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int ntM = cdiv(p.M, BM), ntN = cdiv(p.N, BN);
    timing_begin(TIMER_WIDE_GEMM, st);
    hipLaunchKernelGGL((wide_gemm_nt_kernel<BM, BN, WTM, D>), dim3(ntM * ntN), dim3(THREADS), LDS, st, p, ntM, ntN);
    timing_end(TIMER_WIDE_GEMM, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- NT, 256 x 256 tile, two staggered wave groups ("ping-pong") --------------------------------------------------------
// 8 waves; wave (wr = wave >> 2, wn = wave & 3) owns the 128 x 64 sub-tile (rows 128 wr, columns 64 wn). A 64-deep K tile is
// four phases of 16 MFMAs, one per quadrant of the sub-tile (X half of 64 rows x W half of 32 columns x K = 64):
//   P1 reads X half 0 + W half 0 (12 ds_read_b128), P2 reads W half 1 (4), P3 reads X half 1 (8), P4 reads nothing.
// Every phase is  { fragment reads; two LDS-DMA staging instructions; counted vmcnt; lgkmcnt(0) }  barrier  { 16 MFMAs }
// barrier, and the waves 4-7 run ONE BARRIER behind the waves 0-3: while one wave of a SIMD issues MFMAs its partner reads
// fragments and stages, so the LDS reads, the DMA issue and the waits sit under the partner's matrix work instead of in
// front of the wave's own (one barrier per K step with all eight waves in lockstep left the matrix pipe idle during every
// read burst: 0.27-0.36 of peak). Two 64 KB stages; a region of a stage is restaged as soon as its last reader phase is over:
//   in tile T:  P1 stages X quarters 1, 3 of tile T+1 (other stage; last read in P3 of T-1),  P2 X quarters 0, 2 of T+2 (this
//   stage; last read in P1),  P3 / P4 the W rows of T+2 that are read in P1 / P2. Each wave stages 8 rows of every region, so
//   its own vmcnt covers a slice of everything; loads complete in order, two per phase: vmcnt(10) after a phase's issue says
//   "everything issued five or more phases ago has landed", which is at least one phase earlier than any region is read
//   (6-7 phases after its issue), and a barrier separates that wait from the reads.
// WAR: a region's restaging is issued at least one barrier after the lgkmcnt(0) that retired its last reads in BOTH groups.
#ifdef EGX_STAMPS
__device__ unsigned long long g_ppstamps[8];
#define PPSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ppstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
int debug_read_ppstamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ppstamps), sizeof(g_ppstamps)) == hipSuccess ? 0 : 1; }
#else
#define PPSTAMP(i) do { } while (0)
int debug_read_ppstamps(unsigned long long*) { return 1; }
#endif
// NWT = W tiles (16 output columns) per wave: 4 -> the 256 x 256 tile described above; 3 -> a 256 x 192 tile for outputs whose
// 256-column tiles leave the chip's last round half empty (N = 768: 384 tiles = 1.5 rounds, 512 tiles of 192 columns = 2 full
// rounds of 0.75 the work). Its W half 1 is one tile (P2 / P3: 8 MFMAs), staged by one instruction per wave in P4: seven LDS-DMA
// loads per K tile, so the uniform wait is vmcnt(8) - any five consecutive phases issue at least eight.
// PERSIST: the launch has one workgroup per CU and each walks tiles id, id + grid, ...: the next tile's prologue (14 LDS-DMA loads
// into the K-loop stages) is issued BEFORE the epilogue of the finished tile, which then goes through a small LDS region of its
// own behind the stages (rounds of 16 rows) - the 3.5k-cycle prologue and part of the store drain disappear under the epilogue.
// bytes of a wave's 16-row epilogue region in the persistent kernel: bf16 rows only for the 256-column tile (its fp32 rounds do not fit)
template <int NWT> __host__ __device__ __forceinline__ constexpr int pp_region_bytes() { return NWT == 4 ? 16 * epi_ldb<4>() : epi_lds_f32_bytes<1, NWT>(); }
template <int NWT, bool PERSIST>
__global__ __launch_bounds__(512, 1) void wide_gemm_nt_pp_kernel(WideGemmParams p, int ntM, int ntN) {
    constexpr int BM = 256, BN = 64 * NWT, WC = 16 * NWT, SB = (BM + BN) * TBK * 2, WOFF = BM * TBK * 2;
    constexpr int H1 = NWT - 2;                  // tiles of W half 1 (half 0: tiles 0, 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PPSTAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wn = wave & 3;
    const int ntiles = ntM * ntN;
    int id = blockIdx.x, m0, n0;
    const int nk = p.K / TBK;

    // staging: one instruction = 8 rows x 128 B; lane -> row (lane >> 3), 16-byte chunk (lane & 7) ^ row of the SOURCE
    const int srow = lane >> 3, lch = (lane & 7) ^ srow;
    const bf16_t* srcX[4];      // X quarter q: rows 64 q + 8 wave .. + 8
    const bf16_t* srcW[4];      // W pieces of the P1 rows (j = 0, 1) and of the P2 rows (j = 2, 3; NWT = 3: j = 2 only)
    int dstX[4], dstW[4];
    auto set_tile = [&](int tid_) {
    const int t = xcd_tile(tid_, ntiles);
    m0 = (t / ntN) * BM; n0 = (t % ntN) * BN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 64 * q + 8 * wave;
        int gm = m0 + row + srow; gm = gm < p.M ? gm : p.M - 1;
        srcX[q] = p.A + (size_t)gm * p.lda + lch * 8;
        dstX[q] = row * 128;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // wave-column wn reads W rows WC wn + [0, 32) in P1 and WC wn + [32, WC) in P2; pieces of 8 rows
        int row;
        if (j < 2) { const int idx = wave * 2 + j; row = (idx >> 2) * WC + (idx & 3) * 8; }                       // 16 pieces of half 0
        else if (NWT == 4) { const int idx = wave * 2 + (j & 1); row = (idx >> 2) * WC + 32 + (idx & 3) * 8; }      // 16 pieces of half 1
        else { row = (wave >> 1) * WC + 32 + (wave & 1) * 8; }                                                      // 8 pieces of half 1 (j = 2)
        int gn = n0 + row + srow; gn = gn < p.N ? gn : p.N - 1;
        srcW[j] = p.B + (size_t)gn * p.ldb + lch * 8;
        dstW[j] = WOFF + row * 128;
    }
    };
    set_tile(id);
    auto koff = [&](int kt) { return (kt < nk ? kt : nk - 1) * TBK; };      // tiles past the end re-fetch the last one (never read)
    auto stage_x = [&](int kt, int qa, int qb) {
        unsigned char* st = smem + (kt & 1) * SB;
        const int k0 = koff(kt);
        glds16(srcX[qa] + k0, st + dstX[qa]);
        glds16(srcX[qb] + k0, st + dstX[qb]);
    };
    auto stage_w = [&](int kt, int h) {
        unsigned char* st = smem + (kt & 1) * SB;
        const int k0 = koff(kt);
        glds16(srcW[2 * h] + k0, st + dstW[2 * h]);
        if (h == 0 || NWT == 4) glds16(srcW[2 * h + 1] + k0, st + dstW[2 * h + 1]);
    };

    const int r = lane & 15, g = lane >> 4;
    const int offX = (wr * 128 + r) * 128, offW = WOFF + (wn * WC + r) * 128;
    const int c0 = ((0 * 4 + g) ^ (r & 7)) * 16, c1 = ((1 * 4 + g) ^ (r & 7)) * 16;

    f32x4 acc[NWT][8];

#define EGX_PP_WAIT() do { if (NWT == 4) EGX_WAIT_VM(10); else EGX_WAIT_VM(8); } while (0)
    // prologue, in the order the steady state would have issued them: tile 0 complete, tile 1 up to its P2-P4 regions
    auto prologue = [&]() {
        stage_x(0, 0, 2); stage_w(0, 0); stage_w(0, 1); stage_x(0, 1, 3);
        stage_x(1, 0, 2); stage_w(1, 0); stage_w(1, 1);
    };
    prologue();
    for (;;) {
#pragma unroll
    for (int i = 0; i < NWT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    EGX_PP_WAIT();          // all but the first four (X quarters 0, 2 and W half 0 of tile 0) may still be in flight (after an
                            // epilogue its stores are younger than the prologue: the wait covers some of them too - safe)
    ring_barrier();
    PPSTAMP(1);
    if (wr == 1) ring_barrier();        // the stagger: waves 4-7 run one barrier behind

    bf16x8 fx[4][2], fw[NWT][2];        // X half (4 row tiles x 2 K sub-steps), W tiles (both halves stay live for P4)
#define EGX_PP_PHASE_TAIL()                                  \
    EGX_PP_WAIT();                                           \
    ring_barrier();                                          \
    __builtin_amdgcn_s_setprio(1);
#define EGX_PP_MFMA(XH, I0, NI_)                                                                                         \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                        \
        _Pragma("unroll") for (int i = 0; i < NI_; ++i)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
                acc[I0 + i][XH * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[I0 + i][s], fx[j][s], acc[I0 + i][XH * 4 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    ring_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* st = smem + (kt & 1) * SB;
        // P1: X half 0, W half 0
#pragma unroll
        for (int j = 0; j < 4; ++j) { fx[j][0] = lds_read128(st + offX + j * 2048 + c0); fx[j][1] = lds_read128(st + offX + j * 2048 + c1); }
#pragma unroll
        for (int i = 0; i < 2; ++i) { fw[i][0] = lds_read128(st + offW + i * 2048 + c0); fw[i][1] = lds_read128(st + offW + i * 2048 + c1); }
        stage_x(kt + 1, 1, 3);
        EGX_PP_PHASE_TAIL();
        EGX_PP_MFMA(0, 0, 2);
        // P2: W half 1
#pragma unroll
        for (int i = 2; i < NWT; ++i) { fw[i][0] = lds_read128(st + offW + i * 2048 + c0); fw[i][1] = lds_read128(st + offW + i * 2048 + c1); }
        stage_x(kt + 2, 0, 2);
        EGX_PP_PHASE_TAIL();
        EGX_PP_MFMA(0, 2, H1);
        // P3: X half 1 (the registers of half 0 are dead)
#pragma unroll
        for (int j = 0; j < 4; ++j) { fx[j][0] = lds_read128(st + offX + (4 + j) * 2048 + c0); fx[j][1] = lds_read128(st + offX + (4 + j) * 2048 + c1); }
        stage_w(kt + 2, 0);
        EGX_PP_PHASE_TAIL();
        EGX_PP_MFMA(1, 2, H1);
        // P4: no reads
        stage_w(kt + 2, 1);
        EGX_PP_PHASE_TAIL();
        EGX_PP_MFMA(1, 0, 2);
    }
    PPSTAMP(2);
    if (wr == 0) ring_barrier();        // balance the stagger
    EGX_WAIT_VM(0);                     // no LDS-DMA may outlive the workgroup (the over-fetched K tiles; they are never read)
    PPSTAMP(3);
    if constexpr (PERSIST) {
        ring_barrier();     // every wave's LDS-DMA has landed, every fragment read is over: the stages may be refilled
        const int mc = m0, nc = n0, nid = id + (int)gridDim.x;
        const bool more = nid < ntiles;
        if (more) { set_tile(nid); prologue(); }
        nt_epilogue_lds<8, NWT, true>(p, acc, mc + wr * 128, nc + wn * WC, mc / 64 + wr * 2, r, g, lane, smem + 2 * SB + wave * pp_region_bytes<NWT>());
        if (!more) break;
        id = nid;
    } else {
        if (NWT != 4 || nt_epilogue_simple<8, NWT>(p)) {
            ring_barrier();     // every wave's LDS-DMA has landed (each waited for its own above): the stages are free for the wave regions
            nt_epilogue_lds<8, NWT>(p, acc, m0 + wr * 128, n0 + wn * WC, m0 / 64 + wr * 2, r, g, lane, smem + wave * epi_lds_bytes<8, NWT>());
        } else if constexpr (NWT == 4) {
            nt_epilogue<8>(p, acc, m0 + wr * 128, n0 + wn * 64, m0 / 64 + wr * 2, r, g);
        }
        break;
    }
    }
#undef EGX_PP_PHASE_TAIL
#undef EGX_PP_MFMA
#undef EGX_PP_WAIT
    PPSTAMP(4);
#ifdef EGX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PPSTAMP(5);
#endif
}

// persistent launch (one workgroup per CU walking the tiles): when the output needs more than one round of tiles and the
// epilogue's 16-row regions fit behind the two stages (fp32 rows of the 256-column tile do not: 131 072 + 8 x 4 352 > 160 KB)
// OFF by default (EGX_WIDE_PERSIST=1 enables it): with the uniform vmcnt(10) waits the first phases of the next tile wait for
// the epilogue's stores anyway, and the 16-row epilogue rounds cost more than the hidden prologue saves - same-box C4 9.31 ms
// without, 9.61 ms with; gemm_bench +4 % / -2 % / -2.5 % / +3 %. Worth revisiting with exact store counts.
static int pp_persist_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_WIDE_PERSIST"); v = e ? atoi(e) : 0; }
    return v;
}
template <int NWT, bool PERSIST>
static int launch_nt_pp_impl(const WideGemmParams& p, hipStream_t st) {
    constexpr int BN = 64 * NWT;
    constexpr int STAGES = 2 * (256 + BN) * TBK * 2, EPI = 8 * epi_lds_bytes<8, NWT>();
    constexpr int LDS = PERSIST ? STAGES + 8 * pp_region_bytes<NWT>() : (STAGES > EPI ? STAGES : EPI);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_nt_pp_kernel<NWT, PERSIST>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int ntM = cdiv(p.M, 256), ntN = cdiv(p.N, BN);
    int grid = ntM * ntN;
    if (PERSIST) {
        static int cus = 0;
        if (!cus) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; if (cus < 8) cus = 256; cus &= ~7; }
        if (grid > cus) grid = cus;      // a multiple of 8: the workgroup -> XCD mapping of xcd_tile() stays valid for id + k * grid
    }
    timing_begin(TIMER_WIDE_GEMM, st);
    hipLaunchKernelGGL((wide_gemm_nt_pp_kernel<NWT, PERSIST>), dim3(grid), dim3(512), LDS, st, p, ntM, ntN);
    timing_end(TIMER_WIDE_GEMM, st);
    EGX_LAUNCH_CHECK();
    return 0;
}
template <int NWT>
static int launch_nt_pp(const WideGemmParams& p, hipStream_t st) {
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 64 * NWT);
    const bool fp32_rounds = p.Cf || p.residual || p.mask || p.colsum;
    if (pp_persist_mode() && tiles > 256 && p.epi_lds && (p.N & 7) == 0 && !p.colsum && !(NWT == 4 && fp32_rounds))
        return launch_nt_pp_impl<NWT, true>(p, st);
    return launch_nt_pp_impl<NWT, false>(p, st);
}

// tile choice. 256 x 256 (EGX_WIDE_TILE=512 forces it, =256 / =128 force the others): N a multiple of 256 and at least 3.5
// rounds of tiles over the 256 CUs (its 1.5x lower operand traffic is worth nothing in a half-empty last round)
static int nt_variant(int M, int N) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("EGX_WIDE_TILE"); force = e ? atoi(e) : 0; }
    if (force == 1024) return 3;        // the ping-pong 256 x 256 kernel
    if (force == 512) return 2;
    if (force == 256) return 1;
    if (force == 128) return 0;
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    // the ping-pong kernel once its tiles fill the chip at least once (same-box A/B against the one-barrier 256 x 256 / 256 x 128
    // variants at M = 32768: N = 2048, K = 768: 872 vs 832 TFLOP/s; N = 768, K = 2048 / 2304: 917 / 950 vs 849 / 892; equal
    // on the K = 768, N = 768 / 2304 shapes); below that the 128-row tiles spread the work over more CUs
    if (t256 >= 256 && (N % 256 == 0 || N >= 1024)) return 3;
    return (long)cdiv(M, 256) * cdiv(N, TBN) >= 512 ? 1 : 0;
}
// rows of the `colsum` partial buffer written by wide_gemm_nt for an (M, N) output: one per 64 output rows of every tile
int wide_gemm_nt_colsum_rows(int M, int N) { return nt_variant(M, N) ? cdiv(M, 256) * 4 : cdiv(M, 128) * 2; }     // one per 64 rows of every tile

static int epi_lds_mode() {     // EGX_WIDE_EPI: 0 = accumulator-layout stores, 1 = through LDS without mask / column sums, 2 = through LDS always
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_WIDE_EPI"); v = e ? atoi(e) : 2; }
    return v;
}
int wide_gemm_nt(const WideGemmParams& p_in, hipStream_t st) {
    WideGemmParams p = p_in;
    p.epi_lds = epi_lds_mode();
    EGX_CHECK(p.A && p.B && (p.Cf || p.Cb), "wide_gemm_nt: null operand");
    EGX_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "wide_gemm_nt: empty problem %dx%dx%d", p.M, p.N, p.K);
    EGX_CHECK(p.K % TBK == 0 && p.N % 4 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0,
              "wide_gemm_nt: %dx%dx%d needs K %% 64 == 0, N %% 4 == 0, 16-byte aligned rows", p.M, p.N, p.K);
    const int v = nt_variant(p.M, p.N);
    if (v == 3) {
        // 256 x 192 tiles when they fill the last round better (cost = rounds x tile width): N = 768 is 1.5 rounds of 256-column
        // tiles (2 rounds of time) or exactly 2 rounds of 192-column tiles (1.5). EGX_WIDE_PP192=0 keeps the 256 x 256 tile.
        static int pp192 = -1;
        if (pp192 < 0) { const char* e = getenv("EGX_WIDE_PP192"); pp192 = e ? atoi(e) : 1; }
        if (pp192 && !p.colsum && p.N % 192 == 0) {
            const long tm = cdiv(p.M, 256);
            const long c256 = cdiv(tm * cdiv(p.N, 256), 256) * 256, c192 = cdiv(tm * cdiv(p.N, 192), 256) * 192;
            if (c192 * 5 <= c256 * 4) return launch_nt_pp<3>(p, st);       // a 192-column tile costs ~0.87, not 0.75, of a 256-column one (measured: N = 2304, 0.9 of the rounds x width, lost 9 %)
        }
        return launch_nt_pp<4>(p, st);
    }
    if (v == 2) return launch_nt<256, 256, 128, 2>(p, st);
    // 256-row tiles (8 waves, 3-stage ring) once they fill the chip twice over; 128-row tiles (4 waves) below
    if (v == 1) return launch_nt<256, 128, 64, 3>(p, st);
    // 128 x 128 tiles: two workgroups per CU with two stages each once the tiles outnumber the CUs; a launch that cannot
    // give every CU a second workgroup anyway (the decoder's B * sy <= 2048 target rows: 16-64 tiles) runs four stages deep,
    // so that a K step costs a third of a memory round trip instead of a whole one (EGX_WIDE_SMALL_STAGES = 2 / 4 forces one)
    static int small_stages = -1;
    if (small_stages < 0) { const char* e = getenv("EGX_WIDE_SMALL_STAGES"); small_stages = e ? atoi(e) : 0; }
    const long tiles = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
    // at most one 128 x 128 tile per CU (the decoder's projections of its B * sy target rows: 16-64 tiles; the d = 256 encoder
    // of the HHI EgoT2-g: 180): 64 x 64 tiles, four waves of 16 x 64 and four stages, put four times as many workgroups on a GEMM
    // whose duration is a chain of memory round trips, not MFMA time. Same-box sweep of the threshold (64 / 128 / 256 tiles):
    // C5 HOI 4.02 / 3.98 / 3.99 ms, C5 HHI 2.51 / 2.52 / 2.43 ms, C4 unchanged; without the variant 4.32 / 2.78 ms.
    // (the column-sum epilogue needs 64-row waves: those launches keep the 128 x 128 tiles)
    static int tiny = -1;
    if (tiny < 0) { const char* e = getenv("EGX_WIDE_TINY"); tiny = e ? atoi(e) : 256; }      // threshold in 128 x 128 tiles (0: off)
    if (tiles <= tiny && !p.colsum) return launch_nt<64, 64, 16, 4>(p, st);
    const int stages = small_stages ? small_stages : (tiles <= 256 ? 4 : 2);
    return stages == 2 ? launch_nt<128, 128, 64, 2>(p, st) : launch_nt<128, 128, 64, 4>(p, st);
}

// ---- TN ---------------------------------------------------------------------------------------------------------------
// BM x BN output tile (BM columns of dY against BN columns of X), (BM / WTM) x (BN / 64) waves of WTM x 64, ring of D stages
// as above. A stage holds BM / 128 dY images and BN / 128 X images, each [64 tokens][128 columns] (256-byte rows).
// Variants: 256 x 128 / 128 x 128 (wave 64 x 64) and 256 x 256 (wave 128 x 64, 2 stages): 1.5x fewer operand bytes per FLOP.
// (the body takes the tile and split ids as arguments: wide_gemm_tn_kernel passes its block ids, the GROUPED launch those of the
// problem the workgroup belongs to)
template <int BM, int BN, int WTM, int D>
__device__ __forceinline__ void tn_tile(const WideGemmParams& p, int tile_id, int split, int ntM, int ntN, int kps, float* slabs, size_t slab_stride) {
    constexpr int WM = BM / WTM, NW = WM * (BN / 64);               // waves along m, waves
    constexpr int NIY = BM / 128, NIX = BN / 128, NI = NIY + NIX;   // images per stage
    constexpr int TJ = WTM / 16;
    constexpr int IMG = TBK * 128 * 2;                              // 16 KB
    constexpr int SB = NI * IMG;
    constexpr int PER = NI * 16 / NW;                               // staging instructions per wave per stage
    static_assert(NI * 16 % NW == 0, "staging instructions must divide among the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = tile_id;
    const int m0 = (t / ntN) * BM, n0 = (t % ntN) * BN;
    const int kbeg = split * kps, kend = min(p.K, kbeg + kps);

    // staging: 4 token rows (256 B each) of one image per instruction
    const int slot = lane & 15, half = slot & 1, pch = slot >> 1;
    int rowS[PER], colS[PER], ldS[PER];
    const bf16_t* baseS[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int q = wave * PER + j, img = q >> 4, ii = q & 15;
        const int row = ii * 4 + (lane >> 4);
        const int sw = (row & 3) | (((row >> 3) & 1) << 2);
        rowS[j] = row;
        colS[j] = (pch ^ sw) * 16 + half * 8;
        const bool isx = img >= NIY;
        baseS[j] = isx ? p.B + n0 + (img - NIY) * 128 : p.A + m0 + img * 128;
        ldS[j] = isx ? p.ldb : p.lda;
    }
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(p.zero_page) + slot * 16;
    auto stage = [&](int k0, int buf) {
        unsigned char* dst = smem + buf * SB + wave * PER * 1024;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int tok = k0 + rowS[j];
            const void* a = tok < kend ? (const void*)(baseS[j] + (size_t)tok * ldS[j] + colS[j]) : (const void*)zero;
            glds16(a, dst + j * 1024);
        }
    };

    const int r = lane & 15, g = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    // transposed reads: lane supplies row 8g + (r >> 2) (+4) of the 32-token slice, 4 columns 4 * (r & 3) of its 16-wide tile
    const int swz = (r >> 2) | ((g & 1) << 2);
    const int rowoff = (8 * g + (r >> 2)) * 256 + 8 * (r & 3);
    const int imgY = ((wm * WTM) / 128) * IMG, cY = ((wm * WTM) % 128) / 16;
    const int imgX = (NIY + (wn >> 1)) * IMG, cX = (wn & 1) * 4;

    f32x4 acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int nk = (kend - kbeg + TBK - 1) / TBK;
#pragma unroll
    for (int s0 = 0; s0 < D - 1; ++s0)
        if (s0 < nk) stage(kbeg + s0 * TBK, s0);
    int buf = 0, nbuf = D - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (D >= 4 && kt + 2 < nk) wait_vm<2 * PER>();
        else if (D >= 3 && kt + 1 < nk) wait_vm<PER>();
        else wait_vm<0>();
        ring_barrier();
        if (kt + D - 1 < nk) stage(kbeg + (kt + D - 1) * TBK, nbuf);
        const unsigned char* sy = smem + buf * SB + imgY;              // dY image: [64 tokens][128 m]
        const unsigned char* sx = smem + buf * SB + imgX;              // X  image: [64 tokens][128 n]
        buf = buf + 1 == D ? 0 : buf + 1;
        nbuf = nbuf + 1 == D ? 0 : nbuf + 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fx[4], fy[TJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // MFMA A operand: X^T rows n = wn*64 + i*16 + r
                const unsigned char* q0 = sx + s * 32 * 256 + rowoff + (((cX + i) ^ swz) * 32);
                fx[i] = lds_read_tr2(q0, q0 + 4 * 256);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {      // MFMA B operand: dY columns m = wm*WTM + j*16 + r
                const unsigned char* q0 = sy + s * 32 * 256 + rowoff + (((cY + j) ^ swz) * 32);
                fy[j] = lds_read_tr2(q0, q0 + 4 * 256);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fy[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = slabs + (size_t)split * slab_stride;
    if constexpr (NW * epi_lds_f32_bytes<TJ>() <= D * SB) {
        if (p.epi_lds) {        // the slab tile through LDS: row-major stores of full cache lines (see nt_epilogue_lds)
            WideGemmParams q;
            q.Cf = out; q.ldc = p.N; q.M = p.M; q.N = p.N;
            __syncthreads();    // the last stage's fragment reads are over in every wave
            nt_epilogue_lds<TJ>(q, acc, m0 + wm * WTM, n0 + wn * 64, 0, r, g, lane, smem + wave * epi_lds_f32_bytes<TJ>());
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int m = m0 + wm * WTM + j * 16 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * g;
            *reinterpret_cast<float4*>(out + (size_t)m * p.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

template <int BM, int BN, int WTM, int D>
__global__ __launch_bounds__((BM / WTM) * (BN / 64) * 64, 1) void wide_gemm_tn_kernel(WideGemmParams p, int ntM, int ntN, int kps, float* slabs, size_t slab_stride, int by_slice) {
    const int tiles = ntM * ntN;
    if (by_slice) {
        // round 5: the workgroups sharing an XCD (id % 8) get a contiguous range of (K slice, tile) pairs, slice-major: all tiles of a K
        // slice read its rows of A and B through ONE L2. With (tile, slice) grids the tiles of a slice sat on all eight XCDs and every
        // XCD fetched the slice's B rows from HBM for itself: 471 MB read per launch at C4 against ~170 MB of operands
        // (profiles/r05_pmc_c4_bf16.json), the launches at 4 TB/s.
        const int c = xcd_tile(blockIdx.x, (int)gridDim.x);
        tn_tile<BM, BN, WTM, D>(p, c % tiles, c / tiles, ntM, ntN, kps, slabs, slab_stride);
    } else
        tn_tile<BM, BN, WTM, D>(p, xcd_tile(blockIdx.x, tiles), blockIdx.y, ntM, ntN, kps, slabs, slab_stride);
}

// GROUPED launch: up to WIDE_TN_GROUP_MAX independent weight-gradient problems of ONE tile variant as one grid (workgroup ->
// problem by prefix sums of their tile x split counts). The EgoT2-g decoder's backward issued 35 such GEMMs over its 512 target rows
// per step, 14 us each and almost all of it launch latency and pipeline fill (profiles/r04_bench_c5hhi_kernel_stats.csv): together
// they are ~15 GFLOP — one launch that fills the chip. Their operands live in per-(layer, use) buffers until the call returns.
template <int BM, int BN, int WTM, int D>
__global__ __launch_bounds__((BM / WTM) * (BN / 64) * 64, 1) void wide_gemm_tn_grouped_kernel(WideTnGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.d[i + 1].first_block) ++i;
    const WideTnDesc& q = g.d[i];
    const int local = blockIdx.x - q.first_block, tiles = q.ntM * q.ntN;
    WideGemmParams p;
    p.A = q.A; p.B = q.B; p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.zero_page = g.zero_page; p.epi_lds = g.epi_lds;
    const int c = g.by_slice ? xcd_tile(local, tiles * q.splits) : local;     // slice-major per XCD (see wide_gemm_tn_kernel)
    tn_tile<BM, BN, WTM, D>(p, c % tiles, c / tiles, q.ntM, q.ntN, q.kps, q.slabs, q.slab_stride);
}

// C[m][n] (+)= sum_s slab[s][m][n], fixed order
__global__ __launch_bounds__(256) void wide_slab_reduce_kernel(const float* __restrict__ slabs, size_t slab_stride, int splits,
                                                               float* __restrict__ C, int ldc, int M, int N, int accumulate) {
    const size_t n4 = (size_t)M * N / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4*>(slabs)[i];
        for (int k = 1; k < splits; ++k) {
            float4 v = reinterpret_cast<const float4*>(slabs + (size_t)k * slab_stride)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const size_t e = i * 4;
        const int m = (int)(e / N), n = (int)(e % N);
        float4* dst = reinterpret_cast<float4*>(C + (size_t)m * ldc + n);
        if (accumulate) { float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *dst = s;
    }
}

// the same reduction for several weight gradients in one launch (a backward has 4 per layer + the projections; as separate
// launches of ~10 us they were 2.4 % of the C4 step and 7.6 % of the d = 256 EgoT2-g encoder's)
__global__ __launch_bounds__(256) void wide_slab_reduce_batch_kernel(WideReduceBatch b) {
    int di = 0;
    while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].first_block) ++di;
    const WideReduceDesc& q = b.d[di];
    const int local = blockIdx.x - q.first_block, nb = q.blocks;
    const size_t n4 = (size_t)q.M * q.N / 4;
    for (size_t i = (size_t)local * 256 + threadIdx.x; i < n4; i += (size_t)nb * 256) {
        float4 s = reinterpret_cast<const float4*>(q.slabs)[i];
        for (int k = 1; k < q.splits; ++k) {
            float4 v = reinterpret_cast<const float4*>(q.slabs + (size_t)k * q.stride)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const size_t e = i * 4;
        const int m = (int)(e / q.N), n = (int)(e % q.N);
        float4* dst = reinterpret_cast<float4*>(q.C + (size_t)m * q.ldc + n);
        if (q.accumulate) { float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *dst = s;
    }
}
int wide_reduce_flush(WideReduceBatch& b, hipStream_t st) {
    if (!b.n) return 0;
    hipLaunchKernelGGL(wide_slab_reduce_batch_kernel, dim3(b.total_blocks), dim3(256), 0, st, b);
    EGX_LAUNCH_CHECK();
    b.n = 0; b.total_blocks = 0;
    return 0;
}

// tile variant of a TN problem: 2 = 256 x 256, 1 = 256 x 128, 0 = 128 x 128 (EGX_WIDE_TN_TILE = 512 / 256 / 128 forces one)
static int tn_variant(int M, int N) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("EGX_WIDE_TN_TILE"); force = e ? atoi(e) : 0; }
    const bool ok2 = M % 256 == 0 && N % 256 == 0, ok1 = M % 256 == 0;
    if (force == 512 && ok2) return 2;
    if (force == 256 && ok1) return 1;
    if (force == 128) return 0;
    if (ok2 && (long)(M / 256) * (N / 256) >= 9) return 2;      // fewer tiles: the finer variants split the tokens less
    return ok1 ? 1 : 0;
}
// Split-K count: every workgroup runs one tile over K / splits tokens; the launch lasts about
// ceil(tiles * splits / 256) rounds of K / splits tokens each; more splits also mean more fp32 slab traffic
// (2 x 4 bytes per output element per split against ~6 TB/s). Pick the cheapest count under that model.
static int tn_splits(int M, int N, int K, int* kps_out) {
    const int v = tn_variant(M, N);
    const int tiles = (M / (v ? 256 : 128)) * (N / (v == 2 ? 256 : TBN));
    const double step_us = v == 2 ? 1.5 : 0.9;         // one 64-token K step of a resident workgroup, measured
    const double slab_us = 8.0 * M * N / 6.0e6;         // write + read of one slab
    int best = 1;
    double best_t = 1e30;
    for (int sp = 1; sp <= 64; ++sp) {
        int kps = cdiv(cdiv(K, sp), TBK) * TBK;
        if (kps < 256 && sp > 1) break;
        int real = cdiv(K, kps);
        double t = (double)cdiv(tiles * real, 256) * (kps / TBK) * step_us + (real > 1 ? real * slab_us : 0.0);
        if (t < best_t - 1e-9) { best_t = t; best = real; }
    }
    int kps = cdiv(cdiv(K, best), TBK) * TBK;
    if (kps_out) *kps_out = kps;
    return cdiv(K, kps);
}

size_t wide_gemm_tn_scratch(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)tn_splits(M, N, K, nullptr) * M * N * sizeof(float);
}

static int tn_slice_major() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_TN_SLICE_XCD"); v = e ? atoi(e) : 1; }
    return v;
}

template <int BM, int BN, int WTM, int D>
static int launch_tn(const WideGemmParams& p, int splits, int kps, float* slabs, size_t slab_stride, hipStream_t st) {
    constexpr int LDS = D * (BM / 128 + BN / 128) * TBK * 128 * 2;
    constexpr int THREADS = (BM / WTM) * (BN / 64) * 64;
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_tn_kernel<BM, BN, WTM, D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int ntM = p.M / BM, ntN = p.N / BN;
    const int by_slice = tn_slice_major();
    timing_begin(TIMER_WIDE_GEMM, st);
    if (by_slice)
        hipLaunchKernelGGL((wide_gemm_tn_kernel<BM, BN, WTM, D>), dim3(ntM * ntN * splits), dim3(THREADS), LDS, st, p, ntM, ntN, kps, slabs, slab_stride, 1);
    else
        hipLaunchKernelGGL((wide_gemm_tn_kernel<BM, BN, WTM, D>), dim3(ntM * ntN, splits), dim3(THREADS), LDS, st, p, ntM, ntN, kps, slabs, slab_stride, 0);
    timing_end(TIMER_WIDE_GEMM, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WTM, int D>
static int launch_tn_grouped(const WideTnGroup& g, int blocks, hipStream_t st) {
    constexpr int LDS = D * (BM / 128 + BN / 128) * TBK * 128 * 2;
    constexpr int THREADS = (BM / WTM) * (BN / 64) * 64;
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_tn_grouped_kernel<BM, BN, WTM, D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    WideTnGroup gg = g;
    gg.by_slice = tn_slice_major();
    timing_begin(TIMER_WIDE_GEMM, st);
    hipLaunchKernelGGL((wide_gemm_tn_grouped_kernel<BM, BN, WTM, D>), dim3(blocks), dim3(THREADS), LDS, st, gg);
    timing_end(TIMER_WIDE_GEMM, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

// queue one weight-gradient problem for the grouped launch of its tile variant (the slab reduction is queued in `defer` as
// wide_gemm_tn does); `scratch` = this problem's own slab region. wide_tn_queue_flush() launches what is queued: ONE grid per variant.
int wide_tn_queue_add(WideTnQueue& Q, const WideGemmParams& p, void* scratch, hipStream_t st, WideReduceBatch* defer) {
    EGX_CHECK(p.A && p.B && p.Cf && scratch && p.zero_page && defer, "wide_tn_queue_add: null operand");
    EGX_CHECK(p.M % TBM == 0 && p.N % TBN == 0 && p.K > 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0,
              "wide_tn_queue_add: %dx%dx%d needs M, N multiples of 128 and 16-byte aligned rows", p.M, p.N, p.K);
    int kps = 0;
    int splits = tn_splits(p.M, p.N, p.K, &kps);
    const int v = tn_variant(p.M, p.N);
    WideTnGroup& g = Q.g[v];
    if (g.n == WIDE_TN_GROUP_MAX && wide_tn_queue_flush(Q, st)) return 1;
    if (defer->n == WIDE_REDUCE_MAX) {       // the reduction reads slabs the queued GEMMs have not written yet: flush them first
        if (wide_tn_queue_flush(Q, st) || wide_reduce_flush(*defer, st)) return 1;
    }
    g.zero_page = p.zero_page; g.epi_lds = epi_lds_mode();
    WideTnDesc& q = g.d[g.n++];
    q.A = p.A; q.B = p.B; q.M = p.M; q.N = p.N; q.K = p.K; q.lda = p.lda; q.ldb = p.ldb;
    q.ntM = p.M / (v ? 256 : 128); q.ntN = p.N / (v == 2 ? 256 : TBN); q.kps = kps; q.splits = splits;
    q.slabs = (float*)scratch; q.slab_stride = (size_t)p.M * p.N;
    q.first_block = Q.blocks[v];
    Q.blocks[v] += q.ntM * q.ntN * splits;
    // the slab reduction (even a single slab goes through it: C (+)= slab), batched by the caller
    const size_t n4 = q.slab_stride / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 512) blocks = 512;
    WideReduceDesc& r = defer->d[defer->n++];
    r.slabs = q.slabs; r.stride = q.slab_stride; r.splits = splits; r.C = p.Cf; r.ldc = p.ldc; r.M = p.M; r.N = p.N;
    r.accumulate = p.accumulate; r.first_block = defer->total_blocks; r.blocks = blocks;
    defer->total_blocks += blocks;
    return 0;
}
int wide_tn_queue_flush(WideTnQueue& Q, hipStream_t st) {
    for (int v = 0; v < 3; ++v) {
        if (!Q.g[v].n) continue;
        int rc = v == 2 ? launch_tn_grouped<256, 256, 128, 2>(Q.g[v], Q.blocks[v], st)
               : v == 1 ? launch_tn_grouped<256, 128, 64, 3>(Q.g[v], Q.blocks[v], st) : launch_tn_grouped<128, 128, 64, 4>(Q.g[v], Q.blocks[v], st);
        if (rc) return rc;
        Q.g[v].n = 0; Q.blocks[v] = 0;
    }
    return 0;
}

int wide_gemm_tn(const WideGemmParams& p_in, void* scratch, hipStream_t st, WideReduceBatch* defer) {
    WideGemmParams p = p_in;
    p.epi_lds = epi_lds_mode();
    EGX_CHECK(p.A && p.B && p.Cf && scratch && p.zero_page, "wide_gemm_tn: null operand");
    EGX_CHECK(p.M % TBM == 0 && p.N % TBN == 0 && p.K > 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0,
              "wide_gemm_tn: %dx%dx%d needs M, N multiples of 128 and 16-byte aligned rows", p.M, p.N, p.K);
    int kps = 0;
    int splits = tn_splits(p.M, p.N, p.K, &kps);
    if (p.tn_max_splits > 0 && splits > p.tn_max_splits) {
        kps = cdiv(cdiv(p.K, p.tn_max_splits), TBK) * TBK;
        splits = cdiv(p.K, kps);
    }
    const size_t slab_stride = (size_t)p.M * p.N;
    const int v = tn_variant(p.M, p.N);
    if (splits == 1 && !p.accumulate && p.ldc == p.N) {      // the one slab IS the output
        if (v == 2) return launch_tn<256, 256, 128, 2>(p, 1, kps, p.Cf, slab_stride, st);
        if (v == 1) return launch_tn<256, 128, 64, 3>(p, 1, kps, p.Cf, slab_stride, st);
        return launch_tn<128, 128, 64, 4>(p, 1, kps, p.Cf, slab_stride, st);
    }
    if (v == 2) { if (launch_tn<256, 256, 128, 2>(p, splits, kps, (float*)scratch, slab_stride, st)) return 1; }
    else if (v == 1) { if (launch_tn<256, 128, 64, 3>(p, splits, kps, (float*)scratch, slab_stride, st)) return 1; }
    else if (launch_tn<128, 128, 64, 4>(p, splits, kps, (float*)scratch, slab_stride, st)) return 1;
    const size_t n4 = slab_stride / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (defer) {        // `scratch` is this problem's own slab region: the caller reduces every queued problem in one launch
        if (defer->n == WIDE_REDUCE_MAX && wide_reduce_flush(*defer, st)) return 1;
        if (blocks > 512) blocks = 512;
        WideReduceDesc& q = defer->d[defer->n++];
        q.slabs = (const float*)scratch; q.stride = slab_stride; q.splits = splits; q.C = p.Cf; q.ldc = p.ldc; q.M = p.M; q.N = p.N;
        q.accumulate = p.accumulate; q.first_block = defer->total_blocks; q.blocks = blocks;
        defer->total_blocks += blocks;
        return 0;
    }
    hipLaunchKernelGGL(wide_slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)scratch, slab_stride, splits, p.Cf,
                       p.ldc, p.M, p.N, p.accumulate);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- fp32 -> bf16 cast (+ transpose) -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_cast_kernel(const float* __restrict__ src, int R, int C, int ld,
                                                        bf16_t* __restrict__ dst, bf16_t* __restrict__ dst_t) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 4 columns each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rr = r0 + ty + 16 * i, cc = c0 + 4 * tx;
        float4 v = make_float4(0, 0, 0, 0);
        if (rr < R && cc < C) v = *reinterpret_cast<const float4*>(src + (size_t)rr * ld + cc);     // C % 4 == 0
        bf16_t h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
        if (dst && rr < R && cc < C)
            *reinterpret_cast<uint2*>(dst + (size_t)rr * C + cc) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
        tile[ty + 16 * i][4 * tx + 0] = h0; tile[ty + 16 * i][4 * tx + 1] = h1;
        tile[ty + 16 * i][4 * tx + 2] = h2; tile[ty + 16 * i][4 * tx + 3] = h3;
    }
    if (!dst_t) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int cc = c0 + ty + 16 * i, rr = r0 + 4 * tx;           // output row = source column
        if (cc < C && rr < R) {                                 // R % 4 == 0
            bf16_t a = tile[4 * tx + 0][ty + 16 * i], b = tile[4 * tx + 1][ty + 16 * i];
            bf16_t c = tile[4 * tx + 2][ty + 16 * i], d = tile[4 * tx + 3][ty + 16 * i];
            *reinterpret_cast<uint2*>(dst_t + (size_t)cc * R + rr) = make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)d << 16));
        }
    }
}

// several casts in one launch (all weights of a forward: 4 per layer + the projections were 19 launches of ~5 us at C4)
__global__ __launch_bounds__(256) void wide_cast_batch_kernel(WideCastBatch b) {
    __shared__ bf16_t tile[64][66];
    int di = 0;
    while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].first_block) ++di;
    const WideCastDesc& q = b.d[di];
    const int local = blockIdx.x - q.first_block, nbx = (q.C + 63) / 64;
    const int r0 = (local / nbx) * 64, c0 = (local % nbx) * 64;
    const int R = q.R, C = q.C;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rr = r0 + ty + 16 * i, cc = c0 + 4 * tx;
        float4 v = make_float4(0, 0, 0, 0);
        if (rr < R && cc < C) v = *reinterpret_cast<const float4*>(q.src + (size_t)rr * q.ld + cc);
        const uint32_t lo = pack_bf16x2(v.x, v.y), hi = pack_bf16x2(v.z, v.w);
        if (q.dst && rr < R && cc < C) *reinterpret_cast<uint2*>(q.dst + (size_t)rr * C + cc) = make_uint2(lo, hi);
        tile[ty + 16 * i][4 * tx + 0] = (bf16_t)(lo & 0xffffu); tile[ty + 16 * i][4 * tx + 1] = (bf16_t)(lo >> 16);
        tile[ty + 16 * i][4 * tx + 2] = (bf16_t)(hi & 0xffffu); tile[ty + 16 * i][4 * tx + 3] = (bf16_t)(hi >> 16);
    }
    if (!q.dst_t) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int cc = c0 + ty + 16 * i, rr = r0 + 4 * tx;           // output row = source column
        if (cc < C && rr < R) {
            bf16_t a = tile[4 * tx + 0][ty + 16 * i], bb = tile[4 * tx + 1][ty + 16 * i];
            bf16_t c = tile[4 * tx + 2][ty + 16 * i], d = tile[4 * tx + 3][ty + 16 * i];
            *reinterpret_cast<uint2*>(q.dst_t + (size_t)cc * R + rr) = make_uint2((uint32_t)a | ((uint32_t)bb << 16), (uint32_t)c | ((uint32_t)d << 16));
        }
    }
}
int wide_cast_add(WideCastBatch& b, const float* src, int R, int C, int ld, bf16_t* dst, bf16_t* dst_t, hipStream_t st) {
    EGX_CHECK(src && (dst || dst_t), "wide_cast: null pointer");
    EGX_CHECK(C % 4 == 0 && ld % 4 == 0 && (!dst_t || R % 4 == 0), "wide_cast: %dx%d needs multiples of 4", R, C);
    if (R <= 0 || C <= 0) return 0;
    if (b.n == WIDE_CAST_MAX && wide_cast_flush(b, st)) return 1;
    WideCastDesc& q = b.d[b.n++];
    q.src = src; q.R = R; q.C = C; q.ld = ld; q.dst = dst; q.dst_t = dst_t; q.first_block = b.blocks;
    b.blocks += cdiv(C, 64) * cdiv(R, 64);
    return 0;
}
int wide_cast_flush(WideCastBatch& b, hipStream_t st) {
    if (!b.n) return 0;
    hipLaunchKernelGGL(wide_cast_batch_kernel, dim3(b.blocks), dim3(256), 0, st, b);
    EGX_LAUNCH_CHECK();
    b.n = 0; b.blocks = 0;
    return 0;
}

int wide_cast(const float* src, int R, int C, int ld, bf16_t* dst, bf16_t* dst_t, hipStream_t st) {
    EGX_CHECK(src && (dst || dst_t), "wide_cast: null pointer");
    EGX_CHECK(C % 4 == 0 && ld % 4 == 0 && (!dst_t || R % 4 == 0), "wide_cast: %dx%d needs multiples of 4", R, C);
    if (R <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(wide_cast_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, st, src, R, C, ld, dst, dst_t);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- feature hand-off: temporal mean over `pool` frames fused with the cast to the projection operand -----------------
template <bool SRC_BF16>
__global__ __launch_bounds__(256) void wide_pool_cast_kernel(const void* __restrict__ src, int R, int pool, int C, bf16_t* __restrict__ dst) {
    const int c8 = C / 8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * c8) return;
    const int r = (int)(i / c8), c = (int)(i % c8) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < pool; ++j) {
        const size_t row = (size_t)r * pool + j;
        if constexpr (SRC_BF16) {
            uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(src) + row * C + c);
            acc[0] += bf2f((bf16_t)(v.x & 0xffffu)); acc[1] += bf2f((bf16_t)(v.x >> 16));
            acc[2] += bf2f((bf16_t)(v.y & 0xffffu)); acc[3] += bf2f((bf16_t)(v.y >> 16));
            acc[4] += bf2f((bf16_t)(v.z & 0xffffu)); acc[5] += bf2f((bf16_t)(v.z >> 16));
            acc[6] += bf2f((bf16_t)(v.w & 0xffffu)); acc[7] += bf2f((bf16_t)(v.w >> 16));
        } else {
            const float* p = reinterpret_cast<const float*>(src) + row * C + c;
            float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        }
    }
    const float s = 1.f / (float)pool;
    *reinterpret_cast<uint4*>(dst + (size_t)r * C + c) =
        make_uint4(pack2(acc[0] * s, acc[1] * s), pack2(acc[2] * s, acc[3] * s), pack2(acc[4] * s, acc[5] * s), pack2(acc[6] * s, acc[7] * s));
}

int wide_pool_cast(const void* src, int src_bf16, int R, int pool, int C, bf16_t* dst, hipStream_t st) {
    EGX_CHECK(src && dst && C % 8 == 0 && pool >= 1, "wide_pool_cast: bad arguments");
    if (R <= 0) return 0;
    const size_t n = (size_t)R * (C / 8);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (src_bf16) hipLaunchKernelGGL(wide_pool_cast_kernel<true>, dim3(blocks), dim3(256), 0, st, src, R, pool, C, dst);
    else hipLaunchKernelGGL(wide_pool_cast_kernel<false>, dim3(blocks), dim3(256), 0, st, src, R, pool, C, dst);
    EGX_LAUNCH_CHECK();
    return 0;
}

}  // namespace egx
