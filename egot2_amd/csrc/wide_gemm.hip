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

namespace egx {

namespace {

constexpr int TBM = 128, TBN = 128, TBK = 64;
constexpr int STAGE_BYTES = (TBM + TBN) * TBK * 2;     // 32 KB

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
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

// blocks sharing an XCD (id % 8) get a contiguous range of tiles (bijective for any total)
__device__ __forceinline__ int xcd_tile(int id, int total) {
    int xcd = id & 7, j = id >> 3, q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

}  // namespace

// ---- NT ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wide_gemm_nt_kernel(WideGemmParams p, int ntM, int ntN) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_tile(blockIdx.x, ntM * ntN);
    const int m0 = (t / ntN) * TBM, n0 = (t % ntN) * TBN;

    // staging: wave w copies rows [32w, 32w + 32) of both operand tiles, 8 rows (1 KiB) per instruction
    const int srow = lane >> 3;                       // row within the 8-row group == (row & 7)
    const int lch = (lane & 7) ^ srow;                // logical 16-byte chunk this lane fetches
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int row = (wave * 4 + j) * 8 + srow;
        int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
        srcA[j] = p.A + (size_t)gm * p.lda + lch * 8;
        srcB[j] = p.B + (size_t)gn * p.ldb + lch * 8;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * STAGE_BYTES + wave * 4096;
        unsigned char* sb = sa + TBM * TBK * 2;
        const int k0 = kt * TBK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(srcA[j] + k0, sa + j * 1024);
            glds16(srcB[j] + k0, sb + j * 1024);
        }
    };

    const int r = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    // fragment byte offsets inside a stage: rows of X (m) / W (n); chunk (s * 4 + g) ^ (row & 7), row & 7 == r & 7
    const int offX = (wm * 64 + r) * 128, offW = TBM * TBK * 2 + (wn * 64 + r) * 128;
    const int c0 = ((0 * 4 + g) ^ (r & 7)) * 16, c1 = ((1 * 4 + g) ^ (r & 7)) * 16;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int nk = p.K / TBK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cs = s ? c1 : c0;
            bf16x8 fw[4], fx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = lds_read128(st + offW + i * 2048 + cs);
#pragma unroll
            for (int j = 0; j < 4; ++j) fx[j] = lds_read128(st + offX + j * 2048 + cs);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: lane (r, g) of tile (i, j) holds C[m0 + wm*64 + j*16 + r][n0 + wn*64 + i*16 + 4g .. +3]
    float cs_part[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) cs_part[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + r;
        const bool mv = m < p.M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * g;
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
                drop_scale4(p.drop_key, (uint32_t)m, (uint32_t)n, p.drop_thresh, p.drop_inv, ds);
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
                    cs_part[i][0] += bf2f((bf16_t)(o.x & 0xffffu)); cs_part[i][1] += bf2f((bf16_t)(o.x >> 16));
                    cs_part[i][2] += bf2f((bf16_t)(o.y & 0xffffu)); cs_part[i][3] += bf2f((bf16_t)(o.y >> 16));
                }
            } else if (p.colsum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) cs_part[i][e] += v[e];
            }
        }
    }
    if (p.colsum) {      // one partial row per (row tile, wave row half): [ntM * 2][N]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = cs_part[i][e];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                cs_part[i][e] = s;
            }
        if (r == 0) {
            const int prow = (m0 / TBM) * 2 + wm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + i * 16 + 4 * g;
                if (n < p.N) *reinterpret_cast<float4*>(p.colsum + (size_t)prow * p.N + n) =
                    make_float4(cs_part[i][0], cs_part[i][1], cs_part[i][2], cs_part[i][3]);
            }
        }
    }
}

int wide_gemm_nt(const WideGemmParams& p, hipStream_t st) {
    EGX_CHECK(p.A && p.B && (p.Cf || p.Cb), "wide_gemm_nt: null operand");
    EGX_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "wide_gemm_nt: empty problem %dx%dx%d", p.M, p.N, p.K);
    EGX_CHECK(p.K % TBK == 0 && p.N % 4 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0,
              "wide_gemm_nt: %dx%dx%d needs K %% 64 == 0, N %% 4 == 0, 16-byte aligned rows", p.M, p.N, p.K);
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        attr = true;
    }
    const int ntM = cdiv(p.M, TBM), ntN = cdiv(p.N, TBN);
    timing_begin(TIMER_WIDE_GEMM, st);
    hipLaunchKernelGGL(wide_gemm_nt_kernel, dim3(ntM * ntN), dim3(256), 2 * STAGE_BYTES, st, p, ntM, ntN);
    timing_end(TIMER_WIDE_GEMM, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- TN ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wide_gemm_tn_kernel(WideGemmParams p, int ntM, int ntN, int kps, float* slabs, size_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_tile(blockIdx.x, ntM * ntN);
    const int m0 = (t / ntN) * TBM, n0 = (t % ntN) * TBN;
    const int split = blockIdx.y;
    const int kbeg = split * kps, kend = min(p.K, kbeg + kps);

    // staging: 4 token rows (256 B each) per instruction; wave w copies rows [16w, 16w + 16) of both tiles
    const int slot = lane & 15, half = slot & 1, pch = slot >> 1;
    int rowS[4], colS[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int row = (wave * 4 + j) * 4 + (lane >> 4);
        int sw = (row & 3) | (((row >> 3) & 1) << 2);
        rowS[j] = row;
        colS[j] = ((pch ^ sw) * 16 + half * 8);
    }
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(p.zero_page) + slot * 16;
    auto stage = [&](int k0, int buf) {
        unsigned char* sa = smem + buf * STAGE_BYTES + wave * 4096;
        unsigned char* sb = sa + TBK * TBM * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tok = k0 + rowS[j];
            const bool ok = tok < kend;
            const void* a = ok ? (const void*)(p.A + (size_t)tok * p.lda + m0 + colS[j]) : (const void*)zero;
            const void* b = ok ? (const void*)(p.B + (size_t)tok * p.ldb + n0 + colS[j]) : (const void*)zero;
            glds16(a, sa + j * 1024);
            glds16(b, sb + j * 1024);
        }
    };

    const int r = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    // transposed reads: lane supplies row 8g + (r >> 2) (+4) of the 32-token slice, 4 columns 4 * (r & 3) of its 16-wide tile
    const int swz = (r >> 2) | ((g & 1) << 2);
    const int rowoff = (8 * g + (r >> 2)) * 256 + 8 * (r & 3);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int nk = (kend - kbeg + TBK - 1) / TBK;
    if (nk > 0) stage(kbeg, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kbeg + (kt + 1) * TBK, (kt + 1) & 1);
        const unsigned char* sa = smem + (kt & 1) * STAGE_BYTES;      // dY tile: [64 tokens][128 m]
        const unsigned char* sb = sa + TBK * TBM * 2;                  // X  tile: [64 tokens][128 n]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fx[4], fy[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // MFMA A operand: X^T rows n = wn*64 + i*16 + r
                const unsigned char* q0 = sb + s * 32 * 256 + rowoff + (((wn * 4 + i) ^ swz) * 32);
                fx[i] = lds_read_tr2(q0, q0 + 4 * 256);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {       // MFMA B operand: dY columns m = wm*64 + j*16 + r
                const unsigned char* q0 = sa + s * 32 * 256 + rowoff + (((wm * 4 + j) ^ swz) * 32);
                fy[j] = lds_read_tr2(q0, q0 + 4 * 256);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fy[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = slabs + (size_t)split * slab_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * g;
            *reinterpret_cast<float4*>(out + (size_t)m * p.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
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

static int tn_splits(int M, int N, int K, int* kps_out) {
    const int tiles = (M / TBM) * (N / TBN);
    int splits = cdiv(768, tiles);                     // ~3 workgroups per CU slot pair
    int max_splits = cdiv(K, 256);                     // at least 4 K steps per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int kps = cdiv(cdiv(K, splits), TBK) * TBK;
    splits = cdiv(K, kps);
    if (kps_out) *kps_out = kps;
    return splits;
}

size_t wide_gemm_tn_scratch(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)tn_splits(M, N, K, nullptr) * M * N * sizeof(float);
}

int wide_gemm_tn(const WideGemmParams& p, void* scratch, hipStream_t st) {
    EGX_CHECK(p.A && p.B && p.Cf && scratch && p.zero_page, "wide_gemm_tn: null operand");
    EGX_CHECK(p.M % TBM == 0 && p.N % TBN == 0 && p.K > 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0,
              "wide_gemm_tn: %dx%dx%d needs M, N multiples of 128 and 16-byte aligned rows", p.M, p.N, p.K);
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        attr = true;
    }
    int kps = 0;
    const int splits = tn_splits(p.M, p.N, p.K, &kps);
    const int ntM = p.M / TBM, ntN = p.N / TBN;
    const size_t slab_stride = (size_t)p.M * p.N;
    timing_begin(TIMER_WIDE_GEMM, st);
    hipLaunchKernelGGL(wide_gemm_tn_kernel, dim3(ntM * ntN, splits), dim3(256), 2 * STAGE_BYTES, st, p, ntM, ntN, kps,
                       (float*)scratch, slab_stride);
    timing_end(TIMER_WIDE_GEMM, st);
    const size_t n4 = slab_stride / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
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

int wide_cast(const float* src, int R, int C, int ld, bf16_t* dst, bf16_t* dst_t, hipStream_t st) {
    EGX_CHECK(src && (dst || dst_t), "wide_cast: null pointer");
    EGX_CHECK(C % 4 == 0 && ld % 4 == 0 && (!dst_t || R % 4 == 0), "wide_cast: %dx%d needs multiples of 4", R, C);
    if (R <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(wide_cast_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, st, src, R, C, ld, dst, dst_t);
    EGX_LAUNCH_CHECK();
    return 0;
}

}  // namespace egx
