// Generic MFMA GEMM for the translator's linear layers (forward NT, input-gradient NN, weight-gradient TN).
//
// Replaces the aten::addmm / aten::mm calls issued by nn.Linear and F.multi_head_attention_forward's
// in/out projections (reference call sites: HHI/models/ttm/model_taskspecific.py:238-242).
//
// gfx950 mapping: 256-thread workgroups (4 waves, 2x2), each wave owns a (BM/2)x(BN/2) block of 16x16
// MFMA tiles. fp32 mode uses v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue); bf16 mode converts
// operands to bf16 while staging into LDS and uses v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
// LDS always holds k-contiguous rows ([row][BK+pad]) so every fragment is one ds_read_b128; operands
// that are m-contiguous in HBM (the "MC" side of NN/TN) are transposed in registers as 4x4 blocks on the
// way in (coalesced dwordx4 global loads, b128/b64 LDS writes, no shuffles). Global loads for tile t+1
// are issued before the MFMAs of tile t (register prefetch).
#include "common.h"
#include "kernels.h"

namespace egx {

constexpr int BK = 32;
constexpr int LDF = BK + 4;   // fp32 LDS row stride (floats)
constexpr int LDH = BK + 8;   // bf16 LDS row stride (halfs)

template <bool BF16> struct LdsElem { typedef float type; static constexpr int LD = LDF; };
template <> struct LdsElem<true> { typedef unsigned short type; static constexpr int LD = LDH; };

// ---- tile loaders -----------------------------------------------------------------------------
// KC operand: src[row][k], k contiguous. Thread owns float4 slots f = tid + i*256: row = f>>3, kq = f&7.
template <int R, bool VEC>
__device__ __forceinline__ void load_kc(const float* __restrict__ src, int ld, int row0, int nrows,
                                        int k0, int kend, float4 (&v)[R / 32], int tid) {
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        int f = tid + i * 256;
        int row = row0 + (f >> 3);
        int k = k0 + ((f & 7) << 2);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nrows) {
            const float* p = src + (size_t)row * ld + k;
            if (VEC) {
                if (k < kend) x = *reinterpret_cast<const float4*>(p);   // kend % 4 == 0 on the VEC path
            } else {
                if (k + 0 < kend) x.x = p[0];
                if (k + 1 < kend) x.y = p[1];
                if (k + 2 < kend) x.z = p[2];
                if (k + 3 < kend) x.w = p[3];
            }
        }
        v[i] = x;
    }
}

template <int R, bool BF16>
__device__ __forceinline__ void store_kc(typename LdsElem<BF16>::type* lds, const float4 (&v)[R / 32], int tid) {
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        int f = tid + i * 256;
        int row = f >> 3;
        int k = (f & 7) << 2;
        if constexpr (BF16) {
            uint2 pk;
            pk.x = pack_bf16x2(v[i].x, v[i].y);
            pk.y = pack_bf16x2(v[i].z, v[i].w);
            *reinterpret_cast<uint2*>(lds + row * LDH + k) = pk;
        } else {
            *reinterpret_cast<float4*>(lds + row * LDF + k) = v[i];
        }
    }
}

// MC operand: src[k][m], m contiguous. Thread owns a 4(m) x KT(k) block: mq = tid % (R/4), kq = tid / (R/4).
// KT = R/32 consecutive k rows (R=128: 4, R=64: 2).
template <int R, bool VEC>
__device__ __forceinline__ void load_mc(const float* __restrict__ src, int ld, int m0, int mend,
                                        int k0, int kend, float4 (&v)[R / 32], int tid) {
    constexpr int KT = R / 32;
    int mq = tid % (R / 4);
    int kq = tid / (R / 4);
    int m = m0 + (mq << 2);
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        int k = k0 + kq * KT + i;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) {
            const float* p = src + (size_t)k * ld + m;
            if (VEC) {
                if (m < mend) x = *reinterpret_cast<const float4*>(p);   // mend % 4 == 0 on the VEC path
            } else {
                if (m + 0 < mend) x.x = p[0];
                if (m + 1 < mend) x.y = p[1];
                if (m + 2 < mend) x.z = p[2];
                if (m + 3 < mend) x.w = p[3];
            }
        }
        v[i] = x;
    }
}

template <int R, bool BF16>
__device__ __forceinline__ void store_mc(typename LdsElem<BF16>::type* lds, const float4 (&v)[R / 32], int tid) {
    constexpr int KT = R / 32;
    int mq = tid % (R / 4);
    int kq = tid / (R / 4);
    int kb = kq * KT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int row = (mq << 2) + j;
        float e[KT];
#pragma unroll
        for (int i = 0; i < KT; ++i) e[i] = (j == 0) ? v[i].x : (j == 1) ? v[i].y : (j == 2) ? v[i].z : v[i].w;
        if constexpr (BF16) {
            if constexpr (KT == 4) {
                uint2 pk;
                pk.x = pack_bf16x2(e[0], e[1]);
                pk.y = pack_bf16x2(e[2], e[3]);
                *reinterpret_cast<uint2*>(lds + row * LDH + kb) = pk;
            } else {
                uint32_t pk = pack_bf16x2(e[0], e[1]);
                *reinterpret_cast<uint32_t*>(lds + row * LDH + kb) = pk;
            }
        } else {
            if constexpr (KT == 4) {
                *reinterpret_cast<float4*>(lds + row * LDF + kb) = make_float4(e[0], e[1], e[2], e[3]);
            } else {
                *reinterpret_cast<float2*>(lds + row * LDF + kb) = make_float2(e[0], e[1]);
            }
        }
    }
}

// ---- kernel -----------------------------------------------------------------------------------
// LAYOUT 0: NT  A[M,K] (KC)  B[N,K] (KC)
// LAYOUT 1: NN  A[M,K] (KC)  B[K,N] (MC)
// LAYOUT 2: TN  A[K,M] (MC)  B[K,N] (MC)
template <int BM, int BN, int LAYOUT, bool BF16, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    typedef typename LdsElem<BF16>::type lds_t;
    constexpr int LD = LdsElem<BF16>::LD;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr bool A_MC = (LAYOUT == 2);
    constexpr bool B_MC = (LAYOUT != 0);

    __shared__ __attribute__((aligned(16))) lds_t As[BM * LD];
    __shared__ __attribute__((aligned(16))) lds_t Bs[BN * LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;

    // XCD-aware tile order is not needed here: operands of one launch fit the 256 MiB Infinity Cache.
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    float* __restrict__ C = p.C + (size_t)blockIdx.z * p.slab_stride;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[BM / 32], rb[BN / 32];
    auto gload = [&](int k0) {
        if constexpr (A_MC) load_mc<BM, VEC>(p.A, p.lda, m0, p.M, k0, kend, ra, tid);
        else load_kc<BM, VEC>(p.A, p.lda, m0, p.M, k0, kend, ra, tid);
        if constexpr (B_MC) load_mc<BN, VEC>(p.B, p.ldb, n0, p.N, k0, kend, rb, tid);
        else load_kc<BN, VEC>(p.B, p.ldb, n0, p.N, k0, kend, rb, tid);
    };

    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (nkt > 0) gload(kbeg);
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (A_MC) store_mc<BM, BF16>(As, ra, tid); else store_kc<BM, BF16>(As, ra, tid);
        if constexpr (B_MC) store_mc<BN, BF16>(Bs, rb, tid); else store_kc<BN, BF16>(Bs, rb, tid);
        __syncthreads();
        if (kt + 1 < nkt) gload(kbeg + (kt + 1) * BK);

        const lds_t* ap = As + (wm * WM + r) * LD;
        const lds_t* bp = Bs + (wn * WN + r) * LD;
        if constexpr (BF16) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ap + i * 16 * LD + 8 * q);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bp + j * 16 * LD + 8 * q);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                // lane group q holds k = 16kk + 4q + j for MFMA step j (same permutation on A and B)
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(ap + i * 16 * LD + kk * 16 + 4 * q);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(bp + j * 16 * LD + kk * 16 + 4 * q);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            float av = (s == 0) ? a[i].x : (s == 1) ? a[i].y : (s == 2) ? a[i].z : a[i].w;
                            float bv = (s == 0) ? b[j].x : (s == 1) ? b[j].y : (s == 2) ? b[j].z : b[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int col = n0 + wn * WN + j * 16 + r;
            if (col >= p.N) continue;
            float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int row = m0 + wm * WM + i * 16 + 4 * q + e;
                if (row >= p.M) continue;
                float v = acc[i][j][e] + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.drop_thresh) v *= drop_scale(p.drop_key, (uint32_t)row, (uint32_t)col, p.drop_thresh, p.drop_inv_keep);
                if (p.mask) v = (p.mask[(size_t)row * p.ldm + col] > 0.f) ? v * p.mask_scale : 0.f;
                if (p.residual) v += p.residual[(size_t)row * p.ldr + col];
                if (p.atomic) atomicAdd(C + (size_t)row * p.ldc + col, v);
                else C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

// out[i] (+)= sum_z slabs[z][i]
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                    float* __restrict__ out, size_t n, int accumulate) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i4 = i * 4;
    const bool al = ((((uintptr_t)out) | ((uintptr_t)slabs)) & 15) == 0 && (slab_stride % 4 == 0);
    if (al && i4 + 3 < n) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < nslab; ++z) {
            float4 v = *reinterpret_cast<const float4*>(slabs + z * slab_stride + i4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4* o = reinterpret_cast<float4*>(out + i4);
        if (accumulate) { float4 c = *o; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        *o = s;
    } else {
        for (size_t e = i4; e < n && e < i4 + 4; ++e) {
            float s = 0.f;
            for (int z = 0; z < nslab; ++z) s += slabs[z * slab_stride + e];
            out[e] = accumulate ? out[e] + s : s;
        }
    }
}

template <int BM, int BN, int LAYOUT, bool BF16>
static int launch_cfg(const GemmParams& p, int splits, bool vec, hipStream_t st) {
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), splits);
    if (vec) hipLaunchKernelGGL((gemm_kernel<BM, BN, LAYOUT, BF16, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, LAYOUT, BF16, false>), grid, dim3(256), 0, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

template <int LAYOUT, bool BF16>
static int launch_layout(const GemmParams& p, int splits, bool vec, bool big, hipStream_t st) {
    if (big) return launch_cfg<128, 128, LAYOUT, BF16>(p, splits, vec, st);
    return launch_cfg<64, 64, LAYOUT, BF16>(p, splits, vec, st);
}

static int pick_splits(int tiles, int K) {
    // aim for ~2 waves of workgroups over 256 CUs, K chunks of at least 8 k-tiles
    int want = cdiv(512, tiles);
    int maxs = max(1, K / (BK * 8));
    int s = min(min(want, maxs), 64);
    return max(s, 1);
}

static void choose_tiles(int layout, int M, int N, int K, bool& big, int& splits) {
    long tiles_big = (long)cdiv(M, 128) * cdiv(N, 128);
    big = tiles_big >= 200;
    // weight gradients split K (tokens) across blockIdx.z: the grid fills the chip through the splits, so prefer the
    // 128 x 128 tile (4x the MFMA work per staged byte and per barrier of the 64 x 64 one) whenever splits can supply
    // >= 256 workgroups
    if (layout == 2 && M >= 128 && N >= 128 && tiles_big * (long)min(64, max(1, K / (BK * 8))) >= 256) big = true;
    int bm = big ? 128 : 64;
    int tiles = cdiv(M, bm) * cdiv(N, bm);
    splits = (layout == 2) ? pick_splits(tiles, K) : 1;
    // skinny input gradients with a long reduction (dx of a wide classifier head: 256 x 768 over K = 11 860 classes is 48
    // workgroups of 186 K-steps): split K there too when the caller provides slab scratch (see gemm())
    if (layout == 1 && tiles < 128 && K >= 2048) splits = pick_splits(tiles, K);
}

size_t gemm_scratch_bytes(int layout, int M, int N, int K) {
    if (layout == 0) return 0;
    bool big; int splits;
    choose_tiles(layout, M, N, K, big, splits);
    return (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
}

// C = op(A) op(B) with fused epilogue. For LAYOUT TN (weight gradients) the K range (tokens) is split
// across blockIdx.z into fp32 slabs in `scratch` and summed by reduce_slabs_kernel (deterministic order);
// `accumulate` adds the result into C.
int gemm(int layout, GemmParams p, int compute, int accumulate, void* scratch, size_t scratch_bytes, hipStream_t st) {
    EGX_CHECK(layout >= 0 && layout <= 2, "gemm: bad layout %d", layout);
    EGX_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    bool vec = true;
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    if (!al16(p.A) || !al16(p.B)) vec = false;
    if (p.lda % 4 || p.ldb % 4) vec = false;
    if (layout == 0) { if (p.K % 4) vec = false; }
    else if (layout == 1) { if (p.K % 4 || p.N % 4) vec = false; }
    else { if (p.M % 4 || p.N % 4) vec = false; }

    bool big; int splits;
    choose_tiles(layout, p.M, p.N, p.K, big, splits);
    if (layout == 1 && splits > 1 &&
        (!scratch || scratch_bytes < (size_t)splits * p.M * p.N * sizeof(float) || p.bias || p.residual || p.mask || p.relu || p.drop_thresh || p.ldc != p.N))
        splits = 1;         // split-K needs slab scratch and a plain epilogue: otherwise the single-pass kernel
    float* final_C = p.C;
    int final_ldc = p.ldc;
    bool use_slabs = false;
    // small accumulate-into outputs (<= 256k floats): split-K blocks add straight into C, no slab round trip
    if (!det_on() && !getenv("EGX_NO_ATOMIC_DW") && accumulate && layout == 2 && (size_t)p.M * p.N <= 262144 && !p.bias && !p.residual && !p.mask && !p.relu && !p.drop_thresh) {
        p.atomic = 1;
    } else if (splits > 1 || accumulate) {
        size_t need = (size_t)splits * p.M * p.N * sizeof(float);
        EGX_CHECK(scratch && scratch_bytes >= need, "gemm: scratch too small (%zu < %zu)", scratch_bytes, need);
        EGX_CHECK(!p.bias && !p.residual && !p.mask && !p.relu && !p.drop_thresh, "gemm: epilogue unsupported with split-K");
        use_slabs = true;
        p.C = (float*)scratch;
        p.ldc = p.N;
        p.slab_stride = (size_t)p.M * p.N;
    }
    int kt = cdiv(p.K, BK);
    p.k_chunk = cdiv(kt, splits) * BK;
    splits = cdiv(p.K, p.k_chunk);
    if (!use_slabs) p.slab_stride = 0;

    int rc;
    if (compute == 1) {
        if (layout == 0) rc = launch_layout<0, true>(p, splits, vec, big, st);
        else if (layout == 1) rc = launch_layout<1, true>(p, splits, vec, big, st);
        else rc = launch_layout<2, true>(p, splits, vec, big, st);
    } else {
        if (layout == 0) rc = launch_layout<0, false>(p, splits, vec, big, st);
        else if (layout == 1) rc = launch_layout<1, false>(p, splits, vec, big, st);
        else rc = launch_layout<2, false>(p, splits, vec, big, st);
    }
    if (rc) return rc;
    if (use_slabs) {
        EGX_CHECK(final_ldc == p.N, "gemm: split-K output must be dense (ldc %d != N %d)", final_ldc, p.N);
        size_t n = (size_t)p.M * p.N;
        int blocks = (int)((n / 4 + 255) / 256) + 1;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)scratch, splits,
                           (size_t)p.M * p.N, final_C, n, accumulate);
        EGX_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace egx
