// Cut mode of the per-clip kernels (round 5): the FFN of a layer as launches of its own with EIGHT waves per clip.
//
// The per-clip kernels (fused.hip, fused_bwd.hip) run one 256-thread workgroup per clip — one wave per SIMD, up to 468 registers per
// wave. In-kernel stamps put 55 % of their cycles in the FFN hidden loops, where the matrix pipe is busy half of the time: a wave's own
// epilogue (bias, dropout hash, ReLU bits, three-way split, H-tile transposes: 2.3k of a hidden block's 9.4k cycles in f32s mode, 1.3k of
// 3.1k in bf16) and its LDS fragment reads do not run in the shadow of its own MFMAs; a SECOND wave's do (tools/micro/mfma_dep.hip,
// DESIGN_APPENDIX.md round 4). Two waves per SIMD need <= 256 registers per wave, which the other phases of the clip kernels (token
// preparation, Q | K | V, attention backward) do not fit. So the kernels are cut at the FFN, the way the tiled mode cuts them at the
// attention:
//
//   forward   fused_fwd_kernel<.., CUT>  [token preparation | saved layer input] .. LayerNorm1      -> x1 rows + bf16 operand planes
//             ffn_fwd_kernel             H = relu(W1 x1 + b1), Y = W2 H, second residual, LayerNorm2 -> next layer input | tokens | pooled head
//   backward  ffn_bwd_kernel             [pooled head backward] LayerNorm2 backward, dH = (W2^T g2) . alive, dX1 = W1^T dH -> dy1
//             fused_bwd_kernel<.., CUT>  LayerNorm1 backward .. in-projection input gradient -> d(layer input) | token-preparation backward
//
// The kernels here: 512 threads = 8 waves per clip, each wave every 8th 32-wide hidden block for all 48 tokens (same feature-major
// accumulator -> operand chains as the clip kernels), weight fragments streamed through a RING of D register slots in consumption order
// (fragment f of the stream sits in slot f mod D and is replaced by fragment f + D right behind the MFMAs that consumed it: every load
// unconditional, so hipcc's s_waitcnt counts are exact), the 8 partial outputs summed in two rounds through LDS in a fixed order.
// Everything another kernel reads (H / dH tiles, ReLU bits, operand planes, saved residual sums, per-clip partial rows) has the layout
// the one-launch kernels write: the weight-gradient kernels and the tests do not know which path ran.
// Reference math: torch.nn.TransformerEncoderLayer as built at HHI/models/ttm/model_taskspecific.py:212-215 (linear1 -> ReLU -> dropout ->
// linear2 -> dropout2 -> residual -> norm2) and its autograd.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "fused.h"
#include "fused_dev.h"

namespace egx {

// Development aid (-DEGX_STAMPS): per-phase cycle sums of the hidden loop for waves 0 and 4 (same SIMD) of workgroup 0, forward [0..15] and
// backward [16..31]: slot 8 * (wave / 4) + k; read back by egx_debug_stamps(out, -3000).
__device__ unsigned long long g_cstamps[32];
#ifdef EGX_STAMPS
#define CSTAMP_INIT() unsigned long long ct_prev = __builtin_amdgcn_s_memtime(), ct_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); ct_acc[k] += t_ - ct_prev; ct_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define CSTAMP_FLUSH(base) do { if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) for (int k_ = 0; k_ < 8; ++k_) g_cstamps[(base) + (threadIdx.x >> 8) * 8 + k_] = ct_acc[k_]; } while (0)
#define ASTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_cstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#define CSTAMP_INIT() do { } while (0)
#define CSTAMP(k) do { } while (0)
#define CSTAMP_FLUSH(base) do { } while (0)
#endif
int debug_read_cstamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cstamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
// diagnostic variants (timing only, results wrong): -DEGX_DIAG_NOLOAD: the weight ring is never refilled; -DEGX_DIAG_NOSTORE: no H / dH tiles, no bits
#ifdef EGX_DIAG_NOLOAD
#define RING_REFILL(slot, k, hb, hbn) do { } while (0)
#else
#define RING_REFILL(slot, k, hb, hbn) ring[slot] = frag_load(((k) + D) & 15, (k) + D < 16 ? (hb) : (hbn))
#endif

namespace {
constexpr int CT = 512;                 // threads per clip
constexpr int CNT = 3;                  // 16-token tiles
constexpr int CSP = CNT * 16;
constexpr int CBLK = CSP * LDX;         // one token-major fp32 block (floats)
constexpr int CPS = CSP * LDXH;         // one bf16 operand plane (halfwords)

// fragments in flight per wave (a divisor of the 16 fragments of a hidden block: slot = position mod depth is then static).
// f32s: 12 registers per fragment, f32: 8 — four is what fits 256 registers beside the 96 accumulators; bf16: 4 registers per fragment
// and only 48 matrix-pipe cycles of work per fragment: a whole block ahead.
#ifndef EGX_RING_SPLIT
#define EGX_RING_SPLIT 4
#endif
#ifndef EGX_RING_BF16
#define EGX_RING_BF16 16
#endif
template <int CM> struct RingDepth { static constexpr int v = EGX_RING_SPLIT; };
template <> struct RingDepth<CM_BF16> { static constexpr int v = EGX_RING_BF16; };

// B-operand fragment of token tile `row0` for the K-block at feature k0: from the pre-split planes (f32s: three, bf16: one) or the fp32 block
template <int CM>
__device__ __forceinline__ Frag<CM> operand_frag(const unsigned short* planes, const float* xf, int row, int k0, int q) {
    if constexpr (CM == CM_SPLIT) {
        return load_split_frag(planes, CPS, row, k0, q);
    } else if constexpr (CM == CM_BF16) {
        const unsigned short* b = planes + row * LDXH + k0 + 4 * q;
        const uint2 lo = *reinterpret_cast<const uint2*>(b), hi = *reinterpret_cast<const uint2*>(b + 16);
        Frag<CM_BF16> f;
        f.v = __builtin_bit_cast(bf16x8, (u32x4){lo.x, lo.y, hi.x, hi.y});
        return f;
    } else {
        return load_frag<CM>(xf + row * LDX + k0, q);
    }
}

// dense (48, 128) bf16 plane(s) in HBM -> LDS planes with the conflict-free row stride (rows are 8-byte aligned: two 8-byte writes per 16 bytes)
template <int NPL>
__device__ __forceinline__ void planes_to_lds(unsigned short* dst, const unsigned short* src, size_t plane_stride) {
    for (int i = threadIdx.x; i < NPL * CSP * (FD / 8); i += CT) {
        const int part = i / (CSP * (FD / 8)), rem = i - part * (CSP * (FD / 8));
        const int row = rem >> 4, c8 = rem & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(src + part * plane_stride + (size_t)rem * 8);
        unsigned short* d = dst + part * CPS + row * LDXH + c8 * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(d + 4) = make_uint2(v.z, v.w);
    }
}
template <int NPL>
__device__ __forceinline__ void planes_from_lds(unsigned short* dst, size_t plane_stride, const unsigned short* src) {
    for (int i = threadIdx.x; i < NPL * CSP * (FD / 8); i += CT) {
        const int part = i / (CSP * (FD / 8)), rem = i - part * (CSP * (FD / 8));
        const int row = rem >> 4, c8 = rem & 15;
        const unsigned short* s = src + part * CPS + row * LDXH + c8 * 8;
        const uint2 a = *reinterpret_cast<const uint2*>(s), b = *reinterpret_cast<const uint2*>(s + 4);
        *reinterpret_cast<uint4*>(dst + part * plane_stride + (size_t)rem * 8) = make_uint4(a.x, a.y, b.x, b.y);
    }
}
// rows [0, S) of a dense (S, 128) fp32 array -> token-major LDS block (rows >= S untouched)
__device__ __forceinline__ void rows_to_lds(float* dst, const float* src, int S) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    for (int i = threadIdx.x; i < S * (FD / 4); i += CT) {
        const int row = i >> 5, c = (i & 31) << 2;
        *reinterpret_cast<f32x4*>(dst + row * LDX + c) = s4[i];
    }
}
__device__ __forceinline__ void rows_from_lds(float* dst, const float* src, int S) {
    for (int i = threadIdx.x; i < S * (FD / 4); i += CT) {
        const int row = i >> 5, c = (i & 31) << 2;
        *reinterpret_cast<f32x4*>(dst + (size_t)i * 4) = *reinterpret_cast<const f32x4*>(src + row * LDX + c);
    }
}
// fp32 block rows [0, 48) -> one bf16 LDS plane (rows >= S of the block are zero)
__device__ __forceinline__ void block_to_plane(unsigned short* plane, const float* blk) {
    for (int i = threadIdx.x; i < CSP * (FD / 4); i += CT) {
        const int row = i >> 5, c = (i & 31) << 2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(blk + row * LDX + c);
        *reinterpret_cast<uint2*>(plane + row * LDXH + c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
}
// fp32 block -> the three bf16 parts (rows [0, 48): padded rows are zero and split to zeros)
__device__ __forceinline__ void block_to_split_planes(unsigned short* planes, const float* blk) {
    const int row = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 32;
    if (row < CSP) {
        float g[32];
        uint32_t h[16], m[16], lo[16];
        load32(blk + row * LDX + c0, g);
        split32(g, h, m, lo);
        store_parts32(planes + row * LDXH + c0, (size_t)CPS, h, m, lo);
    }
}

// Which hidden blocks a wave walks. Waves w and w + 4 share SIMD w: the pair owns blocks {w + 4 j : j < 2 nit} (nit = d_ff / 256). The
// matrix pipe and the VALU issue slots of a SIMD are arbitrated oldest-first (MI355X_MICROARCH.md, two waves per SIMD): with an even
// split the older wave (w < 4) ran its 8 blocks of the f32s loop in 83k cycles and then waited 24k at the barrier for the younger
// one (107k; stamps, profiles/r05_cut_stamps.txt). So the older wave takes nit / 8 blocks more, the younger as many fewer (9 : 7 at
// d_ff = 2048) — a STATIC split: sums stay in a fixed order. bf16's short blocks are 12 % apart and keep the even split.
// (round 6: the forward's split moved to 10 : 6 — three same-box pairs -2.5 us on the step; the backward's stays, 10 : 6 and 11 : 5 measured equal)
template <int CM, bool FWD>
__device__ __forceinline__ void cut_walk(int wave, int nit, int& j_begin, int& n_mine, int shift_code = 0) {
    const int shift = shift_code ? shift_code - 1 : (CM == CM_BF16 ? 0 : (FWD ? nit / 4 : nit / 8));
    const int n_old = nit + shift;
    j_begin = wave < 4 ? 0 : n_old;
    n_mine = wave < 4 ? n_old : 2 * nit - n_old;
}
// parameter vectors the tail of the kernels needs, staged in LDS at kernel entry: a global load issued behind the loop's tile stores
// waits for all of them (vmcnt counts in order), and each of these phases is a chain of two or three such round trips otherwise
constexpr int CP_FLOATS = 1024 + 8 * FD;       // [norm_w | norm_b | lin2_b | head ln_w | head ln_b | head b (64) | d_logits (64) | pad][<= 8 head rows]
constexpr int CP_HEAD_ROWS = 8;
__device__ __forceinline__ int cut_rot(int rot_mode, int clip, int nit) {
    return rot_mode == 0 ? (int)((clip * 11u + (clip >> 3) * 5u) % (unsigned)nit)
         : rot_mode == 4 ? (int)(((unsigned)(clip >> 3) & 3u) % (unsigned)nit)
         : rot_mode == 5 ? (int)(((unsigned)(clip >> 3) & 7u) % (unsigned)nit) : 0;
}

// y (8 feature tiles x 3 token tiles of this wave) += the other waves': two rounds through four LDS blocks, fixed order. Afterwards
// blocks 0..3 hold the sums (w, w + 4) of waves w = 0..3; the caller adds the four in order.
__device__ __forceinline__ void reduce_partials8(f32x4 (&y)[8][CNT], float* blk0, int wave, int r, int q) {
    if (wave >= 4) {
        float* mine = blk0 + (wave - 4) * CBLK;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < CNT; ++t)
                *reinterpret_cast<f32x4*>(mine + (t * 16 + r) * LDX + i * 16 + 4 * q) = y[i][t];
    }
    __syncthreads();
    if (wave < 4) {
        float* mine = blk0 + wave * CBLK;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < CNT; ++t) {
                f32x4* a = reinterpret_cast<f32x4*>(mine + (t * 16 + r) * LDX + i * 16 + 4 * q);
                *a = y[i][t] + *a;
            }
    }
    __syncthreads();
}
}  // namespace

// ---- forward ---------------------------------------------------------------------------------------------------------------------------
template <int CM>
__global__ __launch_bounds__(CT) void ffn_fwd_kernel(FusedFwdParams p, int l) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X1 = lds;                        // x1 (fp32), later res2
    float* R = lds + CBLK;                  // four blocks: operand planes during the loop, partial sums afterwards, LayerNorm2 output at the end
    unsigned short* XP = reinterpret_cast<unsigned short*>(R);
    constexpr int NPL = CM == CM_SPLIT ? 3 : 1;
    constexpr int D = RingDepth<CM>::v;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int clip = blockIdx.x, S = p.S;
    const size_t tokbase = (size_t)clip * S;
    const FusedLayer& w = p.layer[l];
    const bool dev_seed = p.seed_ptr != nullptr;
    const uint64_t seed_dev = dev_seed ? *p.seed_ptr : 0ull;
    const uint64_t k_ffn = dev_seed ? site_key(seed_dev, l, SITE_FFN) : w.ffn_key;
    const uint64_t k_res2 = dev_seed ? site_key(seed_dev, l, SITE_RES2) : w.res2_key;

    ASTAMP(16);
    // weight streams of the next launch -> Infinity Cache (TouchList): the oldest loads of this launch, consumed behind the prologue's barrier
    Touched tch = {{0, 0, 0, 0}};
    if (p.touch.n) tch = touch_lines<CT>(p.touch, blockIdx.x, gridDim.x, tid);
    const int nhb = p.d_ff / 32;
    int j_begin_, n_mine_;
    cut_walk<CM, true>(wave, nhb / 8, j_begin_, n_mine_, p.rot_mode >> 8);
    const int nit = __builtin_amdgcn_readfirstlane(n_mine_), j_begin = __builtin_amdgcn_readfirstlane(j_begin_);
    const int pair_s = __builtin_amdgcn_readfirstlane(wave & 3);
    const int rot = cut_rot(p.rot_mode & 255, clip, nit);
    auto hb_of = [&](int it) { int j = it + rot; if (j >= nit) j -= nit; return pair_s + 4 * (j_begin + j); };
    float* CP = lds + 5 * CBLK;
    // LayerNorm2 / bias / head parameters -> LDS: requested here with everything else (unconditional, clamped), stored in front of the first barrier
    const bool cp_hd = p.head.n_out > 0 && l + 1 == p.n_layers;
    const int cp_g = tid >> 5, cp_c4 = (tid & 31) << 2;
    f32x4 cp_v, cp_w;
    float cp_b;
    {
        const float* src = cp_g == 0 ? w.norm2_w : cp_g == 1 ? w.norm2_b : (cp_g == 3 && cp_hd) ? p.head.ln_w : (cp_g == 4 && cp_hd) ? p.head.ln_b : w.lin2_b;
        cp_v = *reinterpret_cast<const f32x4*>(src + cp_c4);
        const int nb = cp_hd ? p.head.n_out : 0, nw = cp_hd && p.head.n_out <= CP_HEAD_ROWS ? p.head.n_out * 32 : 0;
        cp_b = nb ? p.head.b[min(max(tid - 160, 0), nb - 1)] : 0.f;
        cp_w = *reinterpret_cast<const f32x4*>(nw ? p.head.W + (size_t)min(max(tid - 256, 0), nw - 1) * 4 : w.lin2_b);
    }
    auto cp_store = [&] {
        if (cp_g < 3 || (cp_g < 5 && cp_hd)) *reinterpret_cast<f32x4*>(CP + cp_g * FD + cp_c4) = cp_v;
        if (cp_hd) {
            if (tid >= 160 && tid < 160 + p.head.n_out) CP[5 * FD + (tid - 160)] = cp_b;
            if (p.head.n_out <= CP_HEAD_ROWS && tid >= 256 && tid < 256 + p.head.n_out * 32) *reinterpret_cast<f32x4*>(CP + 1024 + (tid - 256) * 4) = cp_w;
        }
    };
    // the weight stream of a hidden block in consumption order: W1 fragments (row tile i, K-block kb) for kb = 0..3, i = 0..1, then the
    // eight W2 fragments (feature tile i); the dropout keep-scale rides on the packed W1 (encoder.hip)
    auto frag_load = [&](int k, int hb) -> WRaw<CM> {
        return k < 8 ? load_w<CM>(w.lin1_wp, hb * 2 + (k & 1), FD / 32, k >> 1, lane) : load_w<CM>(w.lin2_wp, k - 8, nhb, hb, lane);
    };
    WRaw<CM> ring[D];
    float4 b1r[2];
    {
        const int hb0 = hb_of(0);
#pragma unroll
        for (int k = 0; k < D; ++k) ring[k] = frag_load(k, hb0);
#pragma unroll
        for (int i = 0; i < 2; ++i) b1r[i] = *reinterpret_cast<const float4*>(w.lin1_b + hb0 * 32 + i * 16 + 4 * q);
    }
    // x1: fp32 rows (residual; exact-fp32 mode: also the B operand) and the operand planes the attention-side kernel left. Every load
    // is requested before the first one is used (unconditional, clamped: a copy loop of "load, then store to LDS" iterations pays one
    // memory round trip per iteration, and this kernel has nothing else to run meanwhile): one round trip for the whole prologue.
    {
        constexpr int XR = CSP * (FD / 4) / CT;                                     // f32x4 per thread of the x1 block
        constexpr int NPV = CM == CM_F32 ? 1 : (NPL * CSP * (FD / 8) + CT - 1) / CT;  // 16-byte pieces per thread of the planes
        static_assert(CSP * (FD / 4) % CT == 0, "the x1 block must divide among the threads");
        f32x4 xr[XR];
        u32x4 pv[NPV];
        const f32x4* xs = reinterpret_cast<const f32x4*>(p.x1f_out + ((size_t)l * p.Ntok + tokbase) * FD);
        const int nx = S * (FD / 4);
#pragma unroll
        for (int k = 0; k < XR; ++k) { const int i = tid + CT * k; xr[k] = xs[i < nx ? i : nx - 1]; }
        const bool have_planes = CM != CM_F32 && p.x1p_out != nullptr;
        if constexpr (CM != CM_F32) {
            constexpr int NPIECE = NPL * CSP * (FD / 8);
            const size_t plane = CM == CM_SPLIT ? (size_t)p.B * CSP * FD : 0;
            const unsigned short* src = !have_planes ? reinterpret_cast<const unsigned short*>(p.x1f_out)      // (never used: any valid address)
                                      : CM == CM_SPLIT ? p.x1p_out + (size_t)l * 3 * plane + (size_t)clip * CSP * FD
                                                       : p.x1p_out + ((size_t)l * p.B + clip) * CSP * FD;
#pragma unroll
            for (int k = 0; k < NPV; ++k) {
                int i = tid + CT * k;
                i = i < NPIECE ? i : NPIECE - 1;
                const int part = i / (CSP * (FD / 8)), rem = i - part * (CSP * (FD / 8));
                pv[k] = *reinterpret_cast<const u32x4*>(src + (have_planes ? part * plane + (size_t)rem * 8 : 0));
            }
        }
        for (int i = tid; i < (CSP - S) * LDX; i += CT) X1[S * LDX + i] = 0.f;      // padded rows: zero operands
        cp_store();
#pragma unroll
        for (int k = 0; k < XR; ++k) {
            const int i = tid + CT * k, row = i >> 5, c = (i & 31) << 2;
            if (i < nx) *reinterpret_cast<f32x4*>(X1 + row * LDX + c) = xr[k];
        }
        if constexpr (CM != CM_F32) {
            constexpr int NPIECE = NPL * CSP * (FD / 8);
            if (have_planes) {
#pragma unroll
                for (int k = 0; k < NPV; ++k) {
                    const int i = tid + CT * k;
                    const int part = i / (CSP * (FD / 8)), rem = i - part * (CSP * (FD / 8));
                    const int row = rem >> 4, c8 = rem & 15;
                    if (i < NPIECE) {
                        unsigned short* d = XP + part * CPS + row * LDXH + c8 * 8;
                        const bool live = row < S;      // rows >= S: zero operands whatever the producer left there (the bf16 plane is written for rows < S only)
                        *reinterpret_cast<uint2*>(d) = live ? make_uint2(pv[k][0], pv[k][1]) : make_uint2(0, 0);
                        *reinterpret_cast<uint2*>(d + 4) = live ? make_uint2(pv[k][2], pv[k][3]) : make_uint2(0, 0);
                    }
                }
            }
        }
        __syncthreads();
        touch_sink(tch);
        if (CM != CM_F32 && !have_planes) {     // no planes handed over (EGX_FFN_DW_PLANES=0): build them from the fp32 rows
            if constexpr (CM == CM_SPLIT) block_to_split_planes(XP, X1);
            else if constexpr (CM == CM_BF16) block_to_plane(XP, X1);
            __syncthreads();
        }
    }

    ASTAMP(17);
    f32x4 y[8][CNT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < CNT; ++t) y[i][t] = f32x4{0, 0, 0, 0};
    const float bscale = w.ffn_thresh ? w.drop_inv : 1.f;
    const size_t bits_base = ((size_t)l * p.B + clip) * nhb * 64 + lane;
    constexpr int ESZ = CM == CM_BF16 ? 2 : 4;
    const int nht = p.d_ff / 16;
    char* const hid_base = (char*)p.hid_out + ((size_t)l * p.B + clip) * CNT * nht * (size_t)(HTILE_ELEMS * ESZ);

    CSTAMP_INIT();
    for (int it = 0; it < nit; ++it) {
        const int hb = hb_of(it);
        const int hbn = hb_of(it + 1 < nit ? it + 1 : it);      // the last block refills itself (never used)
        __builtin_amdgcn_sched_barrier(0);
        CSTAMP(0);
        f32x4 hacc[2][CNT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < CNT; ++t) hacc[i][t] = f32x4{0, 0, 0, 0};
        Frag<CM> xb[CNT];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = k & 1, kb = k >> 1;
            if (i == 0) {
#pragma unroll
                for (int t = 0; t < CNT; ++t) xb[t] = operand_frag<CM>(XP, X1, t * 16 + r, kb * 32, q);
            }
            pin(ring[k % D]);
            if (k == 0) { CSTAMP(1); }
            Frag<CM> a = w_frag<CM>(ring[k % D]);
#pragma unroll
            for (int t = 0; t < CNT; ++t) mma<CM>(hacc[i][t], a, xb[t]);
            __builtin_amdgcn_sched_barrier(0);
            RING_REFILL(k % D, k, hb, hbn);
            __builtin_amdgcn_sched_barrier(0);
        }
        CSTAMP(2);
        float bv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bv[i][0] = b1r[i].x * bscale; bv[i][1] = b1r[i].y * bscale; bv[i][2] = b1r[i].z * bscale; bv[i][3] = b1r[i].w * bscale;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) b1r[i] = *reinterpret_cast<const float4*>(w.lin1_b + hbn * 32 + i * 16 + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
        if (w.ffn_thresh) {     // one wave-uniform branch per hidden block (no memory operation inside)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t cq = (uint32_t)(hb * 32 + i * 16 + 4 * q) >> 2;
#pragma unroll
                for (int t = 0; t < CNT; ++t) {
                    const uint2 h = rand_quad(k_ffn, (uint32_t)(clip * 64 + t * 16 + r), cq);
                    // dropped units become negative: the ReLU below zeroes them and their sign bit marks them dead
                    hacc[i][t][0] = keep_lo(h.x, w.ffn_thresh) ? hacc[i][t][0] + bv[i][0] : -1.f;
                    hacc[i][t][1] = keep_hi(h.x, w.ffn_thresh) ? hacc[i][t][1] + bv[i][1] : -1.f;
                    hacc[i][t][2] = keep_lo(h.y, w.ffn_thresh) ? hacc[i][t][2] + bv[i][2] : -1.f;
                    hacc[i][t][3] = keep_hi(h.y, w.ffn_thresh) ? hacc[i][t][3] + bv[i][3] : -1.f;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < CNT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hacc[i][t][e] += bv[i][e];
        }
        // "alive" bits (ReLU active AND kept by the dropout = sign bit clear), same word layout as fused_fwd_kernel
        uint32_t dead = 0;
#pragma unroll
        for (int k = 2 * CNT * 4 - 1; k >= 0; --k) {
            const int i = k / (CNT * 4), t = (k / 4) % CNT, e = k & 3;
            dead = __builtin_amdgcn_alignbit(dead, __float_as_uint(hacc[i][t][e]), 31);
        }
        p.relu_bits[bits_base + (size_t)hb * 64] = ~dead & ((1u << (2 * CNT * 4)) - 1u);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < CNT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) hacc[i][t][e] = __int_as_float(max(__float_as_int(hacc[i][t][e]), 0));
        Frag<CM> hbq[CNT];
#pragma unroll
        for (int t = 0; t < CNT; ++t) hbq[t] = chain_frag<CM>(hacc[0][t], hacc[1][t]);
#ifndef EGX_DIAG_NOSTORE
        {       // H tiles for the weight-gradient kernel
            char* hb_base = hid_base + (size_t)hb * 2 * (HTILE_ELEMS * ESZ);
#pragma unroll
            for (int t = 0; t < CNT; ++t) {
                if constexpr (CM == CM_BF16) {
                    const u32x4 u = __builtin_bit_cast(u32x4, hbq[t].v);
                    store_hid_tile_bf16(hb_base + (size_t)t * nht * (HTILE_ELEMS * ESZ), u, lane, S - t * 16);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        store_hid_tile<CM>(hb_base + ((size_t)t * nht + i) * (HTILE_ELEMS * ESZ), hacc[i][t], lane, S - t * 16);
                }
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        CSTAMP(3);
#pragma unroll
        for (int k = 8; k < 16; ++k) {
            pin(ring[k % D]);
            if (k == 8) { CSTAMP(4); }
            Frag<CM> a = w_frag<CM>(ring[k % D]);
#pragma unroll
            for (int t = 0; t < CNT; ++t) mma<CM>(y[k - 8][t], a, hbq[t]);
            __builtin_amdgcn_sched_barrier(0);
            RING_REFILL(k % D, k, hb, hbn);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    CSTAMP(5);
    CSTAMP_FLUSH(0);
    ASTAMP(18);
    __syncthreads();        // every wave is done with the operand planes: the partial sums overwrite them
    ASTAMP(19);
    reduce_partials8(y, R, wave, r, q);
    ASTAMP(20);

    // ---- sum of the partials + bias + dropout2 + residual -> res2 (in X1), saved; LayerNorm2 -> block 0 of R
    for (int i = tid; i < S * (FD / 4); i += CT) {
        const int row = i >> 5, c = (i & 31) << 2, o = row * LDX + c;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(R + o), a1 = *reinterpret_cast<const f32x4*>(R + CBLK + o);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(R + 2 * CBLK + o), a3 = *reinterpret_cast<const f32x4*>(R + 3 * CBLK + o);
        f32x4 f = ((a0 + a1) + (a2 + a3)) + *reinterpret_cast<const f32x4*>(CP + 2 * FD + c);
        if (w.res_thresh) {
            float m4[4];
            drop_scale4(k_res2, (uint32_t)(tokbase + row), (uint32_t)c, w.res_thresh, w.drop_inv, m4);
            f = f * f32x4{m4[0], m4[1], m4[2], m4[3]};
        }
        *reinterpret_cast<f32x4*>(X1 + o) = f + *reinterpret_cast<const f32x4*>(X1 + o);
    }
    __syncthreads();
    ASTAMP(21);
    const bool last = l + 1 == p.n_layers;
    float* Y = R;
    ln_rows(X1, S, CP, CP + FD, p.eps, [&](int row, int c0, float (&x)[32], float (&yv)[32]) {
        if (last && p.tokens_out && row < p.out_T) store32(p.tokens_out + ((size_t)clip * p.out_T + row) * FD + c0, yv);
        store32(Y + row * LDX + c0, yv);
    }, [&] { rows_from_lds(p.saved_res + ((size_t)(2 * l + 1) * p.B + clip) * S * FD, X1, S); });
    __syncthreads();
    ASTAMP(22);
    if (!last) {        // the next layer's input: the attention-side kernel of layer l + 1 starts from it (and the backward reads it)
        rows_from_lds(p.xin_out + ((size_t)(l + 1) * p.B + clip) * S * FD, Y, S);
        return;
    }
    // ---- optional pooled head: logits = Linear(LN(mean_s tokens)) (as fused_fwd_kernel)
    if (p.head.n_out > 0) {
        float* pooled = X1;
        // fused weighted cross entropy (egx_ce): the labels and their class weights are requested here, under the pooling
        const FusedCe ce{p.ce_target, p.ce_weight, p.ce_loss, p.ce_dlogits, p.ce_B, nullptr};
        CeReq rq;
        if (ce.target) { ce_request_labels<CT>(ce, tid, rq); ce_request_weights(ce, p.head.n_out, rq); }
        if (tid < FD) pooled[tid] = colsum_lds(Y, 0, S, tid) * (1.f / (float)S);
        if (ce.target) ce_weight_partials<CT>(ce, p.head.n_out, tid, rq, pooled + 256);
        __syncthreads();
        if (wave == 0) {
            float2 x = *reinterpret_cast<float2*>(pooled + 2 * lane);
            float mean = wsum(x.x + x.y) * (1.f / FD);
            float dx = x.x - mean, dy = x.y - mean;
            float rstd = rsqrtf(wsum(dx * dx + dy * dy) * (1.f / FD) + p.eps);
            float2 lw = *reinterpret_cast<const float2*>(CP + 3 * FD + 2 * lane);
            float2 lb = *reinterpret_cast<const float2*>(CP + 4 * FD + 2 * lane);
            float y0 = dx * rstd * lw.x + lb.x, y1 = dy * rstd * lw.y + lb.y;
            const bool w_lds = p.head.n_out <= CP_HEAD_ROWS;
            float zmine = 0.f;
            for (int o = 0; o < p.head.n_out; ++o) {
                float2 wv = w_lds ? *reinterpret_cast<const float2*>(CP + 1024 + o * FD + 2 * lane) : *reinterpret_cast<const float2*>(p.head.W + (size_t)o * FD + 2 * lane);
                float sdot = wsum(y0 * wv.x + y1 * wv.y) + CP[5 * FD + o];
                if (lane == 0) p.logits_out[(size_t)clip * p.head.n_out + o] = sdot;
                zmine = lane == o ? sdot : zmine;
            }
            if (ce.target) ce_clip<CT>(ce, p.head.n_out, clip, lane, zmine, pooled + 256, true);
        }
    }
    ASTAMP(23);
}

// ---- backward --------------------------------------------------------------------------------------------------------------------------
template <int CM>
__global__ __launch_bounds__(CT) void ffn_bwd_kernel(FusedBwdParams p, int l) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* B1 = lds;                // res2 -> d_res2 (kept until the final sum)
    float* Gs = lds + 1 * CBLK;     // dY of this layer; partial 0
    float* B2 = lds + 2 * CBLK;     // g2 = d_res2 . dropout2 mask; partial 1
    float* B3 = lds + 3 * CBLK;     // dY . xhat, then (with B4) the operand planes of g2; partials 2, 3
    float* B4 = lds + 4 * CBLK;
    unsigned short* GP = reinterpret_cast<unsigned short*>(B3);
    constexpr int D = RingDepth<CM>::v;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int clip = blockIdx.x, S = p.S;
    const size_t tok0 = (size_t)clip * S;
    const FusedBwdLayer& w = p.layer[l];
    const bool dev_seed = p.seed_ptr != nullptr;
    const uint64_t seed_dev = dev_seed ? *p.seed_ptr : 0ull;
    const uint64_t k_res2 = dev_seed ? site_key(seed_dev, l, SITE_RES2) : w.res2_key;
    const bool last = l + 1 == p.n_layers;

    ASTAMP(24);
    Touched tch = {{0, 0, 0, 0}};        // (see ffn_fwd_kernel)
    if (p.touch.n) tch = touch_lines<CT>(p.touch, blockIdx.x, gridDim.x, tid);
    if (p.zero_buf) {       // the caller's flat gradient buffer (first launch of the backward only): accumulated into by later launches
        const size_t n4 = p.zero_n / 4, per = (n4 + gridDim.x - 1) / gridDim.x;
        const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
        for (size_t k = b0 + threadIdx.x; k < b1; k += CT) reinterpret_cast<float4*>(p.zero_buf)[k] = make_float4(0, 0, 0, 0);
    }
    float* part = p.partials + (size_t)clip * p.P;
    float* pl = part + l * FUSED_P_LAYER;

    const int nhb = p.d_ff / 32;
    int j_begin_, n_mine_;
    cut_walk<CM, false>(wave, nhb / 8, j_begin_, n_mine_, p.rot_mode >> 8);
    const int nit = __builtin_amdgcn_readfirstlane(n_mine_), j_begin = __builtin_amdgcn_readfirstlane(j_begin_);
    const int pair_s = __builtin_amdgcn_readfirstlane(wave & 3);
    const int rot = cut_rot(p.rot_mode & 255, clip, nit);
    auto hb_of = [&](int it) { int j = it + rot; if (j >= nit) j -= nit; return pair_s + 4 * (j_begin + j); };
    float* CP = lds + 5 * CBLK;
    // LayerNorm2 weights and the head's parameters -> LDS (see ffn_fwd_kernel): requested here, stored in front of the first barrier
    const bool cp_hd = p.head.n_out > 0 && last;
    const int cp_g = tid >> 5, cp_c4 = (tid & 31) << 2;
    f32x4 cp_v, cp_w;
    float cp_b;
    {
        const float* src = cp_g == 1 ? w.norm2_b : (cp_g == 3 && cp_hd) ? p.head.ln_w : (cp_g == 4 && cp_hd) ? p.head.ln_b : w.norm2_w;
        cp_v = *reinterpret_cast<const f32x4*>(src + cp_c4);
        const int nb = cp_hd ? p.head.n_out : 0, nw = cp_hd && p.head.n_out <= CP_HEAD_ROWS ? p.head.n_out * 32 : 0;
        cp_b = nb ? p.d_logits[(size_t)clip * p.head.n_out + min(max(tid - 160, 0), nb - 1)] : 0.f;
        if (nb && p.d_logits_scale) cp_b *= *p.d_logits_scale;
        cp_w = *reinterpret_cast<const f32x4*>(nw ? p.head.W + (size_t)min(max(tid - 256, 0), nw - 1) * 4 : w.norm2_w);
    }
    auto cp_store = [&] {
        if (cp_g < 2 || ((cp_g == 3 || cp_g == 4) && cp_hd)) *reinterpret_cast<f32x4*>(CP + cp_g * FD + cp_c4) = cp_v;
        if (cp_hd) {
            if (tid >= 160 && tid < 160 + p.head.n_out) CP[5 * FD + 64 + (tid - 160)] = cp_b;
            if (p.head.n_out <= CP_HEAD_ROWS && tid >= 256 && tid < 256 + p.head.n_out * 32) *reinterpret_cast<f32x4*>(CP + 1024 + (tid - 256) * 4) = cp_w;
        }
    };
    // weight stream of a hidden block: W2^T fragments (hidden tile i, K-block kb) for kb = 0..3, i = 0..1 (the dropout keep-scale rides on
    // them), then the eight W1^T fragments (feature tile i)
    auto frag_load = [&](int k, int hb) -> WRaw<CM> {
        return k < 8 ? load_w<CM>(w.lin2_wtp, hb * 2 + (k & 1), FD / 32, k >> 1, lane) : load_w<CM>(w.lin1_wtp, k - 8, nhb, hb, lane);
    };
    const uint32_t* relu_bits = p.relu_bits + ((size_t)l * p.B + clip) * (size_t)nhb * 64 + lane;
    WRaw<CM> ring[D];
    uint32_t relu_word;
    {
        const int hb0 = hb_of(0);
#pragma unroll
        for (int k = 0; k < D; ++k) ring[k] = frag_load(k, hb0);
        relu_word = relu_bits[(size_t)hb0 * 64];
    }
    // res2 and this layer's dY are requested together, ahead of everything that needs them (see ffn_fwd_kernel)
    const float* res2 = p.saved_res + ((size_t)(2 * l + 1) * p.B + clip) * S * FD;
    const bool head_bwd = last && p.head.n_out > 0;
    {
        constexpr int XR = CSP * (FD / 4) / CT;
        f32x4 rr[XR], dr[XR];
        const f32x4* rs = reinterpret_cast<const f32x4*>(res2);
        const int ndy_rows = head_bwd ? 0 : (last ? p.out_T : S);      // rows of dY that come from memory
        const f32x4* ds = head_bwd ? rs : reinterpret_cast<const f32x4*>(last ? p.d_tokens + (size_t)clip * p.out_T * FD : p.dxin + tok0 * FD);
        const int nx = S * (FD / 4), nd = ndy_rows * (FD / 4);
#pragma unroll
        for (int k = 0; k < XR; ++k) {
            const int i = tid + CT * k;
            rr[k] = rs[i < nx ? i : nx - 1];
            dr[k] = ds[i < nd ? i : (nd > 0 ? nd - 1 : 0)];
        }
        for (int i = tid; i < 5 * CBLK / 4; i += CT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0, 0, 0, 0};
        cp_store();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < XR; ++k) {
            const int i = tid + CT * k, row = i >> 5, c = (i & 31) << 2;
            if (i < nx) {
                *reinterpret_cast<f32x4*>(B1 + row * LDX + c) = rr[k];
                if (head_bwd) *reinterpret_cast<f32x4*>(Gs + row * LDX + c) = rr[k];
            }
            if (i < nd) *reinterpret_cast<f32x4*>(Gs + row * LDX + c) = dr[k];      // rows >= out_T carry no upstream gradient (LDS is zero there)
        }
        touch_sink(tch);
    }
    ASTAMP(25);
    if (head_bwd) {
        // fused pooled head backward (as fused_bwd_kernel): rebuild y = LN2(res2), pool, head forward / backward for this clip (wave 0),
        // d(tokens) = d(pooled) / S broadcast into Gs
        float* hp = part + p.head_off;
        __syncthreads();
        ln_rows(Gs, S, CP, CP + FD, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) { store32(Gs + row * LDX + c0, y); });
        __syncthreads();
        float* pooled = B2;        // [0,128): pooled; [128,256): d(pooled)
        if (tid < FD) pooled[tid] = colsum_lds(Gs, 0, S, tid) * (1.f / (float)S);
        __syncthreads();
        if (wave == 0) {
            float2 x = *reinterpret_cast<float2*>(pooled + 2 * lane);
            float mean = wsum(x.x + x.y) * (1.f / FD);
            float xh0 = x.x - mean, xh1 = x.y - mean;
            float rstd = rsqrtf(wsum(xh0 * xh0 + xh1 * xh1) * (1.f / FD) + p.eps);
            xh0 *= rstd; xh1 *= rstd;
            float2 lw = *reinterpret_cast<const float2*>(CP + 3 * FD + 2 * lane);
            float2 lb = *reinterpret_cast<const float2*>(CP + 4 * FD + 2 * lane);
            float y0 = xh0 * lw.x + lb.x, y1 = xh1 * lw.y + lb.y;
            float d0 = 0.f, d1 = 0.f;
            const bool w_lds = p.head.n_out <= CP_HEAD_ROWS;
            for (int o = 0; o < p.head.n_out; ++o) {
                float go = CP[5 * FD + 64 + o];
                float2 wv = w_lds ? *reinterpret_cast<const float2*>(CP + 1024 + o * FD + 2 * lane) : *reinterpret_cast<const float2*>(p.head.W + (size_t)o * FD + 2 * lane);
                d0 += go * wv.x; d1 += go * wv.y;
                *reinterpret_cast<float2*>(hp + 256 + FUSED_HEAD_MAX_OUT + o * FD + 2 * lane) = make_float2(go * y0, go * y1);
                if (lane == 0) hp[256 + o] = go;
            }
            *reinterpret_cast<float2*>(hp + 2 * lane) = make_float2(d0 * xh0, d1 * xh1);          // d(head ln_w)
            *reinterpret_cast<float2*>(hp + 128 + 2 * lane) = make_float2(d0, d1);                // d(head ln_b)
            float g0 = d0 * lw.x, g1 = d1 * lw.y;
            float s1 = wsum(g0 + g1) * (1.f / FD);
            float s2 = wsum(g0 * xh0 + g1 * xh1) * (1.f / FD);
            float inv_s = 1.f / (float)S;
            *reinterpret_cast<float2*>(pooled + 128 + 2 * lane) =
                make_float2(rstd * (g0 - s1 - xh0 * s2) * inv_s, rstd * (g1 - s1 - xh1 * s2) * inv_s);
        }
        __syncthreads();
        f32x4 dp[1];
        {
            const int c = (tid & 31) << 2;
            dp[0] = *reinterpret_cast<const f32x4*>(pooled + 128 + c);
        }
        __syncthreads();        // pooled lives in B2: everybody has its d(pooled) quad before the rows of Gs (and B2's zeros) are rewritten
        for (int i = tid; i < S * (FD / 4); i += CT) {      // (the quad a thread holds is the column quad of every element it writes: CT % 32 == 0)
            const int row = i >> 5, c = (i & 31) << 2;
            *reinterpret_cast<f32x4*>(Gs + row * LDX + c) = dp[0];
        }
        if (tid < 64) *reinterpret_cast<f32x4*>(B2 + 4 * tid) = f32x4{0, 0, 0, 0};      // B2 is an accumulation / operand block again
    }
    __syncthreads();
    ASTAMP(26);
    // P2: LayerNorm2 backward. B1 <- d_res2 (in place), B3 <- dY * xhat, B2 <- g2 = d_res2 .* dropout2 mask
    ln_bwd_rows(S, CP, p.eps,
        [&](int row, int c0, float (&dy)[32], float (&x)[32]) { load32(Gs + row * LDX + c0, dy); load32(B1 + row * LDX + c0, x); },
        [&](int row, int c0, float (&dy)[32], float (&dx)[32], float (&dyx)[32]) {
            store32(B1 + row * LDX + c0, dx);
            store32(B3 + row * LDX + c0, dyx);
            if (w.res_thresh) {
                const uint32_t orow = (uint32_t)(tok0 + row);
#pragma unroll
                for (int j = 0; j < 32; ++j) dx[j] *= drop_scale(k_res2, orow, (uint32_t)(c0 + j), w.res_thresh, w.drop_inv);
            }
            store32(B2 + row * LDX + c0, dx);
        });
    __syncthreads();
    // P3: column sums (norm2_w, norm2_b, lin2_b partials): three groups of 128 threads
    if (tid < 128) pl[0 + tid] = colsum_lds(B3, 0, S, tid);
    else if (tid < 256) pl[128 + (tid - 128)] = colsum_lds(Gs, 0, S, tid - 128);
    else if (tid < 384) pl[256 + (tid - 256)] = colsum_lds(B2, 0, S, tid - 256);
    // g2 leaves for the weight-gradient kernel (fp32 rows, or the operand planes it multiplies), and becomes the loop's B operand
    if (!(CM != CM_F32 && p.xg_planes)) rows_from_lds(w.g2_out + tok0 * FD, B2, S);
    __syncthreads();        // dY . xhat (B3) is consumed: the planes go over B3 / B4
    if constexpr (CM == CM_SPLIT) block_to_split_planes(GP, B2);
    else if constexpr (CM == CM_BF16) block_to_plane(GP, B2);
    __syncthreads();
    if (CM != CM_F32 && p.xg_planes) {
        if constexpr (CM == CM_SPLIT) {
            const size_t plane = (size_t)p.B * CSP * FD;
            planes_from_lds<3>(reinterpret_cast<unsigned short*>(w.g2_out) + (size_t)clip * CSP * FD, plane, GP);
        } else {
            planes_from_lds<1>(reinterpret_cast<unsigned short*>(w.g2_out) + (size_t)clip * CSP * FD, 0, GP);
        }
    }

    ASTAMP(27);
    // P4: FFN input gradient. dH^T = (W2^T g2^T) .* alive; dX1^T += W1^T dH^T
    f32x4 dxa[8][CNT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < CNT; ++t) dxa[i][t] = f32x4{0, 0, 0, 0};
    constexpr int ESZ = CM == CM_BF16 ? 2 : 4;
    const int nht = p.d_ff / 16;
    char* const dhid_base = (char*)p.dhid_out + ((size_t)l * p.B + clip) * CNT * nht * (size_t)(HTILE_ELEMS * ESZ);
    for (int it = 0; it < nit; ++it) {
        const int hb = hb_of(it);
        const int hbn = hb_of(it + 1 < nit ? it + 1 : it);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 dacc[2][CNT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < CNT; ++t) dacc[i][t] = f32x4{0, 0, 0, 0};
        Frag<CM> gb[CNT];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = k & 1, kb = k >> 1;
            if (i == 0) {
#pragma unroll
                for (int t = 0; t < CNT; ++t) gb[t] = operand_frag<CM>(GP, B2, t * 16 + r, kb * 32, q);
            }
            pin(ring[k % D]);
            Frag<CM> a2 = w_frag<CM>(ring[k % D]);
#pragma unroll
            for (int t = 0; t < CNT; ++t) mma<CM>(dacc[i][t], a2, gb[t]);
            __builtin_amdgcn_sched_barrier(0);
            RING_REFILL(k % D, k, hb, hbn);
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t bits = relu_word;
        __builtin_amdgcn_sched_barrier(0);
        relu_word = relu_bits[(size_t)hbn * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < CNT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = (i * CNT + t) * 4 + e;
                    const int32_t m = ((int32_t)(bits << (31 - k))) >> 31;
                    dacc[i][t][e] = __uint_as_float(__float_as_uint(dacc[i][t][e]) & (uint32_t)m);
                }
        Frag<CM> dq_[CNT];
#pragma unroll
        for (int t = 0; t < CNT; ++t) dq_[t] = chain_frag<CM>(dacc[0][t], dacc[1][t]);
        {       // dH tiles for the weight-gradient kernel
            char* hb_base = dhid_base + (size_t)hb * 2 * (HTILE_ELEMS * ESZ);
#pragma unroll
            for (int t = 0; t < CNT; ++t) {
                if constexpr (CM == CM_BF16) {
                    const u32x4 u = __builtin_bit_cast(u32x4, dq_[t].v);
                    store_hid_tile_bf16(hb_base + (size_t)t * nht * (HTILE_ELEMS * ESZ), u, lane, S - t * 16);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        store_hid_tile<CM>(hb_base + ((size_t)t * nht + i) * (HTILE_ELEMS * ESZ), dacc[i][t], lane, S - t * 16);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 8; k < 16; ++k) {
            pin(ring[k % D]);
            Frag<CM> a = w_frag<CM>(ring[k % D]);
#pragma unroll
            for (int t = 0; t < CNT; ++t) mma<CM>(dxa[k - 8][t], a, dq_[t]);
            __builtin_amdgcn_sched_barrier(0);
            RING_REFILL(k % D, k, hb, hbn);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    ASTAMP(28);
    __syncthreads();        // g2 and its planes are consumed by every wave
    ASTAMP(29);
    reduce_partials8(dxa, Gs, wave, r, q);
    ASTAMP(30);
    // dy1 = dX1 (four partial sums, fixed order) + d_res2: what reaches LayerNorm1's output; dense rows for the attention-side kernel
    {
        f32x4* dst = reinterpret_cast<f32x4*>(p.dy1 + tok0 * FD);
        for (int i = tid; i < S * (FD / 4); i += CT) {
            const int o = (i >> 5) * LDX + ((i & 31) << 2);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Gs + o), a1 = *reinterpret_cast<const f32x4*>(B2 + o);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(B3 + o), a3 = *reinterpret_cast<const f32x4*>(B4 + o);
            dst[i] = (((a0 + a1) + a2) + a3) + *reinterpret_cast<const f32x4*>(B1 + o);
        }
    }
    ASTAMP(31);
}

bool ffn_cut_supported(int d_ff) { return d_ff % 256 == 0 && d_ff >= 256; }
size_t ffn_cut_lds_bytes() { return (size_t)(5 * CBLK + CP_FLOATS) * sizeof(float); }

template <int CM>
static int launch_cut_fwd(const FusedFwdParams& p, int l, hipStream_t st) {
    const size_t lds = ffn_cut_lds_bytes();
    static bool attr_set = false;
    if (!attr_set) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fwd_kernel<CM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    timing_begin(TIMER_FFN_FWD, st);
    hipLaunchKernelGGL((ffn_fwd_kernel<CM>), dim3(p.B), dim3(CT), lds, st, p, l);
    timing_end(TIMER_FFN_FWD, st);
    EGX_LAUNCH_CHECK();
    return 0;
}
template <int CM>
static int launch_cut_bwd(const FusedBwdParams& p, int l, hipStream_t st) {
    const size_t lds = ffn_cut_lds_bytes();
    static bool attr_set = false;
    if (!attr_set) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel<CM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    timing_begin(TIMER_FFN_BWD, st);
    hipLaunchKernelGGL((ffn_bwd_kernel<CM>), dim3(p.B), dim3(CT), lds, st, p, l);
    timing_end(TIMER_FFN_BWD, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

int ffn_cut_forward(const FusedFwdParams& p, int l, int compute, hipStream_t st) {
    EGX_CHECK(ffn_cut_supported(p.d_ff) && p.S <= CSP && l >= 0 && l < p.n_layers, "ffn_cut_forward: d_ff=%d S=%d layer %d", p.d_ff, p.S, l);
    EGX_CHECK(p.x1f_out && p.xin_out && p.saved_res && p.relu_bits && p.hid_out, "ffn_cut_forward: missing buffers");
    return compute == CM_BF16 ? launch_cut_fwd<CM_BF16>(p, l, st) : compute == CM_SPLIT ? launch_cut_fwd<CM_SPLIT>(p, l, st) : launch_cut_fwd<CM_F32>(p, l, st);
}
int ffn_cut_backward(const FusedBwdParams& p, int l, int compute, hipStream_t st) {
    EGX_CHECK(ffn_cut_supported(p.d_ff) && p.S <= CSP && l >= 0 && l < p.n_layers, "ffn_cut_backward: d_ff=%d S=%d layer %d", p.d_ff, p.S, l);
    EGX_CHECK(p.dy1 && p.saved_res && p.relu_bits && p.dhid_out && p.partials && (l + 1 == p.n_layers || p.dxin), "ffn_cut_backward: missing buffers");
    return compute == CM_BF16 ? launch_cut_bwd<CM_BF16>(p, l, st) : compute == CM_SPLIT ? launch_cut_bwd<CM_SPLIT>(p, l, st) : launch_cut_bwd<CM_F32>(p, l, st);
}

}  // namespace egx
