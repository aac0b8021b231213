// Device-side building blocks shared by the fused forward (fused.hip) and backward (fused_bwd.hip) kernels:
// the MFMA operand convention, fragment loads, packed-weight access and the row-parallel LayerNorm.
#pragma once
#include "common.h"
#include "fused.h"       // TouchList

namespace egx {

constexpr int FD = 128;        // d_model
constexpr int FH = 4;          // heads
constexpr int FDH = 32;        // head dim
constexpr int LDX = FD + 4;    // token-major LDS row stride (floats)
constexpr int LDV = 64 + 4;    // V^T row stride: keys padded to 64

// ---- compute modes ------------------------------------------------------------------------------
// CM_F32   exact v_mfma_f32_16x16x4_f32 (8 per 32-wide K-block, 32 cycles each).
// CM_BF16  operands rounded to bf16, one v_mfma_f32_16x16x32_bf16 per K-block (16 cycles), fp32 accumulate.
// CM_SPLIT fp32 operands split EXACTLY into three bf16 parts (x = x1 + x2 + x3, 3 x 8 significant bits) and the six products
//          a1b1 a1b2 a2b1 a1b3 a2b2 a3b1 accumulated by six bf16 MFMAs into the fp32 accumulator. The dropped products
//          (a2b3, a3b2, a3b3) are below 2^-23 |a||b|, the size of fp32's own product rounding: fp32-grade results at
//          6 x 16 = 96 matrix-pipe cycles per K-block instead of 8 x 32 = 256.
enum { CM_F32 = 0, CM_BF16 = 1, CM_SPLIT = 2 };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- operand fragments --------------------------------------------------------------------------
template <int CM> struct Frag { float v[8]; };
template <> struct Frag<CM_BF16> { bf16x8 v; };
template <> struct Frag<CM_SPLIT> { bf16x8 p[3]; };      // high, middle, low part

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return pack_bf16x2(a, b); }

// (x, y) -> three packed bf16 pairs with x = x1 + x2 + x3 exactly (round-to-nearest at each level; the last residual has
// at most 8 significant bits, so its conversion is exact)
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& h, uint32_t& m, uint32_t& l) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    h = pack_bf16(x, y);
    f32x2 v = {x, y};
    f32x2 hv = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    f32x2 r = v - hv;
    m = pack_bf16(r[0], r[1]);
    f32x2 mv = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    f32x2 r2 = r - mv;
    l = pack_bf16(r2[0], r2[1]);
}

template <int CM>
__device__ __forceinline__ Frag<CM> make_frag(float4 a, float4 b) {
    Frag<CM> f;
    if constexpr (CM == CM_BF16) {
        u32x4 u = {pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w)};
        f.v = __builtin_bit_cast(bf16x8, u);
    } else if constexpr (CM == CM_SPLIT) {
        uint32_t h[4], m[4], l[4];
        split_pair(a.x, a.y, h[0], m[0], l[0]);
        split_pair(a.z, a.w, h[1], m[1], l[1]);
        split_pair(b.x, b.y, h[2], m[2], l[2]);
        split_pair(b.z, b.w, h[3], m[3], l[3]);
        f.p[0] = __builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]});
        f.p[1] = __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]});
        f.p[2] = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
    } else {
        f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
        f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    }
    return f;
}

// fragment of one row/column for the K-block that starts at p (fp32 memory, 16-byte aligned)
template <int CM>
__device__ __forceinline__ Frag<CM> load_frag(const float* p, int q) {
    float4 a = *reinterpret_cast<const float4*>(p + 4 * q);
    float4 b = *reinterpret_cast<const float4*>(p + 16 + 4 * q);
    return make_frag<CM>(a, b);
}

// fragment over a head dimension of DH (32: a full K-block; 16: the K-block's second half is zero — the shipped PNR / OSCC
// translators run 8 heads of 16, HOI/models/pnr/video_model_transfer_3task.py:231)
template <int CM, int DH>
__device__ __forceinline__ Frag<CM> load_head_frag(const float* p, int q) {
    if constexpr (DH == 32) return load_frag<CM>(p, q);
    else return make_frag<CM>(*reinterpret_cast<const float4*>(p + 4 * q), make_float4(0, 0, 0, 0));
}

template <int CM>
__device__ __forceinline__ Frag<CM> zero_frag() {
    return make_frag<CM>(make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0));
}

// raw (unconverted) fragment data: issued early so that many loads are in flight at once
struct Raw { float4 a, b; };
__device__ __forceinline__ Raw load_raw(const float* p, int q) {
    Raw x;
    x.a = *reinterpret_cast<const float4*>(p + 4 * q);
    x.b = *reinterpret_cast<const float4*>(p + 16 + 4 * q);
    return x;
}
template <int CM>
__device__ __forceinline__ Frag<CM> to_frag(const Raw& x) { return make_frag<CM>(x.a, x.b); }
// Pins a prefetched fragment at this program point: the loads that produce it must have been issued above (hipcc
// otherwise sinks every load down to its convert/MFMA and waits on each one individually), and the single
// s_waitcnt for the whole batch lands here.
__device__ __forceinline__ void pin(Raw& x) {
    asm volatile("" : "+v"(x.a.x), "+v"(x.a.y), "+v"(x.a.z), "+v"(x.a.w), "+v"(x.b.x), "+v"(x.b.y), "+v"(x.b.z), "+v"(x.b.w));
}

// ---- packed weights ---------------------------------------------------------------------------------
// A row-major weight read in MFMA-fragment shape makes every lane of a 16-lane group touch a different cache
// line (16 rows x 64 B per instruction): the texture addresser then delivers ~16 B/clk/CU and the whole kernel
// runs at that rate. pack_weights_kernel therefore rewrites each weight ONCE per step into fragment order:
//   block (tile t of 16 rows, K-block kb of 32) holds the 64 lanes' operands contiguously, so a fragment load is
//   one (bf16: uint4) or two (fp32: float4 planes) perfectly coalesced 1 KiB wave accesses.
//   fp32: float4 plane[half][lane] = W[t*16 + r][kb*32 + half*16 + 4q .. +3]
//   bf16: uint4  [lane]            = bf16 of the same 8 values (half 0 first)
//   split: uint4 [part][lane]           = the three bf16 parts of the same 8 values (pre-split once per step)
template <int CM> struct WRaw { float4 a, b; };
// (128-bit vector members: a fragment stays ONE register tuple from the load to the MFMA operand)
template <> struct WRaw<CM_BF16> { u32x4 v; };
template <> struct WRaw<CM_SPLIT> { u32x4 p[3]; };

template <int CM>
__device__ __forceinline__ WRaw<CM> load_w(const void* packed, int tile, int nkb, int kb, int lane) {
    WRaw<CM> x;
    size_t blk = (size_t)tile * nkb + kb;
    if constexpr (CM == CM_BF16) {
        x.v = reinterpret_cast<const u32x4*>(packed)[blk * 64 + lane];
    } else if constexpr (CM == CM_SPLIT) {
        const u32x4* pl = reinterpret_cast<const u32x4*>(packed) + blk * 192;
        x.p[0] = pl[lane];
        x.p[1] = pl[64 + lane];
        x.p[2] = pl[128 + lane];
    } else {
        const float4* pl = reinterpret_cast<const float4*>(packed) + blk * 128;
        x.a = pl[lane];
        x.b = pl[64 + lane];
    }
    return x;
}
template <int CM>
__device__ __forceinline__ Frag<CM> w_frag(const WRaw<CM>& x) {
    if constexpr (CM == CM_BF16) {
        Frag<CM_BF16> f;
        f.v = __builtin_bit_cast(bf16x8, x.v);
        return f;
    } else if constexpr (CM == CM_SPLIT) {
        Frag<CM_SPLIT> f;
#pragma unroll
        for (int i = 0; i < 3; ++i) f.p[i] = __builtin_bit_cast(bf16x8, x.p[i]);
        return f;
    } else {
        return make_frag<CM_F32>(x.a, x.b);
    }
}
__device__ __forceinline__ void pin(WRaw<CM_F32>& x) {
    asm volatile("" : "+v"(x.a.x), "+v"(x.a.y), "+v"(x.a.z), "+v"(x.a.w), "+v"(x.b.x), "+v"(x.b.y), "+v"(x.b.z), "+v"(x.b.w));
}
__device__ __forceinline__ void pin(WRaw<CM_BF16>& x) {      // one 128-bit operand: pinning the four dwords separately made
    asm volatile("" : "+v"(x.v));                             // the compiler re-pack them with v_mov before every MFMA
}
__device__ __forceinline__ void pin(WRaw<CM_SPLIT>& x) {
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(x.p[i]));
}
template <class T, int N>
__device__ __forceinline__ void pin_all(T (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) pin(x[i]);
}
template <class T, int N, int M>
__device__ __forceinline__ void pin_all(T (&x)[N][M]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) pin(x[i][j]);
}


// ---- pre-split operand planes (CM_SPLIT) ------------------------------------------------------------------------------
// A token-major fp32 LDS block that is the B operand of many MFMAs (the FFN input x1: 64 hidden blocks) is split ONCE
// into three bf16 planes [part][rows][LDXH]; a fragment is then three ds_read2_b64 (hipcc merges the low / high half of a
// K-block, 32 B apart) and no VALU work. ds_read2_b64 is banked over 32 banks in groups of 16 CONSECUTIVE lanes = the 16 rows
// of a fragment at one lane group q: the row stride must be 2 (mod 4) dwords for them to cover all 32 banks. Round 2 used
// 272 B (68 dwords = 4 mod 32, chosen for plain ds_read_b64's 64-bank rule): rows r and r + 8 collided — the 6.0 M / 7.6 M
// SQ_LDS_BANK_CONFLICT cycles of the split-mode clip kernels. 264 B = 66 dwords: lane r reads banks 2r, 2r + 1. Rows are
// then only 8-byte aligned: the planes are written and copied out in 8-byte pieces.
constexpr int LDXH = FD + 4;
__device__ __forceinline__ void split32(const float (&v)[32], uint32_t (&h)[16], uint32_t (&m)[16], uint32_t (&l)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) split_pair(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
}
// 32 consecutive elements of one row, already split: d = &plane0[row][c0] (8-byte aligned)
__device__ __forceinline__ void store_parts32(unsigned short* d, size_t plane_stride, const uint32_t (&h)[16], const uint32_t (&m)[16],
                                              const uint32_t (&l)[16]) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        *reinterpret_cast<uint2*>(d + 4 * g) = make_uint2(h[2 * g], h[2 * g + 1]);
        *reinterpret_cast<uint2*>(d + plane_stride + 4 * g) = make_uint2(m[2 * g], m[2 * g + 1]);
        *reinterpret_cast<uint2*>(d + 2 * plane_stride + 4 * g) = make_uint2(l[2 * g], l[2 * g + 1]);
    }
}
__device__ __forceinline__ void store_split32(unsigned short* planes, int plane_stride, int row, int c0, const float (&v)[32]) {
    uint32_t h[16], m[16], l[16];
    split32(v, h, m, l);
    store_parts32(planes + row * LDXH + c0, (size_t)plane_stride, h, m, l);
}
__device__ __forceinline__ Frag<CM_SPLIT> load_split_frag(const unsigned short* planes, int plane_stride, int row, int k0, int q) {
    Frag<CM_SPLIT> f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned short* b = planes + i * plane_stride + row * LDXH + k0 + 4 * q;
        uint2 lo = *reinterpret_cast<const uint2*>(b);
        uint2 hi = *reinterpret_cast<const uint2*>(b + 16);
        f.p[i] = __builtin_bit_cast(bf16x8, (u32x4){lo.x, lo.y, hi.x, hi.y});
    }
    return f;
}

// two consecutive 16-row C tiles of a feature-major result -> B operand of the next GEMM (K = those 32 rows)
template <int CM>
__device__ __forceinline__ Frag<CM> chain_frag(const f32x4& t0, const f32x4& t1) {
    return make_frag<CM>(make_float4(t0[0], t0[1], t0[2], t0[3]), make_float4(t1[0], t1[1], t1[2], t1[3]));
}

template <int CM>
__device__ __forceinline__ void mma(f32x4& acc, const Frag<CM>& a, const Frag<CM>& b) {
    if constexpr (CM == CM_BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
    } else if constexpr (CM == CM_SPLIT) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], acc, 0, 0, 0);     // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
    }
}

// ---- hidden-activation tiles for ffn_dw ----------------------------------------------------------------------
// The clip-parallel kernels hold the FFN hidden activation (and its gradient) as C tiles with the hidden unit on the
// registers (row 4q+e) and the token on the lane (col r). The weight-gradient kernel needs them as MFMA operands
// with the TOKEN along K, i.e. lane (r' = hidden % 16, q') holding tokens 4q'..4q'+3. A 4x4 transpose inside each
// lane quad (two DPP butterfly steps) converts one into the other, after which the whole 16x16 tile leaves as ONE
// dense store (1 KB fp32 / 512 B bf16): tile[lane' = hid%16 + 16*((tok%16)>>2)][tok&3].
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// afterwards v[j] of quad-lane l holds what v[l] of quad-lane j held. Every register travels through the DPP network
// and the per-lane choice is made afterwards: 4 + 4 selects whose DPP operand the compiler folds into
// v_cndmask_b32_dpp, instead of the select / move / select chains it builds when only the "needed" registers move.
__device__ __forceinline__ void quad_transpose(float (&v)[4], int lane) {
    const bool h2 = (lane & 2) != 0, h1 = (lane & 1) != 0;
    const float t0 = dpp_quad<0x4E>(v[0]), t1 = dpp_quad<0x4E>(v[1]);      // quad_perm [2,3,0,1]: partner lane ^ 2
    const float t2 = dpp_quad<0x4E>(v[2]), t3 = dpp_quad<0x4E>(v[3]);
    const float w0 = h2 ? t2 : v[0], w1 = h2 ? t3 : v[1], w2 = h2 ? v[2] : t0, w3 = h2 ? v[3] : t1;
    const float u0 = dpp_quad<0xB1>(w0), u1 = dpp_quad<0xB1>(w1);          // quad_perm [1,0,3,2]: partner lane ^ 1
    const float u2 = dpp_quad<0xB1>(w2), u3 = dpp_quad<0xB1>(w3);
    v[0] = h1 ? u1 : w0; v[1] = h1 ? w1 : u0; v[2] = h1 ? u3 : w2; v[3] = h1 ? w3 : u2;
}
constexpr int HTILE_ELEMS = 256;   // one (16 tokens x 16 hidden units) tile
// `tile` = start of the (token tile, hidden tile) block; tokens >= n_valid (within this 16-token tile) are stored as 0
// CM_BF16: the tile leaves in ACCUMULATOR layout ([lane = token + 16 * (hidden / 4)][hidden % 4]: no transpose, no DPP
// traffic in the FFN loops); the weight-gradient kernel turns it into token-along-K fragments with the hardware transposed
// LDS read (ds_read_b64_tr_b16) it uses for its other operand anyway (bf16 step 0.259 -> 0.252 ms). fp32 MFMA operands
// cannot take that route (16-bit transposes only), and in CM_SPLIT the three-part staging it needs costs the
// weight-gradient kernel more (+10 us) than the FFN loops save (-4 us): both keep the quad-transposed tile.
template <int CM>
__device__ __forceinline__ void store_hid_tile(void* tile, const f32x4& c, int lane, int n_valid) {
    float v[4] = {c[0], c[1], c[2], c[3]};
    if constexpr (CM == CM_BF16) {
        if (__builtin_amdgcn_readfirstlane(n_valid) < 16) {
            asm volatile("" ::: "memory");
            const bool ok = (lane & 15) < n_valid;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0.f;
        }
        reinterpret_cast<uint2*>(tile)[lane] = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
        return;
    }
    quad_transpose(v, lane);
    const int r = lane & 15, q = lane >> 4;
    const int t0 = 4 * (r >> 2);
    if (__builtin_amdgcn_readfirstlane(n_valid) < 16) {      // wave-uniform: only a clip's last (partial) tile pays for it
        asm volatile("" ::: "memory");                         // keep it a branch (hipcc otherwise if-converts it into 8 selects per tile)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (t0 + j < n_valid) ? v[j] : 0.f;
    }
    const int dst = 4 * q + (r & 3) + 16 * (r >> 2);
    if constexpr (CM == CM_BF16) {
        reinterpret_cast<uint2*>(tile)[dst] = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    } else {
        reinterpret_cast<float4*>(tile)[dst] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
// bf16: the two accumulator-layout tiles (hidden tiles 2 hb, 2 hb + 1; 512 B each, adjacent) of one token tile, taken from
// the packed operand words of the chained GEMM (u = {tile0 lo, tile0 hi, tile1 lo, tile1 hi}): no second conversion
__device__ __forceinline__ void store_hid_tile_bf16(void* tile0, u32x4 u, int lane, int n_valid) {
    if (__builtin_amdgcn_readfirstlane(n_valid) < 16) {      // wave-uniform: only a clip's last (partial) token tile pays for it
        asm volatile("" ::: "memory");
        const bool ok = (lane & 15) < n_valid;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = ok ? u[j] : 0u;
    }
    reinterpret_cast<uint2*>(tile0)[lane] = make_uint2(u[0], u[1]);
    reinterpret_cast<uint2*>(tile0)[64 + lane] = make_uint2(u[2], u[3]);
}
// A/B operand of one K-block of 32 tokens = two consecutive 16-token tiles of one hidden tile
template <int CM>
__device__ __forceinline__ Frag<CM> load_hid_frag(const void* tile_a, const void* tile_b, int lane) {
    if constexpr (CM == CM_BF16) {
        Frag<CM> f;
        uint2 a = reinterpret_cast<const uint2*>(tile_a)[lane];
        uint2 b = reinterpret_cast<const uint2*>(tile_b)[lane];
        u32x4 u = {a.x, a.y, b.x, b.y};
        f.v = __builtin_bit_cast(bf16x8, u);
        return f;
    } else {
        float4 a = reinterpret_cast<const float4*>(tile_a)[lane];
        float4 b = reinterpret_cast<const float4*>(tile_b)[lane];
        return make_frag<CM>(a, b);
    }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// TouchList (fused.h): this workgroup's share of the lines, ONE load per thread and stream, every load unconditional (unused streams repeat
// stream 0; threads beyond the share re-read its last line; a share larger than the workgroup is cut short: this is a hint). The caller keeps
// the values alive up to a point where every younger load of its prologue has been waited for anyway (touch_sink): no extra wait.
struct Touched { uint32_t v[TOUCH_MAX]; };
template <int NTHREADS>
__device__ __forceinline__ Touched touch_lines(const TouchList& tl, int block, int nblocks, int tid) {
    Touched t;
#pragma unroll
    for (int k = 0; k < TOUCH_MAX; ++k) {
        const int kk = k < tl.n ? k : 0;
        const unsigned nl = tl.lines[kk], per = (nl + (unsigned)nblocks - 1) / (unsigned)nblocks;
        unsigned i = (unsigned)block * per + (unsigned)tid;
        i = (i < nl && (unsigned)tid < per) ? i : nl - 1;
        t.v[k] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tl.base[kk]) + ((size_t)i << 7));
    }
    return t;
}
__device__ __forceinline__ void touch_sink(const Touched& t) {
#pragma unroll
    for (int k = 0; k < TOUCH_MAX; ++k) asm volatile("" :: "v"(t.v[k]));
}

__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- weighted cross entropy fused into the pooled head (egx_ce, round 6) --------------------------------------------------------
// loss = sum_i w[y_i] nll_i / sum_i w[y_i] (nn.CrossEntropyLoss(weight), HHI/tasks/ttm/video_task_2loader.py:21-22,34; the arithmetic of
// weighted_ce_kernel, train.hip). The normaliser depends on the labels only, so every clip's workgroup sums it itself (same order in every
// workgroup: identical bits) and its loss term and d loss / d logits are final in the launch that produced the logits.
struct FusedCe {
    const int64_t* target;      // (B) or null = no fused loss
    const float* class_weight;  // (n_out) or null
    float* loss;                // scalar: zero when the launch starts, every clip adds its term
    float* d_logits;            // (B, n_out)
    int B;                      // clips of the BATCH (the normaliser's range)
    unsigned* ticket;           // null, or {arrival counter, accumulator}: see FusedFwdParams::ce_ticket
};
// all NTHREADS threads: per-wave partial sums of the normaliser -> red[wave]; the caller puts a barrier behind it. `y` / `wy` are requested by
// ce_request() ahead of a phase that hides the two dependent round trips (label, then its class weight).
struct CeReq { int64_t y[2]; float w[2]; };
template <int NTHREADS>
__device__ __forceinline__ void ce_request_labels(const FusedCe& ce, int tid, CeReq& rq) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int i = tid + k * NTHREADS; rq.y[k] = ce.target[i < ce.B ? i : ce.B - 1]; }
}
__device__ __forceinline__ void ce_request_weights(const FusedCe& ce, int n_out, CeReq& rq) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t y = rq.y[k];
        const int yc = (y >= 0 && y < n_out) ? (int)y : 0;
        rq.w[k] = ce.class_weight ? ce.class_weight[yc] : 1.f;
    }
}
template <int NTHREADS>
__device__ __forceinline__ void ce_weight_partials(const FusedCe& ce, int n_out, int tid, const CeReq& rq, float* red) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * NTHREADS;
        s += (i < ce.B && rq.y[k] >= 0 && rq.y[k] < n_out) ? rq.w[k] : 0.f;
    }
    for (int i = tid + 2 * NTHREADS; i < ce.B; i += NTHREADS) {     // batches beyond 2 x NTHREADS clips: plain dependent loads
        const int64_t y = ce.target[i];
        if (y >= 0 && y < n_out) s += ce.class_weight ? ce.class_weight[y] : 1.f;
    }
    s = wsum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
}
// wave 0 of the clip's workgroup: lane o < n_out holds logit o in `z`. Writes d_logits of the clip and adds the clip's loss term.
template <int NTHREADS>
__device__ __forceinline__ void ce_clip(const FusedCe& ce, int n_out, int clip, int lane, float z, const float* red, bool add_loss) {
    float wtot = 0.f;
#pragma unroll
    for (int w = 0; w < NTHREADS / 64; ++w) wtot += red[w];
    const int64_t y = ce.target[clip];
    const bool valid = y >= 0 && y < n_out;
    const float wy = valid ? (ce.class_weight ? ce.class_weight[y] : 1.f) : 0.f;
    const float zl = lane < n_out ? z : -INFINITY;
    const float m = wmax(zl);
    const float e = lane < n_out ? __expf(zl - m) : 0.f;
    const float s = wsum(e);
    const float zy = __shfl(zl, valid ? (int)y : 0, 64);
    const float k = wy / wtot;
    if (lane < n_out) ce.d_logits[(size_t)clip * n_out + lane] = valid ? k * (e / s - (lane == (int)y ? 1.f : 0.f)) : 0.f;
    if (lane == 0 && add_loss) {
        float term = valid ? k * (m + __logf(s) - zy) : 0.f;
        if (wtot == 0.f && clip == 0) term = __builtin_nanf("");    // no valid label in the batch: 0 / 0 like the reference
        if (!ce.ticket) {
            atomicAdd(ce.loss, term);
        } else {
            float* acc = reinterpret_cast<float*>(ce.ticket + 1);
            atomicAdd(acc, term);
            __threadfence();
            if (atomicInc(ce.ticket, (unsigned)ce.B - 1u) == (unsigned)ce.B - 1u) {     // last clip of the launch (the counter wraps to zero)
                __threadfence();
                *ce.loss = atomicExch(acc, 0.f);
            }
        }
    }
}


// Row-parallel LayerNorm over a token-major LDS block: 4 adjacent lanes own one row (32 features each), so up
// to 64 rows are normalised in one pass with quad (DPP) reductions. `fn(row, c0, x[32] pre-LN, y[32] post-LN)`
// consumes the result (global saves, embeddings, dropout, write-back).
__device__ __forceinline__ float quad_sum4(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}
template <class Fn, class Hook>
__device__ __forceinline__ void ln_rows(const float* buf, int S, const float* __restrict__ w, const float* __restrict__ b,
                                        float eps, Fn&& fn, Hook&& after_w) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int c0 = part * 32;
    // the lane's weights and biases are requested first and unconditionally, `after_w()` runs right behind the request and
    // outside any branch (see ln_bwd_rows): stores and prefetches issued there are younger than the weights
    f32x4 wq[8], bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wq[j] = *reinterpret_cast<const f32x4*>(w + c0 + 4 * j);
        bq[j] = *reinterpret_cast<const f32x4*>(b + c0 + 4 * j);
    }
    after_w();
    if (row < S) {
        float x[32], y[32];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 v = *reinterpret_cast<const float4*>(buf + row * LDX + c0 + 4 * j);
            x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
            s += (v.x + v.y) + (v.z + v.w);
        }
        float mean = quad_sum4(s) * (1.f / FD);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { float t = x[j] - mean; ss += t * t; }
        float rstd = rsqrtf(quad_sum4(ss) * (1.f / FD) + eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * j + e] = (x[4 * j + e] - mean) * rstd * wq[j][e] + bq[j][e];
        }
        fn(row, c0, x, y);
    } else {
        // keep the quad shuffles convergent for partially filled waves
        (void)quad_sum4(0.f);
        (void)quad_sum4(0.f);
    }
}
template <class Fn>
__device__ __forceinline__ void ln_rows(const float* buf, int S, const float* __restrict__ w, const float* __restrict__ b,
                                        float eps, Fn&& fn) {
    ln_rows(buf, S, w, b, eps, fn, [] {});
}
// The same with the weight and bias vectors staged in LDS (round 6). From global memory every 4-lane row group requests the same 2 x 512
// bytes: 16 load instructions per thread, 64 KB per LayerNorm through the CU's one address path for 1 KB of parameters - 3 - 4k of a
// phase's 11k cycles (stamps: profiles/r06_attn_stamps.txt); here they are ds_read broadcasts taken where they are used. `hook()` runs
// first (the saves / prefetches the global variant issued behind its weight request).
template <class Fn, class Hook>
__device__ __forceinline__ void ln_rows_lds(const float* buf, int S, const float* wl, const float* bl, float eps, Fn&& fn, Hook&& hook) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int c0 = part * 32;
    hook();
    if (row < S) {
        float x[32], y[32];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 v = *reinterpret_cast<const float4*>(buf + row * LDX + c0 + 4 * j);
            x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
            s += (v.x + v.y) + (v.z + v.w);
        }
        float mean = quad_sum4(s) * (1.f / FD);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { float t = x[j] - mean; ss += t * t; }
        float rstd = rsqrtf(quad_sum4(ss) * (1.f / FD) + eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 wq = *reinterpret_cast<const float4*>(wl + c0 + 4 * j), bq = *reinterpret_cast<const float4*>(bl + c0 + 4 * j);
            y[4 * j + 0] = (x[4 * j + 0] - mean) * rstd * wq.x + bq.x;
            y[4 * j + 1] = (x[4 * j + 1] - mean) * rstd * wq.y + bq.y;
            y[4 * j + 2] = (x[4 * j + 2] - mean) * rstd * wq.z + bq.z;
            y[4 * j + 3] = (x[4 * j + 3] - mean) * rstd * wq.w + bq.w;
        }
        fn(row, c0, x, y);
    } else {
        (void)quad_sum4(0.f);
        (void)quad_sum4(0.f);
    }
}
__device__ __forceinline__ void store32(float* dst, const float (&v)[32]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(dst + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

// rows [0, S) of a token-major fp32 LDS block -> dense global rows, 16 bytes per lane in lane order. (store32 from the
// LayerNorm lanes sends every 128-byte line to the L2 as eight partial writes, and the next vmcnt wait of the wave
// sits behind them: 12 such stores per lane cost the forward kernel 10 us, the same bytes in lane order 2.5 us.)
__device__ __forceinline__ void store_block(float* dst, const float* lds_src, int S) {
    for (int i = threadIdx.x; i < S * (FD / 4); i += 256) {
        int row = i >> 5, c4 = i & 31;
        *reinterpret_cast<float4*>(dst + (size_t)i * 4) = *reinterpret_cast<const float4*>(lds_src + row * LDX + c4 * 4);
    }
}

// rows [0, 48) of a token-major fp32 LDS block -> one dense bf16 plane [48][128] in HBM (rows >= S as zeros): the x1 / g2
// operand images of the bf16 weight-gradient kernel, 16 bytes per lane in lane order
__device__ __forceinline__ void store_block_bf16(unsigned short* dst, const float* lds_src, int S) {
    for (int i = threadIdx.x; i < 48 * (FD / 8); i += 256) {
        int row = i >> 4, c8 = i & 15;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < S) {
            float4 a = *reinterpret_cast<const float4*>(lds_src + row * LDX + c8 * 8);
            float4 b = *reinterpret_cast<const float4*>(lds_src + row * LDX + c8 * 8 + 4);
            v = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
        }
        *reinterpret_cast<uint4*>(dst + (size_t)i * 8) = v;
    }
}

__device__ __forceinline__ void load32(const float* src, float (&v)[32]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float4 a = *reinterpret_cast<const float4*>(src + 4 * j);
        v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
    }
}

// Row-parallel LayerNorm backward (4 lanes per row). `get(row, c0, dy[32], x[32])` supplies the upstream gradient
// and the pre-LN input of the lane's 32 features; `put(row, c0, dy, dx, dyxhat)` consumes the input gradient and
// the per-row contribution to d(gamma) (= dy * xhat).
// The lane's 32 weights are requested first and unconditionally; `after_w()` runs right behind that request, outside any branch:
// global loads issued there (prefetches for a later phase) are YOUNGER than the weights, so the wait for the weights leaves
// them in flight (a load issued before them, or under the row < S branch, would be waited for here).
#ifndef LNB_STAMP
#define LNB_STAMP(base, k) do { } while (0)
#endif
template <class Get, class Put, class Hook>
__device__ __forceinline__ void ln_bwd_rows(int S, const float* __restrict__ w, float eps, Get&& get, Put&& put, Hook&& after_w, int stamp_base = -1) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int c0 = part * 32;
    f32x4 wq[8];
    LNB_STAMP(stamp_base, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) wq[j] = *reinterpret_cast<const f32x4*>(w + c0 + 4 * j);
    after_w();
    LNB_STAMP(stamp_base, 1);
    if (row < S) {
        float dy[32], x[32], dx[32];
        get(row, c0, dy, x);
        LNB_STAMP(stamp_base, 2);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) s += x[j];
        float mean = quad_sum4(s) * (1.f / FD);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { x[j] -= mean; ss += x[j] * x[j]; }
        float rstd = rsqrtf(quad_sum4(ss) * (1.f / FD) + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float wj[4] = {wq[j][0], wq[j][1], wq[j][2], wq[j][3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int i = 4 * j + e;
                x[i] *= rstd;                 // xhat
                float g = dy[i] * wj[e];
                dx[i] = g;
                s1 += g;
                s2 += g * x[i];
            }
        }
        s1 = quad_sum4(s1) * (1.f / FD);
        s2 = quad_sum4(s2) * (1.f / FD);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            dx[i] = rstd * (dx[i] - s1 - x[i] * s2);
            x[i] *= dy[i];                    // dy * xhat
        }
        LNB_STAMP(stamp_base, 3);
        put(row, c0, dy, dx, x);
        LNB_STAMP(stamp_base, 4);
    } else {
        (void)quad_sum4(0.f); (void)quad_sum4(0.f); (void)quad_sum4(0.f); (void)quad_sum4(0.f);
    }
}

template <class Get, class Put>
__device__ __forceinline__ void ln_bwd_rows(int S, const float* __restrict__ w, float eps, Get&& get, Put&& put) {
    ln_bwd_rows(S, w, eps, get, put, [] {});
}
// The same with the weight vector staged in LDS (round 6, see ln_rows_lds): `hook()` runs first.
template <class Get, class Put, class Hook>
__device__ __forceinline__ void ln_bwd_rows_lds(int S, const float* wl, float eps, Get&& get, Put&& put, Hook&& hook, int stamp_base = -1) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int c0 = part * 32;
    LNB_STAMP(stamp_base, 0);
    hook();
    LNB_STAMP(stamp_base, 1);
    if (row < S) {
        float dy[32], x[32], dx[32];
        get(row, c0, dy, x);
        LNB_STAMP(stamp_base, 2);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) s += x[j];
        float mean = quad_sum4(s) * (1.f / FD);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { x[j] -= mean; ss += x[j] * x[j]; }
        float rstd = rsqrtf(quad_sum4(ss) * (1.f / FD) + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 wv = *reinterpret_cast<const float4*>(wl + c0 + 4 * j);
            float wj[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int i = 4 * j + e;
                x[i] *= rstd;                 // xhat
                float g = dy[i] * wj[e];
                dx[i] = g;
                s1 += g;
                s2 += g * x[i];
            }
        }
        s1 = quad_sum4(s1) * (1.f / FD);
        s2 = quad_sum4(s2) * (1.f / FD);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            dx[i] = rstd * (dx[i] - s1 - x[i] * s2);
            x[i] *= dy[i];                    // dy * xhat
        }
        LNB_STAMP(stamp_base, 3);
        put(row, c0, dy, dx, x);
        LNB_STAMP(stamp_base, 4);
    } else {
        (void)quad_sum4(0.f); (void)quad_sum4(0.f); (void)quad_sum4(0.f); (void)quad_sum4(0.f);
    }
}

// Column sums of a token-major LDS block over rows [r0, r1): four independent accumulators so the LDS reads
// pipeline instead of forming one dependent latency chain.
__device__ __forceinline__ float colsum_lds(const float* buf, int r0, int r1, int c) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int row = r0;
    for (; row + 3 < r1; row += 4) {
        s0 += buf[(row + 0) * LDX + c];
        s1 += buf[(row + 1) * LDX + c];
        s2 += buf[(row + 2) * LDX + c];
        s3 += buf[(row + 3) * LDX + c];
    }
    for (; row < r1; ++row) s0 += buf[row * LDX + c];
    return (s0 + s1) + (s2 + s3);
}

// Feature-major GEMM against a token-major LDS operand: acc[i][t] += pack(tile0 + i, kb0 + kb) * B(rows t*16.., cols kb*32..).
template <int CM, int NTI, int NT, int NKB>
__device__ __forceinline__ void gemm_pack_lds(f32x4 (&acc)[NTI][NT], const void* pack, int tile0, int nkb_total, int kb0,
                                              const float* ldsB, int r, int q, int lane) {
    WRaw<CM> wv[NTI][NKB];
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) wv[i][kb] = load_w<CM>(pack, tile0 + i, nkb_total, kb0 + kb, lane);
    __builtin_amdgcn_sched_barrier(0);
    pin_all(wv);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        Frag<CM> b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = load_frag<CM>(ldsB + (t * 16 + r) * LDX + kb * 32, q);
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            Frag<CM> a = w_frag<CM>(wv[i][kb]);
#pragma unroll
            for (int t = 0; t < NT; ++t) mma<CM>(acc[i][t], a, b[t]);
        }
    }
}

// The same GEMM with its weight fetch split off: pack_issue() early (before the LayerNorm / column-sum / barrier work that
// precedes the GEMM phase, so the L2 round trip of the fragments is hidden under it), gemm_packed() where the operands are
// ready. One wave per SIMD: nothing else hides that latency.
template <int CM, int NTI, int NKB> struct PackW { WRaw<CM> v[NTI][NKB]; };
template <int CM, int NTI, int NKB>
__device__ __forceinline__ void pack_issue(PackW<CM, NTI, NKB>& w, const void* pack, int tile0, int nkb_total, int kb0) {
    int ln = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) w.v[i][kb] = load_w<CM>(pack, tile0 + i, nkb_total, kb0 + kb, ln);
    __builtin_amdgcn_sched_barrier(0);
}
template <int CM, int NTI, int NT, int NKB>
__device__ __forceinline__ void gemm_packed(f32x4 (&acc)[NTI][NT], PackW<CM, NTI, NKB>& w, const float* ldsB, int r, int q) {
    pin_all(w.v);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        Frag<CM> b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = load_frag<CM>(ldsB + (t * 16 + r) * LDX + kb * 32, q);
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            Frag<CM> a = w_frag<CM>(w.v[i][kb]);
#pragma unroll
            for (int t = 0; t < NT; ++t) mma<CM>(acc[i][t], a, b[t]);
        }
    }
}

// A-operand fragment gathered from a token-major LDS block: element (i = column c0 + r, k = token row of the
// K-block that starts at row k0): 8 ds_read_b32. Rows are clamped to rmax (callers zero the matching B rows).
template <int CM>
__device__ __forceinline__ Frag<CM> gather_frag(const float* buf, int c, int k0, int q, int rmax, int ld = LDX) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int row = k0 + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
        row = row < rmax ? row : rmax;
        v[j] = buf[row * ld + c];
    }
    return make_frag<CM>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
}

// ---- sliced mode: n workgroups share one clip and meet at the end of the FFN loop -----------------------------------------------
// blockIdx -> (clip, slice): the slices of a clip get block ids with the same residue mod 8 (under round-robin dispatch: one XCD, so
// that the clip's features and saved rows are fetched into one L2; nothing depends on it — the linear mapping measured the same).
// Grid = round_up(B, 8) * n; workgroups whose clip is >= B leave at once.
__device__ __forceinline__ void slice_map(int n, int& clip, int& slice) {
    const int b = blockIdx.x, x = b & 7, m = b >> 3;
    slice = m % n;
    clip = (m / n) * 8 + x;
}
// The partial sums of the n slices of a clip meet in `xc` ((n, 48, 128) fp32) behind one "published" word per slice (zeroed
// before the launch). A slice publishes its own block, then waits a BOUNDED time for each of the others; a block that does not
// arrive (its workgroup is not resident yet: another process holds the compute units, the grid is larger than the chip) is
// simply computed here as well — every slice can run the whole FFN — and published for everybody: no launch ever depends on
// all of its workgroups being resident at once, the slicing is an optimisation, not a protocol the hardware has to honour.
// No fences: an agent-scope release / acquire pair on gfx950 is a write-back plus an invalidate of the XCD's whole L2 (the eight
// L2s are not coherent with each other), which throws away the packed weights every clip of the XCD streams from it. The
// exchanged values instead travel as agent-scope relaxed atomics (xchg_store / xchg_load: write-through / coherent reads of just
// those words), ordered by the s_waitcnt + barrier in front of the flag store.
// (one word per lane, consecutive lanes = consecutive words: these accesses are not merged into wider ones)
// This ordering argument is a property of gfx942 / gfx950 (sc1 write-through stores, L2-bypassing coherent loads, in-order completion
// counted by vmcnt), not of the HIP memory model: the device code refuses to build for anything else, the host (encoder.hip
// device_slicing_ok) runs one workgroup per clip on any other device, and tools/micro/slice_litmus.hip shows the protocol failing
// once the s_waitcnt is taken out (tests/test_gpu_sliced.py runs it).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the sliced-mode exchange (slice_publish / slice_wait / slice_gather) is validated on gfx942 / gfx950 only"
#endif
__device__ __forceinline__ void xchg_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float xchg_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr int SLICE_MAX = 8;                        // slices per clip (flag words per (layer, clip))
constexpr unsigned long long SLICE_WAIT_TICKS = 10000;      // 100 us of the 100 MHz wall clock per missing slice
// sum of the four per-wave partial blocks in LDS (row stride ld) -> block `xs` of the exchange buffer, then its flag
__device__ __forceinline__ void slice_publish(const float* p0, const float* p1, const float* p2, const float* p3, int ld, int S, float* xs, unsigned* flag) {
    for (int e = threadIdx.x; e < S * 128; e += 256) {
        const int o = (e >> 7) * ld + (e & 127);
        xchg_store(xs + e, (p0[o] + p1[o]) + (p2[o] + p3[o]));
    }
#ifndef EGX_LITMUS_NO_WAITCNT      // (tools/micro/slice_litmus.hip builds this function without the wait to show what it is there for)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's words have been written through
#endif
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// diagnostic: how many slices a waiting workgroup had to compute itself (a late or absent partner, EGX_SLICE_DROP). A non-zero
// count on a quiet GPU means the launches pay 100 us timeouts + recomputed FFN slices: egx_slices_stolen() makes that visible.
__device__ inline void slice_stolen_note(unsigned* counter) { if (threadIdx.x == 0) atomicAdd(counter, 1u); }
// has the block behind `flag` been published? Waits at most SLICE_WAIT_TICKS; the answer is uniform over the workgroup
__device__ __forceinline__ bool slice_wait(const unsigned* flag) {
    __shared__ unsigned arrived;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned a;
        do {
            a = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a) break;
            __builtin_amdgcn_s_sleep(2);
        } while (wall_clock64() - t0 < SLICE_WAIT_TICKS);
        arrived = a;
    }
    __syncthreads();
    const unsigned a = arrived;
    __syncthreads();
    return a != 0;
}
// the sum of the n published blocks, in slice order (identical bits in every slice) -> p0
__device__ __forceinline__ void slice_gather(const float* xc, int n, float* p0, int ld, int S) {
    constexpr int BLK = 48 * 128, PER = BLK / 256;      // one slice's block per round, its 24 words per thread in flight together
    const int tid = threadIdx.x;                        // (rows >= S: never written, read and dropped)
    float acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = 0.f;
    for (int s = 0; s < n; ++s) {
        float v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = xchg_load(xc + (size_t)s * BLK + tid + 256 * i);
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + 256 * i;
        if (e < S * 128) p0[(e >> 7) * ld + (e & 127)] = acc[i];
    }
    __syncthreads();
}

}  // namespace egx
