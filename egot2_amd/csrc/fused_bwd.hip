// Fused backward kernels of the d_model = 128 translator.
//
//   ffn_dw_kernel    (hidden-parallel)  weight gradients of the FFN: dW1, db1, dW2 from (x1, d_res2) with the hidden
//                                       activation H and its gradient dH recomputed on chip — the (N, 2048) tensors
//                                       never exist in HBM.
//   fused_bwd_kernel (clip-parallel)    everything that is local to a clip: LayerNorm backward, FFN input
//                                       gradient (again recomputing H), attention backward with recomputed
//                                       probabilities, in/out projection input gradients, token-prep backward.
//
// Same operand convention as the forward (fused_dev.h). Reference math: autograd of
// torch.nn.TransformerEncoderLayer as built at HHI/models/ttm/model_taskspecific.py:212-215.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "fused.h"
// bf16 FFN input-gradient loop of the clip kernels software-pipelined across hidden blocks like the forward's (fused_bwd_kernel P4); 0 = block after
// block. OFF: measured no faster — stamps of the phase 38.0k -> 40.5k cycles (16-slot ring) with the LayerNorm1 backward behind it 10.9k -> 17.6k,
// c2 bf16 / C3 steps within the box noise of the plain loop (201.7 vs 201.1, 400.0 vs 390.5, 208.4 vs 217.3 us). Unlike the forward's, this loop's
// epilogue is short (110 VALU per block: masks from the saved alive bits, pack, dH tiles): there is little to hide.
#ifndef EGX_FFN_PIPE_BWD
#define EGX_FFN_PIPE_BWD 0
#endif
#ifndef EGX_FFN_PIPE_BWD_RING
#define EGX_FFN_PIPE_BWD_RING 16      // fragment slots of the pipelined loop: 16 = a ring per weight stream (the backward has the registers), 8 = one shared ring
#endif

#ifdef EGX_STAMPS
namespace egx { extern __device__ unsigned long long g_bstamps[32]; }
#define LNB_STAMP(base, k) do { if ((base) >= 0 && blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); egx::g_bstamps[(base) + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#endif
#include "fused_dev.h"

namespace egx {

// ---- FFN weight gradients -------------------------------------------------------------------------
// grid = (d_ff / (64 * HT), splits). A block owns 64*HT hidden units (wave w: HT tiles of 16) and a contiguous
// range of 32-token K-blocks. Per K-block the block stages x1 and g = d_res2 (32 x 128 fp32 each) in LDS with
// coalesced loads (next K-block prefetched into registers), then every wave computes for its hidden tiles
//     H  [tok][hid] = relu(x1 W1^T + b1)  (A = x1 rows,  B = packed W1 fragment)
//     dH [tok][hid] = (g W2) .* mask      (A = g rows,   B = packed W2^T fragment)
// whose C tiles (token rows on the registers, hidden unit on the lane) chain straight into the A operand of
//     dW1 [hid][in]   += dH^T x1           (B gathered from the LDS tile, 8 ds_read_b32 per fragment)
//     dW2T[hid][dout] += H^T  g
// accumulated in registers over the whole token range and written once as fp32 slabs.
template <int CM, int HT>
__global__ __launch_bounds__(256, HT == 1 ? 2 : 1) void ffn_dw_kernel(FfnDwParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILE = 32 * LDX;             // one tensor, 32 tokens
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int hg = blockIdx.x, split = blockIdx.y;
    const int htile0 = (hg * 4 + wave) * HT;   // first 16-wide hidden tile of this wave
    const int nkb_total = (p.N + 31) / 32;
    const int kb_beg = split * p.kb_per_split;
    const int kb_end = min(nkb_total, kb_beg + p.kb_per_split);
    const uint64_t drop_key = p.seed_ptr ? site_key(*p.seed_ptr, (uint32_t)p.layer, SITE_FFN) : p.drop_key;

    f32x4 accW1[HT][8], accW2[HT][8];
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) { accW1[h][j] = f32x4{0, 0, 0, 0}; accW2[h][j] = f32x4{0, 0, 0, 0}; }
    float accB1[HT];
    float b1v[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) { accB1[h] = 0.f; b1v[h] = p.b1[(htile0 + h) * 16 + r]; }

    // packed weight fragments of this wave's hidden tiles, resident in registers for the whole token range
    // (bf16: 32 VGPRs per tile, fp32: 64; fp32 with HT > 1 re-reads them per K-block instead)
    constexpr bool BF16 = CM == CM_BF16;
    constexpr bool WRES = BF16 || HT == 1;
    constexpr int WR = WRES ? HT : 1;
    WRaw<CM> w1f[WR][4], w2f[WR][4];
    if constexpr (WRES) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                w1f[h][k4] = load_w<CM>(p.w1p, htile0 + h, 4, k4, lane);
                w2f[h][k4] = load_w<CM>(p.w2tp, htile0 + h, 4, k4, lane);
            }
    }

    // LDS image of one K-block: fp32 mode keeps [tensor][32][LDX] floats; bf16 mode converts ONCE while staging into
    // [tensor][32][LDB] bf16 rows, from which row fragments are two ds_read_b64 and the transposed (token-along-K)
    // fragments are two ds_read_b64_tr_b16 hardware-transposed reads: no per-fragment conversion or gather VALU.
    constexpr int LDB = 144;                    // halfwords per row: 288 B, conflict-free for the transposed reads
    typedef short s4v __attribute__((ext_vector_type(4)));
    unsigned short* ldsh = reinterpret_cast<unsigned short*>(lds);
    constexpr int TILEH = 32 * LDB;
    float4 pre[8];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int f = tid + i * 256;
            int tensor = f >> 10, row = (f & 1023) >> 5, c4 = f & 31;
            int n = kb * 32 + row;
            const float* src = (tensor ? p.g : p.x1) + (size_t)n * FD + c4 * 4;
            pre[i] = (n < p.N) ? *reinterpret_cast<const float4*>(src) : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf_idx) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int f = tid + i * 256;
            int tensor = f >> 10, row = (f & 1023) >> 5, c4 = f & 31;
            if constexpr (BF16) {
                uint2 pk = make_uint2(pack_bf16(pre[i].x, pre[i].y), pack_bf16(pre[i].z, pre[i].w));
                *reinterpret_cast<uint2*>(ldsh + (buf_idx * 2 + tensor) * TILEH + row * LDB + c4 * 4) = pk;
            } else {
                *reinterpret_cast<float4*>(lds + (buf_idx * 2 + tensor) * TILE + row * LDX + c4 * 4) = pre[i];
            }
        }
    };
    auto row_frag = [&](int buf_idx, int tensor, int tt, int k4) -> Frag<CM> {
        if constexpr (BF16) {
            const unsigned short* b = ldsh + (buf_idx * 2 + tensor) * TILEH + (tt * 16 + r) * LDB + k4 * 32 + 4 * q;
            s4v lo = *reinterpret_cast<const s4v*>(b);
            s4v hi = *reinterpret_cast<const s4v*>(b + 16);
            Frag<CM_BF16> f;
            f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return f;
        } else {
            return load_frag<CM>(lds + (buf_idx * 2 + tensor) * TILE + (tt * 16 + r) * LDX + k4 * 32, q);
        }
    };
    // fragment with the 32 tokens along K for feature tile jt: element (k = token, j = feature jt*16 + r)
    auto tok_frag = [&](int buf_idx, int tensor, int jt) -> Frag<CM> {
        if constexpr (BF16) {
            const int i = lane & 15;
            const unsigned short* b = ldsh + (buf_idx * 2 + tensor) * TILEH + (4 * q + (i >> 2)) * LDB + jt * 16 + 4 * (i & 3);
            typedef __attribute__((address_space(3))) s4v lds_s4v;
            s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b));
            s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + 16 * LDB));
            Frag<CM_BF16> f;
            f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return f;
        } else {
            const float* c = lds + (buf_idx * 2 + tensor) * TILE + jt * 16 + r;
            const int t0 = 4 * q;
            float4 a = make_float4(c[(t0 + 0) * LDX], c[(t0 + 1) * LDX], c[(t0 + 2) * LDX], c[(t0 + 3) * LDX]);
            float4 b = make_float4(c[(t0 + 16) * LDX], c[(t0 + 17) * LDX], c[(t0 + 18) * LDX], c[(t0 + 19) * LDX]);
            return make_frag<CM>(a, b);
        }
    };

    if (kb_beg < kb_end) gload(kb_beg);
    int cur = 0;
    for (int kb = kb_beg; kb < kb_end; ++kb) {
        lstore(cur);
        __syncthreads();
        if (kb + 1 < kb_end) gload(kb + 1);

        // dropout row keys of this K-block's 8 token rows held by the lane (one short loop instead of a division per element)
        uint32_t rowkey[8];
        if (p.drop_thresh) {
            int base = kb * 32;
            int clip0 = base / p.S;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int off = base - clip0 * p.S + (i >> 2) * 16 + 4 * q + (i & 3);
                int c = clip0;
                while (off >= p.S) { off -= p.S; ++c; }
                rowkey[i] = (uint32_t)(c * 64 + off);
            }
        }
        Frag<CM> aH[HT], aD[HT];
        f32x4 hc[HT][2], dc[HT][2];
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { hc[h][tt] = f32x4{0, 0, 0, 0}; dc[h][tt] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            // token-major A fragments, loaded just in time (keeps the kernel at two waves per SIMD)
            Frag<CM> ax[2], ag[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                ax[tt] = row_frag(cur, 0, tt, k4);
                ag[tt] = row_frag(cur, 1, tt, k4);
            }
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                if constexpr (!WRES) {
                    w1f[0][k4] = load_w<CM>(p.w1p, htile0 + h, 4, k4, lane);
                    w2f[0][k4] = load_w<CM>(p.w2tp, htile0 + h, 4, k4, lane);
                }
                Frag<CM> b1f = w_frag<CM>(w1f[WRES ? h : 0][k4]);
                Frag<CM> b2f = w_frag<CM>(w2f[WRES ? h : 0][k4]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    mma<CM>(hc[h][tt], ax[tt], b1f);
                    mma<CM>(dc[h][tt], ag[tt], b2f);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            float bsum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float hv = fmaxf(hc[h][tt][e] + b1v[h], 0.f);
                    float scale = hv > 0.f ? 1.f : 0.f;
                    if (p.drop_thresh) {
                        float ds = drop_scale(drop_key, rowkey[tt * 4 + e], (uint32_t)((htile0 + h) * 16 + r), p.drop_thresh, p.drop_inv);
                        hv *= ds;
                        scale *= ds;
                    }
                    hc[h][tt][e] = hv;
                    float dv = dc[h][tt][e] * scale;
                    dc[h][tt][e] = dv;
                    bsum += dv;
                }
            accB1[h] += bsum;
            aH[h] = chain_frag<CM>(hc[h][0], hc[h][1]);
            aD[h] = chain_frag<CM>(dc[h][0], dc[h][1]);
        }
        // dW accumulation: B operands with the tokens along K (transposed view of the staged tiles)
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            Frag<CM> bx = tok_frag(cur, 0, jt);
            Frag<CM> bg = tok_frag(cur, 1, jt);
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                mma<CM>(accW1[h][jt], aD[h], bx);
                mma<CM>(accW2[h][jt], aH[h], bg);
            }
        }
        cur ^= 1;
    }

    // write the slabs. dW1: C rows = hidden (4q + e), cols = in (jt*16 + r). dW2 is stored already transposed
    // ([dout][hid], the parameter's own layout): the lane's 4 consecutive hidden units are one 16-byte store.
    float* sw1 = p.slab_w1 + (size_t)split * p.d_ff * FD;
    float* sw2 = p.slab_w2t + (size_t)split * p.d_ff * FD;
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        int hrow = (htile0 + h) * 16 + 4 * q;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sw1[(size_t)(hrow + e) * FD + jt * 16 + r] = accW1[h][jt][e];
            *reinterpret_cast<float4*>(sw2 + (size_t)(jt * 16 + r) * p.d_ff + hrow) =
                make_float4(accW2[h][jt][0], accW2[h][jt][1], accW2[h][jt][2], accW2[h][jt][3]);
        }
        float bs = accB1[h];
        bs += __shfl_xor(bs, 16, 64);
        bs += __shfl_xor(bs, 32, 64);
        if (q == 0) p.slab_b1[(size_t)split * p.d_ff + (htile0 + h) * 16 + r] = bs;
    }
}

// Stored-operand variant: the clip-parallel kernels have already written H (forward) and dH (backward) as
// token-along-K operand tiles (fused_dev.h store_hid_tile), so a wave's A fragments are plain coalesced 1 KB loads
// and only the two weight-gradient GEMMs remain. K-blocks are pairs of 16-token tiles of the clip-padded token
// grid (FUSED_TOK_TILES per clip); x1 / g rows of padding tokens are staged as zeros.
// PART (round 5, f32s): 0 = both weight gradients in one workgroup (the kernel of rounds 3-4); 1 = dW1 (+ db1) only, 2 = dW2 only. A PART workgroup
// keeps one accumulator set, stages one operand image (x1 or g: 27 KB of planes) and reads one of the H / dH tile streams: about half the
// registers and half the LDS, so FOUR workgroups share a CU where two did. ffn_dw_split2_kernel launches both parts as one grid (blockIdx.z).
// NW (round 6, f32s PART workgroups only): 8 = eight waves share the staged operand image (hidden group of 128: the x1 / g planes are fetched
// once per 128 hidden units instead of once per 64 — 805 -> 510 MB through the CUs' address paths per launch)
// NH (round 6, f32s PART workgroups only; tuning aid EGX_FFN_DW_W8=42 | 82): hidden tiles per wave. A wave's 8 token-along-K fragments of the x1 / g
// image (48 transposing LDS reads of 512 B per K-block) meet NH H / dH fragments: at NH = 1 the kernel issues ONE LDS read per MFMA (SQ counters,
// profiles/r06_pmc_c2_f32s.json: 5.0 M LDS instructions for 4.7 M MFMAs, 49 % MFMA busy), NH = 2 halves that without splitting any H / dH tile twice.
// Measured SLOWER: c2 f32s step 351 / 357 us (NH = 1, eight waves) vs 362 / 365 (NH = 2, four waves, three workgroups per CU) vs 371 / 372 (NH = 2, eight
// waves, one per CU): the kernel wants four independent waves per SIMD more than it wants fewer LDS reads. Default unchanged.
template <int CM, int OCC, int PART, int NW = 4, int NH = 1>
__device__ __forceinline__ void ffn_dw_stored_body(const FfnDwParams& p) {
    static_assert(NW == 4 || ((NW == 8 || NW == 16) && CM == CM_SPLIT && PART != 0), "eight / sixteen waves: the f32s PART workgroups only");
    static_assert(NH == 1 || (CM == CM_SPLIT && PART != 0), "several hidden tiles per wave: the f32s PART workgroups only");
    constexpr bool W1 = PART != 2, W2 = PART != 1;
    constexpr bool BF16 = CM == CM_BF16;
    constexpr bool SPLIT = CM == CM_SPLIT;
    // OCC 3: single LDS buffer, no register prefetch, three workgroups per CU. CM_SPLIT: single buffer of three bf16
    // planes per tensor (54 KB), two workgroups per CU.
    constexpr bool PF = OCC <= 2 && !SPLIT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILE = 32 * LDX;
    constexpr int ESZ = BF16 ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    // Workgroups are dispatched round-robin over the 8 XCDs; the 32 hidden groups of one token split read the same x1 / g
    // rows, so each XCD takes a contiguous range of (split, hidden group) pairs and its L2 fetches those rows once.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nwg = gridDim.x * gridDim.y, id = by * gridDim.x + bx;
        if ((nwg & 7) == 0) { const int t = (id & 7) * (nwg >> 3) + (id >> 3); by = t / gridDim.x; bx = t - by * gridDim.x; }
    }
    const int htile = (bx * NW + wave) * NH, split = by;      // (first of the wave's NH hidden tiles)
    const int nht = p.d_ff / 16;
    const int ntile = p.B * FUSED_TOK_TILES;
    const int nkb_total = (ntile + 1) / 2;
    const int kb_beg = split * p.kb_per_split;
    const int kb_end = min(nkb_total, kb_beg + p.kb_per_split);

    f32x4 accW1[W1 ? NH : 1][8], accW2[W2 ? NH : 1][8];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) { if constexpr (W1) accW1[h][j] = f32x4{0, 0, 0, 0}; if constexpr (W2) accW2[h][j] = f32x4{0, 0, 0, 0}; }
    float accB1[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) accB1[h] = 0.f;
    float accB4[4] = {0.f, 0.f, 0.f, 0.f};      // CM_BF16: db1 partials of hidden units 4q..4q+3 over this lane's token

    constexpr int LDB = 144;
    typedef short s4v __attribute__((ext_vector_type(4)));
    unsigned short* ldsh = reinterpret_cast<unsigned short*>(lds);
    constexpr int TILEH = 32 * LDB;
    // per-wave staging of the H / dH tiles (accumulator layout -> token-along-K fragments), behind the x1 / g images
    constexpr int XG_BYTES = SPLIT ? 2 * 3 * TILEH * 2 : (PF ? 4 : 2) * 32 * LDX * 4;      // as allocated by launch_ffn_dw
    unsigned short* tstage = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(lds) + XG_BYTES);
    float4 pre[8];
    // the wave's H / dH operand tiles of the next K-block, as loaded (bf16: already operands; fp32 tiles: converted / split
    // into operands after the prefetch has landed)
    struct HidRaw { float4 a, b; };
    struct HidRawB { uint2 a, b; };
    typename std::conditional<BF16, HidRawB, HidRaw>::type nH[NH], nD[NH];
    // CM_SPLIT: x1 / g arrive as three bf16 planes ([part][N][128], written by the clip-parallel kernels); a thread moves two rows x
    // 16 B of every (tensor, part) image: no split work here (the 32 workgroups of a token range used to repeat it)
    uint4 prs[SPLIT ? 12 : 1];
    auto gload = [&](int kb) {
        if constexpr (SPLIT && NW == 16) {
            // 1024 threads: the image's 3 x 32 x 16 sixteen-byte pieces in two passes (the second half-empty: clamped, not stored)
            const size_t plane = (size_t)p.B * FUSED_TOK_PAD * FD;
            const unsigned short* base = reinterpret_cast<const unsigned short*>(W1 ? p.x1 : p.g);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int idx = tid + k * 1024;
                idx = idx < 1536 ? idx : 1535;
                const int part = idx >> 9, row = (idx & 511) >> 4, c8 = idx & 15;
                const int tile = kb * 2 + (row >> 4);
                const int tc = tile < ntile ? tile : ntile - 1;
                const uint4 v = *reinterpret_cast<const uint4*>(base + part * plane + ((size_t)tc * 16 + (row & 15)) * FD + c8 * 8);
                prs[k] = tile < ntile ? v : make_uint4(0, 0, 0, 0);
            }
        } else if constexpr (SPLIT) {
            const size_t plane = (size_t)p.B * FUSED_TOK_PAD * FD;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int tp = i >> 1, tensor = tp / 3, part = tp - tensor * 3;
                if ((tensor == 0 && !W1) || (tensor == 1 && !W2)) continue;
                if (NW == 8 && (i & 1)) continue;       // 512 threads: one pass covers the 32 rows
                const int row = (tid >> 4) + (i & 1) * 16, c8 = tid & 15;
                int tile = kb * 2 + (row >> 4);
                const unsigned short* src = reinterpret_cast<const unsigned short*>(tensor ? p.g : p.x1) + part * plane +
                                            ((size_t)tile * 16 + (row & 15)) * FD + c8 * 8;
                prs[i] = tile < ntile ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int f = tid + i * 256;
                int tensor = f >> 10, row = (f & 1023) >> 5, c4 = f & 31;
                int tile = kb * 2 + (row >> 4);
                int clip = tile / FUSED_TOK_TILES;
                int tok = (tile - clip * FUSED_TOK_TILES) * 16 + (row & 15);
                const float* src = (tensor ? p.g : p.x1) + ((size_t)clip * p.S + tok) * FD + c4 * 4;
                pre[i] = (tile < ntile && tok < p.S) ? *reinterpret_cast<const float4*>(src) : make_float4(0, 0, 0, 0);
            }
        }
        int ta = kb * 2, tb = min(kb * 2 + 1, ntile - 1);     // a missing second tile meets zero x1 / g rows
        size_t oa = ((size_t)ta * nht + htile) * (HTILE_ELEMS * ESZ), ob = ((size_t)tb * nht + htile) * (HTILE_ELEMS * ESZ);
        if constexpr (BF16) {
            nH[0].a = reinterpret_cast<const uint2*>((const char*)p.hs + oa)[lane];
            nH[0].b = reinterpret_cast<const uint2*>((const char*)p.hs + ob)[lane];
            nD[0].a = reinterpret_cast<const uint2*>((const char*)p.dhs + oa)[lane];
            nD[0].b = reinterpret_cast<const uint2*>((const char*)p.dhs + ob)[lane];
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h) {      // (consecutive hidden tiles of a token tile are consecutive in memory)
                const size_t oh = (size_t)h * (HTILE_ELEMS * ESZ);
                if constexpr (W2) {
                    nH[h].a = reinterpret_cast<const float4*>((const char*)p.hs + oa + oh)[lane];
                    nH[h].b = reinterpret_cast<const float4*>((const char*)p.hs + ob + oh)[lane];
                }
                if constexpr (W1) {
                    nD[h].a = reinterpret_cast<const float4*>((const char*)p.dhs + oa + oh)[lane];
                    nD[h].b = reinterpret_cast<const float4*>((const char*)p.dhs + ob + oh)[lane];
                }
            }
        }
    };
    auto lstore = [&](int buf_idx) {
        if constexpr (SPLIT && NW == 16) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + k * 1024;
                const int part = idx >> 9, row = (idx & 511) >> 4, c8 = idx & 15;
                if (idx < 1536) *reinterpret_cast<uint4*>(ldsh + part * TILEH + row * LDB + c8 * 8) = prs[k];
            }
            return;
        }
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if ((i < 6 && !W1) || (i >= 6 && !W2)) continue;
                if (NW == 8 && (i & 1)) continue;
                const int row = (tid >> 4) + (i & 1) * 16, c8 = tid & 15;
                // (a PART workgroup has ONE operand image: it sits where tensor 0's would)
                *reinterpret_cast<uint4*>(ldsh + ((i >> 1) - (PART == 2 ? 3 : 0)) * TILEH + row * LDB + c8 * 8) = prs[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int f = tid + i * 256;
            int tensor = f >> 10, row = (f & 1023) >> 5, c4 = f & 31;
            if constexpr (BF16) {
                uint2 pk = make_uint2(pack_bf16(pre[i].x, pre[i].y), pack_bf16(pre[i].z, pre[i].w));
                *reinterpret_cast<uint2*>(ldsh + (buf_idx * 2 + tensor) * TILEH + row * LDB + c4 * 4) = pk;
            } else {
                *reinterpret_cast<float4*>(lds + (buf_idx * 2 + tensor) * TILE + row * LDX + c4 * 4) = pre[i];
            }
        }
    };
    typedef __attribute__((address_space(3))) s4v lds_s4v;
    auto tok_frag = [&](int buf_idx, int tensor, int jt) -> Frag<CM> {
        if constexpr (BF16) {
            const int i = lane & 15;
            const unsigned short* b = ldsh + (buf_idx * 2 + tensor) * TILEH + (4 * q + (i >> 2)) * LDB + jt * 16 + 4 * (i & 3);
            s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b));
            s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + 16 * LDB));
            Frag<CM_BF16> f;
            f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return f;
        } else if constexpr (SPLIT) {
            const int i = lane & 15;
            const unsigned short* b = ldsh + (PART == 2 ? 0 : tensor) * 3 * TILEH + (4 * q + (i >> 2)) * LDB + jt * 16 + 4 * (i & 3);
            Frag<CM_SPLIT> f;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + pl * TILEH));
                s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + pl * TILEH + 16 * LDB));
                f.p[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            return f;
        } else {
            const float* c = lds + (buf_idx * 2 + tensor) * TILE + jt * 16 + r;
            const int t0 = 4 * q;
            float4 a = make_float4(c[(t0 + 0) * LDX], c[(t0 + 1) * LDX], c[(t0 + 2) * LDX], c[(t0 + 3) * LDX]);
            float4 b = make_float4(c[(t0 + 16) * LDX], c[(t0 + 17) * LDX], c[(t0 + 18) * LDX], c[(t0 + 19) * LDX]);
            return make_frag<CM>(a, b);
        }
    };

    // CM_SPLIT: register prefetch of the next K-block over its single LDS buffer (the loads fly under this block's MFMAs).
    // Measured on this kernel (79 us): dropping 7/8 of the MFMAs, or all fragment reads but one, or the H / dH split, or half of
    // the H / dH loads, each moves it by < 10 %; two hidden tiles per wave at one workgroup per CU: 90 us; eight waves sharing
    // a three-stage LDS-DMA ring with the tiles two K-blocks ahead: 95 us. SQ counters: MFMA busy 35 us + VALU busy 27 us
    // per SIMD, waves wait 38 % of their time — what helps is two INDEPENDENT workgroups per CU drifting out of phase (one
    // in its split / staging phase while the other runs MFMAs); an 8-wave workgroup runs all its phases in lockstep.
    // (A 2 x 4 hidden x feature tiling per wave, which halves the LDS fragment reads, was slower: 87 -> 100 us — the kernel
    // follows its total VALU + MFMA work, and that tiling splits twice as many H / dH tiles per wave.)
    constexpr bool PF1 = SPLIT;
    if ((PF || PF1) && kb_beg < kb_end) gload(kb_beg);
    int cur = 0;
    for (int kb = kb_beg; kb < kb_end; ++kb) {
        if constexpr (PF1) {
            if (kb > kb_beg) __syncthreads();   // everyone is done reading the previous K-block
        } else if constexpr (!PF) {
            gload(kb);
            __syncthreads();            // everyone is done reading the previous K-block
        }
        lstore(cur);
        const bool has_b = kb * 2 + 1 < ntile;      // odd tile count: the last block's second half is a duplicate
        Frag<CM> aH[NH], aD[NH];
        if constexpr (!BF16) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                if constexpr (W1) {
                    float s0 = (nD[h].a.x + nD[h].a.y) + (nD[h].a.z + nD[h].a.w), s1 = (nD[h].b.x + nD[h].b.y) + (nD[h].b.z + nD[h].b.w);
                    accB1[h] += s0 + (has_b ? s1 : 0.f);
                    aD[h] = make_frag<CM>(nD[h].a, nD[h].b);
                }
                if constexpr (W2) aH[h] = make_frag<CM>(nH[h].a, nH[h].b);
            }
        } else {
            // CM_BF16: tiles arrive in accumulator layout (lane = token r, 4 hidden units 4q..4q+3). Stage this wave's two
            // K-blocks ([32 tokens][16 hidden] bf16, unpadded 32-byte rows: writes and transposed reads are conflict-free) and
            // read them back token-along-K. db1: per-lane sums over the tokens, reduced across the 16 token lanes at the end.
            // (The code below also handles three-part staging; CM_SPLIT does not use it, see store_hid_tile.)
            constexpr int NPL = SPLIT ? 3 : 1;
            unsigned short* wt = tstage + wave * (2 * NPL * 512);      // [tensor][part][32][16] halfwords
            auto put = [&](int tensor, int half, auto ra) {
                unsigned short* d = wt + tensor * NPL * 512 + (half * 16 + r) * 16 + 4 * q;
                if constexpr (BF16) {
                    *reinterpret_cast<uint2*>(d) = (half && !has_b) ? make_uint2(0, 0) : ra;
                } else {
                    float4 v = (half && !has_b) ? make_float4(0, 0, 0, 0) : ra;
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split_pair(v.x, v.y, h0, m0, l0);
                    split_pair(v.z, v.w, h1, m1, l1);
                    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(d + 512) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(d + 1024) = make_uint2(l0, l1);
                }
            };
            put(0, 0, nH[0].a); put(0, 1, nH[0].b); put(1, 0, nD[0].a); put(1, 1, nD[0].b);
            if constexpr (BF16) {
                accB4[0] += __uint_as_float(nD[0].a.x << 16) + (has_b ? __uint_as_float(nD[0].b.x << 16) : 0.f);
                accB4[1] += __uint_as_float(nD[0].a.x & 0xffff0000u) + (has_b ? __uint_as_float(nD[0].b.x & 0xffff0000u) : 0.f);
                accB4[2] += __uint_as_float(nD[0].a.y << 16) + (has_b ? __uint_as_float(nD[0].b.y << 16) : 0.f);
                accB4[3] += __uint_as_float(nD[0].a.y & 0xffff0000u) + (has_b ? __uint_as_float(nD[0].b.y & 0xffff0000u) : 0.f);
            } else {
                accB4[0] += nD[0].a.x + (has_b ? nD[0].b.x : 0.f); accB4[1] += nD[0].a.y + (has_b ? nD[0].b.y : 0.f);
                accB4[2] += nD[0].a.z + (has_b ? nD[0].b.z : 0.f); accB4[3] += nD[0].a.w + (has_b ? nD[0].b.w : 0.f);
            }
            auto get = [&](int tensor) {
                const int i = lane & 15;
                const unsigned short* b = wt + tensor * NPL * 512 + (4 * q + (i >> 2)) * 16 + 4 * (i & 3);
                Frag<CM> f;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + pl * 512));
                    s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + pl * 512 + 16 * 16));
                    if constexpr (SPLIT) f.p[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    else f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                return f;
            };
            aH[0] = get(0);
            aD[0] = get(1);
        }
        __syncthreads();
        if ((PF || PF1) && kb + 1 < kb_end) gload(kb + 1);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            if constexpr (W1) {
                Frag<CM> bx = tok_frag(cur, 0, jt);
#pragma unroll
                for (int h = 0; h < NH; ++h) mma<CM>(accW1[h][jt], aD[h], bx);
            }
            if constexpr (W2) {
                Frag<CM> bg = tok_frag(cur, 1, jt);
#pragma unroll
                for (int h = 0; h < NH; ++h) mma<CM>(accW2[h][jt], aH[h], bg);
            }
        }
        if constexpr (PF) cur ^= 1;
    }

    float* sw1 = p.slab_w1 + (size_t)split * p.d_ff * FD;
    float* sw2 = p.slab_w2t + (size_t)split * p.d_ff * FD;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int hrow = (htile + h) * 16 + 4 * q;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            if constexpr (W1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sw1[(size_t)(hrow + e) * FD + jt * 16 + r] = accW1[h][jt][e];
            }
            if constexpr (W2)
                *reinterpret_cast<float4*>(sw2 + (size_t)(jt * 16 + r) * p.d_ff + hrow) =
                    make_float4(accW2[h][jt][0], accW2[h][jt][1], accW2[h][jt][2], accW2[h][jt][3]);
        }
    }
    if constexpr (!BF16) {
        if constexpr (W1) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float bs = accB1[h];
                bs += __shfl_xor(bs, 16, 64);
                bs += __shfl_xor(bs, 32, 64);
                if (q == 0) p.slab_b1[(size_t)split * p.d_ff + (htile + h) * 16 + r] = bs;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float bs = accB4[e];
            bs += __shfl_xor(bs, 1, 64); bs += __shfl_xor(bs, 2, 64); bs += __shfl_xor(bs, 4, 64); bs += __shfl_xor(bs, 8, 64);
            accB4[e] = bs;
        }
        if (r == 0) *reinterpret_cast<float4*>(p.slab_b1 + (size_t)split * p.d_ff + htile * 16 + 4 * q) = make_float4(accB4[0], accB4[1], accB4[2], accB4[3]);
    }
}

template <int CM, int OCC>
__global__ __launch_bounds__(256, OCC) void ffn_dw_stored_kernel(FfnDwParams p) { ffn_dw_stored_body<CM, OCC, 0>(p); }
// f32s: dW1 (+ db1) workgroups (blockIdx.z = 0) and dW2 workgroups (z = 1) in one grid, four per CU
__global__ __launch_bounds__(256, 4) void ffn_dw_split2_kernel(FfnDwParams p) {
    if (blockIdx.z == 0) ffn_dw_stored_body<CM_SPLIT, 2, 1>(p);
    else ffn_dw_stored_body<CM_SPLIT, 2, 2>(p);
}
__global__ __launch_bounds__(512, 2) void ffn_dw_split2w8_kernel(FfnDwParams p) {
    if (blockIdx.z == 0) ffn_dw_stored_body<CM_SPLIT, 2, 1, 8>(p);
    else ffn_dw_stored_body<CM_SPLIT, 2, 2, 8>(p);
}
// two hidden tiles per wave (NH = 2): four waves per workgroup = a hidden group of 128 like the eight-wave grid, three workgroups per CU (<= 168 registers)
__global__ __launch_bounds__(256, 3) void ffn_dw_split2h2_kernel(FfnDwParams p) {
    if (blockIdx.z == 0) ffn_dw_stored_body<CM_SPLIT, 2, 1, 4, 2>(p);
    else ffn_dw_stored_body<CM_SPLIT, 2, 2, 4, 2>(p);
}
// ... and eight waves: a hidden group of 256
__global__ __launch_bounds__(512, 1) void ffn_dw_split2w8h2_kernel(FfnDwParams p) {
    if (blockIdx.z == 0) ffn_dw_stored_body<CM_SPLIT, 2, 1, 8, 2>(p);
    else ffn_dw_stored_body<CM_SPLIT, 2, 2, 8, 2>(p);
}
__global__ __launch_bounds__(1024, 1) void ffn_dw_split2w16_kernel(FfnDwParams p) {
    if (blockIdx.z == 0) ffn_dw_stored_body<CM_SPLIT, 2, 1, 16>(p);
    else ffn_dw_stored_body<CM_SPLIT, 2, 2, 16>(p);
}

// bf16 stored-operand kernel with everything staged by LDS-DMA. ffn_dw_stored_kernel<CM_BF16> spends a K-block's time on
// 16 MFMAs per wave and a chain of load -> convert -> ds_write -> barrier -> ds_read around them. Here the x1 / g rows
// (bf16 planes on the 48-row clip grid, written by the clip-parallel kernels) and the wave's four H / dH tiles enter a ring
// of three LDS stages by global_load_lds_dwordx4 — no registers, no conversion, no ds_write — two K-blocks ahead; a counted
// s_waitcnt leaves the newer K-block in flight across the one barrier per K-block. Stage = x1 image + g image ([32 tokens]
// [128] bf16, 32-byte column chunks XOR-swizzled with the row so that the transposed fragment reads spread over all banks)
// + 4 waves x 4 tiles of 512 B (accumulator layout, read back token-along-K by ds_read_b64_tr_b16 as they lie). 256 threads,
// two independent workgroups per CU.
#define EGX_RING_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// NW waves per workgroup share the staged x1 / g images (round 6: with four waves 24 KB per K-block for 64 MFMAs, the images 16 KB of it; eight and
// sixteen waves measured SLOWER, 201 -> 205-219 us on the c2 bf16 step: the default stays four); D ring stages.
template <int NW, int D>
__device__ __forceinline__ void ffn_dw_bf16_ring_body(const FfnDwParams& p) {
    constexpr int CM = CM_BF16;
    // (two stages at three workgroups per CU, with 16 or 24 token splits, run at the same 35 us)
    constexpr int IMG = 32 * 256, TILES = NW * 4 * 512, SB = 2 * IMG + TILES;
    constexpr int IMGI = 16 / NW;        // image staging instructions per wave and K-block (two images x 32 rows = 16 wave-loads of 4 rows)
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    typedef short s4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s4v lds_s4v;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    {   // XCD-contiguous (split, hidden group) ranges, as in ffn_dw_stored_kernel
        const int nwg = gridDim.x * gridDim.y, id = by * gridDim.x + bx;
        if ((nwg & 7) == 0) { const int t = (id & 7) * (nwg >> 3) + (id >> 3); by = t / gridDim.x; bx = t - by * gridDim.x; }
    }
    const int htile = bx * NW + wave, split = by;
    const int nht = p.d_ff / 16;
    const int ntile = p.B * FUSED_TOK_TILES;
    const int nkb_total = (ntile + 1) / 2;
    const int kb_beg = split * p.kb_per_split;
    const int kb_end = min(nkb_total, kb_beg + p.kb_per_split);

    f32x4 accW1[8], accW2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { accW1[j] = f32x4{0, 0, 0, 0}; accW2[j] = f32x4{0, 0, 0, 0}; }
    float accB4[4] = {0.f, 0.f, 0.f, 0.f};

    // staging, IMGI + 2 instructions per wave and K-block: j < IMGI moves rows 4 * (idx & 7) .. + 3 of image idx >> 3 (idx = IMGI w + j);
    // the last two move this wave's two H / two dH tiles (lanes 0..31: first token tile, lanes 32..63: second)
    const unsigned char* srcS[IMGI];
    int rowS[IMGI];
#pragma unroll
    for (int j = 0; j < IMGI; ++j) {
        const int idx = wave * IMGI + j, img = idx >> 3;
        const int row = (idx & 7) * 4 + (lane >> 4), slot = lane & 15;
        rowS[j] = row;
        srcS[j] = reinterpret_cast<const unsigned char*>(img ? p.g : p.x1) + (((slot >> 1) ^ (row & 7)) * 32 + (slot & 1) * 16);
    }
    const unsigned char* hsrc = reinterpret_cast<const unsigned char*>(p.hs) + (size_t)htile * 512 + (lane & 31) * 16;
    const unsigned char* dsrc = reinterpret_cast<const unsigned char*>(p.dhs) + (size_t)htile * 512 + (lane & 31) * 16;
    auto stage = [&](int kb, int buf) {
        unsigned char* dst = ring + buf * SB + wave * IMGI * 1024;
#pragma unroll
        for (int j = 0; j < IMGI; ++j) {
            int tile = min(kb * 2 + (rowS[j] >> 4), ntile - 1);     // a missing second tile: finite rows, its H / dH operand half is zeroed
            __builtin_amdgcn_global_load_lds((glb_void*)(srcS[j] + ((size_t)tile * 16 + (rowS[j] & 15)) * (FD * 2)), (lds_void*)(dst + j * 1024), 16, 0, 0);
        }
        const int tt = min(kb * 2 + (lane >> 5), ntile - 1);
        unsigned char* td = ring + buf * SB + 2 * IMG + wave * 2048;
        __builtin_amdgcn_global_load_lds((glb_void*)(hsrc + (size_t)tt * nht * 512), (lds_void*)td, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(dsrc + (size_t)tt * nht * 512), (lds_void*)(td + 1024), 16, 0, 0);
    };
    // x1 / g fragment: lane (r, q) supplies row 4q + (r >> 2) (+ 16) and the 8 bytes 4 (r & 3) of its 32-byte chunk
    const int frow = 4 * q + (r >> 2);
    const int foff = frow * 256 + 8 * (r & 3), fswz = frow & 7;
    auto tok_frag = [&](int buf, int img, int jt) -> Frag<CM> {
        const unsigned char* b = ring + buf * SB + img * IMG + foff + ((jt ^ fswz) * 32);
        s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b));
        s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + 16 * 256));
        Frag<CM> f;
        f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return f;
    };
    // H / dH tile pair as they lie (accumulator layout: [hidden / 4][token][hidden % 4]): lane (r, q) supplies the 4 hidden
    // units 4 (r & 3) .. + 3 of token 4q + (r >> 2); the second tile sits 512 B behind the first
    const int toff = ((r & 3) * 16 + 4 * q + (r >> 2)) * 8;

#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if (kb_beg + s < kb_end) stage(kb_beg + s, s);
    for (int kb0 = kb_beg; kb0 < kb_end; kb0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int kb = kb0 + s;
            if (kb >= kb_end) break;
            // this K-block's loads have landed; the next K-block's (IMGI + 2 per wave) may stay in flight across the barrier
            if (kb + 1 < kb_end) {
                if constexpr (D >= 3) { if constexpr (IMGI == 4) EGX_RING_WAIT(6); else if constexpr (IMGI == 2) EGX_RING_WAIT(4); else EGX_RING_WAIT(3); }
                else EGX_RING_WAIT(0);
            } else EGX_RING_WAIT(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kb + D - 1 < kb_end) stage(kb + D - 1, (s + D - 1) % D);
            const bool has_b = kb * 2 + 1 < ntile;
            const unsigned char* tb = ring + s * SB + 2 * IMG + wave * 2048;
            Frag<CM> aH, aD;
            {
                s4v hlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(tb + toff));
                s4v hhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(tb + 512 + toff));
                s4v dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(tb + 1024 + toff));
                s4v dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(tb + 1536 + toff));
                if (!has_b) { hhi = s4v{0, 0, 0, 0}; dhi = s4v{0, 0, 0, 0}; }
                aH.v = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                aD.v = __builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            {   // db1: this lane's token of hidden units 4q .. 4q + 3 (tile entry [lane]), summed over the 16 token lanes at the end
                uint2 da = *reinterpret_cast<const uint2*>(tb + 1024 + lane * 8);
                uint2 db = *reinterpret_cast<const uint2*>(tb + 1536 + lane * 8);
                if (!has_b) db = make_uint2(0, 0);
                accB4[0] += __uint_as_float(da.x << 16) + __uint_as_float(db.x << 16);
                accB4[1] += __uint_as_float(da.x & 0xffff0000u) + __uint_as_float(db.x & 0xffff0000u);
                accB4[2] += __uint_as_float(da.y << 16) + __uint_as_float(db.y << 16);
                accB4[3] += __uint_as_float(da.y & 0xffff0000u) + __uint_as_float(db.y & 0xffff0000u);
            }
#pragma unroll
            for (int jt = 0; jt < 8; ++jt) {
                Frag<CM> bx1 = tok_frag(s, 0, jt);
                Frag<CM> bg = tok_frag(s, 1, jt);
                mma<CM>(accW1[jt], aD, bx1);
                mma<CM>(accW2[jt], aH, bg);
            }
        }
    }

    float* sw1 = p.slab_w1 + (size_t)split * p.d_ff * FD;
    float* sw2 = p.slab_w2t + (size_t)split * p.d_ff * FD;
    const int hrow = htile * 16 + 4 * q;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sw1[(size_t)(hrow + e) * FD + jt * 16 + r] = accW1[jt][e];
        *reinterpret_cast<float4*>(sw2 + (size_t)(jt * 16 + r) * p.d_ff + hrow) =
            make_float4(accW2[jt][0], accW2[jt][1], accW2[jt][2], accW2[jt][3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float bs = accB4[e];
        bs += __shfl_xor(bs, 1, 64); bs += __shfl_xor(bs, 2, 64); bs += __shfl_xor(bs, 4, 64); bs += __shfl_xor(bs, 8, 64);
        accB4[e] = bs;
    }
    if (r == 0) *reinterpret_cast<float4*>(p.slab_b1 + (size_t)split * p.d_ff + htile * 16 + 4 * q) = make_float4(accB4[0], accB4[1], accB4[2], accB4[3]);
}
__global__ __launch_bounds__(256, 2) void ffn_dw_bf16_ring_kernel(FfnDwParams p) { ffn_dw_bf16_ring_body<4, 3>(p); }
__global__ __launch_bounds__(512, 1) void ffn_dw_bf16_ring8_kernel(FfnDwParams p) { ffn_dw_bf16_ring_body<8, 3>(p); }
__global__ __launch_bounds__(512, 2) void ffn_dw_bf16_ring8d2_kernel(FfnDwParams p) { ffn_dw_bf16_ring_body<8, 2>(p); }
__global__ __launch_bounds__(1024, 1) void ffn_dw_bf16_ring16_kernel(FfnDwParams p) { ffn_dw_bf16_ring_body<16, 3>(p); }

// Three slab reductions in one launch: out_k[i] += sum_z slab_k[z * n_k + i], float4-vectorised. The same launch can
// carry the reduction of the per-clip partial rows (blocks >= slab_blocks; see reduce_partials_kernel).
__device__ __forceinline__ void reduce_slabs_block(const SlabReduce& a, unsigned block) {
    size_t i = ((size_t)block * 256 + threadIdx.x) * 4;
    for (int k = 0; k < a.narr; ++k) {
        if (i < a.n[k]) {
            float4 s = *reinterpret_cast<const float4*>(a.out[k] + i);
            int z = 0;
            for (; z + 8 <= a.nslab; z += 8) {      // eight loads in flight, summed in slab order
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(a.slab[k] + (size_t)(z + j) * a.n[k] + i);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
            }
            for (; z < a.nslab; ++z) {
                float4 v = *reinterpret_cast<const float4*>(a.slab[k] + (size_t)z * a.n[k] + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(a.out[k] + i) = s;
            return;
        }
        i -= a.n[k];
    }
}
// grads[dst] += sum_clip partials[clip][off .. off+len): block (bx, by) of a (P / 64, clip chunks) grid; 256 threads =
// 64 columns x 4 clip lanes, LDS tree, one atomic per column per block (<= 16 adders per address).
__device__ __forceinline__ void reduce_partials_block(const ReducePartialsParams& rp, int bx, int by, int ny) {
    __shared__ float red[4][64];
    int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    int j = bx * 64 + c;
    int per = (rp.B + ny - 1) / ny;
    int b0 = by * per, b1 = min(rp.B, b0 + per);
    float s = 0.f;
    if (j < rp.P)
        for (int b = b0 + g; b < b1; b += 4) s += rp.partials[(size_t)b * rp.P + j];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && j < rp.P) {
        float* dst = nullptr;
        for (int i = 0; i < rp.n; ++i)
            if (j >= rp.d[i].off && j < rp.d[i].off + rp.d[i].len) dst = rp.d[i].dst + (j - rp.d[i].off);
        if (dst) atomicAdd(dst, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
}
__global__ __launch_bounds__(256) void reduce_slabs_add_kernel(SlabReduce a) { reduce_slabs_block(a, blockIdx.x); }
__global__ __launch_bounds__(256) void reduce_tail_kernel(SlabReduce a, ReducePartialsParams rp, unsigned slab_blocks, int chunks) {
    // the partial-row blocks go first: they are chains of dependent loads and atomics that then run under the slab streams
    const unsigned part_blocks = gridDim.x - slab_blocks;
    if (blockIdx.x >= part_blocks) { reduce_slabs_block(a, blockIdx.x - part_blocks); return; }
    reduce_partials_block(rp, (int)(blockIdx.x / chunks), (int)(blockIdx.x % chunks), chunks);
}
static int partial_chunks(int B) { return B >= 64 ? 16 : (B >= 8 ? 4 : 1); }

// fp32 stored-operand kernel: three workgroups per CU (single LDS buffer, no register prefetch) with 24 token splits;
// every other variant: two per CU, 16 splits. EGX_FFN_DW_OCC=2|3 overrides (tuning aid).
static int ffn_dw_occ(bool stored, bool bf16) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("EGX_FFN_DW_OCC"); env = (e && (e[0] == '2' || e[0] == '3')) ? e[0] - '0' : 0; }
    if (!stored) return 2;
    return env ? env : (bf16 ? 2 : 3);
}
// EGX_FFN_DW_RING=0: bf16 runs keep fp32 x1 / g2 hand-overs and ffn_dw_stored_kernel<CM_BF16> (tuning aid; read by the
// encoder when it lays out the buffers, see ffn_dw_bf16_planes())
static bool ffn_dw_ring() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_FFN_DW_RING"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
bool ffn_dw_bf16_planes() { return ffn_dw_ring(); }
// waves per workgroup of the bf16 LDS-ring kernel: 4 (rounds 3-5), 8, 16, or 82 = eight waves with a two-stage ring at two workgroups per CU
// (EGX_FFN_DW_BF16_NW, tuning aid)
static int ffn_dw_bf16_waves(int d_ff) {
    static const int env = [] { const char* e = getenv("EGX_FFN_DW_BF16_NW"); return e ? atoi(e) : 4; }();
    const int nwv = env == 82 ? 8 : env;
    if ((nwv == 8 || nwv == 16) && d_ff % (16 * nwv) == 0) return env;
    return 4;
}
// most token splits any variant uses for a hidden width: a narrow FFN (the PNR / OSCC recipe's d_ff = 256: four hidden groups) needs more of them
// to fill the chip; its slabs are small
static int ffn_dw_max_splits(int d_ff) { return d_ff <= 512 ? 48 : 32; }
// EGX_FFN_DW_SPLIT2=0: the one-workgroup-for-both-gradients kernel of rounds 3-4 in f32s (tuning aid)
static bool ffn_dw_split2() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_FFN_DW_SPLIT2"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
static int ffn_dw_splits(int nkb, int occ, int d_ff, bool split2) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("EGX_FFN_DW_SPLITS"); env = e ? atoi(e) : 0; }     // tuning aid
    const int cap = ffn_dw_max_splits(d_ff);
    if (env > 0) return min(nkb, min(env, cap));
    int sp = occ == 3 ? 24 : 16;
    // f32s (one grid of dW1 and dW2 workgroups, four per CU): aim at ~1024 workgroups. d_ff = 256: 1663 -> 1595 us per PNR step at 24 splits,
    // measured; 2048-wide FFNs keep 16 (12 and 24 both slower on C2)
    // (the other variants: one workgroup per (group, split), two or three per CU: ~512)
    { const int groups = d_ff / 64, per = split2 ? 2 : 1, want = split2 ? 1024 : 512; while (sp < cap && per * groups * sp < want) sp += 8; }
    return min(nkb, min(sp, cap));
}
size_t ffn_dw_scratch_bytes(int N, int d_ff, int* splits_out) {
    int nkb = (N + 31) / 32;
    int splits = min(nkb, ffn_dw_max_splits(d_ff));           // sized for the largest split count any variant uses
    if (splits_out) *splits_out = splits;
    return (size_t)splits * ((size_t)2 * d_ff * FD + d_ff) * sizeof(float);
}

template <int CM>
static int launch_ffn_dw(FfnDwParams p, hipStream_t st) {
    constexpr int HT = 1;
    size_t lds = (size_t)4 * 32 * LDX * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_kernel<CM, HT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(p.d_ff / (64 * HT), p.splits);
    timing_begin(TIMER_FFN_DW, st);
    if (p.hs && CM == CM_BF16 && p.xg_planes && ffn_dw_ring()) {
        const int nw = ffn_dw_bf16_waves(p.d_ff);
        const int nd = nw == 82 ? 2 : 3, nwv = nw == 82 ? 8 : nw;
        const int ring_bytes = nd * (2 * 32 * 256 + nwv * 4 * 512);
        static bool attr3_set = false;
        if (!attr3_set) {
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_bf16_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (2 * 32 * 256 + 4 * 4 * 512)));
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_bf16_ring8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (2 * 32 * 256 + 8 * 4 * 512)));
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_bf16_ring8d2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 32 * 256 + 8 * 4 * 512)));
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_bf16_ring16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (2 * 32 * 256 + 16 * 4 * 512)));
            attr3_set = true;
        }
        const dim3 g2(p.d_ff / (16 * nwv), p.splits);
        if (nw == 16) hipLaunchKernelGGL(ffn_dw_bf16_ring16_kernel, g2, dim3(1024), ring_bytes, st, p);
        else if (nw == 82) hipLaunchKernelGGL(ffn_dw_bf16_ring8d2_kernel, g2, dim3(512), ring_bytes, st, p);
        else if (nw == 8) hipLaunchKernelGGL(ffn_dw_bf16_ring8_kernel, g2, dim3(512), ring_bytes, st, p);
        else hipLaunchKernelGGL(ffn_dw_bf16_ring_kernel, grid, dim3(256), ring_bytes, st, p);
    } else if (p.hs) {
        EGX_CHECK(!p.xg_planes || CM == CM_SPLIT, "ffn_dw: bf16 operand planes are read by the LDS-ring kernel only (unset EGX_FFN_DW_RING)");
        EGX_CHECK(CM != CM_SPLIT || p.xg_planes, "ffn_dw: the split-mode stored-operand kernel reads pre-split x1 / g planes");
        const int occ = CM == CM_SPLIT ? 2 : ffn_dw_occ(true, CM == CM_BF16);
        if (CM == CM_SPLIT) lds = (size_t)2 * 3 * 32 * 144 * sizeof(unsigned short);
        // + per-wave staging of the H / dH tiles (4 waves x 2 tensors x parts x 1 KB)
        const size_t tstage_bytes = CM == CM_BF16 ? (size_t)4 * 2 * 1024 : 0;
        static bool attr2_set = false;
        if (!attr2_set) {
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_stored_kernel<CM, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + tstage_bytes)));
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_stored_kernel<CM, 3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + tstage_bytes)));
            attr2_set = true;
        }
        if (CM == CM_SPLIT && ffn_dw_split2()) {
            static bool attr4_set = false;
            if (!attr4_set) {
                EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_split2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds / 2)));
                attr4_set = true;
            }
            // Round 6: EIGHT waves per workgroup share one staged operand image (hidden group of 128). With four, the launch moved 805 MB through
            // the CUs' address paths (24 KB of x1 / g planes + 8 KB of H / dH tiles per K-block and workgroup), 590 MB of it the planes that every
            // hidden group of a token range re-reads; with eight 510 MB and half the staging instructions per MFMA. 76.5 -> 64.0 us,
            // step 371 -> 360 us (three same-box pairs, profiles/r06_ab_ffn_dw_w8.txt); sixteen waves (one workgroup per CU) measured the same as
            // eight. EGX_FFN_DW_W8 = 0 | 16 selects the four- / sixteen-wave grids (tuning aid).
            static const int w8 = [] { const char* e = getenv("EGX_FFN_DW_W8"); return e ? atoi(e) : 8; }();
            if ((w8 == 42 || w8 == 82) && p.d_ff % 256 == 0) {      // NH = 2 variants (42: four waves, 82: eight)
                static bool attr7_set = false;
                if (!attr7_set) {
                    EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_split2h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds / 2)));
                    EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_split2w8h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds / 2)));
                    attr7_set = true;
                }
                if (w8 == 42) hipLaunchKernelGGL(ffn_dw_split2h2_kernel, dim3(p.d_ff / 128, grid.y, 2), dim3(256), lds / 2, st, p);
                else hipLaunchKernelGGL(ffn_dw_split2w8h2_kernel, dim3(p.d_ff / 256, grid.y, 2), dim3(512), lds / 2, st, p);
            } else
            if (w8 == 16 && p.d_ff % 256 == 0) {
                static bool attr6_set = false;
                if (!attr6_set) {
                    EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_split2w16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds / 2)));
                    attr6_set = true;
                }
                hipLaunchKernelGGL(ffn_dw_split2w16_kernel, dim3(p.d_ff / 256, grid.y, 2), dim3(1024), lds / 2, st, p);
            } else
            if (w8 && p.d_ff % 128 == 0) {
                static bool attr5_set = false;
                if (!attr5_set) {
                    EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_dw_split2w8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds / 2)));
                    attr5_set = true;
                }
                hipLaunchKernelGGL(ffn_dw_split2w8_kernel, dim3(p.d_ff / 128, grid.y, 2), dim3(512), lds / 2, st, p);
            } else
            hipLaunchKernelGGL(ffn_dw_split2_kernel, dim3(grid.x, grid.y, 2), dim3(256), lds / 2, st, p);
        } else if (occ == 3) hipLaunchKernelGGL((ffn_dw_stored_kernel<CM, 3>), grid, dim3(256), lds / 2 + tstage_bytes, st, p);
        else hipLaunchKernelGGL((ffn_dw_stored_kernel<CM, 2>), grid, dim3(256), lds + tstage_bytes, st, p);
    } else {
        hipLaunchKernelGGL((ffn_dw_kernel<CM, HT>), grid, dim3(256), lds, st, p);
    }
    timing_end(TIMER_FFN_DW, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

// dW1 += dH^T x1, db1 += colsum(dH), dW2 += g^T H with H, dH recomputed. `slabs` holds ffn_dw_scratch_bytes().
int ffn_dw(FfnDwParams p, int compute, float* dW1, float* db1, float* dW2, void* slabs, hipStream_t st,
           const ReducePartialsParams* rp, bool deterministic, SlabReduce* defer) {
    EGX_CHECK(p.d_ff % 128 == 0, "ffn_dw: d_ff=%d must be a multiple of 128", p.d_ff);
    EGX_CHECK(!p.hs == !p.dhs, "ffn_dw: H and dH tiles must be given together");
    int nkb = p.hs ? (p.B * FUSED_TOK_TILES + 1) / 2 : (p.N + 31) / 32;
    int splits = ffn_dw_splits((p.N + 31) / 32, (compute == CM_SPLIT && p.hs) ? 2 : ffn_dw_occ(p.hs != nullptr, compute == CM_BF16), p.d_ff,
                               compute == CM_SPLIT && p.hs && p.xg_planes && ffn_dw_split2());
    p.splits = splits;
    p.kb_per_split = cdiv(nkb, splits);
    p.splits = cdiv(nkb, p.kb_per_split);
    splits = p.splits;
    p.slab_w1 = (float*)slabs;
    p.slab_w2t = p.slab_w1 + (size_t)splits * p.d_ff * FD;
    p.slab_b1 = p.slab_w2t + (size_t)splits * p.d_ff * FD;
    int rc = compute == CM_BF16 ? launch_ffn_dw<CM_BF16>(p, st) : compute == CM_SPLIT ? launch_ffn_dw<CM_SPLIT>(p, st) : launch_ffn_dw<CM_F32>(p, st);
    if (rc) return rc;
    SlabReduce local;
    local.narr = 0; local.nslab = splits;
    SlabReduce& a = defer ? *defer : local;
    EGX_CHECK(a.narr + 3 <= SLAB_REDUCE_MAX && (a.narr == 0 || a.nslab == splits), "ffn_dw: deferred slab reductions must share the split count");
    a.nslab = splits;
    const float* sl[3] = {p.slab_w1, p.slab_w2t, p.slab_b1};
    float* out[3] = {dW1, dW2, db1};
    const size_t nn[3] = {(size_t)p.d_ff * FD, (size_t)p.d_ff * FD, (size_t)p.d_ff};
    for (int k = 0; k < 3; ++k) {
        EGX_CHECK(!out[k] || (((uintptr_t)out[k]) & 15) == 0, "ffn_dw: gradient buffers must be 16-byte aligned");
        if (!out[k]) continue;
        a.slab[a.narr] = sl[k]; a.out[a.narr] = out[k]; a.n[a.narr] = nn[k]; ++a.narr;
    }
    if (defer) return 0;
    return ffn_dw_reduce(a, rp, deterministic, st);
}

void small_dw_tail_init(SmallDwTail& t, const SlabReduce& red, const ReducePartialsParams* rp) {
    memset(&t, 0, sizeof(t));
    t.red = red;
    size_t total = 0;
    for (int k = 0; k < red.narr; ++k) total += red.n[k];
    t.slab_blocks = (unsigned)((total / 4 + 255) / 256);
    t.chunks = 1;
    if (rp) { t.rp = *rp; t.chunks = partial_chunks(rp->B); t.rp_units = cdiv(rp->P, 64) * t.chunks; }
}

int ffn_dw_reduce(const SlabReduce& a, const ReducePartialsParams* rp, bool deterministic, hipStream_t st) {
    size_t total = 0;
    for (int k = 0; k < a.narr; ++k) total += a.n[k];
    unsigned slab_blocks = (unsigned)((total / 4 + 255) / 256);
    if (rp) {
        // deterministic: ONE workgroup per 64 partial columns walks all clips (a single adder per gradient element)
        int chunks = deterministic ? 1 : partial_chunks(rp->B);
        hipLaunchKernelGGL(reduce_tail_kernel, dim3(slab_blocks + (unsigned)(cdiv(rp->P, 64) * chunks)), dim3(256), 0, st, a, *rp, slab_blocks, chunks);
    } else if (slab_blocks) {
        hipLaunchKernelGGL(reduce_slabs_add_kernel, dim3(slab_blocks), dim3(256), 0, st, a);
    }
    EGX_LAUNCH_CHECK();
    return 0;
}

__global__ void seed_advance_kernel(uint64_t* seed) { *seed = *seed * 6364136223846793005ull + 1442695040888963407ull; }
__global__ void derive_keys_kernel(uint64_t* seed, uint64_t* table, uint32_t layer0, int n, int advance) {
    uint64_t s = *seed;
    __syncthreads();        // every thread has read the old value before thread 0 replaces it
    if (advance) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        if (threadIdx.x == 0) *seed = s;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) table[i] = site_key(s, layer0 + (uint32_t)(i / 8), (uint32_t)(i % 8));
}
int derive_keys(uint64_t* seed, uint64_t* table, uint32_t layer0, int nlayer, int advance, hipStream_t st) {
    hipLaunchKernelGGL(derive_keys_kernel, dim3(1), dim3(256), 0, st, seed, table, layer0, nlayer * 8, advance);
    EGX_LAUNCH_CHECK();
    return 0;
}
int seed_advance(uint64_t* seed, hipStream_t st) {
    hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, st, seed);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- grouped small weight gradients ------------------------------------------------------------------------
// dW_in, dW_o and the K projection gradients are "G^T X" reductions over all tokens with tiny outputs; as five
// separate split-K GEMMs they cost 15 us each in launch/prologue latency. One launch covers them all: a work item
// is (problem, 64-row group, 128-column half); grid.y splits the tokens. Per 32-token K-block the block stages
// G[32][64] and X[32][128] in LDS (coalesced, next block prefetched in registers); wave w owns rows 16w..16w+15 and
// gathers its A fragment (and the 8 B fragments) transposed from the token-major tiles. Partials are atomically
// added into the zero-initialised gradient buffers (<= `splits` adders per element).
__device__ unsigned long long g_sstamps[16];
#ifdef EGX_STAMPS
#define SSTAMP(i) do { if (blockIdx.x == EGX_SSTAMP_BLOCK && threadIdx.x == 0) g_sstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#ifndef EGX_SSTAMP_BLOCK
#define EGX_SSTAMP_BLOCK 0
#endif
#else
#define SSTAMP(i) do { } while (0)
#endif
int debug_read_sstamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sstamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1; }
template <int CM, bool TAIL>
__global__ __launch_bounds__(256) void small_dw_kernel(SmallDwParams p, SmallDwTail tl) {
    constexpr int LDG = 64 + 4, LDXS = 128 + 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * (32 * LDG + 32 * LDXS)];
    SSTAMP(0);
    if constexpr (TAIL) {
        // egx_config.advance_seed == 2: this is the backward's last launch and reads no dropout key: the step's seed advances here
        if (tl.seed_advance && blockIdx.x == 0 && threadIdx.x == 0)
            *tl.seed_advance = *tl.seed_advance * 6364136223846793005ull + 1442695040888963407ull;
        // the next forward's first weight streams -> Infinity Cache (TouchList); consumed right away: this launch starts with reductions anyway
        if (tl.touch.n) touch_sink(touch_lines<256>(tl.touch, blockIdx.x, gridDim.x, threadIdx.x));
        // The FFN weight-gradient slabs and the per-clip partial rows are summed here, 1 / grid of the units per workgroup, before
        // the workgroup's own GEMM work: the reduction launch of its own cost 14 us of mostly exposed latency. (As EXTRA
        // workgroups of this launch the same units were limited to three per CU by its LDS footprint: +60 us.)
        for (unsigned u = blockIdx.x; u < tl.slab_blocks; u += gridDim.x) reduce_slabs_block(tl.red, u);
        for (int u = blockIdx.x; u < tl.rp_units; u += gridDim.x) {
            reduce_partials_block(tl.rp, u / tl.chunks, u % tl.chunks, tl.chunks);
            __syncthreads();
        }
    }
    SSTAMP(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    int pi = 0;
    while (pi + 1 < p.n && (int)blockIdx.x >= p.pr[pi + 1].first_block) ++pi;
    const SmallDwProblem& pr = p.pr[pi];
    const int local = blockIdx.x - pr.first_block;
    const int item = local / pr.splits, split = local - item * pr.splits;
    const int ncol = (pr.C + 127) / 128;
    const int row0 = (item / ncol) * 64, col0 = (item % ncol) * 128;
    const int nkb = (pr.K + 31) / 32;
    const int kb_beg = split * p.per, kb_end = min(nkb, kb_beg + p.per);
    if (kb_beg >= kb_end) return;

    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
    // operand tiles of K-block kb + 2 are requested while kb is multiplied (two register sets): one block ahead left every
    // iteration waiting a memory round trip for ~1k cycles of MFMA + split work (52 % of the wave cycles parked, 14 % MFMA busy).
    // The loads are unconditional (clamped addresses, zeroed by a select afterwards): a load under a branch is waited for at once.
    float4 pgA[2], pxA[4], pgB[2], pxB[4];
    auto gload = [&](int kb, float4 (&pg)[2], float4 (&px)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int f = tid + i * 256, row = f >> 4, c4 = (f & 15) << 2;
            int n = kb * 32 + row, c = row0 + c4;
            const bool ok = n < pr.K && c < pr.R;
            float4 v = *reinterpret_cast<const float4*>(pr.G + (size_t)(ok ? n : 0) * pr.ldg + (ok ? c : 0));
            pg[i] = ok ? v : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int f = tid + i * 256, row = f >> 5, c4 = (f & 31) << 2;
            int n = kb * 32 + row, c = col0 + c4;
            const bool ok = n < pr.K && c < pr.C;
            float4 v = *reinterpret_cast<const float4*>(pr.X + (size_t)(ok ? n : 0) * pr.ldx + (ok ? c : 0));
            px[i] = ok ? v : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](float* gt, float* xt, const float4 (&pg)[2], const float4 (&px)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { int f = tid + i * 256; *reinterpret_cast<float4*>(gt + (f >> 4) * LDG + ((f & 15) << 2)) = pg[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { int f = tid + i * 256; *reinterpret_cast<float4*>(xt + (f >> 5) * LDXS + ((f & 31) << 2)) = px[i]; }
    };
    gload(kb_beg, pgA, pxA);
    gload(kb_beg + 1 < kb_end ? kb_beg + 1 : kb_beg, pgB, pxB);
    if constexpr (CM == CM_SPLIT || CM == CM_BF16) {
        // Operands are converted (bf16: one plane) or split (f32s: three planes) ONCE per K-block while staging (single-buffered:
        // 14 / 42 KB) and read back as token-along-K fragments by the hardware-transposed ds_read_b64_tr_b16: no per-wave gather,
        // conversion or split work. (bf16 took the fp32-tile path below until round 5: eight scalar LDS reads and four converts per
        // fragment and lane, VALU : MFMA = 23, profiles/r05_pmc_c2_bf16.json.)
        constexpr int NPL = CM == CM_SPLIT ? 3 : 1;
        constexpr int LGH = 80, LXH = 144;            // halfword row strides: conflict-free transposed reads
        constexpr int GPL = 32 * LGH, XPL = 32 * LXH;
        unsigned short* gp = reinterpret_cast<unsigned short*>(lds);
        unsigned short* xp = gp + NPL * GPL;
        static_assert((3 * GPL + 3 * XPL) * 2 <= (int)sizeof(lds), "split planes must fit the staging buffer");
        typedef short s4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s4v lds_s4v;
        const int i16 = lane & 15;
        auto tr_frag = [&](const unsigned short* base, int ld, int plane_stride, int c0) {
            const unsigned short* b = base + (4 * q + (i16 >> 2)) * ld + c0 + 4 * (i16 & 3);
            Frag<CM> f;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + pl * plane_stride));
                s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(b + pl * plane_stride + 16 * ld));
                const bf16x8 w = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (CM == CM_SPLIT) f.p[pl] = w; else f.v = w;
            }
            return f;
        };
        auto put = [&](unsigned short* d, int plane_stride, const float4& v) {
            if constexpr (CM == CM_SPLIT) {
                uint32_t h0, m0, l0, h1, m1, l1;
                split_pair(v.x, v.y, h0, m0, l0);
                split_pair(v.z, v.w, h1, m1, l1);
                *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(d + plane_stride) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(d + 2 * plane_stride) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
            }
        };
        auto block = [&](int kb, float4 (&pg)[2], float4 (&px)[4]) {
            if (kb > kb_beg) __syncthreads();           // everyone is done reading the previous K-block
#pragma unroll
            for (int i = 0; i < 2; ++i) { int f = tid + i * 256; put(gp + (f >> 4) * LGH + ((f & 15) << 2), GPL, pg[i]); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { int f = tid + i * 256; put(xp + (f >> 5) * LXH + ((f & 31) << 2), XPL, px[i]); }
            __syncthreads();
            gload(kb + 2 < kb_end ? kb + 2 : kb, pg, px);      // (past the end: a re-read that nobody uses)
            Frag<CM> a = tr_frag(gp, LGH, GPL, wave * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                Frag<CM> b = tr_frag(xp, LXH, XPL, j * 16);
                mma<CM>(acc[j], a, b);
            }
        };
        SSTAMP(2);
        for (int kb = kb_beg; kb < kb_end; kb += 2) {
            block(kb, pgA, pxA);
            if (kb == kb_beg) SSTAMP(3);
            if (kb + 1 < kb_end) block(kb + 1, pgB, pxB);
        }
        SSTAMP(4);
    } else {
    auto block = [&](int kb, int cur, float4 (&pg)[2], float4 (&px)[4]) {
        float* gt = lds + cur * (32 * LDG + 32 * LDXS);
        float* xt = gt + 32 * LDG;
        lstore(gt, xt, pg, px);
        __syncthreads();
        gload(kb + 2 < kb_end ? kb + 2 : kb, pg, px);
        Frag<CM> a = gather_frag<CM>(gt, wave * 16 + r, 0, q, 31, LDG);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            Frag<CM> b = gather_frag<CM>(xt, j * 16 + r, 0, q, 31, LDXS);
            mma<CM>(acc[j], a, b);
        }
    };
    for (int kb = kb_beg; kb < kb_end; kb += 2) {
        block(kb, 0, pgA, pxA);
        if (kb + 1 < kb_end) block(kb + 1, 1, pgB, pxB);
    }
    }
    if (p.slabs) {      // deterministic: dense tile per workgroup, summed in split order by small_dw_reduce_kernel
        float* tile = p.slabs + (size_t)blockIdx.x * (64 * 128);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(wave * 16 + 4 * q + e) * 128 + j * 16 + r] = acc[j][e];
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int row = row0 + wave * 16 + 4 * q + e, col = col0 + j * 16 + r;
            if (row < pr.R && col < pr.C) atomicAdd(pr.out + (size_t)row * pr.C + col, acc[j][e]);
        }
    SSTAMP(5);
    __builtin_amdgcn_s_waitcnt(0);
    SSTAMP(6);
}

// out[row][col] += sum over the splits of one (problem, item) of its slab tiles, in split order. grid.x = work items, 8 blocks each.
__global__ __launch_bounds__(256) void small_dw_reduce_kernel(SmallDwParams p) {
    const int unit = blockIdx.x >> 3, part = blockIdx.x & 7;      // part: 8 rows of the 64 x 128 tile
    int pi = 0, base = 0;
    for (;; ++pi) {
        const int items = ((p.pr[pi].R + 63) / 64) * ((p.pr[pi].C + 127) / 128);
        if (unit < base + items || pi + 1 >= p.n) break;
        base += items;
    }
    const SmallDwProblem& pr = p.pr[pi];
    const int item = unit - base;
    const int ncol = (pr.C + 127) / 128;
    const int row0 = (item / ncol) * 64, col0 = (item % ncol) * 128;
    const int nkb = (pr.K + 31) / 32;
    const int r = part * 8 + (threadIdx.x >> 5), c = (threadIdx.x & 31) * 4;
    float4 s = make_float4(0, 0, 0, 0);
    for (int sp = 0; sp < pr.splits; ++sp) {
        if (sp * p.per >= nkb) break;                               // workgroups without K-blocks wrote nothing
        const float* tile = p.slabs + (size_t)(pr.first_block + item * pr.splits + sp) * (64 * 128);
        float4 v = *reinterpret_cast<const float4*>(tile + r * 128 + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int row = row0 + r, col = col0 + c;
    if (row < pr.R && col < pr.C) {                                 // C % 4 == 0
        float4* dst = reinterpret_cast<float4*>(pr.out + (size_t)row * pr.C + col);
        float4 o = *dst;
        *dst = make_float4(o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w);
    }
}

int small_dw(SmallDwParams& p, int compute, hipStream_t st, void* slabs, size_t slab_bytes, const SmallDwTail* tail, bool reduce_here) {
    int items[SMALL_DW_MAX], nkb[SMALL_DW_MAX], total_items = 0;
    for (int i = 0; i < p.n; ++i) {
        EGX_CHECK(p.pr[i].R % 4 == 0 && p.pr[i].C % 4 == 0 && p.pr[i].ldg % 4 == 0 && p.pr[i].ldx % 4 == 0,
                  "small_dw: problem %d needs 4-aligned dimensions", i);
        items[i] = cdiv(p.pr[i].R, 64) * cdiv(p.pr[i].C, 128);
        nkb[i] = cdiv(p.pr[i].K, 32);
        total_items += items[i];
    }
    if (!total_items) return 0;
    // the same number of K-blocks per workgroup for every problem (so all workgroups finish together), as small as
    // two workgroups per CU allow
    int per = 1;
    for (;; ++per) {
        int blocks = 0;
        for (int i = 0; i < p.n; ++i) blocks += items[i] * cdiv(nkb[i], per);
        if (blocks <= 512 || per >= 4096) break;
    }
    if (const char* e = getenv("EGX_SMALL_DW_PER")) per = max(1, atoi(e));     // tuning aid
    p.per = per;
    int blocks = 0;
    for (int i = 0; i < p.n; ++i) {
        p.pr[i].first_block = blocks;
        p.pr[i].splits = cdiv(nkb[i], per);
        blocks += items[i] * p.pr[i].splits;
    }
    p.slabs = nullptr;
    if (slabs) {
        EGX_CHECK((size_t)blocks * 64 * 128 * sizeof(float) <= slab_bytes, "small_dw: deterministic mode needs %zu bytes of slab scratch, got %zu",
                  (size_t)blocks * 64 * 128 * sizeof(float), slab_bytes);
        p.slabs = (float*)slabs;
    }
    if (tail) {
        EGX_CHECK(!p.slabs, "small_dw: the reduction tail rides only in the atomic (non-deterministic) variant");
        if (compute == CM_BF16) hipLaunchKernelGGL((small_dw_kernel<CM_BF16, true>), dim3(blocks), dim3(256), 0, st, p, *tail);
        else if (compute == CM_SPLIT) hipLaunchKernelGGL((small_dw_kernel<CM_SPLIT, true>), dim3(blocks), dim3(256), 0, st, p, *tail);
        else hipLaunchKernelGGL((small_dw_kernel<CM_F32, true>), dim3(blocks), dim3(256), 0, st, p, *tail);
    } else {
        static SmallDwTail none;
        if (compute == CM_BF16) hipLaunchKernelGGL((small_dw_kernel<CM_BF16, false>), dim3(blocks), dim3(256), 0, st, p, none);
        else if (compute == CM_SPLIT) hipLaunchKernelGGL((small_dw_kernel<CM_SPLIT, false>), dim3(blocks), dim3(256), 0, st, p, none);
        else hipLaunchKernelGGL((small_dw_kernel<CM_F32, false>), dim3(blocks), dim3(256), 0, st, p, none);
    }
    if (p.slabs && reduce_here) hipLaunchKernelGGL(small_dw_reduce_kernel, dim3(total_items * 8), dim3(256), 0, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- round 6: every cross-workgroup sum of the per-clip backward in ONE fixed-order launch -------------------------------------------------
// Until round 5 the grouped small weight gradients left as 8 192 float atomics per workgroup (16 MB of added bytes per C2 step at the ~1.3 TB/s
// the memory-side atomic units take: 12 us of the launch) with the FFN slab sums and the per-clip partial rows riding in front of them (10 us);
// the deterministic mode ran slab tiles and three slow fixed-order passes instead (25 + 16 + 33 us). Now small_dw always writes its tiles and
// this launch sums, each in a fixed order: (a) the tiles of every (problem, item) over its splits, (b) the FFN weight-gradient slabs, (c) the
// per-clip partial rows (16 clip lanes per column, LDS tree) — 51 MB read, all of it freshly written; it also carries the step's seed advance and
// the next forward's weight prefetch. The per-clip backward is bit-reproducible in every mode as a by-product.
struct TailReduceParams {
    SmallDwParams sp;           // tile sums (sp.slabs = the tiles small_dw wrote); sp.n == 0: none
    SlabReduce red;             // FFN slab sums; red.narr == 0: none
    ReducePartialsParams rp;    // partial rows; rp.n == 0: none
    unsigned tile_blocks, slab_blocks, part_blocks;
    uint64_t* seed_advance;
    TouchList touch;
};
__global__ __launch_bounds__(256) void tail_reduce_kernel(TailReduceParams t) {
    const unsigned b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0 && t.seed_advance)
        *t.seed_advance = *t.seed_advance * 6364136223846793005ull + 1442695040888963407ull;
    if (t.touch.n) touch_sink(touch_lines<256>(t.touch, blockIdx.x, gridDim.x, threadIdx.x));
    if (b < t.part_blocks) {
        // (c) partial rows first (chains of dependent loads that then run under the streams): 16 columns x 16 clip lanes per workgroup
        __shared__ float red[16][17];
        const ReducePartialsParams& rp = t.rp;
        const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
        const int j = (int)b * 16 + c;
        float s = 0.f;
        if (j < rp.P) {
            int clip = g;
            for (; clip + 7 * 16 < rp.B; clip += 8 * 16) {      // eight loads in flight
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = rp.partials[(size_t)(clip + k * 16) * rp.P + j];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            for (; clip < rp.B; clip += 16) s += rp.partials[(size_t)clip * rp.P + j];
        }
        red[g][c] = s;
        __syncthreads();
        if (g == 0 && j < rp.P) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += red[k][c];
            float* dst = nullptr;
            for (int i = 0; i < rp.n; ++i)
                if (j >= rp.d[i].off && j < rp.d[i].off + rp.d[i].len) dst = rp.d[i].dst + (j - rp.d[i].off);
            if (dst) *dst += tot;        // the only adder of this element in this launch
        }
        return;
    }
    if (b < t.part_blocks + t.tile_blocks) {
        // (a) tiles of the grouped small weight gradients: (work item, 8 of its 64 rows) per workgroup, splits in order, eight loads in flight
        const SmallDwParams& p = t.sp;
        const unsigned u = b - t.part_blocks;
        const int unit = (int)(u >> 3), part = (int)(u & 7);
        int pi = 0, base = 0;
        for (;; ++pi) {
            const int items = ((p.pr[pi].R + 63) / 64) * ((p.pr[pi].C + 127) / 128);
            if (unit < base + items || pi + 1 >= p.n) break;
            base += items;
        }
        const SmallDwProblem& pr = p.pr[pi];
        const int item = unit - base;
        const int ncol = (pr.C + 127) / 128;
        const int row0 = (item / ncol) * 64, col0 = (item % ncol) * 128;
        const int nkb = (pr.K + 31) / 32;
        const int nsp = min(pr.splits, (nkb + p.per - 1) / p.per);      // workgroups without K-blocks wrote nothing
        const int r = part * 8 + (threadIdx.x >> 5), c = (threadIdx.x & 31) * 4;
        const float* tile0 = p.slabs + (size_t)(pr.first_block + item * pr.splits) * (64 * 128) + r * 128 + c;
        float4 s = make_float4(0, 0, 0, 0);
        int sp = 0;
        for (; sp + 8 <= nsp; sp += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(tile0 + (size_t)(sp + k) * (64 * 128));
#pragma unroll
            for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
        }
        for (; sp < nsp; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(tile0 + (size_t)sp * (64 * 128));
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int row = row0 + r, col = col0 + c;
        if (row < pr.R && col < pr.C) {                                 // C % 4 == 0
            float4* dst = reinterpret_cast<float4*>(pr.out + (size_t)row * pr.C + col);
            const float4 o = *dst;
            *dst = make_float4(o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w);
        }
        return;
    }
    // (b) FFN weight-gradient slabs
    reduce_slabs_block(t.red, b - t.part_blocks - t.tile_blocks);
}

// small_dw with tiles (sp must have been through small_dw(sp, ..., slabs)) + the fixed-order tail. Any of the three parts may be empty.
int tail_reduce(const SmallDwParams* sp, const SlabReduce* red, const ReducePartialsParams* rp, uint64_t* seed_advance_ptr, const TouchList* touch, hipStream_t st) {
    TailReduceParams t;
    memset(&t, 0, sizeof(t));
    if (sp && sp->n && sp->slabs) {
        t.sp = *sp;
        int items = 0;
        for (int i = 0; i < sp->n; ++i) items += cdiv(sp->pr[i].R, 64) * cdiv(sp->pr[i].C, 128);
        t.tile_blocks = (unsigned)items * 8;
    }
    if (red && red->narr) {
        t.red = *red;
        size_t total = 0;
        for (int k = 0; k < red->narr; ++k) total += red->n[k];
        t.slab_blocks = (unsigned)((total / 4 + 255) / 256);
    }
    if (rp && rp->n) { t.rp = *rp; t.part_blocks = (unsigned)cdiv(rp->P, 16); }
    t.seed_advance = seed_advance_ptr;
    if (touch) t.touch = *touch;
    const unsigned blocks = t.tile_blocks + t.slab_blocks + t.part_blocks;
    if (!blocks) return seed_advance_ptr ? seed_advance(seed_advance_ptr, st) : 0;
    hipLaunchKernelGGL(tail_reduce_kernel, dim3(blocks), dim3(256), 0, st, t);
    EGX_LAUNCH_CHECK();
    return 0;
}

__device__ unsigned g_stolen_bwd;      // slices the backward's waiting workgroups computed themselves
long long slices_stolen_bwd(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_stolen_bwd), sizeof(v)) != hipSuccess) return -1;
    if (reset) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stolen_bwd), &z, sizeof(z)); }
    return v;
}
__device__ unsigned long long g_bstamps[32];
#ifdef EGX_STAMPS
#define BSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_bstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
int debug_read_bstamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * (n > 32 ? 32 : n)) == hipSuccess ? 0 : 1;
}

// ---- per-clip backward --------------------------------------------------------------------------------
// One workgroup per clip. Six token-major LDS blocks (48 x 132 fp32 each) are rotated through the roles noted at
// each phase; small per-head softmax statistics live behind them. Everything that another kernel needs
// (operands of the weight-gradient GEMMs) is written to HBM exactly once.
// TILED (48 < S <= 512, see FusedBwdParams): the workgroup is one 48-token tile and the kernel is cut at the attention backward
// (tiled_attn_bwd): a launch runs [P11 - P12 of layer l_front on the dQ | dK | dV rows that kernel left] + [P1 - P7 of layer
// l_back, leaving d(attention output) and the residual gradient in HBM], or ends with the token-preparation backward.
// CUT (round 5, ffn_cut.hip): the launch runs LayerNorm1 backward .. the in-projection input gradient of layer p.cut_layer on the dy1 rows
// ffn_bwd_kernel left, and leaves d(layer input) in p.dxin for the layer below — or, for layer 0, runs on into the token-preparation backward.
template <int CM, bool TILED, int DH, bool SLICED = false, bool CUT = false>      // SLICED: see fused_fwd_kernel
__global__ __launch_bounds__(256, 1) void fused_bwd_kernel(FusedBwdParams p) {
    static_assert(!(TILED && SLICED), "the tiled launches are not sliced");
    static_assert(!(CUT && (TILED || SLICED)), "the cut launches are neither tiled nor sliced");
    constexpr int HPW = FDH / DH, NHEAD = FH * HPW, NCT = DH / 16;      // heads per wave, heads, 16-channel tiles per head
    constexpr int NT = 3;
    constexpr int SP = NT * 16;
    constexpr int BLK = SP * LDX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Gs = lds;                 // dY of the current layer (rotates with B2)
    float* B1 = lds + 1 * BLK;
    float* B2 = lds + 2 * BLK;
    float* B3 = lds + 3 * BLK;
    float* B4 = lds + 4 * BLK;
    float* B5 = lds + 5 * BLK;
    float* stat = lds + 6 * BLK;     // [FH][3][SP] softmax max / 1/sum / delta per query
    float* PSB = stat + NHEAD * 3 * SP;     // LayerNorm weight vectors staged in LDS (round 6, ln_bwd_rows_lds): row 0 ln_w, row 1 norm1_w of the current layer

    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63, r = lane & 15, q = lane >> 4;
    int clip_ = blockIdx.x, slice_ = 0;     // TILED: the tile (index of every 48-row grid)
    if constexpr (SLICED) slice_map(p.n_slices, clip_, slice_);      // sliced mode (FusedFwdParams): n workgroups per clip
    const int clip = clip_, slice = slice_;
    const int n_slices = SLICED ? p.n_slices : 1;
    int S, c_real, t0;
    size_t tok0;                            // global index of the first token (row of the dense arrays, dropout row key)
    if constexpr (TILED) {
        c_real = clip / p.tpc;
        t0 = (clip - c_real * p.tpc) * 48;
        S = min(48, p.S_clip - t0);
        tok0 = (size_t)c_real * p.S_clip + t0;
    } else {
        c_real = clip; t0 = 0; S = p.S; tok0 = (size_t)clip * S;
    }
    if (!CUT && p.zero_buf) {       // the caller's flat gradient buffer: every accumulation into it happens in later launches (cut mode: ffn_bwd_kernel zeroes it)
        const size_t n4 = p.zero_n / 4, per = (n4 + gridDim.x - 1) / gridDim.x;
        const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
        for (size_t k = b0 + threadIdx.x; k < b1; k += 256) reinterpret_cast<float4*>(p.zero_buf)[k] = make_float4(0, 0, 0, 0);
    }
    if constexpr (SLICED) { if (clip >= p.B || ((p.slice_drop >> slice) & 1)) return; }        // (the grid is round_up(B, 8) * n_slices)

    BSTAMP(12);
    float* part = p.partials + (size_t)clip * p.P;
    // hipcc hoists every lane-constant fragment address of every phase to kernel entry and then spills them around
    // the phases (1 KB of scratch per lane). Re-deriving the lane indices behind an opaque asm at each phase start
    // keeps address arithmetic local to the phase that needs it.
#define EGX_PHASE()                                              \
    do {                                                         \
        int t_ = threadIdx.x;                                    \
        asm volatile("" : "+v"(t_));                             \
        lane = t_ & 63; r = lane & 15; q = lane >> 4;            \
    } while (0)

    const bool dev_seed = p.seed_ptr != nullptr;
    const uint64_t seed_dev = dev_seed ? *p.seed_ptr : 0ull;
    const uint64_t pos_key = dev_seed ? site_key(seed_dev, 0, SITE_POS) : p.pos_key;
    // Saved (S, 128) blocks come in through registers: requested one phase (or more) before their LDS block is free, written
    // when it is. The requests are unconditional (clamped) loads of f32x4 VALUES: a load under a branch is waited for at the
    // join, and float4 struct copies become memcpys through a scratch array.
    constexpr int BPF = SP * (FD / 4) / 256;        // f32x4 per thread of one block
    static_assert(SP * (FD / 4) % 256 == 0, "a block must divide among the threads");
    auto blk_request = [&](f32x4 (&v)[BPF], const float* src) {
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));        // keep the address arithmetic here (see EGX_PHASE)
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
        const int n = S * (FD / 4);
#pragma unroll
        for (int k = 0; k < BPF; ++k) { const int i = t_ + 256 * k; v[k] = s4[i < n ? i : n - 1]; }
    };
    auto blk_store = [&](const f32x4 (&v)[BPF], float* dst, float* dst2) {      // rows >= S stay as they are
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int k = 0; k < BPF; ++k) {
            const int i = t_ + 256 * k, row = i >> 5, c = (i & 31) << 2;
            if (i < S * (FD / 4)) {
                *reinterpret_cast<f32x4*>(dst + row * LDX + c) = v[k];
                if (dst2) *reinterpret_cast<f32x4*>(dst2 + row * LDX + c) = v[k];
            }
        }
    };
    f32x4 pf_a[BPF], pf_b[BPF];       // res2 / res1 of the layer about to be processed (pf_a: `pre` for the token preparation)
    // dense (S, 128) block of saved residual sums: slot 2 l = res1, 2 l + 1 = res2 of layer l
    auto res_ptr = [&](int slot) -> const float* {
        return TILED ? p.saved_res + ((size_t)slot * p.Ntok + tok0) * FD : p.saved_res + ((size_t)slot * p.B + clip) * S * FD;
    };
    const int l_first = CUT ? p.cut_layer : TILED ? p.l_back : p.n_layers - 1;      // the layer whose P1 - P7 this launch runs first (TILED: < 0 = none)
    // (CUT: dy1 in place of res2 — it goes to Gs, the sum LayerNorm1's backward starts from)
    // weight streams of a later launch -> Infinity Cache (TouchList): the oldest loads of this launch, consumed with the blocks below
    Touched tch = {{0, 0, 0, 0}};
    if constexpr (!TILED) { if (p.touch.n) tch = touch_lines<256>(p.touch, blockIdx.x, gridDim.x, tid); }
    blk_request(pf_a, CUT ? p.dy1 + tok0 * FD : l_first >= 0 ? res_ptr(2 * l_first + 1) : p.saved_pre + tok0 * FD);
    blk_request(pf_b, l_first >= 0 ? res_ptr(2 * l_first) : p.saved_pre + tok0 * FD);
    const int ps_g = tid >> 5, ps_c = (tid & 31) << 2;
    const f32x4 ln0_v = *reinterpret_cast<const f32x4*>(p.ln_w + ps_c);
    f32x4 n1_v = *reinterpret_cast<const f32x4*>(p.layer[l_first >= 0 ? l_first : 0].norm1_w + ps_c);
    static_assert((6 * BLK) % 4 == 0, "the blocks are zeroed in 16-byte pieces");
    for (int i = tid; i < 6 * BLK / 4; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0, 0, 0, 0};
    if (ps_g == 0) *reinterpret_cast<f32x4*>(PSB + ps_c) = ln0_v;
    __syncthreads();
    if constexpr (TILED) {
        if (p.l_front >= 0) {
            // dQ | dK | dV rows of this tile (tiled_attn_bwd) -> B4 | B5 | Gs, the residual gradient of layer l_front -> B1; then
            // P11 - P12 of that layer exactly as below
            const FusedBwdLayer& w = p.layer[p.l_front];
            float* pl = part + p.l_front * FUSED_P_LAYER;
            {
                const f32x4* src = reinterpret_cast<const f32x4*>(w.dqkv_out + tok0 * (3 * FD));
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(p.dres + tok0 * FD);
                for (int i = tid; i < S * 96; i += 256) {
                    const int row = i / 96, c = (i - row * 96) * 4;
                    float* dst = c < FD ? B4 + c : (c < 2 * FD ? B5 + (c - FD) : Gs + (c - 2 * FD));
                    *reinterpret_cast<f32x4*>(dst + row * LDX) = src[i];
                }
                for (int i = tid; i < S * (FD / 4); i += 256) {
                    const int row = i >> 5, c = (i & 31) << 2;
                    *reinterpret_cast<f32x4*>(B1 + row * LDX + c) = rsrc[i];
                }
            }
            __syncthreads();
            PackW<CM, 2, 4> wi_pf;
            pack_issue(wi_pf, w.in_proj_wtp, wave * 2, 12, 0);
            if (tid < 128) {
                pl[768 + tid] = colsum_lds(B4, 0, S, tid);
                pl[768 + 256 + tid] = colsum_lds(Gs, 0, S, tid);
            } else {
                pl[768 + 128 + (tid - 128)] = colsum_lds(B5, 0, S, tid - 128);
            }
            EGX_PHASE();
            {
                f32x4 acc[2][NT];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
                PackW<CM, 2, 4> wi2, wi3;
                pack_issue(wi2, w.in_proj_wtp, wave * 2, 12, 4);
                gemm_packed<CM, 2, NT, 4>(acc, wi_pf, B4, r, q);
                pack_issue(wi3, w.in_proj_wtp, wave * 2, 12, 8);
                gemm_packed<CM, 2, NT, 4>(acc, wi2, B5, r, q);
                gemm_packed<CM, 2, NT, 4>(acc, wi3, Gs, r, q);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        int tok = t * 16 + r;
                        int c = (wave * 2 + i) * 16 + 4 * q;
                        float4 rs = *reinterpret_cast<const float4*>(B1 + tok * LDX + c);
                        float4 o = make_float4(acc[i][t][0] + rs.x, acc[i][t][1] + rs.y, acc[i][t][2] + rs.z, acc[i][t][3] + rs.w);
                        if (tok >= S) o = make_float4(0, 0, 0, 0);
                        *reinterpret_cast<float4*>(B2 + tok * LDX + c) = o;
                    }
            }
            __syncthreads();
            { float* t = Gs; Gs = B2; B2 = t; }
        } else if (p.head.n_out > 0) {
            // Tiled mode with the pooled head (round 5): d(tokens) = d(pooled) / S_clip on every row of the tile, from the token mean
            // the forward saved — pool_head_bwd as a launch of its own (9.5 us at 25 clips: one workgroup per clip), the dense d(tokens)
            // rows and the fill of the caller's gradient buffer in front of its atomics are gone. Every tile of a clip repeats the
            // 128-wide head backward (wave 0); the head's parameter gradients leave in the partial row of the clip's FIRST tile.
            float* hp = part + p.head_off;
            float* pooled = B2;        // [0,128): pooled; [128,256): d(pooled) / S_clip (B2 is all zero here and is zeroed again below)
            if (tid < FD) pooled[tid] = p.pooled[(size_t)c_real * FD + tid];
            __syncthreads();
            if (wave == 0) {
                const float first = t0 == 0 ? 1.f : 0.f;
                float2 x = *reinterpret_cast<float2*>(pooled + 2 * lane);
                float mean = wsum(x.x + x.y) * (1.f / FD);
                float xh0 = x.x - mean, xh1 = x.y - mean;
                float rstd = rsqrtf(wsum(xh0 * xh0 + xh1 * xh1) * (1.f / FD) + p.eps);
                xh0 *= rstd; xh1 *= rstd;
                float2 lw = *reinterpret_cast<const float2*>(p.head.ln_w + 2 * lane);
                float2 lb = *reinterpret_cast<const float2*>(p.head.ln_b + 2 * lane);
                float y0 = xh0 * lw.x + lb.x, y1 = xh1 * lw.y + lb.y;
                float d0 = 0.f, d1 = 0.f;
                const float dl_scale = p.d_logits_scale ? *p.d_logits_scale : 1.f;
                for (int o = 0; o < p.head.n_out; ++o) {
                    float go = p.d_logits[(size_t)c_real * p.head.n_out + o] * dl_scale;
                    float2 wv = *reinterpret_cast<const float2*>(p.head.W + (size_t)o * FD + 2 * lane);
                    d0 += go * wv.x; d1 += go * wv.y;
                    *reinterpret_cast<float2*>(hp + 256 + FUSED_HEAD_MAX_OUT + o * FD + 2 * lane) = make_float2(first * go * y0, first * go * y1);
                    if (lane == 0) hp[256 + o] = first * go;
                }
                *reinterpret_cast<float2*>(hp + 2 * lane) = make_float2(first * d0 * xh0, first * d1 * xh1);      // d(head ln_w)
                *reinterpret_cast<float2*>(hp + 128 + 2 * lane) = make_float2(first * d0, first * d1);            // d(head ln_b)
                float g0 = d0 * lw.x, g1 = d1 * lw.y;
                float s1 = wsum(g0 + g1) * (1.f / FD);
                float s2 = wsum(g0 * xh0 + g1 * xh1) * (1.f / FD);
                float inv_s = 1.f / (float)p.S_clip;
                *reinterpret_cast<float2*>(pooled + 128 + 2 * lane) =
                    make_float2(rstd * (g0 - s1 - xh0 * s2) * inv_s, rstd * (g1 - s1 - xh1 * s2) * inv_s);
            }
            __syncthreads();
            for (int i = tid; i < S * (FD / 4); i += 256) {
                int row = i >> 5, c = (i & 31) << 2;
                *reinterpret_cast<float4*>(Gs + row * LDX + c) = *reinterpret_cast<const float4*>(pooled + 128 + c);
            }
            __syncthreads();
            pooled[tid] = 0.f;
        } else {
            for (int i = tid; i < S * (FD / 4); i += 256) {
                int row = i >> 5, c = (i & 31) << 2;
                *reinterpret_cast<float4*>(Gs + row * LDX + c) = *reinterpret_cast<const float4*>(p.d_tokens + (tok0 + row) * FD + c);
            }
        }
        if (l_first >= 0) {
            blk_store(pf_a, B1, nullptr);
            blk_store(pf_b, B5, nullptr);
        }
    }
    if constexpr (CUT) {        // dy1 -> Gs (B1 .. B4 stay zero: P5 adds them), res1 -> B5
        blk_store(pf_a, Gs, nullptr);
        blk_store(pf_b, B5, nullptr);
        touch_sink(tch);
        // x1 = LayerNorm1(res1) is an operand of the FFN weight gradient; without operand planes from the forward (exact-fp32 mode) it is
        // rebuilt here, as P3 of the one-launch kernel does
        if (!(CM != CM_F32 && p.xg_planes)) {
            const FusedBwdLayer& wc = p.layer[p.cut_layer];
            __syncthreads();
            ln_rows(B5, S, wc.norm1_w, wc.norm1_b, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
                store32(wc.x1_out + (tok0 + row) * FD + c0, y);
            });
        }
    }
    // res2 -> B1 (LayerNorm2 backward; with the fused head also -> Gs, normalised in place below), res1 -> B5 (P3 / P5)
    if constexpr (!TILED && !CUT) {
    blk_store(pf_a, B1, (p.head.n_out > 0 || p.tce_W) ? Gs : nullptr);
    blk_store(pf_b, B5, nullptr);
    touch_sink(tch);
    if (p.head.n_out > 0) {
        // Fused pooled head backward: rebuild the last layer's output tokens y = LN2(res2), pool them, run the head
        // forward/backward for this clip (wave 0) and broadcast d(tokens) = d(pooled) / S into Gs.
        const FusedBwdLayer& wl = p.layer[p.n_layers - 1];
        float* hp = part + p.head_off;
        __syncthreads();
        ln_rows(Gs, S, wl.norm2_w, wl.norm2_b, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
            store32(Gs + row * LDX + c0, y);
        });
        __syncthreads();
        float* pooled = B2;        // [0,128): pooled; [128,256): d(pooled)   (B1 holds res2 for LayerNorm2 backward)
        if (tid < FD) pooled[tid] = colsum_lds(Gs, 0, S, tid) * (1.f / (float)S);
        __syncthreads();
        if (wave == 0) {
            float2 x = *reinterpret_cast<float2*>(pooled + 2 * lane);
            float mean = wsum(x.x + x.y) * (1.f / FD);
            float xh0 = x.x - mean, xh1 = x.y - mean;
            float rstd = rsqrtf(wsum(xh0 * xh0 + xh1 * xh1) * (1.f / FD) + p.eps);
            xh0 *= rstd; xh1 *= rstd;
            float2 lw = *reinterpret_cast<const float2*>(p.head.ln_w + 2 * lane);
            float2 lb = *reinterpret_cast<const float2*>(p.head.ln_b + 2 * lane);
            float y0 = xh0 * lw.x + lb.x, y1 = xh1 * lw.y + lb.y;
            float d0 = 0.f, d1 = 0.f;
            const float dl_scale = p.d_logits_scale ? *p.d_logits_scale : 1.f;
            for (int o = 0; o < p.head.n_out; ++o) {
                float go = p.d_logits[(size_t)clip * p.head.n_out + o] * dl_scale;
                float2 wv = *reinterpret_cast<const float2*>(p.head.W + (size_t)o * FD + 2 * lane);
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
        for (int i = tid; i < S * (FD / 4); i += 256) {
            int row = i >> 5, c = (i & 31) << 2;
            *reinterpret_cast<float4*>(Gs + row * LDX + c) = *reinterpret_cast<const float4*>(pooled + 128 + c);
        }
        __syncthreads();
    } else if (p.tce_W) {
        // egx_token_ce: the forward left d loss / d logits of the clip's out_T token rows. Rebuild the tokens y = LN2(res2) (for d W), leave the clip's
        // partial d b / d W rows in the head section of its partial row, then d tokens = g * d_logits W into Gs (rows >= out_T: no upstream gradient)
        const FusedBwdLayer& wl = p.layer[p.n_layers - 1];
        float* hp = part + p.head_off;
        const int C = p.tce_C;
        __syncthreads();
        ln_rows(Gs, S, wl.norm2_w, wl.norm2_b, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
            store32(Gs + row * LDX + c0, y);
        });
        float* gos = B2;           // [row][8]: g * d_logits
        const float g = p.d_logits_scale ? *p.d_logits_scale : 1.f;
        for (int i = tid; i < p.out_T * 8; i += 256) {
            const int row = i >> 3, o = i & 7;
            gos[i] = o < C ? p.tce_dlogits[((size_t)clip * p.out_T + row) * C + o] * g : 0.f;
        }
        __syncthreads();
        if (tid < FD) {
            for (int o = 0; o < C; ++o) {
                float a = 0.f;
                for (int row = 0; row < p.out_T; ++row) a += gos[row * 8 + o] * Gs[row * LDX + tid];
                hp[256 + FUSED_HEAD_MAX_OUT + o * FD + tid] = a;
            }
        } else if (tid < FD + C) {
            const int o = tid - FD;
            float a = 0.f;
            for (int row = 0; row < p.out_T; ++row) a += gos[row * 8 + o];
            hp[256 + o] = a;
        }
        __syncthreads();        // the token rows are consumed
        for (int i = tid; i < S * (FD / 4); i += 256) {
            const int row = i >> 5, c = (i & 31) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.out_T)
                for (int o = 0; o < C; ++o) {
                    const float go = gos[row * 8 + o];
                    const float4 wv = *reinterpret_cast<const float4*>(p.tce_W + (size_t)o * FD + c);
                    v.x += go * wv.x; v.y += go * wv.y; v.z += go * wv.z; v.w += go * wv.w;
                }
            *reinterpret_cast<float4*>(Gs + row * LDX + c) = v;
        }
        __syncthreads();
    } else {        // rows >= out_T carry no upstream gradient (LDS is zero there)
        for (int i = tid; i < p.out_T * (FD / 4); i += 256) {
            int row = i >> 5, c = (i & 31) << 2;
            *reinterpret_cast<float4*>(Gs + row * LDX + c) = *reinterpret_cast<const float4*>(p.d_tokens + ((size_t)clip * p.out_T + row) * FD + c);
        }
    }
    }       // !TILED && !CUT

    for (int l = l_first; l >= 0; --l) {
        const FusedBwdLayer& w = p.layer[l];
        const uint64_t k_attn = dev_seed ? site_key(seed_dev, l, SITE_ATTN) : w.attn_key;
        const uint64_t k_res1 = dev_seed ? site_key(seed_dev, l, SITE_RES1) : w.res1_key;
        const uint64_t k_res2 = dev_seed ? site_key(seed_dev, l, SITE_RES2) : w.res2_key;
        float* pl = part + l * FUSED_P_LAYER;

        BSTAMP(0);
        if (l != l_first) n1_v = *reinterpret_cast<const f32x4*>(w.norm1_w + ps_c);     // (the first layer's was requested at kernel entry)
        // P1: res2 -> B1, res1 -> B5: requested during the previous layer's P11 / P12 (the last layer's before the LDS zero fill)
        if (!TILED && !CUT && l != p.n_layers - 1) {
            blk_store(pf_a, B1, nullptr);
            blk_store(pf_b, B5, nullptr);
        }
        __syncthreads();
        if constexpr (!CUT) {       // P2 - P4 (LayerNorm2 backward, FFN input gradient): ffn_bwd_kernel's in cut mode
        // P2: LayerNorm2 backward. B1 <- d_res2 (in place), B3 <- dY * xhat, B2 <- g2 = d_res2 .* dropout2 mask
        ln_bwd_rows(S, w.norm2_w, p.eps,
            [&](int row, int c0, float (&dy)[32], float (&x)[32]) { load32(Gs + row * LDX + c0, dy); load32(B1 + row * LDX + c0, x); },
            [&](int row, int c0, float (&dy)[32], float (&dx)[32], float (&dyx)[32]) {
                store32(B1 + row * LDX + c0, dx);
                store32(B3 + row * LDX + c0, dyx);
                if (w.res_thresh) {
                    uint32_t orow = (uint32_t)(tok0 + row);
#pragma unroll
                    for (int j = 0; j < 32; ++j) dx[j] *= drop_scale(k_res2, orow, (uint32_t)(c0 + j), w.res_thresh, w.drop_inv);
                }
                store32(B2 + row * LDX + c0, dx);
            });
        __syncthreads();
        if (CM == CM_BF16 && p.xg_planes) store_block_bf16(reinterpret_cast<unsigned short*>(w.g2_out) + (size_t)clip * FUSED_TOK_PAD * FD, B2, S);
        else if (!(CM == CM_SPLIT && p.xg_planes)) store_block(w.g2_out + tok0 * FD, B2, S);
        BSTAMP(1);
        // P3: column sums (norm2_w, norm2_b, lin2_b partials); res1 -> B4 and LayerNorm1 forward in place (x1)
        if (tid < 128) {
            pl[0 + tid] = colsum_lds(B3, 0, S, tid);
            pl[256 + tid] = colsum_lds(B2, 0, S, tid);
        } else {
            pl[128 + (tid - 128)] = colsum_lds(Gs, 0, S, tid - 128);
        }
        // (res1 is in B5 since P1 and stays there until LayerNorm1 backward, P5)
        __syncthreads();
        // x1 is only an operand of the weight-gradient kernel; in split / bf16 mode the forward has saved it, as bf16 planes
        if (!(CM != CM_F32 && p.xg_planes)) {
            ln_rows(B5, S, w.norm1_w, w.norm1_b, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
                store32(w.x1_out + (tok0 + row) * FD + c0, y);
            });
        }
        // CM_SPLIT: g2 (B2) is split once into bf16 operand planes for the P4 loop (over B3 / B4: dY.xhat is consumed, B4 is free)
        unsigned short* GP = reinterpret_cast<unsigned short*>(B3 < B4 ? B3 : B4);
        constexpr int GPS = SP * LDXH;
        if constexpr (CM == CM_SPLIT) {
            static_assert(3 * SP * LDXH * 2 <= 2 * BLK * 4, "operand planes must fit two LDS blocks");
            const int row = tid >> 2, c0 = (tid & 3) * 32;
            if (row < SP) {     // padded rows of B2 are zero
                float g[32];
                uint32_t h[16], m[16], lo[16];
                load32(B2 + row * LDX + c0, g);
                split32(g, h, m, lo);
                store_parts32(GP + row * LDXH + c0, (size_t)GPS, h, m, lo);
            }
        }
        __syncthreads();
        if constexpr (CM == CM_SPLIT) {
            // g2 leaves for the weight-gradient kernel as the same three parts ((3, N, 128) bf16), dense 16-byte pieces in lane order
            if (p.xg_planes) {
                static_assert(SP == FUSED_TOK_PAD, "the operand planes live on the 48-row clip grid");
                const size_t plane = (size_t)p.B * SP * FD;
                unsigned short* dst = reinterpret_cast<unsigned short*>(w.g2_out) + (size_t)clip * SP * FD;
                for (int i = tid; i < 3 * SP * (FD / 8); i += 256) {
                    int part = i / (SP * (FD / 8)), rem = i - part * (SP * (FD / 8));
                    int row = rem >> 4, c8 = rem & 15;
                    const unsigned short* src = GP + part * GPS + row * LDXH + c8 * 8;        // rows are 8-byte aligned
                    const uint2 a = *reinterpret_cast<const uint2*>(src), b = *reinterpret_cast<const uint2*>(src + 4);
                    *reinterpret_cast<uint4*>(dst + part * plane + rem * 8) = make_uint4(a.x, a.y, b.x, b.y);
                }
            }
        }

        BSTAMP(2);
        EGX_PHASE();
        // SLICED: this pass walks the hidden blocks of slice `sl_cur` — its own first; afterwards any whose partial dX1 does not arrive
        // in time is computed here too (slice_wait, fused_dev.h)
        int sl_cur = slice, sl_k = 0;
        for (;;) {
        // P4: FFN input gradient. dH^T = (W2^T g2^T) .* mask (ReLU sign bits saved by the forward); dX1^T += W1^T dH^T
        {
            constexpr bool XRES = CM == CM_BF16;
            constexpr int XR = XRES ? FD / 32 : 1;
            // same schedule as the forward FFN loop (fused.hip): rolling refill of every weight fragment right behind the MFMAs
            // that consumed it, every memory operation unconditional so that the per-fragment waits are exact
            Frag<CM> gb[XR][NT];
            if constexpr (XRES) {
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb)
#pragma unroll
                    for (int t = 0; t < NT; ++t) gb[kb][t] = load_frag<CM>(B2 + (t * 16 + r) * LDX + kb * 32, q);
            }
            f32x4 dxa[8][NT];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) dxa[i][t] = f32x4{0, 0, 0, 0};
            const int nhb = p.d_ff / 32;
            const int nit = nhb / 4 / n_slices;           // sliced mode: blocks [slice * nit, (slice + 1) * nit) of every wave's walk
            const int j0 = sl_cur * nit;
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);
            const int rot = p.rot_mode == 0 ? (int)((clip * 11u + (clip >> 3) * 5u) % (unsigned)nit)
                          : p.rot_mode == 2 ? (int)(((unsigned)(clip >> 3) & 3u) * (unsigned)nit / 4u)
                          : p.rot_mode == 3 ? (int)(((unsigned)(clip >> 3) & 1u) * (unsigned)nit / 2u)
                          : p.rot_mode == 4 ? (int)(((unsigned)(clip >> 3) & 3u) % (unsigned)nit)
                          : p.rot_mode == 5 ? (int)(((unsigned)(clip >> 3) & 7u) % (unsigned)nit) : 0;
            auto hb_of = [&](int it) { int j = it + rot; if (j >= nit) j -= nit; return wave_s + 4 * (j0 + j); };
            WRaw<CM> w2r[2][FD / 32], w3r[8];
            uint32_t relu_word;
            const uint32_t* relu_bits = p.relu_bits + ((size_t)l * p.B + clip) * (size_t)(p.d_ff / 32) * 64 + lane;
            constexpr int ESZ = CM == CM_BF16 ? 2 : 4;
            const int nht = p.d_ff / 16;
            char* const dhid_base = (char*)p.dhid_out + ((size_t)l * p.B + clip) * NT * nht * (size_t)(HTILE_ELEMS * ESZ);
            {
                const int hb0 = hb_of(0);
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb)
#pragma unroll
                    for (int i = 0; i < 2; ++i) w2r[i][kb] = load_w<CM>(w.lin2_wtp, hb0 * 2 + i, FD / 32, kb, lane);
                relu_word = relu_bits[(size_t)hb0 * 64];
#pragma unroll
                for (int i = 0; i < 8; ++i) w3r[i] = load_w<CM>(w.lin1_wtp, i, nhb, hb0, lane);
            }
            bool ffn_done = false;
            if constexpr (CM == CM_BF16 && EGX_FFN_PIPE_BWD && !SLICED) {
                // bf16: the loop software-pipelined across hidden blocks as in the forward (fused.hip): the mask / pack / dH-tile work of block `it` is
                // issued in slices behind the MFMAs of one weight fragment each — GEMM1 of block it + 1 (into a second accumulator set) and GEMM2 of
                // block it - 1 (from the operand fragments the previous block left). One ring of eight fragment slots (w2r) serves both weight streams.
                ffn_done = true;
                f32x4 dc[2][NT], dc2[2][NT];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dc[i][t] = f32x4{0, 0, 0, 0};
                {
                    const int hb1 = hb_of(1 < nit ? 1 : 0);
#pragma unroll
                    for (int kb = 0; kb < FD / 32; ++kb)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            pin(w2r[i][kb]);
                            Frag<CM> a2 = w_frag<CM>(w2r[i][kb]);
#pragma unroll
                            for (int t = 0; t < NT; ++t) mma<CM>(dc[i][t], a2, gb[kb][t]);
                            __builtin_amdgcn_sched_barrier(0);
                            w2r[i][kb] = load_w<CM>(w.lin2_wtp, hb1 * 2 + i, FD / 32, kb, lane);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                Frag<CM> dqA[NT], dqB[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dqA[t] = chain_frag<CM>(f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0});
                auto step = [&](int it, f32x4 (&dcur)[2][NT], f32x4 (&dnext)[2][NT], const Frag<CM> (&dq_prev)[NT], Frag<CM> (&dq)[NT]) {
                    const int hb = hb_of(it);
                    const int hbn = hb_of(it + 1 < nit ? it + 1 : it);
                    const int hb2 = hb_of(it + 2 < nit ? it + 2 : nit - 1);
                    const int hbp = hb_of(it > 0 ? it - 1 : 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t bits = relu_word;
                    __builtin_amdgcn_sched_barrier(0);
                    relu_word = relu_bits[(size_t)hbn * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int t = 0; t < NT; ++t) dnext[i][t] = f32x4{0, 0, 0, 0};
                    static_assert(3 * NT <= 16, "epilogue slices");
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        if (k < 8) {
                            const int kb = k >> 1, i = k & 1;
                            pin(w2r[i][kb]);
                            Frag<CM> a2 = w_frag<CM>(w2r[i][kb]);
#pragma unroll
                            for (int t = 0; t < NT; ++t) mma<CM>(dnext[i][t], a2, gb[kb][t]);
#if EGX_FFN_PIPE_BWD_RING == 16
                            w2r[i][kb] = load_w<CM>(w.lin2_wtp, hb2 * 2 + i, FD / 32, kb, lane);
#else
                            w2r[i][kb] = load_w<CM>(w.lin1_wtp, k, nhb, hbp, lane);             // slot k: W1^T fragment k of the previous block, needed eight steps on
#endif
                        } else {
                            const int j = k - 8, i = j & 1, kb = j >> 1;
#if EGX_FFN_PIPE_BWD_RING == 16
                            pin(w3r[j]);
                            Frag<CM> a = w_frag<CM>(w3r[j]);
#pragma unroll
                            for (int t = 0; t < NT; ++t) mma<CM>(dxa[j][t], a, dq_prev[t]);
                            w3r[j] = load_w<CM>(w.lin1_wtp, j, nhb, hb, lane);      // this block's W1^T rows: GEMM2 of the next iteration
#else
                            pin(w2r[i][kb]);
                            Frag<CM> a = w_frag<CM>(w2r[i][kb]);
#pragma unroll
                            for (int t = 0; t < NT; ++t) mma<CM>(dxa[j][t], a, dq_prev[t]);
                            w2r[i][kb] = load_w<CM>(w.lin2_wtp, hb2 * 2 + i, FD / 32, kb, lane);  // slot j: W2^T fragment j of the block after next
#endif
                        }
                        if (k < 2 * NT) {           // alive bits -> masks of one 16 x 16 tile (see the loop below)
                            const int i = k / NT, t = k % NT;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int kk = (i * NT + t) * 4 + e;
                                const int32_t m = ((int32_t)(bits << (31 - kk))) >> 31;
                                dcur[i][t][e] = __uint_as_float(__float_as_uint(dcur[i][t][e]) & (uint32_t)m);
                            }
                        } else if (k < 3 * NT) {    // operand fragment + dH tile of a token tile
                            const int t = k - 2 * NT;
                            dq[t] = chain_frag<CM>(dcur[0][t], dcur[1][t]);
                            if constexpr (CM == CM_BF16) {
                                const u32x4 u = __builtin_bit_cast(u32x4, dq[t].v);
                                store_hid_tile_bf16(dhid_base + (size_t)hb * 2 * (HTILE_ELEMS * ESZ) + (size_t)t * nht * (HTILE_ELEMS * ESZ), u, lane, S - t * 16);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                auto last_gemm2 = [&](const Frag<CM> (&dq_last)[NT]) {      // (its W1^T rows are requested here: one exposed round trip per layer)
#if EGX_FFN_PIPE_BWD_RING != 16
                    const int hbl = hb_of(nit - 1);
#pragma unroll
                    for (int i = 0; i < 8; ++i) w3r[i] = load_w<CM>(w.lin1_wtp, i, nhb, hbl, lane);
#endif
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        Frag<CM> a = w_frag<CM>(w3r[i]);
#pragma unroll
                        for (int t = 0; t < NT; ++t) mma<CM>(dxa[i][t], a, dq_last[t]);
                    }
                };
                int it = 0;
                for (; it + 1 < nit; it += 2) {
                    step(it, dc, dc2, dqA, dqB);
                    step(it + 1, dc2, dc, dqB, dqA);
                }
                if (it < nit) {
                    step(it, dc, dc2, dqA, dqB);
                    last_gemm2(dqB);
                } else {
                    last_gemm2(dqA);
                }
            }
            if (!ffn_done)
            for (int it = 0; it < nit; ++it) {
                const int hb = hb_of(it);
                const int hbn = hb_of(it + 1 < nit ? it + 1 : it);     // the last block refills itself (never used)
                __builtin_amdgcn_sched_barrier(0);
                f32x4 dacc[2][NT];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dacc[i][t] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb) {
                    if constexpr (CM == CM_SPLIT) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) gb[0][t] = load_split_frag(GP, GPS, t * 16 + r, kb * 32, q);
                    } else if constexpr (!XRES) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) gb[0][t] = load_frag<CM>(B2 + (t * 16 + r) * LDX + kb * 32, q);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        pin(w2r[i][kb]);
                        Frag<CM> a2 = w_frag<CM>(w2r[i][kb]);
#pragma unroll
                        for (int t = 0; t < NT; ++t) mma<CM>(dacc[i][t], a2, gb[XRES ? kb : 0][t]);
                        __builtin_amdgcn_sched_barrier(0);
                        w2r[i][kb] = load_w<CM>(w.lin2_wtp, hbn * 2 + i, FD / 32, kb, lane);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const uint32_t bits = relu_word;
                __builtin_amdgcn_sched_barrier(0);
                relu_word = relu_bits[(size_t)hbn * 64];
                __builtin_amdgcn_sched_barrier(0);
                // alive bits = ReLU active AND kept by the forward's dropout: no RNG here, and no multiply either — the keep-scale
                // 1 / (1 - p) is folded into the packed W2^T (encoder.hip). One sign-extended bit field + one AND per unit.
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = (i * NT + t) * 4 + e;
                            const int32_t m = ((int32_t)(bits << (31 - k))) >> 31;
                            dacc[i][t][e] = __uint_as_float(__float_as_uint(dacc[i][t][e]) & (uint32_t)m);
                        }
                Frag<CM> dq_[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dq_[t] = chain_frag<CM>(dacc[0][t], dacc[1][t]);
                {       // dH tiles for the weight-gradient kernel
                    char* hb_base = dhid_base + (size_t)hb * 2 * (HTILE_ELEMS * ESZ);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
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
                for (int i = 0; i < 8; ++i) {
                    pin(w3r[i]);
                    Frag<CM> a = w_frag<CM>(w3r[i]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mma<CM>(dxa[i][t], a, dq_[t]);
                    __builtin_amdgcn_sched_barrier(0);
                    w3r[i] = load_w<CM>(w.lin1_wtp, i, nhb, hbn, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        BSTAMP(3);
            // cross-wave sum: every wave parks its partial dX1 in a block of its own (dY in Gs, dY.xhat in B3 and B4 are dead;
            // g2 in B2 is dead once every wave has left the loop); LayerNorm1 backward adds the four in a fixed order
            __syncthreads();
            {
                float* mine = wave == 0 ? Gs : (wave == 1 ? B2 : (wave == 2 ? B3 : B4));
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        *reinterpret_cast<float4*>(mine + (t * 16 + r) * LDX + i * 16 + 4 * q) =
                            make_float4(dxa[i][t][0], dxa[i][t][1], dxa[i][t][2], dxa[i][t][3]);
            }
        }
        if constexpr (!SLICED) {
            break;
        } else {
            __syncthreads();
            float* xc = p.xchg + ((size_t)l * p.B + clip) * n_slices * (FUSED_TOK_PAD * FD);
            unsigned* fl = p.xflags + ((size_t)l * p.B + clip) * SLICE_MAX;
            slice_publish(Gs, B2, B3, B4, LDX, S, xc + (size_t)sl_cur * (FUSED_TOK_PAD * FD), fl + sl_cur);
            bool steal = false;
            while (++sl_k < n_slices) {
                const int s2 = slice + sl_k < n_slices ? slice + sl_k : slice + sl_k - n_slices;
                if (!slice_wait(fl + s2)) { sl_cur = s2; steal = true; slice_stolen_note(&g_stolen_bwd); break; }
            }
            if (!steal) {
                // the sum over the waves and over the slices of the clip comes back in Gs (B2, B3, B4 cleared): identical results in every slice
                slice_gather(xc, n_slices, Gs, LDX, S);
                for (int e = tid; e < S * (FD / 4); e += 256) {
                    const int o = (e >> 5) * LDX + (e & 31) * 4;
                    *reinterpret_cast<float4*>(B2 + o) = make_float4(0, 0, 0, 0);
                    *reinterpret_cast<float4*>(B3 + o) = make_float4(0, 0, 0, 0);
                    *reinterpret_cast<float4*>(B4 + o) = make_float4(0, 0, 0, 0);
                }
                break;
            }
            // the missing slice is computed here: the partial blocks overwrote g2 (B2) and its operand planes (B3 / B4); both come
            // back from what this workgroup stored for the weight-gradient kernel before the loop (w.g2_out)
            if (CM == CM_SPLIT && p.xg_planes) {
                const size_t plane = (size_t)p.B * SP * FD;
                const unsigned short* src = reinterpret_cast<const unsigned short*>(w.g2_out) + (size_t)clip * SP * FD;
                unsigned short* GPr = reinterpret_cast<unsigned short*>(B3 < B4 ? B3 : B4);
                for (int i = tid; i < 3 * SP * (FD / 8); i += 256) {
                    const int part = i / (SP * (FD / 8)), rem = i - part * (SP * (FD / 8));
                    const int row = rem >> 4, c8 = rem & 15;
                    const uint4 v = *reinterpret_cast<const uint4*>(src + part * plane + rem * 8);
                    unsigned short* dst = GPr + part * (SP * LDXH) + row * LDXH + c8 * 8;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
                    *reinterpret_cast<uint2*>(dst + 4) = make_uint2(v.z, v.w);
                }
            } else {
                if (CM == CM_BF16 && p.xg_planes) {
                    const unsigned short* src = reinterpret_cast<const unsigned short*>(w.g2_out) + (size_t)clip * FUSED_TOK_PAD * FD;
                    for (int i = tid; i < SP * (FD / 8); i += 256) {
                        const int row = i >> 4, c8 = i & 15;
                        const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)i * 8);       // rows >= S were stored as zeros
                        float* d = B2 + row * LDX + c8 * 8;
                        *reinterpret_cast<float4*>(d) = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
                        *reinterpret_cast<float4*>(d + 4) = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u));
                    }
                } else {
                    for (int i = tid; i < SP * (FD / 4); i += 256) {
                        const int row = i >> 5, c4 = (i & 31) * 4;
                        *reinterpret_cast<float4*>(B2 + row * LDX + c4) =
                            row < S ? *reinterpret_cast<const float4*>(w.g2_out + (tok0 + row) * FD + c4) : make_float4(0, 0, 0, 0);
                    }
                }
                if constexpr (CM == CM_SPLIT) {
                    __syncthreads();
                    unsigned short* GPr = reinterpret_cast<unsigned short*>(B3 < B4 ? B3 : B4);
                    const int row = tid >> 2, c0 = (tid & 3) * 32;
                    if (row < SP) {
                        float g[32];
                        uint32_t h[16], m[16], lo[16];
                        load32(B2 + row * LDX + c0, g);
                        split32(g, h, m, lo);
                        store_parts32(GPr + row * LDXH + c0, (size_t)(SP * LDXH), h, m, lo);
                    }
                }
            }
            __syncthreads();
        }
        }
        }       // !CUT: P2 - P4
        BSTAMP(4);
        // Q | K | V rows of the clip (48 x 384 fp32, saved by the forward): QKV_PF float4 per thread, requested here and parked in
        // registers under P5 - P7 (they take ~10k cycles to arrive from HBM); written to LDS when B4 / B5 / Gs are free, after P7.
        constexpr int QKV_PF = SP * 96 / 256;
        static_assert(SP * 96 % 256 == 0, "the Q | K | V rows must divide among the threads");
        // (f32x4 values, not float4 structs: a struct copy is an llvm.memcpy global -> private -> LDS that pins qv[] in scratch memory)
        const f32x4* qsrc = reinterpret_cast<const f32x4*>(p.saved_qkv + ((size_t)l * p.B + clip) * SP * (3 * FD));   // (L, B, 48, 384)
        f32x4 qv[QKV_PF];
        // P5: LayerNorm1 backward with dy = dX1 (four wave partials in Gs, B2, B3, B4) + d_res2 (B1); x = res1 (B5, from P3).
        if (ps_g == 0) *reinterpret_cast<f32x4*>(PSB + FD + ps_c) = n1_v;
        __syncthreads();
        ln_bwd_rows_lds(S, PSB + FD, p.eps,
            [&](int row, int c0, float (&dy)[32], float (&x)[32]) {
                float t[32];
                load32(Gs + row * LDX + c0, dy);
                load32(B2 + row * LDX + c0, t);
#pragma unroll
                for (int j = 0; j < 32; ++j) dy[j] += t[j];
                load32(B3 + row * LDX + c0, t);
#pragma unroll
                for (int j = 0; j < 32; ++j) dy[j] += t[j];
                load32(B4 + row * LDX + c0, t);
#pragma unroll
                for (int j = 0; j < 32; ++j) dy[j] += t[j];
                load32(B1 + row * LDX + c0, t);
#pragma unroll
                for (int j = 0; j < 32; ++j) dy[j] += t[j];
                load32(B5 + row * LDX + c0, x);
            },
            [&](int row, int c0, float (&dy)[32], float (&dx)[32], float (&dyx)[32]) {
                store32(Gs + row * LDX + c0, dy);      // total dy, for d(norm1_b)
                store32(B1 + row * LDX + c0, dx);      // d_res1 (residual path into the layer input)
                store32(B4 + row * LDX + c0, dyx);
                if (w.res_thresh) {
                    uint32_t orow = (uint32_t)(tok0 + row);
#pragma unroll
                    for (int j = 0; j < 32; ++j) dx[j] *= drop_scale(k_res1, orow, (uint32_t)(c0 + j), w.res_thresh, w.drop_inv);
                }
                store32(B2 + row * LDX + c0, dx);      // g1
            },
            [&] {       // behind the request for the LayerNorm weights: the Q | K | V rows (in flight until after P7)
                if constexpr (!TILED) {
                    int t_ = threadIdx.x;
                    asm volatile("" : "+v"(t_));        // keep the address arithmetic here (see EGX_PHASE)
#pragma unroll
                    for (int k = 0; k < QKV_PF; ++k) qv[k] = qsrc[t_ + 256 * k];
                }
            }, 16);
        BSTAMP(21);
        __syncthreads();
        store_block(w.g1_out + tok0 * FD, B2, S);
        BSTAMP(5);
        PackW<CM, 2, 4> wo_pf;        // W_o^T fragments of P7: in flight under the column sums
        pack_issue(wo_pf, w.out_proj_wtp, wave * 2, 4, 0);
        // P6: column sums (norm1_w, norm1_b, out_proj_b)
        if (tid < 128) {
            pl[384 + tid] = colsum_lds(B4, 0, S, tid);
            pl[640 + tid] = colsum_lds(B2, 0, S, tid);
        } else {
            pl[512 + (tid - 128)] = colsum_lds(Gs, 0, S, tid - 128);
        }
        EGX_PHASE();
        // P7: out-projection input gradient dO^T = W_o^T g1^T -> B3 (token-major)
        __syncthreads();
        {
            f32x4 acc[2][NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
            gemm_packed<CM, 2, NT, 4>(acc, wo_pf, B2, r, q);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<float4*>(B3 + (t * 16 + r) * LDX + (wave * 2 + i) * 16 + 4 * q) =
                        make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
        }
        BSTAMP(6);
        if constexpr (TILED) {
            // the attention backward of the whole clip is another launch (tiled_attn_bwd): d(attention output) and the residual
            // gradient leave as dense rows; the next launch of this kernel picks the residual gradient up again (l_front)
            __syncthreads();
            store_block(p.datt + tok0 * FD, B3, S);
            store_block(p.dres + tok0 * FD, B1, S);
            return;
        }
        // P8 / P9: Q -> B4, K -> B5, V -> Gs, loaded from what the forward saved (FusedFwdParams::qkv_out). Until round 3 the layer
        // input was recomputed here (LayerNorm + embeddings + dropout hash) and projected again: 13k + 6k of the kernel's 136k
        // cycles in bf16 mode, 22k + 17k of 300k in split mode.
        {
            int t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
#pragma unroll
            for (int k = 0; k < QKV_PF; ++k) {
                // rows >= S: bias only (finite; masked like the recomputed ones were)
                const int i = t_ + 256 * k, row = i / 96, c = (i - row * 96) * 4;
                float* dst = c < FD ? B4 + c : (c < 2 * FD ? B5 + (c - FD) : Gs + (c - 2 * FD));
                *reinterpret_cast<f32x4*>(dst + row * LDX) = qv[k];
            }
        }
        __syncthreads();
        BSTAMP(7);
        EGX_PHASE();
        BSTAMP(8);
        EGX_PHASE();
        // P10: attention forward recompute + backward, wave = head (8 heads of 16: two heads per wave, one after the other).
        // Q = B4, K = B5, V = Gs, dO = B3.
#pragma unroll
        for (int hl = 0; hl < HPW; ++hl) {
            const int h = wave * HPW + hl;
            const int hc = h * DH;
            const float scale = DH == 32 ? 0.17677669529663687f : 0.25f;
            float* st_m = stat + (h * 3 + 0) * SP;
            float* st_i = stat + (h * 3 + 1) * SP;
            float* st_d = stat + (h * 3 + 2) * SP;
            // the head's Q, K, V, dO row fragments are loaded (CM_SPLIT: split) ONCE and kept in registers: both passes below use
            // each of them three times, and a reload through the LDS statistics stores in between cannot be elided
            Frag<CM> fq[NT], fk[NT], fv[NT], fdo[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fq[t] = load_head_frag<CM, DH>(B4 + (t * 16 + r) * LDX + hc, q);
                fk[t] = load_head_frag<CM, DH>(B5 + (t * 16 + r) * LDX + hc, q);
                fv[t] = load_head_frag<CM, DH>(Gs + (t * 16 + r) * LDX + hc, q);
                fdo[t] = load_head_frag<CM, DH>(B3 + (t * 16 + r) * LDX + hc, q);
            }
            auto ldq = [&](int t) -> const Frag<CM>& { return fq[t]; };
            auto ldk = [&](int t) -> const Frag<CM>& { return fk[t]; };
            auto ldv = [&](int t) -> const Frag<CM>& { return fv[t]; };
            auto ldo = [&](int t) -> const Frag<CM>& { return fdo[t]; };
            // orientation T: rows = key, cols = query
            f32x4 pt[NT][NT], dpt[NT][NT];
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                Frag<CM> bq = ldq(qt), bdo = ldo(qt);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    pt[kt][qt] = f32x4{0, 0, 0, 0};
                    dpt[kt][qt] = f32x4{0, 0, 0, 0};
                    mma<CM>(pt[kt][qt], ldk(kt), bq);      // S^T = K Q^T
                    mma<CM>(dpt[kt][qt], ldv(kt), bdo);    // dP^T = V dO^T
                }
            }
            auto keep = [&](int query, int key) -> float {   // attention-dropout keep-scale (regenerated, never stored)
                return w.attn_thresh ? drop_scale(k_attn, (uint32_t)((clip * NHEAD + h) * 64 + query), (uint32_t)key, w.attn_thresh, w.drop_inv) : 1.f;
            };
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int key = kt * 16 + 4 * q + e;
                        float sv = (key < S) ? pt[kt][qt][e] * scale : -INFINITY;
                        pt[kt][qt][e] = sv;
                        m = fmaxf(m, sv);
                    }
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float pv = __expf(pt[kt][qt][e] - m); pt[kt][qt][e] = pv; sum += pv; }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                float inv = 1.f / sum;
                int query = qt * 16 + r;
                float dl = 0.f;
                float ksv[NT][4];                      // keep-scale of this query column, computed once
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pv = pt[kt][qt][e] * inv;
                        float ks = keep(query, kt * 16 + 4 * q + e);
                        ksv[kt][e] = ks;
                        pt[kt][qt][e] = pv;
                        dpt[kt][qt][e] *= ks;          // mask .* dP^T
                        dl += pv * dpt[kt][qt][e];
                    }
                dl += __shfl_xor(dl, 16, 64);
                dl += __shfl_xor(dl, 32, 64);
                if (q == 0) { st_m[query] = m; st_i[query] = inv; st_d[query] = dl; }
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pv = pt[kt][qt][e];
                        dpt[kt][qt][e] = pv * (dpt[kt][qt][e] - dl) * scale;                      // dS^T (scaled)
                        pt[kt][qt][e] = pv * ksv[kt][e];                                           // dropped P^T
                    }
            }
            // O^T = V^T (P^T .* mask) -> attn_o (HBM);  dQ^T = K^T dS^T
            f32x4 oq[NCT][NT], dqa[NCT][NT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) { oq[ct][qt] = f32x4{0, 0, 0, 0}; dqa[ct][qt] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag<CM> av[NCT], ak[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    av[ct] = gather_frag<CM>(Gs, hc + ct * 16 + r, kb * 32, q, SP - 1);
                    ak[ct] = gather_frag<CM>(B5, hc + ct * 16 + r, kb * 32, q, SP - 1);
                }
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    f32x4 z = f32x4{0, 0, 0, 0};
                    Frag<CM> bp = chain_frag<CM>(pt[2 * kb][qt], (2 * kb + 1 < NT) ? pt[(2 * kb + 1 < NT) ? 2 * kb + 1 : 0][qt] : z);
                    Frag<CM> bs = chain_frag<CM>(dpt[2 * kb][qt], (2 * kb + 1 < NT) ? dpt[(2 * kb + 1 < NT) ? 2 * kb + 1 : 0][qt] : z);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        mma<CM>(oq[ct][qt], av[ct], bp);
                        mma<CM>(dqa[ct][qt], ak[ct], bs);
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    int tok = qt * 16 + r;
                    if (tok < S)
                        *reinterpret_cast<float4*>(w.attn_o_out + (tok0 + tok) * FD + hc + ct * 16 + 4 * q) =
                            make_float4(oq[ct][qt][0], oq[ct][qt][1], oq[ct][qt][2], oq[ct][qt][3]);
                }
            // orientation N: rows = query, cols = key. P and dS rebuilt from the per-query statistics.
            f32x4 pn[NT][NT], dsn[NT][NT];
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                float4 m4 = *reinterpret_cast<const float4*>(st_m + qt * 16 + 4 * q);
                float4 i4 = *reinterpret_cast<const float4*>(st_i + qt * 16 + 4 * q);
                float4 d4 = *reinterpret_cast<const float4*>(st_d + qt * 16 + 4 * q);
                float mq[4] = {m4.x, m4.y, m4.z, m4.w}, iq[4] = {i4.x, i4.y, i4.z, i4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    f32x4 sN = f32x4{0, 0, 0, 0}, dN = f32x4{0, 0, 0, 0};
                    mma<CM>(sN, ldq(qt), ldk(kt));    // S = Q K^T
                    mma<CM>(dN, ldo(qt), ldv(kt));    // dP = dO V^T
                    int key = kt * 16 + r;
                    // the lane's four elements are four mask ROWS (queries) of one key: one hash per lane and tile, exchanged inside the quad
                    // (common.h tile_keep_rows), instead of four
                    uint32_t m4[4] = {0xFu, 0xFu, 0xFu, 0xFu};
                    if (w.attn_thresh) tile_keep_rows(k_attn, (uint32_t)((clip * NHEAD + h) * 64 + qt * 16), (uint32_t)(kt * 4), r, q, w.attn_thresh, m4);
                    const float kinv = w.attn_thresh ? w.drop_inv : 1.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int query = qt * 16 + 4 * q + e;
                        bool ok = (key < S) && (query < S);
                        float pv = ok ? __expf(sN[e] * scale - mq[e]) * iq[e] : 0.f;
                        float ks = ((m4[e] >> (r & 3)) & 1u) ? kinv : 0.f;
                        pn[qt][kt][e] = pv * ks;
                        dsn[qt][kt][e] = pv * (ks * dN[e] - dq4[e]) * scale;
                    }
                }
            }
            // dV^T = dO^T P ; dK^T = Q^T dS   (A gathered from token-major dO / Q; K dimension = query)
            f32x4 dva[NCT][NT], dka[NCT][NT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) { dva[ct][kt] = f32x4{0, 0, 0, 0}; dka[ct][kt] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag<CM> ad[NCT], aq[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    ad[ct] = gather_frag<CM>(B3, hc + ct * 16 + r, kb * 32, q, SP - 1);
                    aq[ct] = gather_frag<CM>(B4, hc + ct * 16 + r, kb * 32, q, SP - 1);
                }
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    f32x4 z = f32x4{0, 0, 0, 0};
                    Frag<CM> bp = chain_frag<CM>(pn[2 * kb][kt], (2 * kb + 1 < NT) ? pn[(2 * kb + 1 < NT) ? 2 * kb + 1 : 0][kt] : z);
                    Frag<CM> bs = chain_frag<CM>(dsn[2 * kb][kt], (2 * kb + 1 < NT) ? dsn[(2 * kb + 1 < NT) ? 2 * kb + 1 : 0][kt] : z);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        mma<CM>(dva[ct][kt], ad[ct], bp);
                        mma<CM>(dka[ct][kt], aq[ct], bs);
                    }
                }
            }
            // all operands of this head are consumed: overwrite Q/K/V columns with dQ/dK/dV and emit dqkv
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    int tok = t * 16 + r;
                    int c = hc + ct * 16 + 4 * q;
                    bool tv = tok < S;
                    float4 vq = tv ? make_float4(dqa[ct][t][0], dqa[ct][t][1], dqa[ct][t][2], dqa[ct][t][3]) : make_float4(0, 0, 0, 0);
                    float4 vk = tv ? make_float4(dka[ct][t][0], dka[ct][t][1], dka[ct][t][2], dka[ct][t][3]) : make_float4(0, 0, 0, 0);
                    float4 vv = tv ? make_float4(dva[ct][t][0], dva[ct][t][1], dva[ct][t][2], dva[ct][t][3]) : make_float4(0, 0, 0, 0);
                    *reinterpret_cast<float4*>(B4 + tok * LDX + c) = vq;
                    *reinterpret_cast<float4*>(B5 + tok * LDX + c) = vk;
                    *reinterpret_cast<float4*>(Gs + tok * LDX + c) = vv;
                    if (tv) {
                        float* o = w.dqkv_out + (tok0 + tok) * (3 * FD) + c;
                        *reinterpret_cast<float4*>(o) = vq;
                        *reinterpret_cast<float4*>(o + FD) = vk;
                        *reinterpret_cast<float4*>(o + 2 * FD) = vv;
                    }
                }
        }
        __syncthreads();
        BSTAMP(9);
        PackW<CM, 2, 4> wi_pf;        // W_in^T fragments (dQ part) of P12: in flight under the column sums
        pack_issue(wi_pf, w.in_proj_wtp, wave * 2, 12, 0);
        // what the next stretch loads first, requested now (lands under P11 / P12): the next layer's residual sums, or `pre`
        // (no branch around the requests: after layer 0 the second one re-reads `pre` and is dropped)
        blk_request(pf_a, l > 0 ? res_ptr(2 * (l - 1) + 1) : p.saved_pre + tok0 * FD);
        blk_request(pf_b, l > 0 ? res_ptr(2 * (l - 1)) : p.saved_pre + tok0 * FD);
        // P11: in_proj_b partials
        if (tid < 128) {
            pl[768 + tid] = colsum_lds(B4, 0, S, tid);
            pl[768 + 256 + tid] = colsum_lds(Gs, 0, S, tid);
        } else {
            pl[768 + 128 + (tid - 128)] = colsum_lds(B5, 0, S, tid - 128);
        }
        EGX_PHASE();
        // P12: in-projection input gradient + residual d_res1 (B1) -> B2 (next layer's dY)
        {
            f32x4 acc[2][NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
            PackW<CM, 2, 4> wi2, wi3;
            pack_issue(wi2, w.in_proj_wtp, wave * 2, 12, 4);
            gemm_packed<CM, 2, NT, 4>(acc, wi_pf, B4, r, q);
            pack_issue(wi3, w.in_proj_wtp, wave * 2, 12, 8);
            gemm_packed<CM, 2, NT, 4>(acc, wi2, B5, r, q);
            gemm_packed<CM, 2, NT, 4>(acc, wi3, Gs, r, q);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    int tok = t * 16 + r;
                    int c = (wave * 2 + i) * 16 + 4 * q;
                    float4 rs = *reinterpret_cast<const float4*>(B1 + tok * LDX + c);
                    float4 o = make_float4(acc[i][t][0] + rs.x, acc[i][t][1] + rs.y, acc[i][t][2] + rs.z, acc[i][t][3] + rs.w);
                    if (tok >= S) o = make_float4(0, 0, 0, 0);
                    *reinterpret_cast<float4*>(B2 + tok * LDX + c) = o;
                }
        }
        __syncthreads();
        { float* t = Gs; Gs = B2; B2 = t; }
        if constexpr (CUT) {
            if (l > 0) {        // d(layer input): the dY of the layer below, picked up by its ffn_bwd_kernel
                store_block(p.dxin + tok0 * FD, Gs, S);
                return;
            }
        }
    }

        BSTAMP(10);
    // ---- token preparation backward: Gs = d(x0)
    {
        float* pg = part + p.n_layers * FUSED_P_LAYER;
        blk_store(pf_a, B1, nullptr);       // `pre`, requested before the last P11
        __syncthreads();
        ln_bwd_rows_lds(S, PSB, p.eps,
            [&](int row, int c0, float (&dy)[32], float (&x)[32]) {
                load32(Gs + row * LDX + c0, dy);
                if (p.pos_thresh) {
                    uint32_t orow = (uint32_t)(tok0 + row);
#pragma unroll
                    for (int j = 0; j < 32; ++j) dy[j] *= drop_scale(pos_key, orow, (uint32_t)(c0 + j), p.pos_thresh, p.pos_inv);
                }
                load32(B1 + row * LDX + c0, x);
            },
            [&](int row, int c0, float (&dy)[32], float (&dx)[32], float (&dyx)[32]) {
                store32(Gs + row * LDX + c0, dy);
                store32(B3 + row * LDX + c0, dyx);
                if (p.feat_thresh) {        // d(projection output) = d(LayerNorm input) .* feature-dropout mask (the forward's keying)
#pragma unroll
                    for (int si = 0; si < FUSED_MAX_SEG; ++si)
                        if (si < p.nseg && t0 + row >= p.seg[si].off && t0 + row < p.seg[si].off + p.seg[si].T) {
                            const size_t frow = (size_t)c_real * p.seg[si].T + (t0 + row - p.seg[si].off);
                            const uint64_t fk = dev_seed ? site_key(seed_dev, (uint32_t)si, SITE_FEAT) : p.feat_key[si];
#pragma unroll
                            for (int j = 0; j < 32; ++j) dx[j] *= drop_scale(fk, (uint32_t)frow, (uint32_t)(c0 + j), p.feat_thresh, p.feat_inv);
                        }
                }
                store32(B1 + row * LDX + c0, dx);       // (masked) gradient of the projection output: proj_b partials and d(seg) rows below
            }, [] {}, 22);
        BSTAMP(27);
        __syncthreads();
        // d(seg): the rows of every segment out of B1, 16 bytes per lane in lane order (round 6: stored from the LayerNorm lanes each store
        // instruction touched 64 different 128-byte lines: 8.5k of this phase's 19k cycles, profiles/r06_attn_stamps.txt)
#pragma unroll
        for (int si = 0; si < FUSED_MAX_SEG; ++si)
            if (si < p.nseg) {
                int r0 = p.seg[si].off - t0, r1 = r0 + p.seg[si].T;        // the segment's rows within this tile
                const int lo = max(r0, 0), hi = min(r1, S);
                float* dst = p.dseg_out[si] + ((size_t)c_real * p.seg[si].T + (lo - r0)) * FD;
                for (int i = tid; i < (hi - lo) * (FD / 4); i += 256) {
                    const int row = lo + (i >> 5), c4 = i & 31;
                    *reinterpret_cast<float4*>(dst + (size_t)i * 4) = *reinterpret_cast<const float4*>(B1 + row * LDX + c4 * 4);
                }
            }
        if (p.dx0_out) store_block(p.dx0_out + tok0 * FD, Gs, S);      // d(token-prep output) behind its dropout mask: learned-position gradient
        if (tid < 128) pg[tid] = colsum_lds(B3, 0, S, tid);
        else pg[128 + (tid - 128)] = colsum_lds(Gs, 0, S, tid - 128);
        for (int si = 0; si < p.nseg; ++si) {
            int r0 = p.seg[si].off - t0, r1 = r0 + p.seg[si].T;      // the segment's rows within this tile (empty ranges sum to zero)
            r0 = max(r0, 0); r1 = min(r1, S);
            if (tid < 128) pg[256 + si * 256 + tid] = colsum_lds(Gs, r0, r1, tid);
            else pg[256 + si * 256 + 128 + (tid - 128)] = colsum_lds(B1, r0, r1, tid - 128);
        }
    }
    BSTAMP(11);
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(ReducePartialsParams rp) {
    reduce_partials_block(rp, blockIdx.x, blockIdx.y, gridDim.y);
}

int reduce_partials(const ReducePartialsParams& rp, hipStream_t st, bool deterministic) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(rp.P, 64), deterministic ? 1 : partial_chunks(rp.B)), dim3(256), 0, st, rp);
    EGX_LAUNCH_CHECK();
    return 0;
}

template <int CM, bool TILED, int DH>
static int launch_bwd(const FusedBwdParams& p, hipStream_t st) {
    size_t lds = (size_t)(6 * 48 * LDX + (FH * FDH / DH) * 3 * 48 + 2 * FD) * sizeof(float);        // + the staged LayerNorm weight rows
    static bool attr_set = false;
    if (!attr_set) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_bwd_kernel<CM, TILED, DH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if constexpr (!TILED)
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_bwd_kernel<CM, false, DH, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    timing_begin(TIMER_FUSED_BWD, st);
    bool sliced = false;
    if constexpr (!TILED) sliced = p.n_slices > 1;
    if constexpr (!TILED) {
        if (p.cut) {
            static bool cut_attr = false;
            if (!cut_attr) {
                EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_bwd_kernel<CM, false, DH, false, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                cut_attr = true;
            }
            EGX_CHECK(!sliced && p.dy1 && (p.cut_layer == 0 || p.dxin), "cut mode: one workgroup per clip, dy1 / dxin set");
            hipLaunchKernelGGL((fused_bwd_kernel<CM, false, DH, false, true>), dim3(p.B), dim3(256), lds, st, p);
            timing_end(TIMER_FUSED_BWD, st);
            EGX_LAUNCH_CHECK();
            return 0;
        }
    }
    if (sliced) {
        if constexpr (!TILED)
            hipLaunchKernelGGL((fused_bwd_kernel<CM, false, DH, true>), dim3((p.B + 7) / 8 * 8 * p.n_slices), dim3(256), lds, st, p);
    } else {
        hipLaunchKernelGGL((fused_bwd_kernel<CM, TILED, DH>), dim3(p.B), dim3(256), lds, st, p);
    }
    timing_end(TIMER_FUSED_BWD, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

int fused_backward(const FusedBwdParams& p, int compute, hipStream_t st) {
    EGX_CHECK(p.S <= 48, "fused backward: S=%d > 48", p.S);
    if (p.tiled) {
        EGX_CHECK(compute == CM_BF16 || compute == CM_SPLIT, "tiled mode: compute must be bf16 or f32s");
        return compute == CM_BF16 ? launch_bwd<CM_BF16, true, 32>(p, st) : launch_bwd<CM_SPLIT, true, 32>(p, st);
    }
    if (p.n_heads == 2 * FH)
        return compute == CM_BF16 ? launch_bwd<CM_BF16, false, 16>(p, st) : compute == CM_SPLIT ? launch_bwd<CM_SPLIT, false, 16>(p, st) : launch_bwd<CM_F32, false, 16>(p, st);
    return compute == CM_BF16 ? launch_bwd<CM_BF16, false, 32>(p, st) : compute == CM_SPLIT ? launch_bwd<CM_SPLIT, false, 32>(p, st) : launch_bwd<CM_F32, false, 32>(p, st);
}

}  // namespace egx
