// Fused per-clip translator kernels for the small-d regime (d_model = 128, 4 heads of 32, S <= 64 tokens).
//
// One 256-thread workgroup (4 waves, one per SIMD, one workgroup per CU) owns ONE clip and carries its packed
// (S, d) token block through token preparation, the encoder layers and back entirely on chip: activations
// live in LDS (135 KB of the CU's 160 KB) and in MFMA accumulators, weights stream from L2 exactly once per
// clip straight into MFMA A-operand registers, and nothing but the saved pre-LayerNorm residuals touches HBM.
// B = 256 clips fill the 256 CUs of an MI355X with a single wave of workgroups.
//
// Everything is computed FEATURE-MAJOR (Y^T = W X^T): the 16x16 MFMA C tile then has the token on the lane
// column and four consecutive output features in the lane's 4 registers, which is exactly the B-operand layout
// of the next GEMM whose K dimension is that feature axis. GEMM chains (W1 -> ReLU -> W2, QK^T -> softmax ->
// PV) therefore run accumulator -> operand with no LDS round trip and no shuffles. One operand convention serves
// fp32 (v_mfma_f32_16x16x4_f32, exact) and bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   a K-block is 32 wide; lane group q = lane >> 4 holds k in {4q..4q+3} U {16+4q..16+4q+3} of the block.
//
// Reference math: HHI/models/ttm/model_taskspecific.py:222-226,238-242 + torch.nn.TransformerEncoderLayer.
#include "common.h"
#include "kernels.h"
#include "fused.h"
#include "fused_dev.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>

// bf16 FFN loop of the clip kernels software-pipelined across hidden blocks (see fused_fwd_kernel); 0 = the block-after-block loop of rounds 2-5.
// OFF: the stamps of workgroup 0 show the phase at 49.3k -> 45.7k cycles, but four interleaved same-box pairs of the whole step do not
// (profiles/r06_ab_second_half.txt): c2 bf16 +4.3 us (+2 %) with it, C3 -1.8 us (-0.4 %). Kept as a build switch with the measurements.
#ifndef EGX_FFN_PIPE
#define EGX_FFN_PIPE 0
#endif
#ifndef EGX_FFN_PIPE_RING
#define EGX_FFN_PIPE_RING 8
#endif
#ifndef EGX_FFN_PIPE_VALU
#define EGX_FFN_PIPE_VALU 6
#endif

namespace egx {

__global__ __launch_bounds__(64) void pack_weights_kernel(PackParams pp) {
    int blk = blockIdx.x;
    if (pp.seed_advance && blk == 0 && threadIdx.x == 0)      // same LCG as seed_advance_kernel
        *pp.seed_advance = *pp.seed_advance * 6364136223846793005ull + 1442695040888963407ull;
    if (pp.zero_words)
        for (int i = blk * 64 + threadIdx.x; i < pp.n_zero; i += gridDim.x * 64) pp.zero_words[i] = 0u;
    if (pp.zero_word2 && blk == 0 && threadIdx.x == 0) *pp.zero_word2 = 0.f;
    if (pp.zero_ctl && blk == 0 && threadIdx.x < 8) pp.zero_ctl[threadIdx.x] = 0u;     // (words 0-1: ce_ticket, 4-6: tce_ticket)
    if (pp.n == 0) return;      // nothing to pack (weight cache valid): the launch only carries the side jobs above
    int di = 0;
    while (di + 1 < pp.n && blk >= pp.d[di + 1].first_block) ++di;
    const PackDesc& d = pp.d[di];
    int local = blk - d.first_block;
    int nkb = d.K / 32;
    int t = local / nkb, kb = local % nkb;
    int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    float v[8];
    const float sc = d.scale == 0.f ? 1.f : d.scale;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = t * 16 + r, k = kb * 32 + half * 16 + 4 * q + j;
            v[half * 4 + j] = sc * (d.transpose ? d.src[(size_t)k * d.ld + row] : d.src[(size_t)row * d.ld + k]);
        }
    if (pp.mode == CM_BF16) {
        uint4 o = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        reinterpret_cast<uint4*>(d.dst)[(size_t)local * 64 + lane] = o;
    } else if (pp.mode == CM_SPLIT) {
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
        uint4* pl = reinterpret_cast<uint4*>(d.dst) + (size_t)local * 192;
        pl[lane] = make_uint4(h[0], h[1], h[2], h[3]);
        pl[64 + lane] = make_uint4(m[0], m[1], m[2], m[3]);
        pl[128 + lane] = make_uint4(l[0], l[1], l[2], l[3]);
    } else {
        float4* pl = reinterpret_cast<float4*>(d.dst) + (size_t)local * 128;
        pl[lane] = make_float4(v[0], v[1], v[2], v[3]);
        pl[64 + lane] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

int pack_weights(PackParams& pp, hipStream_t st) {
    int blocks = 0;
    for (int i = 0; i < pp.n; ++i) {
        EGX_CHECK(pp.d[i].R % 16 == 0 && pp.d[i].K % 32 == 0, "pack: matrix %d is %dx%d (needs R%%16==0, K%%32==0)", i, pp.d[i].R, pp.d[i].K);
        pp.d[i].first_block = blocks;
        blocks += (pp.d[i].R / 16) * (pp.d[i].K / 32);
    }
    if (!blocks && !pp.seed_advance && !pp.zero_words && !pp.zero_word2) return 0;
    if (!blocks) blocks = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(64), 0, st, pp);
    EGX_LAUNCH_CHECK();
    return 0;
}

__device__ unsigned g_stolen_fwd;      // slices the forward's waiting workgroups computed themselves (fused_dev.h slice_stolen_note)
long long slices_stolen_fwd(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_stolen_fwd), sizeof(v)) != hipSuccess) return -1;
    if (reset) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stolen_fwd), &z, sizeof(z)); }
    return v;
}
// Development aid: phase timestamps of workgroup 0 / wave 0 (s_memtime), read back by egx_debug_stamps().
__device__ unsigned long long g_stamps[32];
#ifdef EGX_STAMPS
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// inner-loop accounting of workgroup 0 / wave 0: LSTAMP(k) adds the time since the previous LSTAMP to slot 16 + k
#ifdef EGX_STAMPS
#define LSTAMP_INIT() unsigned long long lt_prev = __builtin_amdgcn_s_memtime(), lt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define LSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); lt_acc[k] += t_ - lt_prev; lt_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define LSTAMP_FLUSH() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) if (lt_acc[k_]) g_stamps[16 + k_] = lt_acc[k_]; } while (0)
#else
#define LSTAMP_INIT() do { } while (0)
#define LSTAMP(k) do { } while (0)
#define LSTAMP_FLUSH() do { } while (0)
#endif
int debug_read_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n > 32 ? 32 : n)) == hipSuccess ? 0 : 1;
}

// ---- forward kernel -------------------------------------------------------------------------------
// TILED (48 < S <= 512, see FusedFwdParams): the workgroup is one 48-token tile of a clip and the kernel is cut at the attention:
// FUSED_MODE_PRE = token preparation + Q | K | V of layer 0; FUSED_MODE_POST = out-projection .. LayerNorm2 of layer l0 on the
// attention output tiled_attn_fwd left in `attn_in`, then Q | K | V of layer l0 + 1 (or the output tokens). The full-clip
// instantiation (TILED = false) compiles to the code it was before the tiled mode existed.
// SLICED: the small-batch instantiation (n workgroups per clip, FusedFwdParams::n_slices > 1). A template parameter, not a run-time
// test: with the slice code compiled into the one-workgroup-per-clip kernels their B = 256 step was 1 % (f32s) / 2.5 % (bf16) slower.
// CUT (round 5, ffn_cut.hip): the launch runs [token preparation (l0 = 0) | the saved input of layer l0] .. LayerNorm1 of layer l0 and
// leaves x1; the FFN, the second residual and LayerNorm2 are ffn_fwd_kernel's (eight waves per clip).
template <int CM, int NT, bool TILED, int DH, bool SLICED = false, bool CUT = false>
__global__ __launch_bounds__(256, 1) void fused_fwd_kernel(FusedFwdParams p) {
    static_assert(!(TILED && SLICED), "the tiled launches are not sliced");
    static_assert(!(CUT && (TILED || SLICED)), "the cut launches are neither tiled nor sliced");
    constexpr int HPW = FDH / DH;               // heads per wave: 1 (4 heads of 32) or 2 (8 heads of 16)
    constexpr int NHEAD = FH * HPW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SP = NT * 16;                 // padded token count
    float* Xs = lds;                            // [SP][LDX] layer input (later: FFN partial 0)
    float* Qs = Xs + SP * LDX;                  // [SP][LDX] Q, later attention output O
    float* Ks = Qs + SP * LDX;                  // [SP][LDX]
    float* Vt = Ks + SP * LDX;                  // [FD][LDV] V^T, keys >= S zero
    float* X1 = Vt + FD * LDV;                  // [SP][LDX] res1 / x1
    float* Part = Qs;                           // FFN partials 1..3 alias Q/K/Vt (needs 3*SP*LDX <= 2*SP*LDX + FD*LDV)
    // segment descriptors in LDS: `p.seg[i]` with a run-time i is a chain of dependent scalar loads from the kernel-argument
    // segment (advance / source / store touch ~10 fields per step: 2.5-3k cycles per step, 15-18k of the 35-40k cycles of the
    // token preparation by the stamps); one ds_read burst + v_readfirstlane per step instead
    FusedSeg* segtab = reinterpret_cast<FusedSeg*>(X1 + SP * LDX);
    // small parameter vectors staged in LDS (round 6, see ln_rows_lds): rows of 128 floats behind the segment table
    //   0 ln_w, 1 ln_b, 2..5 task-embedding row of segment 0..3 | per layer: 6 norm1_w, 7 norm1_b, 8 out_proj_b, 9 norm2_w, 10 norm2_b, 11 lin2_b
    float* PS = reinterpret_cast<float*>(reinterpret_cast<char*>(segtab) + FUSED_MAX_SEG * sizeof(FusedSeg) + 16);
    static_assert((FUSED_MAX_SEG * sizeof(FusedSeg)) % 16 == 0, "the parameter rows are read in 16-byte pieces");

    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63, r = lane & 15, q = lane >> 4;
    int clip_ = blockIdx.x, slice_ = 0;     // TILED: the tile ("virtual clip"): index of every 48-row grid
    if constexpr (SLICED) {
        slice_map(p.n_slices, clip_, slice_);
        if (clip_ >= p.B || ((p.slice_drop >> slice_) & 1)) return;
    }
    const int clip = clip_, slice = slice_;
    const int n_slices = SLICED ? p.n_slices : 1;
    int S, c_real, t0;                      // tokens of this workgroup, the clip they belong to, their first token within it
    size_t tokbase;                         // global index of the first token: row of the dense (Ntok, .) arrays, dropout row key
    if constexpr (TILED) {
        c_real = clip / p.tpc;
        t0 = (clip - c_real * p.tpc) * 48;
        S = min(48, p.S_clip - t0);
        tokbase = (size_t)c_real * p.S_clip + t0;
    } else {
        c_real = clip; t0 = 0; S = p.S; tokbase = (size_t)clip * S;
    }
    // keep lane-constant fragment addresses local to their phase (hipcc otherwise hoists them all to kernel entry
    // and spills them around the phases)
#define EGX_PHASE()                                              \
    do {                                                         \
        int t_ = threadIdx.x;                                    \
        asm volatile("" : "+v"(t_));                             \
        lane = t_ & 63; r = lane & 15; q = lane >> 4;            \
    } while (0)

    STAMP(0);
    if constexpr (CUT) { if (p.zero_word && blockIdx.x == 0 && tid == 0) *p.zero_word = 0.f; }
    // weight streams of a later launch (or of this launch's later phases) -> Infinity Cache (TouchList): the oldest loads of this launch
    Touched tch = {{0, 0, 0, 0}};
    if constexpr (!TILED) { if (p.touch.n) tch = touch_lines<256>(p.touch, blockIdx.x, gridDim.x, tid); }
    const bool dev_seed = p.seed_ptr != nullptr;
    const uint64_t seed_dev = dev_seed ? *p.seed_ptr : 0ull;
    const uint64_t pos_key = dev_seed ? site_key(seed_dev, 0, SITE_POS) : p.pos_key;
    int* const nseg_slot = reinterpret_cast<int*>(segtab + FUSED_MAX_SEG);
    if constexpr (TILED) {
        // the segments that intersect this tile, as descriptors of their own: frames [row0, row0 + T) at tile rows [off, off + T)
        if (tid == 0) {
            int n = 0;
            for (int i = 0; i < p.nseg; ++i) {
                FusedSeg o = p.seg[i];
                const int lo = max(o.off, t0), hi = min(o.off + o.T, t0 + 48);
                if (hi > lo) { o.Tfull = o.T; o.row0 = lo - o.off; o.T = hi - lo; o.off = lo - t0; o.seg_id = i; segtab[n++] = o; }
            }
            *nseg_slot = n;
        }
    } else {
        if (tid < FUSED_MAX_SEG) segtab[tid] = p.seg[tid];
    }
    // egx_token_ce: the normaliser sum_i w[y_i] over every label of the batch and this thread's row label are fetched HERE, under the kernel's first
    // loads — at the tail, where the classifier runs, each would be a bare memory round trip at the end of the launch (13 us per launch, measured)
    float tce_sw = 0.f;         // (the wave's partial sum, in every lane)
    int64_t tce_y = -1;
    int64_t tce_lab[16];
    if constexpr (!TILED && !SLICED && !CUT) if (p.tce_W) {
        const int M = p.B * p.out_T;
        float cw[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) cw[c] = c < p.tce_C ? (p.tce_cw ? p.tce_cw[c] : 1.f) : 0.f;
        tce_y = p.tce_target[(size_t)clip * p.out_T + min(tid >> 2, p.out_T - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) tce_lab[u] = tid + u * 256 < M ? p.tce_target[tid + u * 256] : -1;     // (requested now, summed behind the staging below)
        for (int i0 = tid + 256 * 16; i0 < M; i0 += 256 * 16) {       // batches beyond 4096 token rows: plain passes
            int64_t y[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) y[u] = i0 + u * 256 < M ? p.tce_target[i0 + u * 256] : -1;
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int c = 0; c < 8; ++c) if (c < p.tce_C) tce_sw += y[u] == c ? cw[c] : 0.f;
        }
        // the classifier itself into LDS (rows 12.. of the staged parameters: C weight rows, then [bias (8) | class weights (8)])
        float* TW = PS + 12 * FD;
        if (tid < p.tce_C * 32) *reinterpret_cast<f32x4*>(TW + tid * 4) = *reinterpret_cast<const f32x4*>(p.tce_W + tid * 4);
        if (tid < 8) { TW[8 * FD + tid] = (tid < p.tce_C && p.tce_b) ? p.tce_b[tid] : 0.f; TW[8 * FD + 8 + tid] = tid < p.tce_C ? (p.tce_cw ? p.tce_cw[tid] : 1.f) : 0.f; }
    }
    const int ps_g = tid >> 5, ps_c = (tid & 31) << 2;      // staging: thread (row group, 4 columns)
    {   // shared LayerNorm + task-embedding rows (a missing one repeats ln_w: never read)
        const int sg_i = ps_g - 2;
        const float* av = sg_i == 0 ? p.seg[0].add_vec : sg_i == 1 ? p.seg[1].add_vec : sg_i == 2 ? p.seg[2].add_vec : sg_i == 3 ? p.seg[3].add_vec : nullptr;
        const float* src = ps_g == 1 ? p.ln_b : (av && sg_i < p.nseg) ? av : p.ln_w;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + ps_c);
        if (ps_g < 6) *reinterpret_cast<f32x4*>(PS + ps_g * FD + ps_c) = v;
    }
    if constexpr (!TILED && !SLICED && !CUT) if (p.tce_W) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < p.tce_C) {      // class weight by compare chain: no load that depends on the label
                const float cwc = p.tce_cw ? p.tce_cw[c] : 1.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) tce_sw += tce_lab[u] == c ? cwc : 0.f;
            }
        tce_sw = wsum(tce_sw);
    }
    __syncthreads();
    const int nseg_t = TILED ? *nseg_slot : p.nseg;
    auto seg_of = [&](int i) {      // wave-uniform copy of descriptor i in scalar registers
        static_assert(sizeof(FusedSeg) % 4 == 0, "descriptor is copied word by word");
        FusedSeg o;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(segtab + i);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(FusedSeg) / 4); ++k) dst[k] = __builtin_amdgcn_readfirstlane(src[k]);
        return o;
    };
    STAMP(10);

    bool skip_front = false;        // TILED / POST: the first layer of this launch starts behind its attention
    int l_begin = 0;
    if constexpr (TILED) {
        if (p.mode == FUSED_MODE_POST) {
            // layer input -> Xs (residual), attention output -> Qs (operand of the out-projection); padded rows zero
            skip_front = true; l_begin = p.l0;
            for (int i = tid; i < SP * LDX / 4; i += 256) {
                reinterpret_cast<f32x4*>(Xs)[i] = f32x4{0, 0, 0, 0};
                reinterpret_cast<f32x4*>(Qs)[i] = f32x4{0, 0, 0, 0};
                reinterpret_cast<f32x4*>(X1)[i] = f32x4{0, 0, 0, 0};
            }
            __syncthreads();
            const f32x4* xs = reinterpret_cast<const f32x4*>(p.xin_out + ((size_t)p.l0 * p.Ntok + tokbase) * FD);
            const f32x4* as = reinterpret_cast<const f32x4*>(p.attn_in + ((size_t)p.l0 * p.Ntok + tokbase) * FD);
            for (int i = tid; i < S * (FD / 4); i += 256) {
                const int row = i >> 5, c = (i & 31) << 2;
                *reinterpret_cast<f32x4*>(Xs + row * LDX + c) = xs[i];
                *reinterpret_cast<f32x4*>(Qs + row * LDX + c) = as[i];
            }
        }
    }
    if constexpr (CUT) {
        l_begin = p.l0;
        if (p.l0 > 0) {     // the layer input ffn_fwd_kernel of the layer below left (padded rows zero)
            for (int i = tid; i < SP * LDX / 4; i += 256) {
                reinterpret_cast<f32x4*>(Xs)[i] = f32x4{0, 0, 0, 0};
                reinterpret_cast<f32x4*>(X1)[i] = f32x4{0, 0, 0, 0};
            }
            __syncthreads();
            const f32x4* xs = reinterpret_cast<const f32x4*>(p.xin_out + ((size_t)l_begin * p.B + clip) * S * FD);
            for (int i = tid; i < S * (FD / 4); i += 256) {
                const int row = i >> 5, c = (i & 31) << 2;
                *reinterpret_cast<f32x4*>(Xs + row * LDX + c) = xs[i];
            }
        }
    }
    // In-projection weights of a layer's first half (3 feature tiles per wave, K = 128) and its biases: requested one phase ahead of the
    // Q | K | V GEMM (round 6: under the token preparation's LayerNorm for the first layer: the request + round trip was 5k of the phase's 24k cycles)
    WRaw<CM> wa[3][FD / 32];
    float4 bbn[3];
    auto request_qkv = [&](const FusedLayer& wl, int half) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int kb = 0; kb < FD / 32; ++kb)
                wa[i][kb] = load_w<CM>(wl.in_proj_wp, wave * 6 + half * 3 + i, FD / 32, kb, lane);
            bbn[i] = *reinterpret_cast<const float4*>(wl.in_proj_b + (wave * 6 + half * 3 + i) * 16 + 4 * q);
        }
    };
    bool qkv_requested = false;
    if ((!TILED || !skip_front) && !(CUT && p.l0 > 0)) {
    // ---- token preparation: proj GEMM (feature-major) -> LDS token-major -> LN + task embedding + position.
    // Steps = (segment, 16-row tile, 128-wide K chunk), two register sets. ROLLING REFILL as in the FFN loop: a fragment's
    // registers are reloaded with the same fragment of step i + 2 right behind the MFMAs that consumed it, every load is
    // unconditional (steps past the end re-fetch the last one) so that the per-fragment s_waitcnt counts are exact. Round 2
    // issued a whole step's 12 fragments under `if (valid(next))`: every wait then drained all loads in flight and the six
    // steps of a clip each paid a full HBM / L2 round trip (40k cycles for 0.8 GF; the weight stream alone needs ~12k).
    {
        struct Step { int sgi, t0, k0; };
        auto valid = [&](const Step& s) { return s.sgi < nseg_t; };
        auto advance = [&](Step s) {
            s.k0 += 128;
            const int d_in = __builtin_amdgcn_readfirstlane(segtab[s.sgi].d_in), T = __builtin_amdgcn_readfirstlane(segtab[s.sgi].T);
            if (s.k0 >= d_in) { s.k0 = 0; s.t0 += 16; if (s.t0 >= T) { s.t0 = 0; ++s.sgi; } }
            return s;
        };
        auto clamped = [&](const Step& s, const Step& fallback) { return valid(s) ? s : fallback; };
        struct Src { const float* frow; const void* wp; const float* bias; int nkb, kb0; };
        auto source = [&](const Step& s) {
            const FusedSeg sg = seg_of(s.sgi);
            const int trow = s.t0 + r;
            Src o;
            o.frow = sg.feat + ((size_t)c_real * (TILED ? sg.Tfull : sg.T) + (TILED ? sg.row0 : 0) + (trow < sg.T ? trow : 0)) * sg.d_in + s.k0;   // rows >= T read row 0 (discarded at the store)
            o.wp = sg.proj_wp; o.nkb = sg.d_in / 32; o.kb0 = s.k0 / 32;
            o.bias = sg.proj_b + wave * 32 + 4 * q;
            return o;
        };
        Raw rb[2][4];
        WRaw<CM> ra0[2][4], ra1[2][4];
        float4 pb[2][2];        // projection bias of the step's two feature tiles: fetched WITH the step's fragments (a load issued in
                                // the store epilogue would be the youngest in flight: its wait drained every prefetched fragment)
        auto issue_all = [&](auto SET, const Step& s) {
            constexpr int b = decltype(SET)::value;
            const Src o = source(s);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rb[b][j] = load_raw(o.frow + 32 * j, q);
                ra0[b][j] = load_w<CM>(o.wp, wave * 2 + 0, o.nkb, o.kb0 + j, lane);
                ra1[b][j] = load_w<CM>(o.wp, wave * 2 + 1, o.nkb, o.kb0 + j, lane);
            }
            pb[b][0] = *reinterpret_cast<const float4*>(o.bias);
            pb[b][1] = *reinterpret_cast<const float4*>(o.bias + 16);
        };
        f32x4 acc[2];
        LSTAMP_INIT();
        auto step = [&](auto SET, const Step& s, const Step& refill) {
            constexpr int b = decltype(SET)::value;
            const FusedSeg sg = seg_of(s.sgi);
            const Src o = source(refill);
            if (s.k0 == 0) { acc[0] = f32x4{0, 0, 0, 0}; acc[1] = f32x4{0, 0, 0, 0}; }
            LSTAMP(5);
#ifdef EGX_STAMPS
            if (s.sgi == 0 && s.t0 == 0 && s.k0 == 0 && blockIdx.x == 0 && threadIdx.x == 0) g_stamps[11] = lt_acc[5];
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pin(rb[b][j]); pin(ra0[b][j]); pin(ra1[b][j]);
                if (j == 0) { LSTAMP(6); }
                Frag<CM> bf = to_frag<CM>(rb[b][j]);
                mma<CM>(acc[0], w_frag<CM>(ra0[b][j]), bf);
                mma<CM>(acc[1], w_frag<CM>(ra1[b][j]), bf);
                __builtin_amdgcn_sched_barrier(0);
                rb[b][j] = load_raw(o.frow + 32 * j, q);
                ra0[b][j] = load_w<CM>(o.wp, wave * 2 + 0, o.nkb, o.kb0 + j, lane);
                ra1[b][j] = load_w<CM>(o.wp, wave * 2 + 1, o.nkb, o.kb0 + j, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float4 bb0 = pb[b][0], bb1 = pb[b][1];
            __builtin_amdgcn_sched_barrier(0);
            pb[b][0] = *reinterpret_cast<const float4*>(o.bias);
            pb[b][1] = *reinterpret_cast<const float4*>(o.bias + 16);
            __builtin_amdgcn_sched_barrier(0);
            LSTAMP(7);
            if (s.k0 + 128 >= sg.d_in) {
                const int trow = s.t0 + r;
                if (trow < sg.T) {
                    const int f0 = wave * 32 + 4 * q;
                    float4 o0 = make_float4(acc[0][0] + bb0.x, acc[0][1] + bb0.y, acc[0][2] + bb0.z, acc[0][3] + bb0.w);
                    float4 o1 = make_float4(acc[1][0] + bb1.x, acc[1][1] + bb1.y, acc[1][2] + bb1.z, acc[1][3] + bb1.w);
                    if (p.feat_thresh) {        // feature dropout on the projection output (wave-uniform branch, no memory operation inside)
                        const uint64_t fk = dev_seed ? site_key(seed_dev, (uint32_t)sg.seg_id, SITE_FEAT) : p.feat_key[sg.seg_id];
                        const uint32_t frow = (uint32_t)((size_t)c_real * (TILED ? sg.Tfull : sg.T) + (TILED ? sg.row0 : 0) + trow);
                        float m0[4], m1[4];
                        drop_scale4(fk, frow, (uint32_t)f0, p.feat_thresh, p.feat_inv, m0);
                        drop_scale4(fk, frow, (uint32_t)(f0 + 16), p.feat_thresh, p.feat_inv, m1);
                        o0 = make_float4(o0.x * m0[0], o0.y * m0[1], o0.z * m0[2], o0.w * m0[3]);
                        o1 = make_float4(o1.x * m1[0], o1.y * m1[1], o1.z * m1[2], o1.w * m1[3]);
                    }
                    *reinterpret_cast<float4*>(Xs + (sg.off + trow) * LDX + f0) = o0;
                    *reinterpret_cast<float4*>(Xs + (sg.off + trow) * LDX + f0 + 16) = o1;
                }
            }
        };
        typedef std::integral_constant<int, 0> SA;
        typedef std::integral_constant<int, 1> SB;
        Step cur{0, 0, 0};
        {
            const Step n1 = clamped(advance(cur), cur);
            STAMP(12);
            issue_all(SA{}, cur);
            issue_all(SB{}, n1);
            STAMP(13);
        }
        // zero the padded rows once so that padded tokens stay finite everywhere: 16-byte writes, under the first steps' loads
        static_assert((SP * LDX) % 4 == 0, "blocks are zeroed in 16-byte pieces");
        for (int i = tid; i < SP * LDX / 4; i += 256) {
            reinterpret_cast<f32x4*>(Xs)[i] = f32x4{0, 0, 0, 0};
            reinterpret_cast<f32x4*>(X1)[i] = f32x4{0, 0, 0, 0};
        }
        STAMP(14);
        __syncthreads();
        STAMP(15);
        for (;;) {
            if (!valid(cur)) break;
            {
                const Step n1 = advance(cur);
                step(SA{}, cur, clamped(valid(n1) ? advance(n1) : n1, cur));
                cur = n1;
            }
            if (!valid(cur)) break;
            {
                const Step n1 = advance(cur);
                step(SB{}, cur, clamped(valid(n1) ? advance(n1) : n1, cur));
                cur = n1;
            }
        }
        LSTAMP_FLUSH();
    }
    __syncthreads();
    touch_sink(tch);
    if constexpr (CUT) { request_qkv(p.layer[l_begin], 0); qkv_requested = true; }      // (one layer per launch: nothing is carried around a layer loop)
    STAMP(1);
    // save pre-LN projections (token order) and apply the shared LN + embeddings. The task-embedding and positional rows of the
    // lane's token are requested unconditionally behind the LayerNorm weights (a missing table reads the LayerNorm weights and
    // is scaled by zero), the save follows them: nothing in this phase waits for a store or for a load under a branch.
    {
        f32x4 pv[8];
        float a_on = 0.f, p_on = 0.f;
        const float* a_row = PS;        // the lane's task-embedding row in LDS (set by the hook)
        ln_rows_lds(Xs, S, PS, PS + FD, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 avj = *reinterpret_cast<const f32x4*>(a_row + c0 + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[4 * j + e] = __builtin_fmaf(p_on, pv[j][e], __builtin_fmaf(a_on, avj[e], y[4 * j + e]));     // (y + emb) + pos
            }
            if (p.pos_thresh) {
                uint32_t orow = (uint32_t)(tokbase + row);
#pragma unroll
                for (int j = 0; j < 32; ++j) y[j] *= drop_scale(pos_key, orow, (uint32_t)(c0 + j), p.pos_thresh, p.pos_inv);
            }
            store32(Xs + row * LDX + c0, y);
        }, [&] {
            int t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
            int row = t_ >> 2;
            row = row < S ? row : S - 1;
            const int c0 = (t_ & 3) * 32;
            int sgi = 0;
            while (sgi + 1 < nseg_t && row >= segtab[sgi + 1].off) ++sgi;
            const FusedSeg sg = segtab[sgi];
            const float* pp = sg.pos ? sg.pos + (size_t)(row - sg.off + (TILED ? sg.row0 : 0)) * sg.pos_stride + c0 : p.ln_w + c0;
            a_row = PS + (2 + sg.seg_id) * FD;
            a_on = sg.add_vec ? 1.f : 0.f;
            p_on = sg.pos ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = *reinterpret_cast<const f32x4*>(pp + 4 * j);
            store_block(p.saved_pre + tokbase * FD, Xs, S);
            __syncthreads();        // the LayerNorm below writes Xs in place
        });
    }
    }       // token preparation
    __syncthreads();

    for (int l = l_begin; l < p.n_layers; ++l) {
        const FusedLayer& w = p.layer[l];
        const uint64_t k_attn = dev_seed ? site_key(seed_dev, l, SITE_ATTN) : w.attn_key;
        const uint64_t k_res1 = dev_seed ? site_key(seed_dev, l, SITE_RES1) : w.res1_key;
        const uint64_t k_ffn = dev_seed ? site_key(seed_dev, l, SITE_FFN) : w.ffn_key;
        const uint64_t k_res2 = dev_seed ? site_key(seed_dev, l, SITE_RES2) : w.res2_key;
        float* sv_res1 = TILED ? p.saved_res + ((size_t)(2 * l) * p.Ntok + tokbase) * FD : p.saved_res + ((size_t)(2 * l) * p.B + clip) * S * FD;
        float* sv_res2 = TILED ? p.saved_res + ((size_t)(2 * l + 1) * p.Ntok + tokbase) * FD : p.saved_res + ((size_t)(2 * l + 1) * p.B + clip) * S * FD;
        // this layer's small parameter vectors: requested here, stored into LDS rows 6..11 in front of the out-projection
        f32x4 pl_v;
        {
            const float* src = ps_g == 1 ? w.norm1_b : ps_g == 2 ? w.out_proj_b : ps_g == 3 ? w.norm2_w : ps_g == 4 ? w.norm2_b : ps_g == 5 ? w.lin2_b : w.norm1_w;
            pl_v = *reinterpret_cast<const f32x4*>(src + ps_c);
        }

        WRaw<CM> wo[2][FD / 32];        // out-projection weights, 2 feature tiles per wave
        auto request_wo = [&] {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb)
                    wo[i][kb] = load_w<CM>(w.out_proj_wp, wave * 2 + i, FD / 32, kb, lane);
        };
        const bool skip_front_wo = TILED && skip_front;
        if (skip_front_wo) request_wo();
        if (!TILED || !skip_front) {
        STAMP(2);
        EGX_PHASE();
        // Q | K | V (with bias) leave for the backward pass, which loads them instead of recomputing the layer input (LayerNorm +
        // embeddings + dropout hash) and its projection; the layer input itself follows after the projection (operand of the
        // in-projection weight gradient). Order matters: a global load that is used while stores are in flight waits for the
        // stores too, so the second half's weights and both halves' biases are requested BEFORE the stores they would wait for.
        float* qkv_g = p.qkv_out + ((size_t)l * p.B + clip) * SP * (3 * FD);       // all 48 rows of the clip grid: unconditional stores
        // ---- QKV projection: 24 feature tiles, 6 per wave, K = 128
        {
            auto request = [&](int half) { request_qkv(w, half); };
            if (!(CUT && qkv_requested)) request(0);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 acc[3][NT];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
                __builtin_amdgcn_sched_barrier(0);
                pin_all(wa);
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb) {
                    Frag<CM> b[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) b[t] = load_frag<CM>(Xs + (t * 16 + r) * LDX + kb * 32, q);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        Frag<CM> a = w_frag<CM>(wa[i][kb]);
#pragma unroll
                        for (int t = 0; t < NT; ++t) mma<CM>(acc[i][t], a, b[t]);
                    }
                }
                float4 bbv[3] = {bbn[0], bbn[1], bbn[2]};
                __builtin_amdgcn_sched_barrier(0);
                if (half == 0) request(1);          // into the registers the MFMAs above have just released
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    int f0 = (wave * 6 + half * 3 + i) * 16 + 4 * q;      // 0..383
                    const float4 bb = bbv[i];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        int tok = t * 16 + r;
                        float4 o = make_float4(acc[i][t][0] + bb.x, acc[i][t][1] + bb.y, acc[i][t][2] + bb.z, acc[i][t][3] + bb.w);
#ifndef EGX_DIAG_NOQKVSTORE
                        *reinterpret_cast<float4*>(qkv_g + (size_t)tok * (3 * FD) + f0) = o;
#endif
                        if (f0 < FD) {
                            *reinterpret_cast<float4*>(Qs + tok * LDX + f0) = o;
                        } else if (f0 < 2 * FD) {
                            *reinterpret_cast<float4*>(Ks + tok * LDX + (f0 - FD)) = o;
                        } else {
                            int c = f0 - 2 * FD;
                            bool kv = tok < S;
                            Vt[(c + 0) * LDV + tok] = kv ? o.x : 0.f;
                            Vt[(c + 1) * LDV + tok] = kv ? o.y : 0.f;
                            Vt[(c + 2) * LDV + tok] = kv ? o.z : 0.f;
                            Vt[(c + 3) * LDV + tok] = kv ? o.w : 0.f;
                        }
                    }
                }
            }
            if (!(CUT && l > 0))    // (cut mode, l > 0: the input was loaded from there)
            store_block(TILED ? p.xin_out + ((size_t)l * p.Ntok + tokbase) * FD : p.xin_out + ((size_t)l * p.B + clip) * S * FD, Xs, S);      // no global load follows before the out-projection
            if constexpr (TILED) return;        // the attention of the whole clip is another launch (tiled_attn_fwd)
            if (NT < 4) {   // zero the key padding columns SP..63 of V^T
                for (int i = tid; i < FD * (64 - SP); i += 256) {
                    int c = i / (64 - SP), k = SP + i % (64 - SP);
                    Vt[c * LDV + k] = 0.f;
                }
            }
        }
        __syncthreads();

        STAMP(3);
        EGX_PHASE();
        if constexpr (CUT) request_wo();       // the out-projection's weights arrive under the attention (round 6)
        // ---- attention: wave = head. S^T = K Q^T (key rows, query columns), softmax over rows, O^T = V^T P^T
        if constexpr (!TILED) {
#pragma unroll
          for (int hl = 0; hl < HPW; ++hl) {
            const int h = wave * HPW + hl, hc = h * DH;
            const float scale = DH == 32 ? 0.17677669529663687f : 0.25f;   // 1/sqrt(d_h)
            f32x4 sc[NT][NT];                            // [key tile][query tile]
            Frag<CM> kq[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) kq[t] = load_head_frag<CM, DH>(Qs + (t * 16 + r) * LDX + hc, q);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                Frag<CM> a = load_head_frag<CM, DH>(Ks + (kt * 16 + r) * LDX + hc, q);
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    sc[kt][qt] = f32x4{0, 0, 0, 0};
                    mma<CM>(sc[kt][qt], a, kq[qt]);
                }
            }
            // softmax over keys for every query column; this lane holds keys kt*16 + 4q + e
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int key = kt * 16 + 4 * q + e;
                        float s = (key < S) ? sc[kt][qt][e] * scale : -INFINITY;
                        sc[kt][qt][e] = s;
                        m = fmaxf(m, s);
                    }
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pv = __expf(sc[kt][qt][e] - m);
                        sc[kt][qt][e] = pv;
                        sum += pv;
                    }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                float inv = 1.f / sum;
                int query = qt * 16 + r;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pv = sc[kt][qt][e] * inv;
                        if (w.attn_thresh) {
                            int key = kt * 16 + 4 * q + e;
                            pv *= drop_scale(k_attn, (uint32_t)((clip * NHEAD + h) * 64 + query), (uint32_t)key, w.attn_thresh, w.drop_inv);
                        }
                        sc[kt][qt][e] = pv;
                    }
            }
            // O^T[c][query] = sum_key V^T[c][key] P^T[key][query]; keys in K-blocks of 32 = key-tile pairs
            constexpr int NCT = DH / 16;
            f32x4 oc[NCT][NT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) oc[ct][qt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < (NT + 1) / 2; ++kb) {
                Frag<CM> a[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[ct] = load_frag<CM>(Vt + (hc + ct * 16 + r) * LDV + kb * 32, q);
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    f32x4 z = f32x4{0, 0, 0, 0};
                    Frag<CM> b = chain_frag<CM>(sc[2 * kb][qt], (2 * kb + 1 < NT) ? sc[(2 * kb + 1 < NT) ? 2 * kb + 1 : 0][qt] : z);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) mma<CM>(oc[ct][qt], a[ct], b);
                }
            }
            // write O token-major over this head's Q columns (only this wave reads/writes them)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    int tok = qt * 16 + r;
                    *reinterpret_cast<float4*>(Qs + tok * LDX + hc + ct * 16 + 4 * q) =
                        make_float4(oc[ct][qt][0], oc[ct][qt][1], oc[ct][qt][2], oc[ct][qt][3]);
                }
          }
        }
        __syncthreads();
        }       // Q | K | V + attention
        skip_front = false;

        STAMP(4);
        EGX_PHASE();
        if (ps_g < 6) *reinterpret_cast<f32x4*>(PS + (6 + ps_g) * FD + ps_c) = pl_v;
        // ---- out-projection + residual -> res1 (X1 region), 2 feature tiles per wave
        {
            f32x4 acc[2][NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
            if constexpr (!CUT) { if (!(TILED && skip_front_wo)) request_wo(); }
            __builtin_amdgcn_sched_barrier(0);
            pin_all(wo);
#pragma unroll
            for (int kb = 0; kb < FD / 32; ++kb) {
                Frag<CM> b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = load_frag<CM>(Qs + (t * 16 + r) * LDX + kb * 32, q);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Frag<CM> a = w_frag<CM>(wo[i][kb]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mma<CM>(acc[i][t], a, b[t]);
                }
            }
            __syncthreads();        // the staged rows 6..11 are visible (the epilogue reads out_proj_b from them)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int f0 = (wave * 2 + i) * 16 + 4 * q;
                float4 bb = *reinterpret_cast<const float4*>(PS + 8 * FD + f0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    int tok = t * 16 + r;
                    if (tok < S) {
                        float o[4] = {acc[i][t][0] + bb.x, acc[i][t][1] + bb.y, acc[i][t][2] + bb.z, acc[i][t][3] + bb.w};
                        float4 xr = *reinterpret_cast<const float4*>(Xs + tok * LDX + f0);
                        if (w.res_thresh) {
                            uint32_t orow = (uint32_t)(tokbase + tok);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] *= drop_scale(k_res1, orow, (uint32_t)(f0 + e), w.res_thresh, w.drop_inv);
                        }
                        *reinterpret_cast<float4*>(X1 + tok * LDX + f0) = make_float4(o[0] + xr.x, o[1] + xr.y, o[2] + xr.z, o[3] + xr.w);
                    }
                }
            }
        }
        __syncthreads();
        STAMP(5);
        // ---- LayerNorm1 in place (res1 saved to HBM for the backward)
        // CM_SPLIT: x1 is also split into bf16 operand planes for the FFN loop (over Q / K, dead since the out-projection)
        unsigned short* XP = reinterpret_cast<unsigned short*>(Qs);
        constexpr int XPS = SP * LDXH;
        ln_rows_lds(X1, S, PS + 6 * FD, PS + 7 * FD, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
            store32(X1 + row * LDX + c0, y);
            if constexpr (CM == CM_SPLIT) {
                uint32_t h[16], m[16], lo[16];
                split32(y, h, m, lo);
                store_parts32(XP + row * LDXH + c0, (size_t)XPS, h, m, lo);
            }
        }, [&] {       // behind the request for the LayerNorm weights, so that their wait does not include these stores
            store_block(sv_res1, X1, S);
            __syncthreads();        // X1 is normalised in place
        });
        if constexpr (CM == CM_SPLIT) {     // padded rows of the planes: zero operands
            for (int i = tid; i < 3 * (SP - S) * (LDXH / 4); i += 256) {
                int pl = i / ((SP - S) * (LDXH / 4)), rem = i - pl * ((SP - S) * (LDXH / 4));
                *reinterpret_cast<uint2*>(XP + pl * XPS + (S + rem / (LDXH / 4)) * LDXH + (rem % (LDXH / 4)) * 4) = make_uint2(0, 0);
            }
        }
        __syncthreads();
        if constexpr (CM == CM_BF16) {      // bf16 mode: one bf16 plane (L, B*48, 128), same hand-over
            if (p.x1p_out) store_block_bf16(p.x1p_out + ((size_t)l * p.B + clip) * FUSED_TOK_PAD * FD, X1, S);
        }
        if constexpr (CM == CM_SPLIT) {
            // x1 leaves for the weight-gradient kernel in the same three parts ((L, 3, N, 128) bf16): dense 16-byte pieces in
            // lane order out of the LDS planes (stores from the LayerNorm lanes, 64 B per lane and part, cost the kernel 10 us)
            if (p.x1p_out) {
                static_assert(SP == FUSED_TOK_PAD, "the operand planes live on the 48-row clip grid");
                const size_t plane = (size_t)p.B * SP * FD;
                unsigned short* dst = p.x1p_out + (size_t)l * 3 * plane + (size_t)clip * SP * FD;
                for (int i = tid; i < 3 * SP * (FD / 8); i += 256) {
                    int part = i / (SP * (FD / 8)), rem = i - part * (SP * (FD / 8));
                    int row = rem >> 4, c8 = rem & 15;
                    const unsigned short* src = XP + part * XPS + row * LDXH + c8 * 8;        // rows are 8-byte aligned
                    const uint2 a = *reinterpret_cast<const uint2*>(src), b = *reinterpret_cast<const uint2*>(src + 4);
                    *reinterpret_cast<uint4*>(dst + part * plane + rem * 8) = make_uint4(a.x, a.y, b.x, b.y);
                }
            }
        }

        if constexpr (CUT) {        // x1 as fp32 rows: ffn_fwd_kernel's residual (and, in exact-fp32 mode, its operand; the weight-gradient kernel's too)
            store_block(p.x1f_out + ((size_t)l * p.Ntok + tokbase) * FD, X1, S);
            STAMP(6);
            return;
        }
        STAMP(6);
        EGX_PHASE();
        // SLICED: this pass walks the hidden blocks of slice `sl_cur` — its own first; afterwards any whose partial sum does not
        // arrive in time is computed here too (slice_wait): the launch does not depend on its workgroups being resident together
        int sl_cur = slice, sl_k = 0;
        for (;;) {
        // ---- FFN: hidden blocks of 32 split across waves; H^T = relu(W1 x1^T + b1) chained into Y^T += W2 H^T
        {
            // x1 as B operand: resident in registers for bf16 (48 VGPRs); re-read from LDS per hidden block in fp32,
            // where one block's MFMAs take 12k cycles and the 24 ds_read_b128 are free
            constexpr bool XRES = CM == CM_BF16;
            constexpr int XR = XRES ? FD / 32 : 1;
            Frag<CM> xb[XR][NT];
            if constexpr (XRES) {
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb)
#pragma unroll
                    for (int t = 0; t < NT; ++t) xb[kb][t] = load_frag<CM>(X1 + (t * 16 + r) * LDX + kb * 32, q);
            }
            f32x4 y[8][NT];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) y[i][t] = f32x4{0, 0, 0, 0};
            const int nhb = p.d_ff / 32;
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // hidden-block indices stay scalar: SGPR-based weight addresses
            // ROLLING REFILL: the registers of a weight fragment are reloaded with the same fragment of the NEXT hidden block right
            // behind the MFMAs that consumed it. Every fragment then has a whole block (3k-10k cycles) to arrive, the loads are
            // spread over the MFMA phases instead of queueing as one batch in front of the CU's one vector-memory pipe (18-50
            // wave loads issued back to back stall the wave 0.8-1.9k cycles per block: 64 B/clk shared by four waves), and no
            // second register set is needed. Every memory operation of the loop is UNCONDITIONAL (the last block refills itself,
            // H tiles and alive bits always leave): with a load or store under a branch hipcc's s_waitcnt insertion assumes the
            // fewest outstanding operations of any path and a fragment wait also drains everything issued after the fragment.
            WRaw<CM> w1r[2][FD / 32];   // W1 rows of the current hidden block
            WRaw<CM> w2r[8];            // W2 columns of the current hidden block
            float4 b1r[2];
            const int nit = nhb / 4 / n_slices;           // sliced mode: blocks [slice * nit, (slice + 1) * nit) of every wave's walk
            const int j0 = sl_cur * nit;
            // Every CU walks the same weights. All in step (rot_mode 1) they hammer the same few L2 lines at once (+2.5 % step
            // time); every clip at its own starting block (rot_mode 0, round 2) the XCD's instantaneous working set is the whole
            // 3-7.5 MB of packed weights and the 4 MB L2 thrashes (+150 MB of re-fetches per step). Default (rot_mode 4): the
            // clips of an XCD (clip >> 3 counts them) start 0..3 blocks apart — same speed as mode 0, the traffic of mode 1.
            const int rot = p.rot_mode == 0 ? (int)((clip * 11u + (clip >> 3) * 5u) % (unsigned)nit)
                          : p.rot_mode == 2 ? (int)(((unsigned)(clip >> 3) & 3u) * (unsigned)nit / 4u)
                          : p.rot_mode == 3 ? (int)(((unsigned)(clip >> 3) & 1u) * (unsigned)nit / 2u)
                          : p.rot_mode == 4 ? (int)(((unsigned)(clip >> 3) & 3u) % (unsigned)nit)
                          : p.rot_mode == 5 ? (int)(((unsigned)(clip >> 3) & 7u) % (unsigned)nit) : 0;
            auto hb_of = [&](int it) { int j = it + rot; if (j >= nit) j -= nit; return wave_s + 4 * (j0 + j); };
            // the dropout keep-scale 1 / (1 - p) is folded into the packed W1 (encoder.hip) and, here, into b1:
            // relu(s (W1 x + b1)) = s relu(W1 x + b1) for s > 0, so the epilogue has no multiply
            const float bscale = w.ffn_thresh ? w.drop_inv : 1.f;
            const size_t bits_base = ((size_t)l * p.B + clip) * nhb * 64 + lane;
            constexpr int ESZ = CM == CM_BF16 ? 2 : 4;
            const int nht = p.d_ff / 16;
            char* const hid_base = (char*)p.hid_out + ((size_t)l * p.B + clip) * NT * nht * (size_t)(HTILE_ELEMS * ESZ);
            {
                const int hb0 = hb_of(0);
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb)
#pragma unroll
                    for (int i = 0; i < 2; ++i) w1r[i][kb] = load_w<CM>(w.lin1_wp, hb0 * 2 + i, FD / 32, kb, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i) b1r[i] = *reinterpret_cast<const float4*>(w.lin1_b + hb0 * 32 + i * 16 + 4 * q);
#pragma unroll
                for (int i = 0; i < 8; ++i) w2r[i] = load_w<CM>(w.lin2_wp, i, nhb, hb0, lane);
            }
            LSTAMP_INIT();
            // (training with dropout only: the epilogue without the mask is half as long, and a second instantiation of the pipelined loop costs ~60
            // registers of hoisted loop invariants; the sliced instantiation has no registers left for the second set either: 37 spills)
            bool ffn_done = false;
            if constexpr (CM == CM_BF16 && EGX_FFN_PIPE && !SLICED) if (w.ffn_thresh) {
                ffn_done = true;
                // bf16 (round 6): ONE wave per SIMD and 16 cycles of matrix pipe per MFMA against ~280 VALU instructions of epilogue per hidden
                // block: run block after block as [W1 x1 | epilogue | W2 H] and the matrix pipe idles through every epilogue (stamps: GEMM1 11.5k +
                // epilogue 21.8k + GEMM2 10.6k of the 49k-cycle loop; 26 % busy). The three are independent ACROSS blocks, so the loop is
                // software-pipelined: the epilogue of block `it` is issued together with GEMM1 of block it + 1 (into a second accumulator set)
                // and GEMM2 of block it - 1 (from the operand fragments the previous epilogue left), one MFMA per ~6 VALU instructions
                // (sched_group_barrier). The first GEMM2 meets zero fragments and the last GEMM1 is discarded: +2 / 16 of the MFMAs, all hidden.
                {
                    const uint32_t rowbase = TILED ? (uint32_t)tokbase : (uint32_t)(clip * 64);
                    f32x4 hc[2][NT];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int t = 0; t < NT; ++t) hc[i][t] = f32x4{0, 0, 0, 0};
                    {
                        const int hb1 = hb_of(1 < nit ? 1 : 0);
#pragma unroll
                        for (int kb = 0; kb < FD / 32; ++kb)
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                pin(w1r[i][kb]);
                                Frag<CM> a = w_frag<CM>(w1r[i][kb]);
#pragma unroll
                                for (int t = 0; t < NT; ++t) mma<CM>(hc[i][t], a, xb[kb][t]);
                                __builtin_amdgcn_sched_barrier(0);
                                w1r[i][kb] = load_w<CM>(w.lin1_wp, hb1 * 2 + i, FD / 32, kb, lane);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                    // two register sets that swap roles from block to block (no copies): (pre-activations, operand fragments) of this block / the next
                    f32x4 hc2[2][NT];
                    Frag<CM> hqA[NT], hqB[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) hqA[t].v = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
                    auto step = [&](int it, f32x4 (&hc)[2][NT], f32x4 (&hn)[2][NT], const Frag<CM> (&hq_prev)[NT], Frag<CM> (&hq)[NT]) {
                        const int hb = hb_of(it);
                        const int hbn = hb_of(it + 1 < nit ? it + 1 : it);
                        const int hb2 = hb_of(it + 2 < nit ? it + 2 : nit - 1);
                        const int hbp = hb_of(it > 0 ? it - 1 : 0);
                        __builtin_amdgcn_sched_barrier(0);
                        LSTAMP(0);
                        float bv[2][4];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            bv[i][0] = b1r[i].x * bscale; bv[i][1] = b1r[i].y * bscale; bv[i][2] = b1r[i].z * bscale; bv[i][3] = b1r[i].w * bscale;
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 2; ++i) b1r[i] = *reinterpret_cast<const float4*>(w.lin1_b + hbn * 32 + i * 16 + 4 * q);
                        __builtin_amdgcn_sched_barrier(0);
                        // 16 steps: one weight fragment's MFMAs (steps 0-7: GEMM1 of the NEXT block -> hn, the last iteration's is discarded; 8-15:
                        // GEMM2 of the PREVIOUS block) + one slice of this block's epilogue. ONE ring of eight fragment slots (w1r) serves both
                        // streams: a slot's fragment is requested eight steps (~1k cycles) before its MFMAs — a second set of eight costs 32 registers
                        // the kernel does not have (the two accumulator / operand sets already take it to 512)
                        // (2 NT slices bias + dropout of one 16 x 16 tile, NT slices alive bits + ReLU of eight units, NT slices operand fragment +
                        // H tile of a token tile)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int t = 0; t < NT; ++t) hn[i][t] = f32x4{0, 0, 0, 0};
                        uint32_t dead = 0;
                        static_assert(4 * NT <= 16, "epilogue slices");
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            if (k < 8) {
                                const int kb = k >> 1, i = k & 1;
                                pin(w1r[i][kb]);
                                Frag<CM> a = w_frag<CM>(w1r[i][kb]);
#pragma unroll
                                for (int t = 0; t < NT; ++t) mma<CM>(hn[i][t], a, xb[kb][t]);
#if EGX_FFN_PIPE_RING == 16
                                w1r[i][kb] = load_w<CM>(w.lin1_wp, hb2 * 2 + i, FD / 32, kb, lane);
#else
                                w1r[i][kb] = load_w<CM>(w.lin2_wp, k, nhb, hbp, lane);             // slot k: W2 fragment k, needed eight steps on
#endif
                            } else {
                                const int j = k - 8, i = j & 1, kb = j >> 1;
#if EGX_FFN_PIPE_RING == 16
                                pin(w2r[j]);
                                Frag<CM> a = w_frag<CM>(w2r[j]);
#pragma unroll
                                for (int t = 0; t < NT; ++t) mma<CM>(y[j][t], a, hq_prev[t]);
                                w2r[j] = load_w<CM>(w.lin2_wp, j, nhb, hb, lane);
#else
                                pin(w1r[i][kb]);
                                Frag<CM> a = w_frag<CM>(w1r[i][kb]);
#pragma unroll
                                for (int t = 0; t < NT; ++t) mma<CM>(y[j][t], a, hq_prev[t]);
                                w1r[i][kb] = load_w<CM>(w.lin1_wp, hb2 * 2 + i, FD / 32, kb, lane);   // slot j: W1 fragment j of the block after next
#endif
                            }
                            if (k < 2 * NT) {
                                const int i = k / NT, t = k % NT;
                                if constexpr (true) {
                                    const uint32_t cq = (uint32_t)(hb * 32 + i * 16 + 4 * q) >> 2;
                                    const uint2 h = rand_quad(k_ffn, rowbase + (uint32_t)(t * 16 + r), cq);
                                    // dropped units become negative: the ReLU zeroes them and their sign bit marks them dead
                                    hc[i][t][0] = keep_lo(h.x, w.ffn_thresh) ? hc[i][t][0] + bv[i][0] : -1.f;
                                    hc[i][t][1] = keep_hi(h.x, w.ffn_thresh) ? hc[i][t][1] + bv[i][1] : -1.f;
                                    hc[i][t][2] = keep_lo(h.y, w.ffn_thresh) ? hc[i][t][2] + bv[i][2] : -1.f;
                                    hc[i][t][3] = keep_hi(h.y, w.ffn_thresh) ? hc[i][t][3] + bv[i][3] : -1.f;
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) hc[i][t][e] += bv[i][e];
                                }
                            } else if (k < 3 * NT) {
                                const int j = k - 2 * NT;       // units 8 (NT - j) - 1 .. 8 (NT - 1 - j): the word's bit order is the loop's below
#pragma unroll
                                for (int kk = 8 * (NT - j) - 1; kk >= 8 * (NT - 1 - j); --kk) {
                                    const int i = kk / (NT * 4), t = (kk / 4) % NT, e = kk & 3;
                                    dead = __builtin_amdgcn_alignbit(dead, __float_as_uint(hc[i][t][e]), 31);
                                    hc[i][t][e] = __int_as_float(max(__float_as_int(hc[i][t][e]), 0));
                                }
                                if (k == 3 * NT - 1) p.relu_bits[bits_base + (size_t)hb * 64] = ~dead & ((1u << (2 * NT * 4)) - 1u);
                            } else if (k < 4 * NT) {
                                const int t = k - 3 * NT;
                                hq[t] = chain_frag<CM>(hc[0][t], hc[1][t]);
                                const u32x4 u = __builtin_bit_cast(u32x4, hq[t].v);
                                store_hid_tile_bf16(hid_base + (size_t)hb * 2 * (HTILE_ELEMS * ESZ) + (size_t)t * nht * (HTILE_ELEMS * ESZ), u, lane, S - t * 16);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        LSTAMP(3);
                    };
                    auto last_gemm2 = [&](const Frag<CM> (&hq_last)[NT]) {      // (its W2 columns are requested here: one exposed round trip per layer)
#if EGX_FFN_PIPE_RING != 16
                        const int hbl = hb_of(nit - 1);
#pragma unroll
                        for (int i = 0; i < 8; ++i) w2r[i] = load_w<CM>(w.lin2_wp, i, nhb, hbl, lane);
#endif
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            Frag<CM> a = w_frag<CM>(w2r[i]);
#pragma unroll
                            for (int t = 0; t < NT; ++t) mma<CM>(y[i][t], a, hq_last[t]);
                        }
                    };
                    int it = 0;
                    for (; it + 1 < nit; it += 2) {
                        step(it, hc, hc2, hqA, hqB);
                        step(it + 1, hc2, hc, hqB, hqA);
                    }
                    if (it < nit) {     // odd block count (narrow FFNs, deep slicing)
                        step(it, hc, hc2, hqA, hqB);
                        last_gemm2(hqB);
                    } else {
                        last_gemm2(hqA);
                    }
                    LSTAMP(4);
                }
            }
            if (!ffn_done)
            for (int it = 0; it < nit; ++it) {
                const int hb = hb_of(it);
                const int hbn = hb_of(it + 1 < nit ? it + 1 : it);     // the last block refills itself (never used)
                __builtin_amdgcn_sched_barrier(0);
                LSTAMP(0);
                f32x4 hacc[2][NT];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) hacc[i][t] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < FD / 32; ++kb) {
                    if constexpr (CM == CM_SPLIT) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) xb[0][t] = load_split_frag(XP, XPS, t * 16 + r, kb * 32, q);
                    } else if constexpr (!XRES) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) xb[0][t] = load_frag<CM>(X1 + (t * 16 + r) * LDX + kb * 32, q);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        pin(w1r[i][kb]);
                        Frag<CM> a = w_frag<CM>(w1r[i][kb]);
#pragma unroll
                        for (int t = 0; t < NT; ++t) mma<CM>(hacc[i][t], a, xb[XRES ? kb : 0][t]);
                        __builtin_amdgcn_sched_barrier(0);
                        w1r[i][kb] = load_w<CM>(w.lin1_wp, hbn * 2 + i, FD / 32, kb, lane);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                float bv[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bv[i][0] = b1r[i].x * bscale; bv[i][1] = b1r[i].y * bscale; bv[i][2] = b1r[i].z * bscale; bv[i][3] = b1r[i].w * bscale;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) b1r[i] = *reinterpret_cast<const float4*>(w.lin1_b + hbn * 32 + i * 16 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
                LSTAMP(1);
                LSTAMP(2);
                if (w.ffn_thresh) {     // one wave-uniform branch per hidden block (no memory operation inside)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const uint32_t cq = (uint32_t)(hb * 32 + i * 16 + 4 * q) >> 2;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
#ifdef EGX_DIAG_NOHASH
                            const uint2 h = make_uint2(cq * 0x9E3779B1u + t, cq * 0x85EBCA77U + r);
#else
                            const uint2 h = rand_quad(k_ffn, TILED ? (uint32_t)(tokbase + t * 16 + r) : (uint32_t)(clip * 64 + t * 16 + r), cq);
#endif
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
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e) hacc[i][t][e] += bv[i][e];
                }
                // "Alive" bits for the backward: ReLU active AND kept by the dropout = sign bit clear (a pre-activation of exactly
                // +0 counts as alive: measure zero). The backward then needs neither the H GEMM nor the dropout RNG:
                // dH = alive ? dY W2 / (1 - p) : 0. One v_alignbit per unit shifts its sign into the word; 256 B per wave, coalesced.
                uint32_t dead = 0;
#pragma unroll
                for (int k = 2 * NT * 4 - 1; k >= 0; --k) {
                    const int i = k / (NT * 4), t = (k / 4) % NT, e = k & 3;
                    dead = __builtin_amdgcn_alignbit(dead, __float_as_uint(hacc[i][t][e]), 31);
                }
                p.relu_bits[bits_base + (size_t)hb * 64] = ~dead & ((1u << (2 * NT * 4)) - 1u);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e)     // ReLU as ONE integer max on the bit pattern (a float max pays a canonicalising second op)
                            hacc[i][t][e] = __int_as_float(max(__float_as_int(hacc[i][t][e]), 0));
                Frag<CM> hbq[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) hbq[t] = chain_frag<CM>(hacc[0][t], hacc[1][t]);
#ifndef EGX_DIAG_NOSTORE
                {       // H tiles for the weight-gradient kernel
                    char* hb_base = hid_base + (size_t)hb * 2 * (HTILE_ELEMS * ESZ);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (CM == CM_BF16) {      // the packed operand words ARE the two accumulator-layout tiles
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
                LSTAMP(3);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pin(w2r[i]);
                    Frag<CM> a = w_frag<CM>(w2r[i]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mma<CM>(y[i][t], a, hbq[t]);
                    __builtin_amdgcn_sched_barrier(0);
                    w2r[i] = load_w<CM>(w.lin2_wp, i, nhb, hbn, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
                LSTAMP(4);
            }
            LSTAMP_FLUSH();
            STAMP(7);
            if constexpr (CM == CM_SPLIT) __syncthreads();      // the partials below overwrite the operand planes
            // cross-wave reduction through LDS: wave 0 -> Xs region, waves 1..3 -> Part (aliases Q/K/V^T, now dead)
            float* mine = (wave == 0) ? Xs : Part + (wave - 1) * SP * LDX;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<float4*>(mine + (t * 16 + r) * LDX + i * 16 + 4 * q) =
                        make_float4(y[i][t][0], y[i][t][1], y[i][t][2], y[i][t][3]);
        }
        __syncthreads();
        if constexpr (!SLICED) {
            break;
        } else {
            float* xc = p.xchg + ((size_t)l * p.B + clip) * n_slices * (FUSED_TOK_PAD * FD);
            unsigned* fl = p.xflags + ((size_t)l * p.B + clip) * SLICE_MAX;
            slice_publish(Xs, Part, Part + SP * LDX, Part + 2 * SP * LDX, LDX, S, xc + (size_t)sl_cur * (FUSED_TOK_PAD * FD), fl + sl_cur);
            bool steal = false;
            while (++sl_k < n_slices) {
                const int s2 = slice + sl_k < n_slices ? slice + sl_k : slice + sl_k - n_slices;
                if (!slice_wait(fl + s2)) { sl_cur = s2; steal = true; slice_stolen_note(&g_stolen_fwd); break; }
            }
            if (!steal) {       // the sum over the waves AND the slices of the clip comes back in Xs
                slice_gather(xc, n_slices, Xs, LDX, S);
                break;
            }
            if constexpr (CM == CM_SPLIT) {     // the partial blocks overwrote the operand planes of x1: split it again
                const int row = tid >> 2, c0 = (tid & 3) * 32;
                if (row < S) {
                    float xv[32];
                    uint32_t h[16], m[16], lo[16];
                    load32(X1 + row * LDX + c0, xv);
                    split32(xv, h, m, lo);
                    store_parts32(XP + row * LDXH + c0, (size_t)XPS, h, m, lo);
                }
                for (int i = tid; i < 3 * (SP - S) * (LDXH / 4); i += 256) {
                    int pl = i / ((SP - S) * (LDXH / 4)), rem = i - pl * ((SP - S) * (LDXH / 4));
                    *reinterpret_cast<uint2*>(XP + pl * XPS + (S + rem / (LDXH / 4)) * LDXH + (rem % (LDXH / 4)) * 4) = make_uint2(0, 0);
                }
                __syncthreads();
            }
        }
        }
        STAMP(8);
        // ---- sum partials + bias + residual -> res2 (in X1), then LayerNorm2 -> next layer input / tokens_out
        {
            const int row = tid >> 2, c0 = (tid & 3) * 32;
            if (row < S) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int o = row * LDX + c0 + 4 * j;
                    float4 a0, a1, a2, a3;
                    if (n_slices > 1) {
                        a0 = *reinterpret_cast<const float4*>(Xs + o);
                        a1 = make_float4(0, 0, 0, 0); a2 = a1; a3 = a1;
                    } else {
                        a0 = *reinterpret_cast<const float4*>(Xs + o);
                        a1 = *reinterpret_cast<const float4*>(Part + o);
                        a2 = *reinterpret_cast<const float4*>(Part + SP * LDX + o);
                        a3 = *reinterpret_cast<const float4*>(Part + 2 * SP * LDX + o);
                    }
                    float4 x1 = *reinterpret_cast<const float4*>(X1 + o);
                    float4 b2 = *reinterpret_cast<const float4*>(PS + 11 * FD + c0 + 4 * j);
                    float f[4] = {a0.x + a1.x + a2.x + a3.x + b2.x, a0.y + a1.y + a2.y + a3.y + b2.y,
                                  a0.z + a1.z + a2.z + a3.z + b2.z, a0.w + a1.w + a2.w + a3.w + b2.w};
                    if (w.res_thresh) {
                        uint32_t orow = (uint32_t)(tokbase + row);
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] *= drop_scale(k_res2, orow, (uint32_t)(c0 + 4 * j + e), w.res_thresh, w.drop_inv);
                    }
                    *reinterpret_cast<float4*>(X1 + o) = make_float4(f[0] + x1.x, f[1] + x1.y, f[2] + x1.z, f[3] + x1.w);
                }
            }
        }
        __syncthreads();
        {
            bool last = (l + 1 == p.n_layers);
            ln_rows_lds(X1, S, PS + 9 * FD, PS + 10 * FD, p.eps, [&](int row, int c0, float (&x)[32], float (&y)[32]) {
                if (last && p.tokens_out && (TILED || row < p.out_T)) store32(p.tokens_out + (TILED ? tokbase + row : (size_t)clip * p.out_T + row) * FD + c0, y);
                if (!last || p.head.n_out > 0 || p.tce_W) store32(Xs + row * LDX + c0, y);
            }, [&] { store_block(sv_res2, X1, S); });        // (the LayerNorm leaves X1 alone)
        }
        __syncthreads();
        STAMP(9);
        if (l + 1 < p.n_layers) {
            for (int i = tid; i < (SP - S) * LDX; i += 256) Xs[S * LDX + i] = 0.f;
            __syncthreads();
        }
    }

    // ---- optional per-token classifier + weighted cross entropy on the returned tokens (egx_token_ce; the arithmetic of linear_ce_fwd_kernel,
    // train.hip): four lanes per token row, the clip's loss / correct-frame terms through the arrival counter (no launch in front that could zero them)
    if constexpr (!TILED && !SLICED && !CUT) if (p.tce_W) {
        constexpr int MAXC = 8;
        float* red = X1;                        // scratch (X1 is dead)
        const int C = p.tce_C;
        const float* TW = PS + 12 * FD;         // classifier rows, bias, class weights: staged at kernel entry
        float cw[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) cw[c] = TW[8 * FD + 8 + c];
        auto wof = [&](int64_t y) {             // class weight by compare chain: no load that depends on the label
            float wv = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) wv = y == c ? cw[c] : wv;
            return wv;
        };
        STAMP(11);
        if (lane == 0) red[wave] = tce_sw;      // the normaliser: every workgroup sums it itself, in the same order (at kernel entry)
        const int row = tid >> 2, part = tid & 3;
        const bool live = row < p.out_T;
        const size_t grow = (size_t)clip * p.out_T + (live ? row : 0);
        const int64_t yrow = tce_y;
        float acc[MAXC];
        {
            float x[32];
            load32(Xs + (live ? row : 0) * LDX + part * 32, x);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                acc[c] = 0.f;
                if (c < C) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 wv = *reinterpret_cast<const float4*>(TW + c * FD + part * 32 + 4 * j);
                        a += (x[4 * j] * wv.x + x[4 * j + 1] * wv.y) + (x[4 * j + 2] * wv.z + x[4 * j + 3] * wv.w);
                    }
                    acc[c] = quad_sum4(a) + TW[8 * FD + c];
                }
            }
        }
        __syncthreads();
        STAMP(12);
        const float wtot = (red[0] + red[1]) + (red[2] + red[3]);
        const float inv = 1.f / wtot;
        float lterm = 0.f, cterm = 0.f;
        if (live && part == 0) {
            float m = acc[0];
#pragma unroll
            for (int c = 1; c < MAXC; ++c) if (c < C) m = fmaxf(m, acc[c]);
            float e[MAXC], ssum = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) { e[c] = c < C ? __expf(acc[c] - m) : 0.f; ssum += e[c]; }
            const float rs = 1.f / ssum;
            const bool ok = yrow >= 0 && yrow < C;
            const float wy = ok ? wof(yrow) : 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    p.tce_logits[grow * C + c] = acc[c];
                    if (p.tce_probs) p.tce_probs[grow * C + c] = e[c] * rs;
                    p.tce_dlogits[grow * C + c] = ok ? wy * inv * (e[c] * rs - (c == (int)yrow ? 1.f : 0.f)) : 0.f;
                    if (ok && c == (int)yrow) lterm += wy * (m + __logf(ssum) - acc[c]);
                }
            const float pl = C > 1 ? rintf(e[1] * rs) : 0.f;
            if (p.tce_pred) p.tce_pred[grow] = pl;
            if (C > 1 && pl == (float)yrow) cterm += 1.f;
        }
        lterm = wsum(lterm); cterm = wsum(cterm);
        if (lane == 0) { red[4 + wave] = lterm; red[8 + wave] = cterm; }
        __syncthreads();
        STAMP(13);
        if (tid == 0) {
            const float l = ((red[4] + red[5]) + (red[6] + red[7])) * inv, cn = (red[8] + red[9]) + (red[10] + red[11]);
            // ONE returning atomic per clip: the arrival word counts clips in its low 12 bits and correct frames above them (every atomic on
            // this line waits for the other clips' — 256 of them finish together: three per clip cost 8 us at the end of the launch, measured)
            float* tacc = reinterpret_cast<float*>(p.tce_ticket + 1);
            atomicAdd(tacc, l);
            __threadfence();
            const unsigned mine = 1u + ((unsigned)cn << 12);
            const unsigned old = atomicAdd(p.tce_ticket, mine);
            if ((old & 4095u) == (unsigned)p.B - 1u) {      // last clip of the launch: publish, leave zeros
                __threadfence();
                *p.tce_loss = atomicExch(tacc, 0.f);
                if (p.tce_correct) *p.tce_correct = (float)((old + mine) >> 12);
                atomicExch(p.tce_ticket, 0u);
            }
        }
        STAMP(14);
    }

    // ---- optional pooled head: logits = Linear(LN(mean_s tokens)); tokens of the last layer are in Xs
    if (!TILED && p.head.n_out > 0) {
        float* pooled = X1;                      // 128 floats of scratch (X1 is dead)
        // fused weighted cross entropy (egx_ce): the labels and their class weights are requested here, under the pooling
        const FusedCe ce{p.ce_target, p.ce_weight, p.ce_loss, p.ce_dlogits, p.ce_B, p.ce_ticket};
        CeReq rq;
        if (ce.target) { ce_request_labels<256>(ce, tid, rq); ce_request_weights(ce, p.head.n_out, rq); }
        if (tid < FD) pooled[tid] = colsum_lds(Xs, 0, S, tid) * (1.f / (float)S);
        if (ce.target) ce_weight_partials<256>(ce, p.head.n_out, tid, rq, pooled + 256);
        __syncthreads();
        if (wave == 0) {
            float2 x = *reinterpret_cast<float2*>(pooled + 2 * lane);
            float mean = wsum(x.x + x.y) * (1.f / FD);
            float dx = x.x - mean, dy = x.y - mean;
            float rstd = rsqrtf(wsum(dx * dx + dy * dy) * (1.f / FD) + p.eps);
            float2 lw = *reinterpret_cast<const float2*>(p.head.ln_w + 2 * lane);
            float2 lb = *reinterpret_cast<const float2*>(p.head.ln_b + 2 * lane);
            float y0 = dx * rstd * lw.x + lb.x, y1 = dy * rstd * lw.y + lb.y;
            float zmine = 0.f;
            for (int o = 0; o < p.head.n_out; ++o) {
                float2 wv = *reinterpret_cast<const float2*>(p.head.W + (size_t)o * FD + 2 * lane);
                float sdot = wsum(y0 * wv.x + y1 * wv.y) + p.head.b[o];
                if (lane == 0) p.logits_out[(size_t)clip * p.head.n_out + o] = sdot;
                zmine = lane == o ? sdot : zmine;
            }
            if (ce.target) ce_clip<256>(ce, p.head.n_out, clip, lane, zmine, pooled + 256, slice == 0);
        }
    }
}

// ---- optional device timing ----------------------------------------------------------------------------
namespace {
// events are created on demand: a wide-path step records a few hundred GEMM launches per timer
struct TimerSlot { std::vector<hipEvent_t> a, b; int n = 0; };
TimerSlot g_timers[TIMER_COUNT];
int g_timing_on = 0;
constexpr int TIMER_MAX_EVENTS = 4096;
}
void timing_enable(int on) {
    g_timing_on = on;
    for (int i = 0; i < TIMER_COUNT; ++i) g_timers[i].n = 0;
}
void timing_begin(int which, hipStream_t st) {
    if (!g_timing_on) return;
    TimerSlot& t = g_timers[which];
    if (t.n >= TIMER_MAX_EVENTS) return;
    if ((int)t.a.size() <= t.n) {
        hipEvent_t ea, eb;
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        t.a.push_back(ea); t.b.push_back(eb);
    }
    (void)hipEventRecord(t.a[t.n], st);
}
void timing_end(int which, hipStream_t st) {
    if (!g_timing_on) return;
    TimerSlot& t = g_timers[which];
    if (t.n < (int)t.a.size() && t.n < TIMER_MAX_EVENTS) { (void)hipEventRecord(t.b[t.n], st); ++t.n; }
}
int timing_read(int which, double* total_ms, int* count) {
    if (which < 0 || which >= TIMER_COUNT) return 1;
    TimerSlot& t = g_timers[which];
    double tot = 0;
    for (int i = 0; i < t.n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(t.b[i]) != hipSuccess || hipEventElapsedTime(&ms, t.a[i], t.b[i]) != hipSuccess) return 1;
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = t.n;
    return 0;
}

size_t fused_lds_bytes(int NT) {
    int SP = NT * 16;
    return (size_t)(4 * SP * LDX + FD * LDV) * sizeof(float) + FUSED_MAX_SEG * sizeof(FusedSeg) + 16 + 21 * FD * sizeof(float);     // + the staged parameter rows (12) and the token classifier (8 + 1)
}

bool fused_supported(int d_model, int n_heads, int d_ff, int S, int nseg, const int* d_in, const int* T, const bool* has_proj) {
    if (d_model != FD || (n_heads != FH && n_heads != 2 * FH)) return false;
    if (d_ff % 128 != 0 || d_ff < 128) return false;
    if (S < 1 || S > 48) return false;   // NT = 4 would need 170 KB of LDS
    for (int i = 0; i < nseg; ++i) {
        if (!has_proj[i]) return false;
        if (d_in[i] % 128 != 0) return false;
        if (T[i] < 1) return false;
    }
    return true;
}

template <int CM, bool TILED, int DH>
static int launch_fwd(const FusedFwdParams& p, hipStream_t st) {
    // NT = 2 (S <= 32) is not instantiated: shorter sequences run the 48-row kernel with masked padding.
    EGX_CHECK(p.S <= 48, "fused: S=%d unsupported", p.S);
    const size_t lds = fused_lds_bytes(3);
    static bool attr_set = false;
    if (!attr_set) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_kernel<CM, 3, TILED, DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if constexpr (!TILED)
            EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_kernel<CM, 3, false, DH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    timing_begin(TIMER_FUSED_FWD, st);
    bool sliced = false;
    if constexpr (!TILED) sliced = p.n_slices > 1;
    if constexpr (!TILED) {
        if (p.mode == FUSED_MODE_ATTN) {
            static bool cut_attr = false;
            if (!cut_attr) {
                EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_kernel<CM, 3, false, DH, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                cut_attr = true;
            }
            EGX_CHECK(!sliced && p.x1f_out, "cut mode: one workgroup per clip, x1f_out set");
            hipLaunchKernelGGL((fused_fwd_kernel<CM, 3, false, DH, false, true>), dim3(p.B), dim3(256), lds, st, p);
            timing_end(TIMER_FUSED_FWD, st);
            EGX_LAUNCH_CHECK();
            return 0;
        }
    }
    if (sliced) {
        if constexpr (!TILED)
            hipLaunchKernelGGL((fused_fwd_kernel<CM, 3, false, DH, true>), dim3((p.B + 7) / 8 * 8 * p.n_slices), dim3(256), lds, st, p);
    } else {
        hipLaunchKernelGGL((fused_fwd_kernel<CM, 3, TILED, DH>), dim3(p.B), dim3(256), lds, st, p);
    }
    timing_end(TIMER_FUSED_FWD, st);
    EGX_LAUNCH_CHECK();
    return 0;
}

int ffn_rot_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_FFN_ROT"); v = e ? atoi(e) : 4; }
    return v;
}

int fused_forward(const FusedFwdParams& p, int compute, hipStream_t st) {
    if (p.mode != FUSED_MODE_FULL && p.mode != FUSED_MODE_ATTN) {        // tiled mode: bf16 and split only (the exact-fp32 mode stays on the generic kernels for S > 48)
        EGX_CHECK(compute == CM_BF16 || compute == CM_SPLIT, "tiled mode: compute must be bf16 or f32s");
        EGX_CHECK(p.n_heads == FH, "tiled mode: 4 heads of 32");
        return compute == CM_BF16 ? launch_fwd<CM_BF16, true, 32>(p, st) : launch_fwd<CM_SPLIT, true, 32>(p, st);
    }
    if (p.n_heads == 2 * FH)        // 8 heads of 16 (the HOI PNR / OSCC and action-recognition translators)
        return compute == CM_BF16 ? launch_fwd<CM_BF16, false, 16>(p, st) : compute == CM_SPLIT ? launch_fwd<CM_SPLIT, false, 16>(p, st) : launch_fwd<CM_F32, false, 16>(p, st);
    return compute == CM_BF16 ? launch_fwd<CM_BF16, false, 32>(p, st) : compute == CM_SPLIT ? launch_fwd<CM_SPLIT, false, 32>(p, st) : launch_fwd<CM_F32, false, 32>(p, st);
}

}  // namespace egx
