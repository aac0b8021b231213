// C ABI + host orchestration of the translator (see include/egot2x.h for the reference call sites each
// entry point replaces). Everything here only enqueues kernels on the caller's stream: no allocation,
// no synchronisation, hipGraph-capturable.
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/egot2x.h"
#include "common.h"
#include "kernels.h"
#include "fused.h"
#include "wide.h"
#include "wide_host.h"

// 1: with a valid weight cache the one-launch forward publishes the fused loss through an arrival counter; 0: a 4-byte memset node in front of it (A/B switch)
#ifndef EGX_CE_TICKET
#define EGX_CE_TICKET 1
#endif

namespace egx {

static thread_local char g_err[1024] = "";

static long long g_launches = 0;
void count_launch() { ++g_launches; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- workspace plan ---------------------------------------------------------------------------
struct LayerOff {
    size_t x_in, qkv, lse, attn_o, res1, stats1, x1, hid, res2, stats2;
};
struct Plan {
    int B = 0, S = 0, d = 0, H = 0, dff = 0, L = 0, nseg = 0;
    int vB = 0, tpc = 1;    // fused kernels: workgroups ("virtual clips") and 48-token tiles per clip; vB = B unless the tiled mode (S > 48) is planned
    size_t N = 0;
    int seg_off[EGX_MAX_SEGMENTS];
    size_t seg_pre[EGX_MAX_SEGMENTS], seg_stats[EGX_MAX_SEGMENTS];
    LayerOff layer[64];
    size_t saved_bytes = 0;
    // scratch
    size_t s_dA = 0, s_dB = 0, s_dqkv = 0, s_dhid = 0, s_slab = 0, slab_bytes = 0, s_det = 0, det_bytes = 0;
    size_t scratch_bytes = 0;
};

// NOT ::max: in hipcc host code the unqualified call resolves to max(int, int) and truncates sizes above 2 GiB
static inline size_t size_max(size_t a, size_t b) { return a > b ? a : b; }

static size_t take(size_t& cur, size_t bytes) {
    size_t o = cur;
    cur = align_up(cur + bytes, 256);
    return o;
}

static int make_plan(const egx_config* cfg, const egx_segment* segs, int B, Plan& pl) {
    EGX_CHECK(cfg && segs, "null config/segments");
    EGX_CHECK(cfg->n_segments >= 1 && cfg->n_segments <= EGX_MAX_SEGMENTS, "n_segments=%d out of range", cfg->n_segments);
    EGX_CHECK(cfg->n_layers >= 0 && cfg->n_layers <= 64, "n_layers=%d out of range", cfg->n_layers);
    EGX_CHECK(cfg->d_model > 0 && cfg->d_model % 4 == 0 && cfg->d_model <= 1024, "d_model=%d unsupported (multiple of 4, <= 1024)", cfg->d_model);
    EGX_CHECK(cfg->n_heads > 0 && cfg->d_model % cfg->n_heads == 0, "d_model=%d not divisible by n_heads=%d", cfg->d_model, cfg->n_heads);
    EGX_CHECK(cfg->d_ff > 0 && cfg->d_ff % 4 == 0, "d_ff=%d must be a positive multiple of 4", cfg->d_ff);
    EGX_CHECK(cfg->compute == EGX_F32 || cfg->compute == EGX_BF16 || cfg->compute == EGX_F32_SPLIT, "compute=%d unknown", cfg->compute);
    EGX_CHECK(B > 0, "empty batch (B=%d)", B);
    EGX_CHECK(cfg->out_tokens >= 0, "out_tokens=%d", cfg->out_tokens);
    pl.B = B; pl.vB = B; pl.tpc = 1; pl.d = cfg->d_model; pl.H = cfg->n_heads; pl.dff = cfg->d_ff; pl.L = cfg->n_layers; pl.nseg = cfg->n_segments;
    int S = 0;
    for (int i = 0; i < pl.nseg; ++i) {
        EGX_CHECK(segs[i].T > 0, "segment %d has T=%d", i, segs[i].T);
        EGX_CHECK(segs[i].d_in > 0, "segment %d has d_in=%d", i, segs[i].d_in);
        EGX_CHECK(segs[i].proj_w || segs[i].d_in == pl.d, "segment %d: identity projection needs d_in == d_model", i);
        EGX_CHECK(segs[i].pool >= 0 && (segs[i].proj_w || (!segs[i].feat_bf16 && segs[i].pool <= 1)),
                  "segment %d: bf16 / frame-pooled features need a projection", i);
        pl.seg_off[i] = S;
        S += segs[i].T;
    }
    pl.S = S;
    pl.N = (size_t)B * S;
    size_t d = pl.d, N = pl.N;
    size_t cur = 0;
    for (int i = 0; i < pl.nseg; ++i) {
        size_t rows = (size_t)B * segs[i].T;
        pl.seg_pre[i] = take(cur, rows * d * 4);
        pl.seg_stats[i] = take(cur, rows * 2 * 4);
    }
    for (int l = 0; l < pl.L; ++l) {
        LayerOff& o = pl.layer[l];
        o.x_in = take(cur, N * d * 4);
        o.qkv = take(cur, N * 3 * d * 4);
        o.lse = take(cur, (size_t)B * pl.H * S * 4);
        o.attn_o = take(cur, N * d * 4);
        o.res1 = take(cur, N * d * 4);
        o.stats1 = take(cur, N * 2 * 4);
        o.x1 = take(cur, N * d * 4);
        o.hid = take(cur, N * (size_t)pl.dff * 4);
        o.res2 = take(cur, N * d * 4);
        o.stats2 = take(cur, N * 2 * 4);
    }
    pl.saved_bytes = cur;

    size_t sc = 0;
    pl.s_dA = take(sc, N * d * 4);
    pl.s_dB = take(sc, N * d * 4);
    pl.s_dqkv = take(sc, N * 3 * d * 4);
    pl.s_dhid = take(sc, N * (size_t)pl.dff * 4);
    size_t slab = 0;
    auto upd = [&](int M, int Nn, int K) { slab = size_max(slab, gemm_scratch_bytes(2, M, Nn, K)); };
    upd(3 * pl.d, pl.d, (int)N);
    upd(pl.d, pl.d, (int)N);
    upd(pl.dff, pl.d, (int)N);
    upd(pl.d, pl.dff, (int)N);
    for (int i = 0; i < pl.nseg; ++i)
        if (segs[i].proj_w) upd(pl.d, segs[i].d_in, B * segs[i].T);
    pl.slab_bytes = slab;
    pl.s_slab = take(sc, slab);
    pl.det_bytes = cfg->deterministic ? generic_det_scratch_bytes(B, pl.d, pl.dff) : 0;
    pl.s_det = take(sc, pl.det_bytes);
    pl.scratch_bytes = sc;
    return 0;
}

// ---- fused per-clip path (fused.hip) ---------------------------------------------------------------
static bool packed_feats(const egx_segment* segs, int nseg) {
    for (int i = 0; i < nseg; ++i)
        if (segs[i].feat_bf16 || segs[i].pool > 1) return true;
    return false;
}
static bool fused_ok(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    if (pl.nseg > FUSED_MAX_SEG || pl.L > FUSED_MAX_LAYERS || pl.L < 1) return false;
    if (packed_feats(segs, pl.nseg)) return false;
    int d_in[EGX_MAX_SEGMENTS], T[EGX_MAX_SEGMENTS];
    bool hp[EGX_MAX_SEGMENTS];
    for (int i = 0; i < pl.nseg; ++i) { d_in[i] = segs[i].d_in; T[i] = segs[i].T; hp[i] = segs[i].proj_w != nullptr; }
    return fused_supported(pl.d, pl.H, pl.dff, pl.S, pl.nseg, d_in, T, hp);
}
static size_t fused_mask_words(const Plan& pl) { return (size_t)pl.L * pl.vB * (pl.dff / 32) * 64; }
static size_t fused_res_bytes(const Plan& pl) { return align_up((size_t)(1 + 2 * pl.L) * pl.N * pl.d * 4, 256); }
// saved = [pre + 2L residual blocks][ReLU sign bits of the FFN hidden: one u32 per (layer, clip, hidden block, lane)][packs]
static size_t fused_act_bytes(const Plan& pl) { return fused_res_bytes(pl) + align_up(fused_mask_words(pl) * 4, 256); }
// Fragment-packed weight copies kept behind the saved activations (written by the forward, reused by the
// backward): per segment the projection, per layer each matrix in both orientations.
struct FusedPackLayout {
    void* proj[EGX_MAX_SEGMENTS];
    struct { void* in_w; void* in_wt; void* out_w; void* out_wt; void* lin1_w; void* lin1_wt; void* lin2_w; void* lin2_wt; } layer[FUSED_MAX_LAYERS];
    size_t bytes;
};
static FusedPackLayout fused_pack_layout(const egx_config* cfg, const egx_segment* segs, const Plan& pl, char* base) {
    FusedPackLayout L;
    memset(&L, 0, sizeof(L));
    int bf = cfg->compute;      // packed element format follows the compute mode (fused_dev.h)
    size_t cur = 0;
    auto take_p = [&](int R, int K) -> void* { void* q = base ? base + cur : nullptr; cur += align_up(packed_bytes(R, K, bf), 256); return q; };
    for (int i = 0; i < pl.nseg; ++i) L.proj[i] = take_p(pl.d, segs[i].d_in);
    for (int l = 0; l < pl.L && l < FUSED_MAX_LAYERS; ++l) {
        L.layer[l].in_w = take_p(3 * pl.d, pl.d);
        L.layer[l].in_wt = take_p(pl.d, 3 * pl.d);
        L.layer[l].out_w = take_p(pl.d, pl.d);
        L.layer[l].out_wt = take_p(pl.d, pl.d);
        L.layer[l].lin1_w = take_p(pl.dff, pl.d);
        L.layer[l].lin1_wt = take_p(pl.d, pl.dff);
        L.layer[l].lin2_w = take_p(pl.d, pl.dff);
        L.layer[l].lin2_wt = take_p(pl.dff, pl.d);
    }
    L.bytes = cur;
    return L;
}
// where the packed copies live: the caller's persistent weight cache (egx_config.weight_cache, ABI v15) or behind the saved activations
static char* fused_pack_base(const egx_config* cfg, const void* saved, const Plan& vp) {
    return cfg->weight_cache ? (char*)cfg->weight_cache : (char*)const_cast<void*>(saved) + fused_act_bytes(vp);
}
// The FFN hidden activation H (forward) and its gradient dH (backward) are handed to the weight-gradient kernel as
// operand tiles instead of being recomputed there (ffn_dw_kernel, the recompute variant, is kept as the egx_ffn_dw unit hook).
static bool store_hidden() { return true; }     // (the recompute variant of the clip kernels was dropped in round 3: their FFN loops store unconditionally)
static size_t fused_hid_total(const egx_config* cfg, const Plan& pl) {
    return store_hidden() ? align_up((size_t)pl.L * fused_hid_bytes(pl.vB, pl.dff, cfg->compute == EGX_BF16), 256) : 0;
}
// saved = [activations][packed weights][H tiles]
static size_t fused_hid_offset(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    return align_up(fused_act_bytes(pl) + fused_pack_layout(cfg, segs, pl, nullptr).bytes, 256);
}
// split / bf16 mode: the FFN input x1 of every layer as bf16 planes (L, 3 or 1, B * 48, d), behind the hidden tiles
static bool split_planes(const egx_config* cfg) {
    return (cfg->compute == EGX_F32_SPLIT || (cfg->compute == EGX_BF16 && ffn_dw_bf16_planes())) && store_hidden();
}
static size_t plane_elem_bytes(const egx_config* cfg) { return cfg->compute == EGX_BF16 ? 2 : 6; }   // one bf16 plane, or the three parts
static size_t fused_x1p_offset(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    return fused_hid_offset(cfg, segs, pl) + fused_hid_total(cfg, pl);
}
// behind them: the input of every layer (L, N, d) and its Q | K | V rows (L, B, 48, 3d), fp32 (the backward loads instead of recomputing)
static size_t fused_xin_offset(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    return fused_x1p_offset(cfg, segs, pl) + (split_planes(cfg) ? align_up((size_t)pl.L * pl.vB * FUSED_TOK_PAD * pl.d * plane_elem_bytes(cfg), 256) : 0);
}
static size_t fused_qkv_offset(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    return fused_xin_offset(cfg, segs, pl) + align_up((size_t)pl.L * pl.N * pl.d * 4, 256);
}
// behind them: x1 = LayerNorm1 output of every layer as fp32 rows (L, N, d): what the cut mode's attention-side launch hands to ffn_fwd_kernel
static size_t fused_x1f_offset(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    return fused_qkv_offset(cfg, segs, pl) + align_up((size_t)pl.L * pl.vB * FUSED_TOK_PAD * 3 * pl.d * 4, 256);
}
static size_t fused_core_bytes(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    return fused_x1f_offset(cfg, segs, pl) + (pl.tpc > 1 ? (size_t)0 : align_up((size_t)pl.L * pl.N * pl.d * 4, 256));
}
// Cut mode (ffn_cut.hip): the per-clip kernels cut at the FFN, whose loops run as launches of their own with eight waves per clip. One
// workgroup per clip only (the sliced mode of small batches keeps the one-launch kernels). Policy = where it measured faster on MI355X
// (profiles/r05_cut_ab.txt): the f32s arithmetic (six MFMAs per K-block: the hidden loops are 55 % of the one-launch kernels and gain
// 25 % from the second wave per SIMD) with one layer per launch pair and the reference's d_ff; bf16's short loops and deeper / narrower
// stacks pay more for the two extra launches per layer than the loops gain. EGX_FFN_CUT=1 forces it wherever it is supported (the
// parity tests run every mode through it), =0 keeps the one-launch kernels.
// Kernel-selection switches of the per-clip path (development / test aids): EGX_FFN_CUT = 0 | 1, EGX_FFN_SLICES = 1 | 2 | 4 | 8,
// EGX_SLICE_DROP = <hex mask>. The product library reads the environment ONCE, at first use (round 6; rounds 4-5 called getenv in every
// forward and backward); egx_tuning_reload() re-reads it — the parity tests, which compare the modes inside one process, call it after
// changing os.environ (egot2_amd/functional.py reload_tuning_each_call). No workspace layout depends on any of them.
struct Tuning { int ffn_cut = -1; int slices_cap = 0; bool has_slices = false; long slice_drop = 0; bool has_drop = false; };
static Tuning g_tuning;
static bool g_tuning_loaded = false;
static void tuning_load() {
    Tuning t;
    if (const char* e = getenv("EGX_FFN_CUT")) t.ffn_cut = e[0] != '0' ? 1 : 0;
    if (const char* e = getenv("EGX_FFN_SLICES")) { t.has_slices = true; t.slices_cap = atoi(e); }
    if (const char* e = getenv("EGX_SLICE_DROP")) { t.has_drop = true; t.slice_drop = strtol(e, nullptr, 16); }
    g_tuning = t;
    g_tuning_loaded = true;
}
static const Tuning& tuning() { if (!g_tuning_loaded) tuning_load(); return g_tuning; }
static bool use_cut(const Plan& pl, int n_slices, bool tiled, int compute) {
    if (tiled || n_slices != 1 || !ffn_cut_supported(pl.dff)) return false;
    if (tuning().ffn_cut >= 0) return tuning().ffn_cut != 0;
    return compute == EGX_F32_SPLIT && pl.L == 1 && pl.dff >= 1024;
}
static int fused_slices(const Plan& pl, int compute);
static int fused_slices_layout(const Plan& pl);
static size_t sliced_xchg_bytes(const Plan& pl, int n);
static size_t sliced_flag_bytes(const Plan& pl, int n);
// The exchange buffer and the flag words are laid out for the LARGEST slice count the shape admits on any device and under any
// EGX_FFN_SLICES (fused_slices_layout: a function of the shape alone), so that the workspace query, the forward and the backward —
// which each pick their own run-time count <= that bound (fused_slices) — agree on every offset whatever the environment or the
// current device did between the calls (a changed count used to move the flag words past the caller's buffer).
static size_t fused_saved_bytes(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    const int n = (pl.tpc > 1 || pl.S > FUSED_TOK_PAD) ? 1 : fused_slices_layout(pl);
    return fused_core_bytes(cfg, segs, pl) + sliced_xchg_bytes(pl, n) + sliced_flag_bytes(pl, n);
}
// ---- sliced mode (small batches): n workgroups per clip, each 1 / n of the FFN hidden blocks (FusedFwdParams::n_slices). Only when
// the workgroups of the launch can all be resident at once: round_up(B, 8) * n <= CUs (a performance rule, not a protocol: a slice
// whose partial sum does not arrive within 100 us is computed by the waiting workgroup itself, fused_dev.h).
// EGX_FFN_SLICES=1 turns it off, =2 / 4 / 8 caps n.
static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = v;
    }
    return cus[dev];
}
// EGX_SLICE_DROP=<hex mask> (testing aid): the workgroups of those slices leave at once; the others must compute their share
static int slice_drop_mask(int n) {
    if (!tuning().has_drop) return 0;
    const int m = (int)tuning().slice_drop & ((1 << n) - 1);
    return m == (1 << n) - 1 ? 0 : m;      // at least one slice has to run
}
// The exchange is built on gfx942 / gfx950 behaviour (write-through relaxed agent-scope atomics, s_waitcnt ordering; fused_dev.h): any other
// device runs one workgroup per clip.
static bool device_slicing_ok() {
    static int ok[64] = {0};        // 0 unknown, 1 yes, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!ok[dev]) {
        hipDeviceProp_t pr;
        ok[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && (strncmp(pr.gcnArchName, "gfx950", 6) == 0 || strncmp(pr.gcnArchName, "gfx942", 6) == 0)) ? 1 : 2;
    }
    return ok[dev] == 1;
}
constexpr int SLICE_LAYOUT_CUS = 304;       // the largest compute-unit count of the supported parts (gfx942: 304, gfx950: 256) bounds the slice count the layout
                                            // provides for (512 reserved a two-slice exchange buffer, 12.6 MB per layer, for the B = 256 headline batch)
static int fused_slices_layout(const Plan& pl) {
    const int nit = pl.dff / 128, bq = (pl.B + 7) / 8 * 8;
    int n = 1;
    while (n * 2 <= 8 && nit % (n * 2) == 0 && bq * n * 2 <= SLICE_LAYOUT_CUS) n *= 2;
    return n;
}
static int fused_slices(const Plan& pl, int compute) {
    if (!device_slicing_ok()) return 1;
    const bool e = tuning().has_slices;            // (EGX_FFN_SLICES; the LAYOUT does not depend on it)
    // measured at B = 32 (profiles/r04_sliced.txt): the exchange costs ~10 us per kernel; with bf16's short FFN loop eight slices lose to four
    int cap = e ? tuning().slices_cap : (compute == EGX_BF16 ? 4 : 8);
    if (cap < 1) cap = 1;
    const int cus = device_cus(), nit = pl.dff / 128, bq = (pl.B + 7) / 8 * 8;
    int n = 1;
    while (n * 2 <= cap && n * 2 <= 8 && nit % (n * 2) == 0 && bq * n * 2 <= cus) n *= 2;
    if (compute == EGX_BF16 && n == 2 && !e) n = 1;     // two slices of the short bf16 loop do not pay for the exchange (B = 128: +3 % / -3 %)
    const int cap_layout = fused_slices_layout(pl);
    return n < cap_layout ? n : cap_layout;
}
// behind the fused layout: the forward's exchange buffer (L, B, n, 48, d) and the "published" words of forward and backward (2, L, B, 8)
static size_t sliced_xchg_bytes(const Plan& pl, int n) { return n > 1 ? align_up((size_t)pl.L * pl.B * n * FUSED_TOK_PAD * pl.d * 4, 256) : 0; }
static size_t sliced_flag_words(const Plan& pl) { return (size_t)pl.L * pl.B * 8; }       // one "published" word per (layer, clip, slice): SLICE_MAX = 8
static size_t sliced_flag_bytes(const Plan& pl, int n) { return n > 1 ? align_up(2 * sliced_flag_words(pl) * 4, 256) : 0; }

// ---- tiled mode (d = 128, 48 < S <= 512): the same kernels over 48-token tiles, attention between the launches. Behind the fused
// layout of the tile grid: every layer's attention output (L, N, d) and log-sum-exp (L, B, H, S), then room for the output tokens
// and the pooled vector of a translator call
static void plan_tiled(Plan& pl) { pl.tpc = cdiv(pl.S, FUSED_TOK_PAD); pl.vB = pl.B * pl.tpc; }
static size_t tiled_attn_offset(const egx_config* cfg, const egx_segment* segs, const Plan& vp) { return align_up(fused_saved_bytes(cfg, segs, vp), 256); }
static size_t tiled_lse_offset(const egx_config* cfg, const egx_segment* segs, const Plan& vp) { return tiled_attn_offset(cfg, segs, vp) + align_up((size_t)vp.L * vp.N * vp.d * 4, 256); }
static size_t tiled_tokens_offset(const egx_config* cfg, const egx_segment* segs, const Plan& vp) { return tiled_lse_offset(cfg, segs, vp) + align_up((size_t)vp.L * vp.B * vp.H * vp.S * 4, 256); }
static size_t tiled_saved_bytes(const egx_config* cfg, const egx_segment* segs, const Plan& vp) {
    return tiled_tokens_offset(cfg, segs, vp) + align_up((vp.N + (size_t)vp.B) * vp.d * 4, 256);
}
static bool tiled_ok(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    if (pl.d != 128 || pl.H != 4 || pl.dff % 128 != 0 || pl.dff < 128) return false;
    if (pl.S <= FUSED_TOK_PAD || pl.S > TILED_MAX_S) return false;
    if (pl.nseg > FUSED_MAX_SEG || pl.L > FUSED_MAX_LAYERS || pl.L < 1) return false;
    if (packed_feats(segs, pl.nseg) || cfg->p_feat > 0.f) return false;
    if (cfg->compute != EGX_BF16 && cfg->compute != EGX_F32_SPLIT) return false;     // exact-fp32 MFMA: the generic kernels
    if (cfg->compute == EGX_BF16 && !ffn_dw_bf16_planes()) return false;
    if (cfg->out_tokens != 0 && cfg->out_tokens != pl.S) return false;
    for (int i = 0; i < pl.nseg; ++i)
        if (!segs[i].proj_w || segs[i].d_in % 128 != 0) return false;
    return true;
}
static bool use_tiled(const egx_config* cfg, const egx_segment* segs, const Plan& pl, bool* err) {
    *err = false;
    const bool ok = tiled_ok(cfg, segs, pl);
    if (cfg->impl == EGX_IMPL_TILED) {
        if (!ok) { set_error("tiled implementation does not support this configuration (needs d=128, h=4, 48 < S <= %d, compute bf16 or f32s, d_ff%%128==0, projected segments, <=4 layers)", TILED_MAX_S); *err = true; }
        return ok;
    }
    return cfg->impl == EGX_IMPL_AUTO && ok;
}

// scratch of the fused backward: per layer the operands of the weight-gradient kernels, then d(seg), the per-clip
// partial sums, and the slab area shared by ffn_dw and the split-K GEMMs.
struct FusedBwdScratch {
    size_t x1[FUSED_MAX_LAYERS], g2[FUSED_MAX_LAYERS], attn_o[FUSED_MAX_LAYERS], g1[FUSED_MAX_LAYERS], dqkv[FUSED_MAX_LAYERS];
    size_t dseg[EGX_MAX_SEGMENTS];
    size_t partials, slabs, slab_bytes, dhid, bytes, dx0;
    size_t sdw_tiles, sdw_bytes;            // round 6: the tiles small_dw leaves for the fixed-order tail launch (NOT the shared slab area: layer 0's FFN slabs live there)
    size_t dy1, dxin;                       // cut mode: what the FFN-side and the attention-side launches of the backward hand each other
    size_t xchg;                            // sliced mode only
    size_t datt, dres, delta, dtok;         // tiled mode only
    size_t ffn_slab[FUSED_MAX_LAYERS];      // slab area of each layer's FFN weight gradient (layer 0: `slabs`): one reduction launch sums them all
    int P;
};
static FusedBwdScratch fused_bwd_scratch(const egx_config* cfg, const egx_segment* segs, const Plan& pl, int head_n_out = 0) {
    FusedBwdScratch s;
    memset(&s, 0, sizeof(s));
    size_t cur = 0;
    size_t nd = pl.N * pl.d * 4;
    size_t nd3 = (size_t)pl.vB * FUSED_TOK_PAD * pl.d * 6;        // g2 leaves as three bf16 planes on the 48-row clip grid in split mode
    for (int l = 0; l < pl.L && l < FUSED_MAX_LAYERS; ++l) {
        s.x1[l] = take(cur, nd); s.g2[l] = take(cur, nd3); s.attn_o[l] = take(cur, nd);
        s.g1[l] = take(cur, nd); s.dqkv[l] = take(cur, 3 * nd);
    }
    for (int i = 0; i < pl.nseg; ++i) s.dseg[i] = take(cur, (size_t)pl.B * segs[i].T * pl.d * 4);
    s.P = fused_partial_len(pl.L, pl.nseg) + fused_head_partial_len(head_n_out);
    s.partials = take(cur, (size_t)pl.vB * s.P * 4);
    size_t slab = ffn_dw_scratch_bytes((int)pl.N, pl.dff, nullptr);
    slab = size_max(slab, gemm_scratch_bytes(2, 3 * pl.d, pl.d, (int)pl.N));
    slab = size_max(slab, gemm_scratch_bytes(2, pl.d, pl.d, (int)pl.N));
    for (int i = 0; i < pl.nseg; ++i) slab = size_max(slab, gemm_scratch_bytes(2, pl.d, segs[i].d_in, pl.B * segs[i].T));
    slab = size_max(slab, (size_t)(512 + SMALL_DW_MAX * 16) * 64 * 128 * sizeof(float));   // deterministic small_dw: one tile per workgroup
    s.slab_bytes = slab;
    s.slabs = take(cur, slab);
    s.ffn_slab[0] = s.slabs;
    for (int l = 1; l < pl.L && l < FUSED_MAX_LAYERS; ++l) s.ffn_slab[l] = take(cur, ffn_dw_scratch_bytes((int)pl.N, pl.dff, nullptr));
    s.sdw_bytes = (size_t)(512 + SMALL_DW_MAX * 16) * 64 * 128 * sizeof(float);
    s.sdw_tiles = take(cur, s.sdw_bytes);
    s.dhid = take(cur, fused_hid_total(cfg, pl));
    s.dx0 = take(cur, nd);         // d(token-prep output) behind its dropout mask (learned positional table gradient)
    s.dy1 = take(cur, nd); s.dxin = take(cur, nd);
    if (pl.tpc > 1 || pl.S > FUSED_TOK_PAD) {       // tiled mode: d(attention output), the residual gradient, delta, d(tokens) of a translator call
        s.datt = take(cur, nd); s.dres = take(cur, nd);
        s.delta = take(cur, (size_t)pl.B * pl.H * pl.S * 4);
        s.dtok = take(cur, nd);
    } else {
        s.xchg = take(cur, sliced_xchg_bytes(pl, fused_slices_layout(pl)));
    }
    s.bytes = cur;
    return s;
}

// EGX_REDUCE_RIDES=0: the slab / partial-row reductions keep their own launch (tuning aid)
static bool reduce_rides() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("EGX_REDUCE_RIDES"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
static bool use_fused(const egx_config* cfg, const egx_segment* segs, const Plan& pl, bool* err) {
    *err = false;
    bool ok = fused_ok(cfg, segs, pl);
    if (cfg->impl == EGX_IMPL_FUSED) {
        if (!ok) { set_error("fused implementation does not support this configuration (needs d=128, h=4 or 8, S<=48, d_ff%%128==0, projected segments, <=%d layers)", FUSED_MAX_LAYERS); *err = true; }
        return ok;
    }
    return cfg->impl == EGX_IMPL_AUTO && ok;   // auto: fused per-clip kernels whenever the shape allows
}

// wide bf16 path (wide_host.hip): bf16 compute outside the fused kernels' shape, whenever its alignment rules hold
static bool use_wide(const egx_config* cfg, const egx_segment* segs, const Plan& pl, bool* err) {
    *err = false;
    const bool ok = wide_ok(cfg, segs, pl.B);
    if (cfg->impl == EGX_IMPL_WIDE) {
        if (!ok) { set_error("wide implementation does not support this configuration (needs compute = bf16, d_model >= 256, d_model / d_ff / projected d_in multiples of 128, S <= 128 with head dim 32 / 64 / 96 / 128 or S <= 480 with head dim 32 / 64)"); *err = true; }
        return ok;
    }
    return cfg->impl == EGX_IMPL_AUTO && ok && !fused_ok(cfg, segs, pl);
}

static inline float* fptr(void* base, size_t off) { return (float*)((char*)base + off); }
static inline const float* cfptr(const void* base, size_t off) { return (const float*)((const char*)base + off); }

struct Drop {
    uint64_t key = 0;
    uint32_t thresh = 0;
    float inv_keep = 1.f;
};
static Drop make_drop(int training, float p, uint64_t seed, uint32_t layer, uint32_t site) {
    Drop dr;
    if (training && p > 0.f) {
        dr.key = site_key(seed, layer, site);
        dr.thresh = drop_threshold(p);
        dr.inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
    }
    return dr;
}

static int linear_nt(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, int relu,
                     const Drop& dr, const float* residual, int compute, hipStream_t st) {
    GemmParams g;
    g.A = x; g.B = W; g.C = y;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.bias = bias;
    g.relu = relu;
    g.drop_key = dr.key; g.drop_thresh = dr.thresh; g.drop_inv_keep = dr.inv_keep;
    g.residual = residual; g.ldr = N;
    return gemm(0, g, compute, 0, nullptr, 0, st);
}

// dx[M,K] = dy[M,N] W[N,K]  (+ mask/scale, + residual)
static int linear_dx(const float* dy, const float* W, float* dx, int M, int N, int K, const float* mask, float mask_scale,
                     const float* residual, int compute, hipStream_t st, void* slab = nullptr, size_t slab_bytes = 0) {
    GemmParams g;
    g.A = dy; g.B = W; g.C = dx;
    g.M = M; g.N = K; g.K = N;
    g.lda = N; g.ldb = K; g.ldc = K;
    g.mask = mask; g.ldm = K; g.mask_scale = mask_scale;
    g.residual = residual; g.ldr = K;
    return gemm(1, g, compute, 0, slab, slab_bytes, st);    // slab scratch (optional): lets a skinny dx split its reduction
}

// dW[N,K] += dy[M,N]^T x[M,K]
static int linear_dw(const float* dy, const float* x, float* dW, int M, int N, int K, int compute, void* slab,
                     size_t slab_bytes, hipStream_t st) {
    GemmParams g;
    g.A = dy; g.B = x; g.C = dW;
    g.M = N; g.N = K; g.K = M;
    g.lda = N; g.ldb = K; g.ldc = K;
    return gemm(2, g, compute, 1, slab, slab_bytes, st);
}

int debug_read_ppstamps(unsigned long long* out);       // wide_gemm.hip (development aid)
int debug_read_cstamps(unsigned long long* out);        // ffn_cut.hip (development aid)
int debug_read_sstamps(unsigned long long* out);        // small_dw_kernel (development aid)

}  // namespace egx

using namespace egx;

extern "C" {

int egx_abi_version(void) { return EGX_ABI_VERSION; }
void egx_tuning_reload(void) { tuning_load(); }
long long egx_launch_count(int reset) { long long n = g_launches; if (reset) g_launches = 0; return n; }
int egx_debug_stamps(unsigned long long* out, int n) { return n == -1000 ? debug_read_ppstamps(out) : n == -3000 ? debug_read_cstamps(out) : n == -4000 ? debug_read_sstamps(out) : (n < 0 ? debug_read_bstamps(out, -n) : debug_read_stamps(out, n)); }
long long egx_slices_stolen(int reset) {
    const long long a = slices_stolen_fwd(reset), b = slices_stolen_bwd(reset);
    return a < 0 || b < 0 ? -1 : a + b;
}
int egx_seed_advance(uint64_t* seed, void* stream) { EGX_CHECK(seed, "null seed"); return seed_advance(seed, (hipStream_t)stream); }
void egx_timing_enable(int on) { timing_enable(on); }
int egx_timing_read(int which, double* total_ms, int* count) { return timing_read(which, total_ms, count); }

// Unit-test hook for the fused FFN weight-gradient kernel. scratch: packed W1 + packed W2^T + slabs.
size_t egx_ffn_dw_scratch(int N, int d_ff, int compute) {
    int bf = compute;
    return 2 * align_up(packed_bytes(d_ff, 128, bf), 256) + ffn_dw_scratch_bytes(N, d_ff, nullptr);
}
int egx_ffn_dw(const float* x1, const float* g, const float* W1, const float* b1, const float* W2, int N, int S, int d_ff,
               float p_drop, uint64_t seed, float* dW1, float* db1, float* dW2, int compute, void* scratch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int bf = compute;
    PackParams pk;
    memset(&pk, 0, sizeof(pk));
    pk.mode = bf;
    char* cur = (char*)scratch;
    pk.d[0].src = W1; pk.d[0].dst = cur; pk.d[0].R = d_ff; pk.d[0].K = 128; pk.d[0].ld = 128; pk.d[0].transpose = 0;
    cur += align_up(packed_bytes(d_ff, 128, bf), 256);
    pk.d[1].src = W2; pk.d[1].dst = cur; pk.d[1].R = d_ff; pk.d[1].K = 128; pk.d[1].ld = d_ff; pk.d[1].transpose = 1;
    cur += align_up(packed_bytes(d_ff, 128, bf), 256);
    pk.n = 2;
    if (pack_weights(pk, st)) return 1;
    FfnDwParams fp;
    memset(&fp, 0, sizeof(fp));
    fp.x1 = x1; fp.g = g; fp.w1p = pk.d[0].dst; fp.w2tp = pk.d[1].dst; fp.b1 = b1;
    fp.N = N; fp.S = S; fp.d_ff = d_ff;
    Drop dh = make_drop(p_drop > 0.f, p_drop, seed, 0, SITE_FFN);
    fp.drop_key = dh.key; fp.drop_thresh = dh.thresh; fp.drop_inv = dh.inv_keep;
    return ffn_dw(fp, compute, dW1, db1, dW2, cur, st);
}
const char* egx_last_error(void) { return g_err; }

// ---- unit hooks of the wide bf16 path ----------------------------------------------------------------------------
size_t egx_wide_gemm_scratch(int layout, int M, int N, int K) { return 1024 + (layout == 2 ? wide_gemm_tn_scratch(M, N, K) : 0); }
int egx_wide_gemm(int layout, const void* A, const void* B, float* Cf, void* Cb, int M, int N, int K, const float* bias,
                  int relu, const float* residual, void* scratch, void* stream) {
    EGX_CHECK(layout == 0 || layout == 2, "egx_wide_gemm: layout %d (0 = NT, 2 = TN)", layout);
    EGX_CHECK(scratch, "egx_wide_gemm: null scratch");
    hipStream_t st = (hipStream_t)stream;
    EGX_HIP(hipMemsetAsync(scratch, 0, 1024, st));
    WideGemmParams g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.M = M; g.N = N; g.K = K;
    g.Cf = Cf; g.Cb = (bf16_t*)Cb; g.ldc = N; g.bias = bias; g.relu = relu; g.residual = residual; g.ldr = N;
    g.zero_page = scratch;
    if (layout == 0) { g.lda = K; g.ldb = K; return wide_gemm_nt(g, st); }
    g.lda = M; g.ldb = N;
    return wide_gemm_tn(g, (char*)scratch + 1024, st);
}
static int wide_attn_hook(const void* qkv, void* out, float* lse, const void* d_out, void* d_qkv, float* delta, int B, int S, int H, int d,
                          float p_drop, uint64_t seed, void* stream, bool bwd) {
    WideAttnParams a;
    a.qkv = (const bf16_t*)qkv; a.out = (bf16_t*)out; a.lse = lse; a.d_out = (const bf16_t*)d_out; a.d_qkv = (bf16_t*)d_qkv; a.delta = delta;
    a.B = B; a.S = S; a.H = H; a.d = d;
    if (p_drop > 0.f) { a.drop_key = site_key(seed, 0, SITE_ATTN); a.drop_thresh = drop_threshold(p_drop); a.drop_inv = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f; }
    return bwd ? wide_attn_bwd(a, (hipStream_t)stream) : wide_attn_fwd(a, (hipStream_t)stream);
}
int egx_wide_attention_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int d, float p_drop, uint64_t seed, void* stream) {
    return wide_attn_hook(qkv, out, lse, nullptr, nullptr, nullptr, B, S, H, d, p_drop, seed, stream, false);
}
int egx_wide_attention_bwd(const void* qkv, const void* out, const float* lse, const void* d_out, void* d_qkv, float* delta, int B, int S,
                           int H, int d, float p_drop, uint64_t seed, void* stream) {
    return wide_attn_hook(qkv, const_cast<void*>(out), const_cast<float*>(lse), d_out, d_qkv, delta, B, S, H, d, p_drop, seed, stream, true);
}

int egx_encoder_workspace(const egx_config* cfg, const egx_segment* segs, int B, size_t* saved_bytes, size_t* scratch_bytes) {
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return 1;
    size_t wsv = 0, wsc = 0;
    if (wide_ok(cfg, segs, B)) wide_workspace(cfg, segs, B, &wsv, &wsc);
    size_t tsv = 0, tsc = 0;
    if (tiled_ok(cfg, segs, pl)) {
        Plan vp = pl;
        plan_tiled(vp);
        tsv = tiled_saved_bytes(cfg, segs, vp);
        tsc = fused_bwd_scratch(cfg, segs, vp, FUSED_HEAD_MAX_OUT).bytes;       // (the pooled head's partial-row section: sized for the largest head)
    }
    if (saved_bytes) *saved_bytes = size_max(size_max(size_max(pl.saved_bytes, wsv), tsv), fused_ok(cfg, segs, pl) ? fused_saved_bytes(cfg, segs, pl) : (size_t)0);
    if (scratch_bytes) *scratch_bytes = size_max(size_max(size_max(pl.scratch_bytes, wsc), tsc), fused_ok(cfg, segs, pl) ? fused_bwd_scratch(cfg, segs, pl, FUSED_HEAD_MAX_OUT).bytes : (size_t)0);
    return 0;
}

// egx_config.token_ce is evaluated by the per-clip kernels of the one-launch mode only (one workgroup per clip, no cut, no slices, no pooled head),
// with the arrival counter of the weight cache's control block
static bool token_ce_ok(const egx_config* cfg, const egx_segment* segs, const Plan& pl) {
    bool ferr;
    if (!use_fused(cfg, segs, pl, &ferr)) return false;
    if (!cfg->weight_cache || cfg->deterministic) return false;
    if (pl.B > 4095) return false;      // (the arrival word: 12 bits of clips, 20 bits of correct frames)
    const int ns = fused_slices(pl, cfg->compute);
    return ns == 1 && !use_cut(pl, ns, false, cfg->compute);
}

// Bytes of the persistent packed-weight cache (egx_config.weight_cache): the fragment-packed copies of the per-clip / tiled kernels. The layout
// is a function of the model dimensions and the compute mode only (fused_pack_layout), so one buffer serves every batch size.
size_t egx_weight_cache_bytes(const egx_config* cfg, const egx_segment* segs) {
    if (!cfg || !segs) return 0;
    Plan pl;
    if (make_plan(cfg, segs, 1, pl)) return 0;
    if (pl.d != 128 || pl.nseg > FUSED_MAX_SEG || pl.L > FUSED_MAX_LAYERS || pl.L < 1 || pl.dff % 128 != 0) return 0;
    for (int i = 0; i < pl.nseg; ++i)
        if (!segs[i].proj_w || segs[i].d_in % 128 != 0) return 0;
    return align_up(fused_pack_layout(cfg, segs, pl, nullptr).bytes, 256) + 256;      // + the control block (FusedFwdParams::ce_ticket / tce_ticket)
}

int egx_encoder_token_ce_ok(const egx_config* cfg, const egx_segment* segs, int B) {
    Plan pl;
    if (!cfg || !segs || make_plan(cfg, segs, B, pl)) return 0;
    return token_ce_ok(cfg, segs, pl) ? 1 : 0;
}

int egx_encoder_uses_fused(const egx_config* cfg, const egx_segment* segs, int B) {
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return 0;
    bool ferr;
    return use_fused(cfg, segs, pl, &ferr) ? 1 : 0;
}

int egx_encoder_slices(const egx_config* cfg, const egx_segment* segs, int B) {
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return -1;
    bool ferr;
    return use_fused(cfg, segs, pl, &ferr) ? fused_slices(pl, cfg->compute) : 1;
}

int egx_encoder_impl(const egx_config* cfg, const egx_segment* segs, int B) {
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return -1;
    bool ferr;
    if (use_fused(cfg, segs, pl, &ferr)) return EGX_IMPL_FUSED;
    if (ferr) return -1;
    if (use_tiled(cfg, segs, pl, &ferr)) return EGX_IMPL_TILED;
    if (ferr) return -1;
    if (use_wide(cfg, segs, pl, &ferr)) return EGX_IMPL_WIDE;
    if (ferr) return -1;
    return EGX_IMPL_GENERIC;
}

}  // extern "C"

// Shared body of egx_encoder_fwd (head == null) and egx_translator_fwd (pooled head fused or appended).
static int encoder_fwd_impl(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                            const egx_layer* layers, const egx_head* head, int B, float* tokens_out, float* logits_out,
                            void* saved, void* scratch, int training, uint64_t seed, void* stream) {
    (void)scratch;
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return 1;
    EGX_CHECK(saved && ln_w && ln_b, "null pointer argument");
    EGX_CHECK(pl.L == 0 || layers, "null layers");
    hipStream_t st = (hipStream_t)stream;
    const int d = pl.d, S = pl.S, comp = cfg->compute;
    const int N = (int)pl.N;
    const bool with_head = head && head->W;
    EGX_CHECK(with_head ? (logits_out != nullptr) : (tokens_out != nullptr), "null output pointer");
    const egx_ce* ce = cfg->ce;
    EGX_CHECK(!ce || with_head, "egx_config.ce: the fused cross entropy needs the pooled head (egx_translator_fwd)");
    EGX_CHECK(!ce || (ce->target && ce->loss && ce->d_logits), "egx_config.ce: target, loss and d_logits must be set");
    EGX_CHECK(!cfg->weight_cache_valid || cfg->weight_cache, "weight_cache_valid without a weight_cache");
    const egx_token_ce* tce = cfg->token_ce;
    if (tce) {
        EGX_CHECK(!with_head && token_ce_ok(cfg, segs, pl), "egx_config.token_ce: not on this configuration (egx_encoder_token_ce_ok)");
        EGX_CHECK(tce->W && tce->target && tce->logits && tce->loss && tce->d_logits && tce->C >= 1 && tce->C <= 8,
                  "egx_config.token_ce: W, target, logits, loss, d_logits and 1 <= C <= 8 must be set");
    }
    EGX_CHECK(!with_head || (head->ln_w && head->ln_b && head->b && head->n_out >= 1 && head->n_out <= FUSED_HEAD_MAX_OUT),
              "head needs ln_w, ln_b, W, b and 1 <= n_out <= %d", FUSED_HEAD_MAX_OUT);
    // A HOST seed is baked into a captured graph: every replay would draw the SAME dropout masks — training that runs, converges worse and
    // says nothing. Refused (as the decoder does, wide_decoder.hip refuse_captured_dropout); with egx_config.seed_ptr the seed lives in device
    // memory and is advanced on the stream, so replays draw fresh masks (model.enable_device_seed(), train.GraphedStep).
    if (training && !cfg->seed_ptr && (cfg->p_drop > 0.f || cfg->p_pos > 0.f || cfg->p_feat > 0.f)) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        EGX_CHECK(cs == hipStreamCaptureStatusNone, "training-mode dropout (p > 0) with a host seed cannot be captured in a hipGraph: every replay would "
                  "repeat the same masks; pass egx_config.seed_ptr (model.enable_device_seed()), capture with p = 0, or launch eagerly");
    }
    bool ferr, terr = false;
    const bool fused = use_fused(cfg, segs, pl, &ferr);
    const bool tiled = !fused && !ferr && use_tiled(cfg, segs, pl, &terr);
    if (fused || tiled) {
        Plan vp = pl;               // sizes of the tile grid: vp.vB workgroups ("virtual clips"); vp.B / vp.N stay the real clips / tokens
        if (tiled) plan_tiled(vp);
        FusedFwdParams fp;
        memset(&fp, 0, sizeof(fp));
        // rewrite the weights into MFMA-fragment order (once per forward; they live behind the saved activations)
        PackParams pk;
        memset(&pk, 0, sizeof(pk));
        pk.mode = comp;
        pk.seed_advance = (cfg->advance_seed == 1 && cfg->seed_ptr && training) ? const_cast<uint64_t*>(cfg->seed_ptr) : nullptr;
        FusedPackLayout PL = fused_pack_layout(cfg, segs, vp, fused_pack_base(cfg, saved, vp));
        // weight cache valid (egx_config.weight_cache_valid): the packed copies of exactly these weights are in place, nothing is packed
        const bool cache_hit = cfg->weight_cache && cfg->weight_cache_valid;
        auto add_pack = [&](const float* src, void* dst, int R, int K, int ld, int transpose, float scale = 1.f) -> const void* {
            if (cache_hit) return dst;
            PackDesc& dsc = pk.d[pk.n++];
            dsc.src = src; dsc.dst = dst; dsc.R = R; dsc.K = K; dsc.ld = ld; dsc.transpose = transpose; dsc.scale = scale;
            return dst;
        };
        // the keep-scale of the FFN hidden dropout rides on the packed W1 (forward: relu(s (W1 x + b1)) = s relu(W1 x + b1)) and
        // W2^T (backward: dH = alive ? s W2^T g : 0): the FFN epilogues of the clip kernels have no multiply
        const Drop dffn = make_drop(training, cfg->p_drop, seed, 0, SITE_FFN);
        const float ffn_scale = dffn.thresh ? dffn.inv_keep : 1.f;
        for (int i = 0; i < pl.nseg; ++i) {
            FusedSeg& fs = fp.seg[i];
            fs.feat = segs[i].feat; fs.proj_wp = add_pack(segs[i].proj_w, PL.proj[i], d, segs[i].d_in, segs[i].d_in, 0); fs.proj_b = segs[i].proj_b;
            fs.add_vec = segs[i].add_vec; fs.pos = segs[i].pos;
            fs.T = segs[i].T; fs.d_in = segs[i].d_in; fs.off = pl.seg_off[i]; fs.pos_stride = segs[i].pos_stride;
            fs.row0 = 0; fs.Tfull = segs[i].T; fs.seg_id = i;
            Drop df = make_drop(training, cfg->p_feat, seed, (uint32_t)i, SITE_FEAT);
            fp.feat_key[i] = df.key; fp.feat_thresh = df.thresh; fp.feat_inv = df.inv_keep;
        }
        fp.n_heads = pl.H;
        for (int l = 0; l < pl.L; ++l) {
            FusedLayer& fl = fp.layer[l];
            const egx_layer& w = layers[l];
            fl.in_proj_wp = add_pack(w.in_proj_w, PL.layer[l].in_w, 3 * d, d, d, 0); fl.in_proj_b = w.in_proj_b;
            fl.out_proj_wp = add_pack(w.out_proj_w, PL.layer[l].out_w, d, d, d, 0); fl.out_proj_b = w.out_proj_b;
            fl.lin1_wp = add_pack(w.lin1_w, PL.layer[l].lin1_w, pl.dff, d, d, 0, ffn_scale); fl.lin1_b = w.lin1_b;
            fl.lin2_wp = add_pack(w.lin2_w, PL.layer[l].lin2_w, d, pl.dff, pl.dff, 0); fl.lin2_b = w.lin2_b;
            {   // transposed copies for the backward kernels
                add_pack(w.in_proj_w, PL.layer[l].in_wt, d, 3 * d, d, 1);
                add_pack(w.out_proj_w, PL.layer[l].out_wt, d, d, d, 1);
                add_pack(w.lin1_w, PL.layer[l].lin1_wt, d, pl.dff, d, 1);
                add_pack(w.lin2_w, PL.layer[l].lin2_wt, pl.dff, d, pl.dff, 1, ffn_scale);
            }
            fl.norm1_w = w.norm1_w; fl.norm1_b = w.norm1_b; fl.norm2_w = w.norm2_w; fl.norm2_b = w.norm2_b;
            Drop da = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_ATTN);
            fl.attn_key = da.key; fl.attn_thresh = da.thresh; fl.drop_inv = da.inv_keep;
            fl.res_thresh = da.thresh; fl.ffn_thresh = da.thresh;
            fl.res1_key = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES1).key;
            fl.ffn_key = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_FFN).key;
            fl.res2_key = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES2).key;
        }
        fp.ln_w = ln_w; fp.ln_b = ln_b; fp.eps = cfg->ln_eps;
        fp.nseg = pl.nseg; fp.n_layers = pl.L; fp.B = vp.vB; fp.S = tiled ? FUSED_TOK_PAD : S; fp.d_ff = pl.dff;
        fp.tpc = vp.tpc; fp.S_clip = S; fp.Ntok = pl.N; fp.mode = FUSED_MODE_FULL;
        fp.tokens_out = tokens_out;
        fp.out_T = cfg->out_tokens > 0 ? cfg->out_tokens : S;
        if (with_head && !tiled) {
            fp.head.ln_w = head->ln_w; fp.head.ln_b = head->ln_b; fp.head.W = head->W; fp.head.b = head->b; fp.head.n_out = head->n_out;
            fp.logits_out = logits_out;
        }
        fp.saved_pre = (float*)saved;
        fp.saved_res = (float*)saved + (size_t)N * d;
        fp.relu_bits = (uint32_t*)((char*)saved + fused_res_bytes(vp));
        fp.hid_out = store_hidden() ? (char*)saved + fused_hid_offset(cfg, segs, vp) : nullptr;
        fp.x1p_out = split_planes(cfg) ? (unsigned short*)((char*)saved + fused_x1p_offset(cfg, segs, vp)) : nullptr;
        fp.xin_out = (float*)((char*)saved + fused_xin_offset(cfg, segs, vp));
        fp.qkv_out = (float*)((char*)saved + fused_qkv_offset(cfg, segs, vp));
        Drop dpz = make_drop(training, cfg->p_pos, seed, 0, SITE_POS);
        fp.pos_key = dpz.key; fp.pos_thresh = dpz.thresh; fp.pos_inv = dpz.inv_keep;
        fp.seed_ptr = cfg->seed_ptr;
        fp.rot_mode = ffn_rot_mode();
        { static const int sh = [] { const char* e = getenv("EGX_CUT_SHIFT_F"); return e ? atoi(e) + 1 : 0; }(); fp.rot_mode |= sh << 8; }    // tuning aid: older : younger wave split of ffn_fwd_kernel
        fp.n_slices = tiled ? 1 : fused_slices(pl, comp);
        if (fp.n_slices > 1) {
            fp.xchg = (float*)((char*)saved + fused_core_bytes(cfg, segs, vp));
            fp.xflags = (unsigned*)((char*)saved + fused_core_bytes(cfg, segs, vp) + sliced_xchg_bytes(pl, fused_slices_layout(pl)));
            pk.zero_words = fp.xflags; pk.n_zero = (int)sliced_flag_words(pl);        // the packing launch (always in front) zeroes the flags
            fp.slice_drop = slice_drop_mask(fp.n_slices);
        }
        fp.x1f_out = tiled ? nullptr : (float*)((char*)saved + fused_x1f_offset(cfg, segs, vp));
        const bool cut = use_cut(pl, fp.n_slices, tiled, comp);
        // fused weighted cross entropy (egx_ce): in the epilogue of the launch that writes the logits; every clip adds its term into *loss,
        // which the packing launch zeroes when there is one, else the first launch of a cut-mode forward, else a memset node
        const bool ce_fused = ce && with_head && !tiled && !cfg->deterministic;
        if (ce_fused) {
            fp.ce_target = ce->target; fp.ce_weight = ce->class_weight; fp.ce_loss = ce->loss; fp.ce_dlogits = ce->d_logits; fp.ce_B = B;
            if (pk.n || pk.seed_advance || pk.zero_words) pk.zero_word2 = ce->loss;
            else if (cut) fp.zero_word = ce->loss;
            else if (cfg->weight_cache && EGX_CE_TICKET) fp.ce_ticket = (unsigned*)((char*)cfg->weight_cache + align_up(PL.bytes, 256));   // no earlier launch: arrival counter instead of a memset node
            else EGX_HIP(hipMemsetAsync(ce->loss, 0, sizeof(float), st));
        }
        if (tce) {
            fp.tce_W = tce->W; fp.tce_b = tce->b; fp.tce_target = tce->target; fp.tce_cw = tce->class_weight; fp.tce_C = tce->C;
            fp.tce_logits = tce->logits; fp.tce_probs = tce->probs; fp.tce_pred = tce->pred; fp.tce_loss = tce->loss; fp.tce_correct = tce->correct;
            fp.tce_dlogits = tce->d_logits;
            fp.tce_ticket = (unsigned*)((char*)cfg->weight_cache + align_up(PL.bytes, 256)) + 4;      // words 4..6 of the control block
        }
        if (cfg->weight_cache && pk.n) pk.zero_ctl = (unsigned*)((char*)cfg->weight_cache + align_up(PL.bytes, 256));     // a launch that fills the cache also resets its control block
        if (pack_weights(pk, st)) return 1;
        // With a persistent weight cache nothing has just written the packed copies: every launch brings the streams its successor reads
        // first into the Infinity Cache (TouchList, fused.h)
        const bool touch = cfg->weight_cache != nullptr && !tiled;
        const size_t ffn_pb = packed_bytes(pl.dff, d, comp), in_pb = packed_bytes(3 * d, d, comp), out_pb = packed_bytes(d, d, comp);
        if (cut) {
            // cut mode: per layer [token preparation | layer input .. LayerNorm1] (4 waves per clip) + [FFN .. LayerNorm2 (+ pooled head)] (8 waves)
            fp.mode = FUSED_MODE_ATTN;
            for (int l = 0; l < pl.L; ++l) {
                fp.l0 = l;
                memset(&fp.touch, 0, sizeof(fp.touch));
                if (touch) { touch_add(fp.touch, PL.layer[l].lin1_w, ffn_pb); touch_add(fp.touch, PL.layer[l].lin2_w, ffn_pb); }
                if (fused_forward(fp, comp, st)) return 1;
                fp.zero_word = nullptr;
                memset(&fp.touch, 0, sizeof(fp.touch));
                if (touch && l + 1 < pl.L) { touch_add(fp.touch, PL.layer[l + 1].in_w, in_pb); touch_add(fp.touch, PL.layer[l + 1].out_w, out_pb); }
                else if (touch && training) { touch_add(fp.touch, PL.layer[l].lin2_wt, ffn_pb); touch_add(fp.touch, PL.layer[l].lin1_wt, ffn_pb); }
                if (ffn_cut_forward(fp, l, comp, st)) return 1;
            }
            return 0;
        }
        if (!tiled) {
            if (touch) {        // one launch: its own FFN streams (read ~25 us in) and, in training, the backward's first
                touch_add(fp.touch, PL.layer[0].lin1_w, ffn_pb); touch_add(fp.touch, PL.layer[0].lin2_w, ffn_pb);
                if (training) { touch_add(fp.touch, PL.layer[pl.L - 1].lin2_wt, ffn_pb); touch_add(fp.touch, PL.layer[pl.L - 1].lin1_wt, ffn_pb); }
            }
            if (fused_forward(fp, comp, st)) return 1;
            if (ce && with_head && !ce_fused) return weighted_ce(logits_out, ce->target, ce->class_weight, B, head->n_out, ce->loss, ce->d_logits, st);
            return 0;
        }
        // tiled mode: token preparation + Q | K | V of layer 0, then per layer [attention of every clip] [out-projection .. LayerNorm2
        // + Q | K | V of the next layer] (2 L + 1 launches), then the pooled head on the output tokens
        float* attn = (float*)((char*)saved + tiled_attn_offset(cfg, segs, vp));
        float* lse = (float*)((char*)saved + tiled_lse_offset(cfg, segs, vp));
        float* extra = (float*)((char*)saved + tiled_tokens_offset(cfg, segs, vp));
        if (!fp.tokens_out) fp.tokens_out = extra;
        fp.attn_in = attn;
        fp.mode = FUSED_MODE_PRE;
        if (fused_forward(fp, comp, st)) return 1;
        for (int l = 0; l < pl.L; ++l) {
            TiledAttnParams ap;
            memset(&ap, 0, sizeof(ap));
            ap.qkv = fp.qkv_out + (size_t)l * vp.vB * FUSED_TOK_PAD * 3 * d;
            ap.attn_o = attn + (size_t)l * N * d;
            ap.lse = lse + (size_t)l * B * pl.H * S;
            ap.B = B; ap.S = S; ap.tpc = vp.tpc;
            ap.drop_key = fp.layer[l].attn_key; ap.drop_thresh = fp.layer[l].attn_thresh; ap.drop_inv = fp.layer[l].drop_inv;
            ap.seed_ptr = cfg->seed_ptr; ap.layer = l;
            if (tiled_attn_fwd(ap, comp, st)) return 1;
            fp.mode = FUSED_MODE_POST; fp.l0 = l;
            if (fused_forward(fp, comp, st)) return 1;
        }
        if (with_head) {
            if (pool_head_fwd(fp.tokens_out, B, S, d, head->ln_w, head->ln_b, cfg->ln_eps, head->W, head->b, head->n_out, extra + (size_t)N * d, logits_out, st)) return 1;
            if (ce) return weighted_ce(logits_out, ce->target, ce->class_weight, B, head->n_out, ce->loss, ce->d_logits, st);
        }
        return 0;
    }
    if (ferr || terr) return 1;
    {
        bool werr;
        EGX_CHECK(cfg->out_tokens == 0 || cfg->out_tokens == S, "out_tokens is implemented by the fused per-clip kernels only (egx_encoder_impl() == EGX_IMPL_FUSED)");
        if (use_wide(cfg, segs, pl, &werr)) {
            size_t wsv = 0, wsc = 0;
            wide_workspace(cfg, segs, B, &wsv, &wsc);
            float* tk = tokens_out;
            float* pooled = nullptr;
            if (with_head) {        // tokens and the pooled vector live behind the wide path's saved block
                float* extra = fptr(saved, align_up(wsv, 256));
                pooled = extra + (size_t)N * d;
                if (!tk) tk = extra;
            }
            if (wide_encoder_fwd(cfg, segs, ln_w, ln_b, layers, B, tk, saved, training, seed, st)) return 1;
            if (with_head) {
                if (pool_head_fwd(tk, B, S, d, head->ln_w, head->ln_b, cfg->ln_eps, head->W, head->b, head->n_out, pooled, logits_out, st)) return 1;
                if (ce) return weighted_ce(logits_out, ce->target, ce->class_weight, B, head->n_out, ce->loss, ce->d_logits, st);
            }
            return 0;
        }
        if (werr) return 1;
    }
    EGX_CHECK(!(cfg->seed_ptr && training && (cfg->p_drop > 0.f || cfg->p_pos > 0.f || cfg->p_feat > 0.f)),
              "device-resident dropout seed (seed_ptr) is only supported by the fused kernels");
    EGX_CHECK(!packed_feats(segs, pl.nseg), "bf16 / frame-pooled features (egx_segment.feat_bf16 / pool) are only supported by the wide bf16 path");

    // generic path with a head: tokens and the pooled vector live behind the layer intermediates in `saved`
    float* head_pooled = nullptr;
    if (with_head) {
        float* tk = fptr(saved, align_up(pl.saved_bytes, 256));
        head_pooled = tk + (size_t)N * d;
        if (!tokens_out) tokens_out = tk;
    }
    float* x0 = pl.L > 0 ? fptr(saved, pl.layer[0].x_in) : tokens_out;
    for (int i = 0; i < pl.nseg; ++i) {
        const egx_segment& sg = segs[i];
        int rows = B * sg.T;
        const float* pre = sg.feat;
        if (sg.proj_w) {
            float* po = fptr(saved, pl.seg_pre[i]);
            Drop df = make_drop(training, cfg->p_feat, seed, (uint32_t)i, SITE_FEAT);
            if (linear_nt(sg.feat, sg.proj_w, sg.proj_b, po, rows, d, sg.d_in, 0, df, nullptr, comp, st)) return 1;
            pre = po;
        }
        LnFwdParams lp;
        lp.x = pre; lp.w = ln_w; lp.b = ln_b; lp.eps = cfg->ln_eps;
        lp.stats = fptr(saved, pl.seg_stats[i]);
        lp.y = x0; lp.rows = rows; lp.d = d;
        lp.T = sg.T; lp.S = S; lp.off = pl.seg_off[i];
        lp.add_vec = sg.add_vec; lp.pos = sg.pos; lp.pos_stride = sg.pos_stride;
        Drop dp = make_drop(training, cfg->p_pos, seed, 0, SITE_POS);
        lp.drop_key = dp.key; lp.drop_thresh = dp.thresh; lp.drop_inv_keep = dp.inv_keep;
        if (layernorm_fwd(lp, st)) return 1;
    }

    for (int l = 0; l < pl.L; ++l) {
        const LayerOff& o = pl.layer[l];
        const egx_layer& w = layers[l];
        const float* x_in = cfptr(saved, o.x_in);
        float* x_out = (l + 1 < pl.L) ? fptr(saved, pl.layer[l + 1].x_in) : tokens_out;
        Drop none;
        if (linear_nt(x_in, w.in_proj_w, w.in_proj_b, fptr(saved, o.qkv), N, 3 * d, d, 0, none, nullptr, comp, st)) return 1;
        Drop da = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_ATTN);
        if (attention_fwd(cfptr(saved, o.qkv), fptr(saved, o.attn_o), fptr(saved, o.lse), B, S, pl.H, d, da.key, da.thresh,
                          da.inv_keep, st)) return 1;
        Drop d1 = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES1);
        if (linear_nt(cfptr(saved, o.attn_o), w.out_proj_w, w.out_proj_b, fptr(saved, o.res1), N, d, d, 0, d1, x_in, comp, st)) return 1;
        LnFwdParams l1;
        l1.x = cfptr(saved, o.res1); l1.w = w.norm1_w; l1.b = w.norm1_b; l1.eps = cfg->ln_eps;
        l1.stats = fptr(saved, o.stats1); l1.y = fptr(saved, o.x1); l1.rows = N; l1.d = d;
        if (layernorm_fwd(l1, st)) return 1;
        Drop dh = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_FFN);
        if (linear_nt(cfptr(saved, o.x1), w.lin1_w, w.lin1_b, fptr(saved, o.hid), N, pl.dff, d, 1, dh, nullptr, comp, st)) return 1;
        Drop d2 = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES2);
        if (linear_nt(cfptr(saved, o.hid), w.lin2_w, w.lin2_b, fptr(saved, o.res2), N, d, pl.dff, 0, d2, cfptr(saved, o.x1), comp, st)) return 1;
        LnFwdParams l2;
        l2.x = cfptr(saved, o.res2); l2.w = w.norm2_w; l2.b = w.norm2_b; l2.eps = cfg->ln_eps;
        l2.stats = fptr(saved, o.stats2); l2.y = x_out; l2.rows = N; l2.d = d;
        if (layernorm_fwd(l2, st)) return 1;
    }
    if (with_head) {
        if (pool_head_fwd(tokens_out, B, S, d, head->ln_w, head->ln_b, cfg->ln_eps, head->W, head->b, head->n_out,
                          head_pooled, logits_out, st)) return 1;
        if (ce) return weighted_ce(logits_out, ce->target, ce->class_weight, B, head->n_out, ce->loss, ce->d_logits, st);
    }
    return 0;
}

static int encoder_bwd_impl(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                            const egx_layer* layers, const egx_head* head, int B, float* d_tokens, const float* d_logits,
                            const void* saved, void* scratch, const egx_segment_grads* seg_grads, float* d_ln_w,
                            float* d_ln_b, const egx_layer_grads* layer_grads, const egx_head_grads* head_grads,
                            int training, uint64_t seed, void* stream) {
    (void)ln_b;
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return 1;
    const bool with_head = head && head->W;
    EGX_CHECK((with_head ? (const void*)d_logits : cfg->token_ce ? (const void*)cfg->token_ce->d_logits : (const void*)d_tokens) && saved && scratch && ln_w, "null pointer argument");
    {
        bool ferr, terr = false;
        const bool fused = use_fused(cfg, segs, pl, &ferr);
        const bool tiled = !fused && !ferr && use_tiled(cfg, segs, pl, &terr);
        if (fused || tiled) {
            hipStream_t st = (hipStream_t)stream;
            const int d = pl.d, S = pl.S, comp = cfg->compute;
            const int N = (int)pl.N;
            Plan vp = pl;
            if (tiled) plan_tiled(vp);
            FusedPackLayout PL = fused_pack_layout(cfg, segs, vp, fused_pack_base(cfg, saved, vp));
            const egx_token_ce* tce = cfg->token_ce;
            if (tce) {
                EGX_CHECK(!with_head && !tiled && token_ce_ok(cfg, segs, pl), "egx_config.token_ce: not on this configuration (egx_encoder_token_ce_ok)");
                EGX_CHECK(tce->W && tce->d_logits && tce->C >= 1 && tce->C <= 8, "egx_config.token_ce: W, d_logits and 1 <= C <= 8 must be set");
            }
            FusedBwdScratch SC = fused_bwd_scratch(cfg, segs, vp, with_head ? head->n_out : tce ? tce->C : 0);
            FusedBwdParams bp;
            memset(&bp, 0, sizeof(bp));
            const bool touch = cfg->weight_cache != nullptr && !tiled;      // (see the forward)
            const size_t ffn_pb = packed_bytes(pl.dff, d, comp), in_pb = packed_bytes(3 * d, d, comp), out_pb = packed_bytes(d, d, comp);
            for (int i = 0; i < pl.nseg; ++i) {
                FusedSeg& fs = bp.seg[i];
                fs.add_vec = segs[i].add_vec; fs.pos = segs[i].pos; fs.T = segs[i].T; fs.d_in = segs[i].d_in;
                fs.off = pl.seg_off[i]; fs.pos_stride = segs[i].pos_stride; fs.row0 = 0; fs.Tfull = segs[i].T; fs.seg_id = i;
                bp.dseg_out[i] = fptr(scratch, SC.dseg[i]);
                Drop df = make_drop(training, cfg->p_feat, seed, (uint32_t)i, SITE_FEAT);
                bp.feat_key[i] = df.key; bp.feat_thresh = df.thresh; bp.feat_inv = df.inv_keep;
            }
            bp.n_heads = pl.H;
            bool want_pos = false;
            for (int i = 0; i < pl.nseg && seg_grads; ++i) want_pos = want_pos || seg_grads[i].pos;
            bp.dx0_out = want_pos ? fptr(scratch, SC.dx0) : nullptr;
            for (int l = 0; l < pl.L; ++l) {
                FusedBwdLayer& fl = bp.layer[l];
                const egx_layer& w = layers[l];
                fl.in_proj_wp = PL.layer[l].in_w; fl.in_proj_wtp = PL.layer[l].in_wt; fl.out_proj_wtp = PL.layer[l].out_wt;
                fl.lin1_wp = PL.layer[l].lin1_w; fl.lin2_wtp = PL.layer[l].lin2_wt; fl.lin1_wtp = PL.layer[l].lin1_wt;
                fl.in_proj_b = w.in_proj_b; fl.lin1_b = w.lin1_b;
                fl.norm1_w = w.norm1_w; fl.norm1_b = w.norm1_b; fl.norm2_w = w.norm2_w; fl.norm2_b = w.norm2_b;
                Drop da = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_ATTN);
                fl.attn_key = da.key; fl.attn_thresh = da.thresh; fl.drop_inv = da.inv_keep;
                fl.res_thresh = da.thresh; fl.ffn_thresh = da.thresh;
                fl.res1_key = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES1).key;
                fl.ffn_key = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_FFN).key;
                fl.res2_key = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES2).key;
                fl.x1_out = fptr(scratch, SC.x1[l]); fl.g2_out = fptr(scratch, SC.g2[l]); fl.attn_o_out = fptr(scratch, SC.attn_o[l]);
                fl.g1_out = fptr(scratch, SC.g1[l]); fl.dqkv_out = fptr(scratch, SC.dqkv[l]);
                fl.x_in_out = const_cast<float*>((const float*)((const char*)saved + fused_xin_offset(cfg, segs, vp))) + (size_t)l * N * d;      // saved by the forward
                // tiled mode: the attention output was saved by the forward too (the per-clip kernels recompute it in P10)
                if (tiled) fl.attn_o_out = const_cast<float*>((const float*)((const char*)saved + tiled_attn_offset(cfg, segs, vp))) + (size_t)l * N * d;
            }
            bp.ln_w = ln_w; bp.ln_b = ln_b; bp.eps = cfg->ln_eps;
            bp.nseg = pl.nseg; bp.n_layers = pl.L; bp.B = vp.vB; bp.S = tiled ? FUSED_TOK_PAD : S; bp.d_ff = pl.dff;
            bp.tiled = tiled ? 1 : 0; bp.tpc = vp.tpc; bp.S_clip = S; bp.Ntok = pl.N;
            bp.d_tokens = d_tokens;
            bp.out_T = cfg->out_tokens > 0 ? cfg->out_tokens : S;
            if (with_head) {        // (tiled mode since round 5 too: the first tile launch runs the head backward from the saved token means)
                bp.head.ln_w = head->ln_w; bp.head.ln_b = head->ln_b; bp.head.W = head->W; bp.head.b = head->b; bp.head.n_out = head->n_out;
                bp.d_logits = d_logits;
                bp.d_logits_scale = cfg->d_logits_scale;
                bp.head_off = fused_partial_len(pl.L, pl.nseg);
                if (tiled) bp.pooled = (const float*)((const char*)saved + tiled_tokens_offset(cfg, segs, vp)) + (size_t)N * d;
            }
            if (tce) {
                bp.tce_W = tce->W; bp.tce_dlogits = tce->d_logits; bp.tce_C = tce->C;
                bp.d_logits_scale = cfg->d_logits_scale;
                bp.head_off = fused_partial_len(pl.L, pl.nseg);
            }
            bp.saved_pre = (const float*)saved;
            bp.saved_res = (const float*)saved + (size_t)N * d;
            bp.saved_qkv = (const float*)((const char*)saved + fused_qkv_offset(cfg, segs, vp));
            bp.relu_bits = (const uint32_t*)((const char*)saved + fused_res_bytes(vp));
            bp.dhid_out = store_hidden() ? (char*)scratch + SC.dhid : nullptr;
            bp.xg_planes = split_planes(cfg) ? 1 : 0;
            EGX_CHECK(cfg->zero_bytes % 16 == 0 && (((uintptr_t)cfg->zero_buf) & 15) == 0, "zero_buf must be 16-byte aligned and sized");
            bp.zero_buf = (float*)cfg->zero_buf; bp.zero_n = cfg->zero_buf ? cfg->zero_bytes / 4 : 0;
            bp.partials = fptr(scratch, SC.partials); bp.P = SC.P;
            Drop dpz = make_drop(training, cfg->p_pos, seed, 0, SITE_POS);
            bp.pos_key = dpz.key; bp.pos_thresh = dpz.thresh; bp.pos_inv = dpz.inv_keep;
            bp.seed_ptr = cfg->seed_ptr;
            bp.rot_mode = ffn_rot_mode();
            { static const int sh = [] { const char* e = getenv("EGX_CUT_SHIFT_B"); return e ? atoi(e) + 1 : 0; }(); bp.rot_mode |= sh << 8; }    // tuning aid: ffn_bwd_kernel
            const int stage = cfg->bwd_stage;
            EGX_CHECK(stage >= 0 && stage <= 2, "bwd_stage=%d", stage);
            if (stage == 2) bp.zero_buf = nullptr;
            bp.n_slices = tiled ? 1 : fused_slices(pl, comp);
            if (bp.n_slices > 1 && stage != 2) {
                bp.xchg = fptr(scratch, SC.xchg);
                bp.xflags = (unsigned*)((char*)saved + fused_core_bytes(cfg, segs, vp) + sliced_xchg_bytes(pl, fused_slices_layout(pl))) + sliced_flag_words(pl);
                EGX_HIP(hipMemsetAsync(bp.xflags, 0, sliced_flag_words(pl) * 4, st));
                bp.slice_drop = slice_drop_mask(bp.n_slices);
            }
            // FFN weight gradient of layer l (dW1, db1, dW2 from the stored H / dH tiles and the x1 / g2 planes); every layer's slabs are summed by ONE
            // launch behind the last one. (Round 6 tried it right behind ffn_bwd_kernel of the same layer, while dH and g2 are still in the Infinity
            // Cache: three same-box pairs 372.1 / 374.9 / 371.9 vs 372.2 / 370.3 / 376.1 us, no difference; and the exact-fp32 mode needs x1 from the
            // attention-side launch first. It stays behind the whole backward.)
            SlabReduce red;
            red.narr = 0; red.nslab = 0;
            auto launch_ffn_dw = [&](int l) -> int {
                const egx_layer& w = layers[l];
                const egx_layer_grads& gw = layer_grads[l];
                if (!(gw.lin1_w || gw.lin1_b || gw.lin2_w)) return 0;
                FfnDwParams fp;
                memset(&fp, 0, sizeof(fp));
                fp.x1 = bp.layer[l].x1_out; fp.g = bp.layer[l].g2_out;
                fp.w1p = PL.layer[l].lin1_w; fp.w2tp = PL.layer[l].lin2_wt; fp.b1 = w.lin1_b;
                fp.N = N; fp.S = S; fp.d_ff = pl.dff;
                fp.drop_key = bp.layer[l].ffn_key; fp.drop_thresh = bp.layer[l].ffn_thresh; fp.drop_inv = bp.layer[l].drop_inv;
                fp.seed_ptr = cfg->seed_ptr; fp.layer = l;
                if (store_hidden()) {
                    size_t lo = (size_t)l * fused_hid_bytes(vp.vB, pl.dff, comp == EGX_BF16);
                    fp.hs = (const char*)saved + fused_hid_offset(cfg, segs, vp) + lo;
                    fp.dhs = (const char*)scratch + SC.dhid + lo;
                    fp.B = vp.vB;
                    fp.xg_planes = bp.xg_planes;
                    if (fp.xg_planes) fp.x1 = (const float*)((const char*)saved + fused_x1p_offset(cfg, segs, vp) + (size_t)l * vp.vB * FUSED_TOK_PAD * d * plane_elem_bytes(cfg));
                }
                return ffn_dw(fp, comp, gw.lin1_w, gw.lin1_b, gw.lin2_w, (char*)scratch + SC.ffn_slab[l], st, nullptr, cfg->deterministic != 0, &red);
            };
            if (stage != 2 && use_cut(pl, bp.n_slices, tiled, comp)) {
                // cut mode: per layer, top down, [LayerNorm2 backward + FFN input gradient] (8 waves per clip) + [LayerNorm1 backward .. the layer
                // input's gradient, or the token-preparation backward] (4 waves)
                bp.cut = 1; bp.dy1 = fptr(scratch, SC.dy1); bp.dxin = fptr(scratch, SC.dxin);
                for (int l = pl.L - 1; l >= 0; --l) {
                    bp.cut_layer = l;
                    memset(&bp.touch, 0, sizeof(bp.touch));
                    if (touch) { touch_add(bp.touch, PL.layer[l].out_wt, out_pb); touch_add(bp.touch, PL.layer[l].in_wt, in_pb); }
                    if (ffn_cut_backward(bp, l, comp, st)) return 1;
                    bp.zero_buf = nullptr;
                    memset(&bp.touch, 0, sizeof(bp.touch));
                    if (touch && l > 0) { touch_add(bp.touch, PL.layer[l - 1].lin2_wt, ffn_pb); touch_add(bp.touch, PL.layer[l - 1].lin1_wt, ffn_pb); }
                    if (fused_backward(bp, comp, st)) return 1;
                }
            } else
            if (stage != 2 && !tiled) {
                if (touch) { touch_add(bp.touch, PL.layer[pl.L - 1].out_wt, out_pb); touch_add(bp.touch, PL.layer[pl.L - 1].in_wt, in_pb); }
                if (fused_backward(bp, comp, st)) return 1;
            }
            if (stage != 2 && tiled) {
                // L + 1 launches of the tile kernel with the attention backward of every clip between them
                bp.datt = fptr(scratch, SC.datt); bp.dres = fptr(scratch, SC.dres);
                const float* lse = (const float*)((const char*)saved + tiled_lse_offset(cfg, segs, vp));
                for (int l = pl.L - 1; l >= 0; --l) {
                    bp.l_back = l; bp.l_front = l + 1 < pl.L ? l + 1 : -1;
                    if (fused_backward(bp, comp, st)) return 1;
                    bp.zero_buf = nullptr;
                    TiledAttnParams ap;
                    memset(&ap, 0, sizeof(ap));
                    ap.qkv = bp.saved_qkv + (size_t)l * vp.vB * FUSED_TOK_PAD * 3 * d;
                    ap.attn_o = bp.layer[l].attn_o_out;
                    ap.lse = const_cast<float*>(lse) + (size_t)l * B * pl.H * S;
                    ap.d_o = bp.datt; ap.delta = fptr(scratch, SC.delta); ap.dqkv = bp.layer[l].dqkv_out;
                    ap.B = B; ap.S = S; ap.tpc = vp.tpc;
                    ap.drop_key = bp.layer[l].attn_key; ap.drop_thresh = bp.layer[l].attn_thresh; ap.drop_inv = bp.layer[l].drop_inv;
                    ap.seed_ptr = cfg->seed_ptr; ap.layer = l;
                    if (tiled_attn_bwd(ap, comp, st)) return 1;
                }
                bp.l_back = -1; bp.l_front = 0;
                if (fused_backward(bp, comp, st)) return 1;
            }

            // small parameter gradients: sum the per-clip partials
            ReducePartialsParams rp;
            memset(&rp, 0, sizeof(rp));
            rp.B = vp.vB; rp.P = SC.P; rp.partials = bp.partials;
            auto add_dst = [&](float* dst, int off, int len) { if (dst) { rp.d[rp.n].dst = dst; rp.d[rp.n].off = off; rp.d[rp.n].len = len; ++rp.n; } };
            for (int l = 0; l < pl.L; ++l) {
                const egx_layer_grads& gw = layer_grads[l];
                int o = l * FUSED_P_LAYER;
                add_dst(gw.norm2_w, o + 0, 128); add_dst(gw.norm2_b, o + 128, 128); add_dst(gw.lin2_b, o + 256, 128);
                add_dst(gw.norm1_w, o + 384, 128); add_dst(gw.norm1_b, o + 512, 128); add_dst(gw.out_proj_b, o + 640, 128);
                add_dst(gw.in_proj_b, o + 768, 384);
            }
            int og = pl.L * FUSED_P_LAYER;
            add_dst(d_ln_w, og, 128); add_dst(d_ln_b, og + 128, 128);
            for (int i = 0; i < pl.nseg; ++i) {
                if (!seg_grads) break;
                EGX_CHECK(!seg_grads[i].feat, "fused backward: feature gradients are not supported (use impl=generic)");
                add_dst(seg_grads[i].add_vec, og + 256 + i * 256, 128);
                add_dst(seg_grads[i].proj_b, og + 256 + i * 256 + 128, 128);
            }
            if (with_head && head_grads) {
                int oh = fused_partial_len(pl.L, pl.nseg);
                add_dst(head_grads->ln_w, oh, 128); add_dst(head_grads->ln_b, oh + 128, 128);
                add_dst(head_grads->b, oh + 256, head->n_out);
                add_dst(head_grads->W, oh + 256 + FUSED_HEAD_MAX_OUT, head->n_out * 128);
            }
            if (tce) {
                int oh = fused_partial_len(pl.L, pl.nseg);
                add_dst(tce->d_b, oh + 256, tce->C);
                add_dst(tce->d_W, oh + 256 + FUSED_HEAD_MAX_OUT, tce->C * 128);
            }
            // learned positional table (the HOI translators' `pe`): sum d(token-prep output) over the clips, per segment
            if (bp.dx0_out && stage != 2)
                for (int i = 0; i < pl.nseg; ++i)
                    if (seg_grads[i].pos && pos_grad_accum(bp.dx0_out, B, S, pl.seg_off[i], segs[i].T, d, seg_grads[i].pos, segs[i].pos_stride, 0, 0, 1.f, st)) return 1;
            // the partial-row reduction rides in the slab-reduction launch of the first FFN weight gradient
            bool rp_pending = stage != 2;
            void* slab = (char*)scratch + SC.slabs;
            for (int l = 0; l < pl.L && stage != 2; ++l)
                if (launch_ffn_dw(l)) return 1;
            // One-stage backward outside the deterministic mode: the reductions ride in the grouped small-gradient launch below
            // (its workgroups each sum 1 / grid of the slabs and partial rows first). Otherwise (two-stage backward: the late
            // region must be complete when the exchange starts; deterministic: fixed single-adder order) they get their own launch.
            bool any_small = false;
            for (int l = 0; l < pl.L; ++l) any_small = any_small || layer_grads[l].out_proj_w || layer_grads[l].in_proj_w;
            for (int i = 0; i < pl.nseg && seg_grads; ++i) any_small = any_small || seg_grads[i].proj_w;
            SmallDwTail tail;
            // egx_config.advance_seed == 2: the backward advances the device seed behind its last reader (the last launch that can run here)
            uint64_t* adv = (cfg->advance_seed == 2 && cfg->seed_ptr && training && stage != 1) ? const_cast<uint64_t*>(cfg->seed_ptr) : nullptr;
            // Round 6, one-stage backward: small_dw writes tiles and ONE fixed-order launch sums them, the FFN slabs and the partial rows (tail_reduce,
            // fused_bwd.hip): no float atomics, bit-reproducible in every mode. EGX_TAIL_REDUCE=0: the round-5 launches (atomics with the reductions
            // riding in small_dw; the slow three-pass path in deterministic mode) — tuning aid. The staged backward (stage 1 / 2) keeps its own launches.
            static const bool tail_env = [] { const char* e = getenv("EGX_TAIL_REDUCE"); return !(e && e[0] == '0'); }();
            if (stage == 0 && tail_env) {
                TouchList tl;
                memset(&tl, 0, sizeof(tl));
                if (touch) {    // the next forward starts with the projections and layer 0's in-projection (one contiguous run of the cache) and out-projection
                    touch_add(tl, PL.proj[0], (size_t)((const char*)PL.layer[0].in_wt - (const char*)PL.proj[0]));
                    touch_add(tl, PL.layer[0].out_w, out_pb);
                }
                SmallDwParams sp;
                memset(&sp, 0, sizeof(sp));
                bool first = true;
                auto flush = [&]() -> int {
                    if (sp.n && small_dw(sp, comp, st, (char*)scratch + SC.sdw_tiles, SC.sdw_bytes, nullptr, false)) return 1;
                    const int rc = (sp.n || first) ? tail_reduce(sp.n ? &sp : nullptr, first ? &red : nullptr, first ? &rp : nullptr, first ? adv : nullptr, first ? &tl : nullptr, st) : 0;
                    first = false;
                    memset(&sp, 0, sizeof(sp));
                    return rc;
                };
                auto add = [&](const float* G, int ldg, const float* X, int ldx, float* out, int R, int Cc, int K) -> int {
                    if (!out) return 0;
                    if (sp.n == SMALL_DW_MAX && flush()) return 1;
                    SmallDwProblem& q = sp.pr[sp.n++];
                    q.G = G; q.X = X; q.out = out; q.R = R; q.C = Cc; q.K = K; q.ldg = ldg; q.ldx = ldx;
                    return 0;
                };
                for (int l = 0; l < pl.L; ++l) {
                    const egx_layer_grads& gw = layer_grads[l];
                    if (add(bp.layer[l].g1_out, d, bp.layer[l].attn_o_out, d, gw.out_proj_w, d, d, N)) return 1;
                    if (add(bp.layer[l].dqkv_out, 3 * d, bp.layer[l].x_in_out, d, gw.in_proj_w, 3 * d, d, N)) return 1;
                }
                for (int i = 0; i < pl.nseg; ++i)
                    if (seg_grads && add(bp.dseg_out[i], d, segs[i].feat, segs[i].d_in, seg_grads[i].proj_w, d, segs[i].d_in, B * segs[i].T)) return 1;
                return flush();
            }
            const bool ride = stage == 0 && !cfg->deterministic && any_small && (red.narr || rp_pending) && reduce_rides();
            if (ride) {
                small_dw_tail_init(tail, red, rp_pending ? &rp : nullptr);
                tail.seed_advance = adv; adv = nullptr;
                if (touch) {    // the next forward starts with the projections and layer 0's in-projection (one contiguous run of the cache) and out-projection
                    touch_add(tail.touch, PL.proj[0], (size_t)((const char*)PL.layer[0].in_wt - (const char*)PL.proj[0]));
                    touch_add(tail.touch, PL.layer[0].out_w, out_pb);
                }
                rp_pending = false;
            } else if (red.narr) {
                if (ffn_dw_reduce(red, rp_pending ? &rp : nullptr, cfg->deterministic != 0, st)) return 1;
                rp_pending = false;
            }
            if (rp_pending && reduce_partials(rp, st, cfg->deterministic != 0)) return 1;
            bool tail_pending = ride;
            // every remaining weight gradient (dW_o, dW_in per layer, dW_proj per segment) in grouped launches
            if (stage != 1) {
                SmallDwParams sp;
                memset(&sp, 0, sizeof(sp));
                auto flush = [&]() -> int {
                    int rc = sp.n ? small_dw(sp, comp, st, cfg->deterministic ? slab : nullptr, SC.slab_bytes, tail_pending ? &tail : nullptr) : 0;
                    if (sp.n) tail_pending = false;
                    memset(&sp, 0, sizeof(sp));
                    return rc;
                };
                auto add = [&](const float* G, int ldg, const float* X, int ldx, float* out, int R, int Cc, int K) -> int {
                    if (!out) return 0;
                    if (sp.n == SMALL_DW_MAX && flush()) return 1;
                    SmallDwProblem& q = sp.pr[sp.n++];
                    q.G = G; q.X = X; q.out = out; q.R = R; q.C = Cc; q.K = K; q.ldg = ldg; q.ldx = ldx;
                    return 0;
                };
                for (int l = 0; l < pl.L; ++l) {
                    const egx_layer_grads& gw = layer_grads[l];
                    if (add(bp.layer[l].g1_out, d, bp.layer[l].attn_o_out, d, gw.out_proj_w, d, d, N)) return 1;
                    if (add(bp.layer[l].dqkv_out, 3 * d, bp.layer[l].x_in_out, d, gw.in_proj_w, 3 * d, d, N)) return 1;
                }
                for (int i = 0; i < pl.nseg; ++i)
                    if (seg_grads && add(bp.dseg_out[i], d, segs[i].feat, segs[i].d_in, seg_grads[i].proj_w, d, segs[i].d_in, B * segs[i].T)) return 1;
                if (flush()) return 1;
            }
            if (adv && seed_advance(adv, st)) return 1;
            return 0;
        }
        if (ferr || terr) return 1;
    }
    {
        bool werr;
        if (use_wide(cfg, segs, pl, &werr)) {
            EGX_CHECK(layers && layer_grads, "null layers / layer_grads");
            if (cfg->bwd_stage == 2) return 0;      // no deferred part
            hipStream_t st = (hipStream_t)stream;
            size_t wsv = 0, wsc = 0;
            wide_workspace(cfg, segs, B, &wsv, &wsc);
            const float* dtok = d_tokens;
            bool zeroed = false;
            if (with_head) {
                const float* extra = cfptr(saved, align_up(wsv, 256));
                const float* pooled = extra + (size_t)pl.N * pl.d;
                float* dt = fptr(scratch, align_up(wsc, 256));
                if (cfg->zero_buf && cfg->zero_bytes) { EGX_HIP(hipMemsetAsync(cfg->zero_buf, 0, cfg->zero_bytes, st)); zeroed = true; }
                if (pool_head_bwd(d_logits, pooled, B, pl.S, pl.d, head->ln_w, head->ln_b, cfg->ln_eps, head->W, head->n_out, dt,
                                  head_grads ? head_grads->ln_w : nullptr, head_grads ? head_grads->ln_b : nullptr,
                                  head_grads ? head_grads->W : nullptr, head_grads ? head_grads->b : nullptr, st)) return 1;
                dtok = dt;
            }
            egx_config c2 = *cfg;
            if (zeroed) { c2.zero_buf = nullptr; c2.zero_bytes = 0; }
            return wide_encoder_bwd(&c2, segs, ln_w, layers, B, dtok, saved, scratch, seg_grads, d_ln_w, d_ln_b, layer_grads, training, seed, st);
        }
        if (werr) return 1;
    }
    EGX_CHECK(pl.L == 0 || (layers && layer_grads), "null layers / layer_grads");
    if (cfg->bwd_stage == 2) return 0;      // the generic path has no deferred part
    hipStream_t st = (hipStream_t)stream;
    const int d = pl.d, S = pl.S, comp = cfg->compute, dff = pl.dff;
    const int N = (int)pl.N;
    if (cfg->zero_buf && cfg->zero_bytes) EGX_HIP(hipMemsetAsync(cfg->zero_buf, 0, cfg->zero_bytes, st));
    // deterministic mode: every cross-workgroup sum below goes through partial buffers and fixed-order reductions
    DetScope det_scope(cfg->deterministic ? (char*)scratch + pl.s_det : nullptr, pl.det_bytes);
    if (with_head) {
        const float* tk = cfptr(saved, align_up(pl.saved_bytes, 256));
        const float* pooled = tk + (size_t)N * d;
        d_tokens = fptr(scratch, align_up(pl.scratch_bytes, 256));
        if (pool_head_bwd(d_logits, pooled, B, S, d, head->ln_w, head->ln_b, cfg->ln_eps, head->W, head->n_out, d_tokens,
                          head_grads ? head_grads->ln_w : nullptr, head_grads ? head_grads->ln_b : nullptr,
                          head_grads ? head_grads->W : nullptr, head_grads ? head_grads->b : nullptr, st)) return 1;
    }
    float* dA = fptr(scratch, pl.s_dA);
    float* dBm = fptr(scratch, pl.s_dB);
    float* dqkv = fptr(scratch, pl.s_dqkv);
    float* dhid = fptr(scratch, pl.s_dhid);
    void* slab = (char*)scratch + pl.s_slab;
    float* g = d_tokens;

    for (int l = pl.L - 1; l >= 0; --l) {
        const LayerOff& o = pl.layer[l];
        const egx_layer& w = layers[l];
        const egx_layer_grads& gw = layer_grads[l];
        // LayerNorm2 backward: dA = d(res2)
        LnBwdParams b2;
        b2.dy = g; b2.pre = cfptr(saved, o.res2); b2.stats = cfptr(saved, o.stats2); b2.w = w.norm2_w;
        b2.dx = dA; b2.dw = gw.norm2_w; b2.db = gw.norm2_b; b2.rows = N; b2.d = d;
        if (layernorm_bwd(b2, st)) return 1;
        // dropout2 on the FFN branch
        const float* dbr = dA;
        Drop d2 = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES2);
        if (d2.thresh) {
            EGX_HIP(hipMemcpyAsync(dBm, dA, (size_t)N * d * 4, hipMemcpyDeviceToDevice, st));
            if (apply_dropout_mask(dBm, N, d, d2.key, d2.thresh, d2.inv_keep, st)) return 1;
            dbr = dBm;
        }
        if (gw.lin2_b && colsum_accum(dbr, N, d, d, gw.lin2_b, st)) return 1;
        if (gw.lin2_w && linear_dw(dbr, cfptr(saved, o.hid), gw.lin2_w, N, d, dff, comp, slab, pl.slab_bytes, st)) return 1;
        Drop dh = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_FFN);
        if (linear_dx(dbr, w.lin2_w, dhid, N, d, dff, cfptr(saved, o.hid), dh.inv_keep, nullptr, comp, st)) return 1;
        if (gw.lin1_b && colsum_accum(dhid, N, dff, dff, gw.lin1_b, st)) return 1;
        if (gw.lin1_w && linear_dw(dhid, cfptr(saved, o.x1), gw.lin1_w, N, dff, d, comp, slab, pl.slab_bytes, st)) return 1;
        // dx1 = dhid W1 + d(res2)  -> g
        if (linear_dx(dhid, w.lin1_w, g, N, dff, d, nullptr, 1.f, dA, comp, st)) return 1;
        // LayerNorm1 backward: dA = d(res1)
        LnBwdParams b1;
        b1.dy = g; b1.pre = cfptr(saved, o.res1); b1.stats = cfptr(saved, o.stats1); b1.w = w.norm1_w;
        b1.dx = dA; b1.dw = gw.norm1_w; b1.db = gw.norm1_b; b1.rows = N; b1.d = d;
        if (layernorm_bwd(b1, st)) return 1;
        dbr = dA;
        Drop d1 = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES1);
        if (d1.thresh) {
            EGX_HIP(hipMemcpyAsync(dBm, dA, (size_t)N * d * 4, hipMemcpyDeviceToDevice, st));
            if (apply_dropout_mask(dBm, N, d, d1.key, d1.thresh, d1.inv_keep, st)) return 1;
            dbr = dBm;
        }
        if (gw.out_proj_b && colsum_accum(dbr, N, d, d, gw.out_proj_b, st)) return 1;
        if (gw.out_proj_w && linear_dw(dbr, cfptr(saved, o.attn_o), gw.out_proj_w, N, d, d, comp, slab, pl.slab_bytes, st)) return 1;
        // d(attn_o) -> g
        if (linear_dx(dbr, w.out_proj_w, g, N, d, d, nullptr, 1.f, nullptr, comp, st)) return 1;
        Drop da = make_drop(training, cfg->p_drop, seed, (uint32_t)l, SITE_ATTN);
        if (attention_bwd(cfptr(saved, o.qkv), cfptr(saved, o.attn_o), cfptr(saved, o.lse), g, dqkv, B, S, pl.H, d, da.key,
                          da.thresh, da.inv_keep, st)) return 1;
        if (gw.in_proj_b && colsum_accum(dqkv, N, 3 * d, 3 * d, gw.in_proj_b, st)) return 1;
        if (gw.in_proj_w && linear_dw(dqkv, cfptr(saved, o.x_in), gw.in_proj_w, N, 3 * d, d, comp, slab, pl.slab_bytes, st)) return 1;
        // dx_in = dqkv Win + d(res1) -> g
        if (linear_dx(dqkv, w.in_proj_w, g, N, 3 * d, d, nullptr, 1.f, dA, comp, st)) return 1;
    }

    // token preparation backward
    Drop dp = make_drop(training, cfg->p_pos, seed, 0, SITE_POS);
    for (int i = 0; i < pl.nseg; ++i) {
        const egx_segment& sg = segs[i];
        egx_segment_grads sgr;
        memset(&sgr, 0, sizeof(sgr));
        if (seg_grads) sgr = seg_grads[i];
        int rows = B * sg.T;
        if (sgr.pos && pos_grad_accum(g, B, S, pl.seg_off[i], sg.T, d, sgr.pos, sg.pos_stride, dp.key, dp.thresh, dp.inv_keep, st)) return 1;
        bool need_dx = (sg.proj_w && (sgr.proj_w || sgr.proj_b || sgr.feat)) || (!sg.proj_w && sgr.feat);
        bool need_any = need_dx || d_ln_w || d_ln_b || sgr.add_vec;
        if (!need_any) continue;
        float* dseg = (!sg.proj_w && sgr.feat) ? sgr.feat : dA;
        LnBwdParams bp;
        bp.dy = g;
        bp.pre = sg.proj_w ? cfptr(saved, pl.seg_pre[i]) : sg.feat;
        bp.stats = cfptr(saved, pl.seg_stats[i]);
        bp.w = ln_w; bp.dx = dseg; bp.dw = d_ln_w; bp.db = d_ln_b; bp.dadd = sgr.add_vec;
        bp.rows = rows; bp.d = d; bp.T = sg.T; bp.S = S; bp.off = pl.seg_off[i];
        bp.drop_key = dp.key; bp.drop_thresh = dp.thresh; bp.drop_inv_keep = dp.inv_keep;
        if (sg.proj_w) {
            Drop df = make_drop(training, cfg->p_feat, seed, (uint32_t)i, SITE_FEAT);
            bp.out_drop_key = df.key; bp.out_drop_thresh = df.thresh; bp.out_drop_inv_keep = df.inv_keep;
        }
        if (layernorm_bwd(bp, st)) return 1;
        if (sg.proj_w) {
            if (sgr.proj_b && colsum_accum(dseg, rows, d, d, sgr.proj_b, st)) return 1;
            if (sgr.proj_w && linear_dw(dseg, sg.feat, sgr.proj_w, rows, d, sg.d_in, comp, slab, pl.slab_bytes, st)) return 1;
            if (sgr.feat && linear_dx(dseg, sg.proj_w, sgr.feat, rows, d, sg.d_in, nullptr, 1.f, nullptr, comp, st)) return 1;
        }
    }
    return 0;
}

extern "C" {

int egx_encoder_fwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                    const egx_layer* layers, int B, float* tokens_out, void* saved, void* scratch, int training,
                    uint64_t seed, void* stream) {
    return encoder_fwd_impl(cfg, segs, ln_w, ln_b, layers, nullptr, B, tokens_out, nullptr, saved, scratch, training, seed, stream);
}

int egx_encoder_bwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                    const egx_layer* layers, int B, float* d_tokens, const void* saved, void* scratch,
                    const egx_segment_grads* seg_grads, float* d_ln_w, float* d_ln_b,
                    const egx_layer_grads* layer_grads, int training, uint64_t seed, void* stream) {
    return encoder_bwd_impl(cfg, segs, ln_w, ln_b, layers, nullptr, B, d_tokens, nullptr, saved, scratch, seg_grads, d_ln_w,
                            d_ln_b, layer_grads, nullptr, training, seed, stream);
}

int egx_translator_workspace(const egx_config* cfg, const egx_segment* segs, int B, size_t* saved_bytes, size_t* scratch_bytes) {
    Plan pl;
    if (make_plan(cfg, segs, B, pl)) return 1;
    size_t sv = 0, sc = 0;
    if (egx_encoder_workspace(cfg, segs, B, &sv, &sc)) return 1;
    // generic path extras: tokens + pooled behind `saved`, d_tokens behind `scratch`
    size_t wsv = 0, wsc = 0;
    if (wide_ok(cfg, segs, B)) wide_workspace(cfg, segs, B, &wsv, &wsc);
    size_t extra_sv = align_up(size_max(pl.saved_bytes, wsv), 256) + (pl.N + (size_t)B) * pl.d * 4 + 256;
    size_t extra_sc = align_up(size_max(pl.scratch_bytes, wsc), 256) + pl.N * pl.d * 4 + 256;
    if (saved_bytes) *saved_bytes = size_max(sv, extra_sv);
    if (scratch_bytes) *scratch_bytes = size_max(sc, extra_sc);
    return 0;
}

int egx_translator_fwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                       const egx_layer* layers, const egx_head* head, int B, float* logits_out, float* tokens_out,
                       void* saved, void* scratch, int training, uint64_t seed, void* stream) {
    EGX_CHECK(head && head->W, "egx_translator_fwd needs a head (use egx_encoder_fwd otherwise)");
    return encoder_fwd_impl(cfg, segs, ln_w, ln_b, layers, head, B, tokens_out, logits_out, saved, scratch, training, seed, stream);
}

int egx_translator_bwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                       const egx_layer* layers, const egx_head* head, int B, const float* d_logits, const void* saved,
                       void* scratch, const egx_segment_grads* seg_grads, float* d_ln_w, float* d_ln_b,
                       const egx_layer_grads* layer_grads, const egx_head_grads* head_grads, int training,
                       uint64_t seed, void* stream) {
    EGX_CHECK(head && head->W, "egx_translator_bwd needs a head (use egx_encoder_bwd otherwise)");
    return encoder_bwd_impl(cfg, segs, ln_w, ln_b, layers, head, B, nullptr, d_logits, saved, scratch, seg_grads, d_ln_w,
                            d_ln_b, layer_grads, head_grads, training, seed, stream);
}

int egx_pool_head_fwd(const float* tokens, int B, int S, int d, const float* ln_w, const float* ln_b, float ln_eps,
                      const float* W, const float* b, int n_out, float* pooled_saved, float* out, void* stream) {
    EGX_CHECK(tokens && pooled_saved && out, "null pointer argument");
    return pool_head_fwd(tokens, B, S, d, ln_w, ln_b, ln_eps, W, b, n_out, pooled_saved, out, (hipStream_t)stream);
}

int egx_pool_head_bwd(const float* d_out, const float* pooled_saved, int B, int S, int d, const float* ln_w,
                      const float* ln_b, float ln_eps, const float* W, int n_out, float* d_tokens, float* d_ln_w,
                      float* d_ln_b, float* d_W, float* d_b, void* stream) {
    EGX_CHECK(d_out && pooled_saved && d_tokens, "null pointer argument");
    return pool_head_bwd(d_out, pooled_saved, B, S, d, ln_w, ln_b, ln_eps, W, n_out, d_tokens, d_ln_w, d_ln_b, d_W, d_b,
                         (hipStream_t)stream);
}

int egx_linear_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int relu, int compute,
                   void* stream) {
    EGX_CHECK(x && W && y, "null pointer argument");
    Drop none;
    return linear_nt(x, W, b, y, M, N, K, relu, none, nullptr, compute, (hipStream_t)stream);
}
int egx_linear_residual_fwd(const float* x, const float* W, const float* b, const float* residual, float* y, int M, int N, int K,
                            int compute, void* stream) {
    EGX_CHECK(x && W && y && residual, "null pointer argument");
    Drop none;
    return linear_nt(x, W, b, y, M, N, K, 0, none, residual, compute, (hipStream_t)stream);
}
int egx_gelu_fwd(const float* z, float* h, size_t n, void* stream) { return gelu_fwd(z, h, n, (hipStream_t)stream); }
int egx_gelu_bwd(const float* z, const float* dh, float* dz, size_t n, void* stream) { return gelu_bwd(z, dh, dz, n, (hipStream_t)stream); }

size_t egx_linear_bwd_scratch(int M, int N, int K) { return size_max(gemm_scratch_bytes(2, N, K, M), gemm_scratch_bytes(1, M, K, N)); }

int egx_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int M, int N, int K,
                   int compute, void* scratch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    EGX_CHECK(dy, "null dy");
    if (db && colsum_accum(dy, M, N, N, db, st)) return 1;
    if (dW) {
        EGX_CHECK(x && scratch, "linear_bwd: dW needs x and scratch");
        if (linear_dw(dy, x, dW, M, N, K, compute, scratch, gemm_scratch_bytes(2, N, K, M), st)) return 1;
    }
    if (dx) {
        EGX_CHECK(W, "linear_bwd: dx needs W");
        if (linear_dx(dy, W, dx, M, N, K, nullptr, 1.f, nullptr, compute, st, scratch, scratch ? egx_linear_bwd_scratch(M, N, K) : 0)) return 1;
    }
    return 0;
}

int egx_gemm(int layout, const float* A, const float* Bm, float* C, int M, int N, int K, const float* bias, int relu,
             int compute, void* scratch, size_t scratch_bytes, void* stream) {
    EGX_CHECK(A && Bm && C, "null pointer argument");
    GemmParams g;
    g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K;
    g.lda = (layout == 2) ? M : K;
    g.ldb = (layout == 0) ? K : N;
    g.ldc = N;
    g.bias = bias; g.relu = relu;
    return gemm(layout, g, compute, 0, scratch, scratch_bytes, (hipStream_t)stream);
}

int egx_layernorm_fwd(const float* x, const float* res, const float* w, const float* b, float eps, float* pre,
                      float* stats, float* y, int rows, int d, void* stream) {
    EGX_CHECK(x && w && b && y, "null pointer argument");
    LnFwdParams p;
    p.x = x; p.res = res; p.w = w; p.b = b; p.eps = eps; p.pre = pre; p.stats = stats; p.y = y; p.rows = rows; p.d = d;
    return layernorm_fwd(p, (hipStream_t)stream);
}

int egx_layernorm_bwd(const float* dy, const float* pre, const float* stats, const float* w, float* dx, float* dw,
                      float* db, int rows, int d, void* stream) {
    EGX_CHECK(dy && pre && stats && w && dx, "null pointer argument");
    LnBwdParams p;
    p.dy = dy; p.pre = pre; p.stats = stats; p.w = w; p.dx = dx; p.dw = dw; p.db = db; p.rows = rows; p.d = d;
    return layernorm_bwd(p, (hipStream_t)stream);
}

int egx_attention_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int d, float p_drop, uint64_t seed,
                      void* stream) {
    EGX_CHECK(qkv && out && lse, "null pointer argument");
    Drop da = make_drop(p_drop > 0.f, p_drop, seed, 0, SITE_ATTN);
    return attention_fwd(qkv, out, lse, B, S, H, d, da.key, da.thresh, da.inv_keep, (hipStream_t)stream);
}

int egx_attention_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, float* d_qkv, int B,
                      int S, int H, int d, float p_drop, uint64_t seed, void* stream) {
    EGX_CHECK(qkv && out && lse && d_out && d_qkv, "null pointer argument");
    Drop da = make_drop(p_drop > 0.f, p_drop, seed, 0, SITE_ATTN);
    return attention_bwd(qkv, out, lse, d_out, d_qkv, B, S, H, d, da.key, da.thresh, da.inv_keep, (hipStream_t)stream);
}

int egx_weighted_ce(const float* logits, const int64_t* target, const float* weight, int B, int C, float* loss,
                    float* d_logits, void* stream) {
    return weighted_ce(logits, target, weight, B, C, loss, d_logits, (hipStream_t)stream);
}

size_t egx_linear_ce_scratch(int M, int K, int C) { return linear_ce_scratch_bytes(M, K, C); }
int egx_linear_ce_fwd(const float* x, const float* W, const float* b, const int64_t* target, const float* weight, int M, int K,
                      int C, float* logits, float* probs, float* d_logits, float* loss, float* correct, float* pred_label,
                      void* scratch, void* stream) {
    return linear_ce_fwd(x, W, b, target, weight, M, K, C, logits, probs, d_logits, loss, correct, pred_label, scratch, (hipStream_t)stream);
}
int egx_linear_ce_bwd(const float* x, const float* W, const float* d_logits, const float* grad_scale, int M, int K, int C,
                      float* dx, float* dW, float* db, void* scratch, void* stream) {
    return linear_ce_bwd(x, W, d_logits, grad_scale, M, K, C, dx, dW, db, scratch, (hipStream_t)stream);
}

int egx_counter_add(int64_t* counter, int64_t inc, void* stream) { return counter_add(counter, inc, (hipStream_t)stream); }

int egx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, const int64_t* step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                  void* stream) {
    EGX_CHECK(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adam_step: invalid hyper-parameters");
    return adam_step(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale,
                     (hipStream_t)stream);
}

// ---- EgoT2-g sequence decoder pieces (decoder.hip) ---------------------------------------------------------------
static SmallAttnParams small_attn_params(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int B, int Sq,
                                         int Sk, int H, int dh, int causal, float p_drop, uint64_t seed, uint32_t site) {
    SmallAttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
    p.B = B; p.Sq = Sq; p.Sk = Sk; p.H = H; p.dh = dh; p.causal = causal;
    Drop dr = make_drop(p_drop > 0.f, p_drop, seed, site >> 8, site & 0xffu);
    p.drop_key = dr.key; p.drop_thresh = dr.thresh; p.drop_inv = dr.inv_keep;
    return p;
}

int egx_small_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, int B,
                            int Sq, int Sk, int H, int dh, int causal, float p_drop, uint64_t seed, uint32_t site, void* stream) {
    SmallAttnParams p = small_attn_params(q, ldq, k, ldk, v, ldv, B, Sq, Sk, H, dh, causal, p_drop, seed, site);
    p.o = o; p.ldo = ldo;
    return small_attention_fwd(p, (hipStream_t)stream);
}

int egx_small_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* d_o, int ldo,
                            float* dq, float* dk, float* dv, int B, int Sq, int Sk, int H, int dh, int causal, float p_drop,
                            uint64_t seed, uint32_t site, void* stream) {
    SmallAttnParams p = small_attn_params(q, ldq, k, ldk, v, ldv, B, Sq, Sk, H, dh, causal, p_drop, seed, site);
    p.d_o = d_o; p.ldo = ldo; p.dq = dq; p.dk = dk; p.dv = dv;
    return small_attention_bwd(p, (hipStream_t)stream);
}

int egx_embed_pos_fwd(const int64_t* tokens, const float* emb, const float* pe, int pe_stride, float scale, float* out, int B,
                      int sy, int d, int V, float p_drop, uint64_t seed, void* stream) {
    Drop dr = make_drop(p_drop > 0.f, p_drop, seed, 0xDECu, SITE_POS);
    return embed_pos_fwd(tokens, emb, pe, pe_stride, scale, out, B, sy, d, V, dr.key, dr.thresh, dr.inv_keep, (hipStream_t)stream);
}

int egx_embed_pos_bwd(const int64_t* tokens, const float* dy, float* d_emb, float scale, int B, int sy, int d, int V, float p_drop,
                      uint64_t seed, void* stream) {
    Drop dr = make_drop(p_drop > 0.f, p_drop, seed, 0xDECu, SITE_POS);
    return embed_pos_bwd(tokens, dy, d_emb, scale, B, sy, d, V, dr.key, dr.thresh, dr.inv_keep, (hipStream_t)stream);
}

int egx_relu_mask(float* dy, const float* y, size_t n, void* stream) { return relu_mask(dy, y, n, (hipStream_t)stream); }

int egx_dropout(float* x, int rows, int cols, float p_drop, uint64_t seed, uint32_t site, void* stream) {
    EGX_CHECK(x || rows * cols == 0, "egx_dropout: null pointer");
    Drop dr = make_drop(p_drop > 0.f, p_drop, seed, site >> 8, site & 0xffu);
    return apply_dropout_mask(x, rows, cols, dr.key, dr.thresh, dr.inv_keep, (hipStream_t)stream);
}

}  // extern "C"
