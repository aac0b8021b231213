// Host orchestration of the wide bf16 path (see wide.h): token preparation + L post-LN encoder layers over all B*S tokens
// as large bf16 MFMA GEMMs with fused epilogues, MFMA attention per (clip, head), row kernels with bf16 side outputs.
// Same call sites as the generic path (include/egot2x.h: egx_encoder_fwd / egx_encoder_bwd); only enqueues kernels.
#include <string.h>

#include "../../include/egot2x.h"
#include "common.h"
#include "kernels.h"
#include "wide.h"
#include "wide_host.h"

namespace egx {

namespace {

struct WLayer {
    size_t w_in, w_in_t, w_o, w_o_t, w1, w1_t, w2, w2_t;                     // bf16 weight copies
    size_t qkv, lse, attn, res1, stats1, x1_32, x1_16, hid, res2, stats2;    // saved activations
    size_t x16;                                                               // bf16 copy of the layer input
    size_t x32;                                                               // fp32 layer input (layer 0: x0; else previous output)
};
struct WPlan {
    int B, S, d, H, dff, L, nseg;
    size_t N;
    int seg_off[EGX_MAX_SEGMENTS];
    size_t keys;        // dropout keys derived from a device-resident seed (derive_keys): (max(L, nseg) layers) x 8 slots
    size_t zero, seg_w[EGX_MAX_SEGMENTS], seg_feat16[EGX_MAX_SEGMENTS], seg_pre[EGX_MAX_SEGMENTS], seg_stats[EGX_MAX_SEGMENTS];
    WLayer layer[64];
    size_t saved_bytes;
    // scratch
    size_t gA, gB, dres, dy16, dhid16, dattn16, dqkv16, adelta, dseg16, slabs, slab_all, slab_all_bytes, lnpart, cspart, rowpart, rowpart_bytes, scratch_bytes;
};

size_t take(size_t& cur, size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; }
size_t smax(size_t a, size_t b) { return a > b ? a : b; }

void make_wplan(const egx_config* cfg, const egx_segment* segs, int B, WPlan& pl) {
    memset(&pl, 0, sizeof(pl));
    pl.B = B; pl.d = cfg->d_model; pl.H = cfg->n_heads; pl.dff = cfg->d_ff; pl.L = cfg->n_layers; pl.nseg = cfg->n_segments;
    int S = 0;
    for (int i = 0; i < pl.nseg; ++i) { pl.seg_off[i] = S; S += segs[i].T; }
    pl.S = S; pl.N = (size_t)B * S;
    const size_t d = pl.d, N = pl.N, dff = pl.dff;
    size_t cur = 0;
    pl.zero = take(cur, 1024);
    pl.keys = take(cur, (size_t)64 * DROP_KEY_SLOTS * sizeof(uint64_t));
    for (int i = 0; i < pl.nseg; ++i) {
        size_t rows = (size_t)B * segs[i].T;
        if (segs[i].proj_w) {
            pl.seg_w[i] = take(cur, d * segs[i].d_in * 2);
            // bf16 copy of the features (the projection GEMM's operand); features that arrive in bf16 and un-pooled are used in place
            if (!(segs[i].feat_bf16 && segs[i].pool <= 1)) pl.seg_feat16[i] = take(cur, rows * segs[i].d_in * 2);
            pl.seg_pre[i] = take(cur, rows * d * 4);
        }
        pl.seg_stats[i] = take(cur, rows * 2 * 4);
    }
    size_t x0_32 = take(cur, N * d * 4);
    for (int l = 0; l < pl.L; ++l) {
        WLayer& o = pl.layer[l];
        o.w_in = take(cur, 3 * d * d * 2); o.w_in_t = take(cur, 3 * d * d * 2);
        o.w_o = take(cur, d * d * 2); o.w_o_t = take(cur, d * d * 2);
        o.w1 = take(cur, dff * d * 2); o.w1_t = take(cur, dff * d * 2);
        o.w2 = take(cur, dff * d * 2); o.w2_t = take(cur, dff * d * 2);
        o.x32 = l == 0 ? x0_32 : take(cur, N * d * 4);
        o.x16 = take(cur, N * d * 2);
        o.qkv = take(cur, N * 3 * d * 2);
        o.lse = take(cur, (size_t)B * pl.H * S * 4);
        o.attn = take(cur, N * d * 2);
        o.res1 = take(cur, N * d * 4);
        o.stats1 = take(cur, N * 2 * 4);
        o.x1_32 = take(cur, N * d * 4);
        o.x1_16 = take(cur, N * d * 2);
        o.hid = take(cur, N * dff * 2);
        o.res2 = take(cur, N * d * 4);
        o.stats2 = take(cur, N * 2 * 4);
    }
    pl.saved_bytes = cur;

    size_t sc = 0;
    pl.gA = take(sc, N * d * 4);
    pl.gB = take(sc, N * d * 4);
    pl.dres = take(sc, N * d * 4);
    pl.dy16 = take(sc, N * d * 2);
    pl.dhid16 = take(sc, N * dff * 2);
    pl.dattn16 = take(sc, N * d * 2);
    pl.dqkv16 = take(sc, N * 3 * d * 2);
    pl.adelta = take(sc, wide_attn_delta_bytes(B, pl.H, S));
    size_t segrows = 0;
    for (int i = 0; i < pl.nseg; ++i) segrows = smax(segrows, (size_t)B * segs[i].T);
    pl.dseg16 = take(sc, segrows * d * 2);
    size_t slab = 0;
    slab = smax(slab, wide_gemm_tn_scratch(3 * pl.d, pl.d, (int)N));
    slab = smax(slab, wide_gemm_tn_scratch(pl.d, pl.d, (int)N));
    slab = smax(slab, wide_gemm_tn_scratch(pl.dff, pl.d, (int)N));
    slab = smax(slab, wide_gemm_tn_scratch(pl.d, pl.dff, (int)N));
    for (int i = 0; i < pl.nseg; ++i)
        if (segs[i].proj_w) slab = smax(slab, wide_gemm_tn_scratch(pl.d, segs[i].d_in, B * segs[i].T));
    pl.slabs = take(sc, slab);
    // one slab region per weight gradient of a backward (their reductions run as ONE launch at its end)
    size_t all = 0;
    auto add = [&](int M, int Nn, int K) { all += (wide_gemm_tn_scratch(M, Nn, K) + 255) / 256 * 256; };
    for (int l = 0; l < pl.L; ++l) { add(pl.d, pl.dff, (int)N); add(pl.dff, pl.d, (int)N); add(pl.d, pl.d, (int)N); add(3 * pl.d, pl.d, (int)N); }
    for (int i = 0; i < pl.nseg; ++i)
        if (segs[i].proj_w) add(pl.d, segs[i].d_in, B * segs[i].T);
    if (all > ((size_t)400 << 20)) all = 0;      // measured break-even (wide_encoder_bwd): 283 MB pays, 764 MB does not
    pl.slab_all = take(sc, all);
    pl.slab_all_bytes = all;
    pl.lnpart = take(sc, wide_ln_bwd_scratch((int)N, pl.d));
    size_t cs = smax(wide_colsum_scratch((int)N, 3 * pl.d), (size_t)(4 * cdiv((int)N, 256) + 4) * pl.dff * 4);
    for (int i = 0; i < pl.nseg; ++i) cs = smax(cs, wide_pos_grad_scratch(B, segs[i].T, pl.d));
    pl.cspart = take(sc, cs);
    // a partial buffer per deferred row reduction of a backward (wide_row_reduce_flush: ONE launch sums them all at its end):
    // two LayerNorm backwards + the lin1 / in-projection bias column sums per layer
    {
        const size_t ln = (wide_ln_bwd_scratch((int)N, pl.d) + 255) / 256 * 256;
        const size_t c1 = ((size_t)(4 * cdiv((int)N, 256) + 4) * pl.dff * 4 + 255) / 256 * 256, c2 = (wide_colsum_scratch((int)N, 3 * pl.d) + 255) / 256 * 256;
        size_t all_r = (size_t)pl.L * (2 * ln + c1 + c2);
        if (all_r > ((size_t)256 << 20)) all_r = 0;
        pl.rowpart = take(sc, all_r);
        pl.rowpart_bytes = all_r;
    }
    pl.scratch_bytes = sc;
}

struct Drop { uint64_t key = 0; uint32_t thresh = 0; float inv = 1.f; };
// g_keys != null (device-resident seed): the key is the ADDRESS of its slot in the table derive_keys() filled on the stream
thread_local const uint64_t* g_keys = nullptr;
Drop mkdrop(int training, float p, uint64_t seed, uint32_t layer, uint32_t site) {
    Drop dr;
    if (training && p > 0.f) {
        dr.key = g_keys ? key_slot(g_keys, layer, site) : site_key(seed, layer, site);
        dr.thresh = drop_threshold(p); dr.inv = p < 1.f ? 1.f / (1.f - p) : 0.f;
    }
    return dr;
}
struct KeyScope {       // sets the table for the mkdrop calls of one forward / backward
    KeyScope(const uint64_t* t) { g_keys = t; }
    ~KeyScope() { g_keys = nullptr; }
};

template <class T> T* at(void* base, size_t off) { return reinterpret_cast<T*>((char*)base + off); }
template <class T> const T* cat(const void* base, size_t off) { return reinterpret_cast<const T*>((const char*)base + off); }

}  // namespace

bool wide_ok(const egx_config* cfg, const egx_segment* segs, int B) {
    (void)B;
    if (cfg->compute != EGX_BF16) return false;
    const int d = cfg->d_model, dff = cfg->d_ff;
    // impl = auto keeps d_model = 128 on the per-clip / fp32-storage kernels (the wide path's extra bf16 roundings sit at the
    // 1e-2 bound there); impl = wide may force it
    if ((d < 256 && cfg->impl != EGX_IMPL_WIDE) || d % 128 != 0 || dff % 128 != 0 || d > 1024 || cfg->n_layers < 1 || cfg->n_layers > 64) return false;
    if (cfg->n_heads <= 0 || d % cfg->n_heads != 0) return false;
    int S = 0;
    for (int i = 0; i < cfg->n_segments; ++i) {
        if (segs[i].proj_w ? (segs[i].d_in % 128 != 0) : (segs[i].d_in != d)) return false;
        S += segs[i].T;
    }
    return wide_attn_supported(S, d / cfg->n_heads);
}

void wide_workspace(const egx_config* cfg, const egx_segment* segs, int B, size_t* saved, size_t* scratch) {
    WPlan pl;
    make_wplan(cfg, segs, B, pl);
    *saved = pl.saved_bytes;
    *scratch = pl.scratch_bytes;
}

int wide_encoder_fwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b, const egx_layer* layers,
                     int B, float* tokens_out, void* saved, int training, uint64_t seed, hipStream_t st) {
    WPlan pl;
    make_wplan(cfg, segs, B, pl);
    const int d = pl.d, S = pl.S, dff = pl.dff, N = (int)pl.N;
    // device-resident seed: one small launch advances it (training forward with advance_seed) and derives every key of this
    // call into `saved`; the kernels below read their keys from there, and so does the backward
    const bool dev_keys = cfg->seed_ptr && training && (cfg->p_drop > 0.f || cfg->p_pos > 0.f || cfg->p_feat > 0.f);
    if (dev_keys && derive_keys(const_cast<uint64_t*>(cfg->seed_ptr), at<uint64_t>(saved, pl.keys), 0, 64, cfg->advance_seed, st)) return 1;
    KeyScope key_scope(dev_keys ? cat<uint64_t>(saved, pl.keys) : nullptr);
    EGX_HIP(hipMemsetAsync(at<char>(saved, pl.zero), 0, 1024, st));
    const void* zero = at<char>(saved, pl.zero);

    // token preparation
    float* x0 = at<float>(saved, pl.layer[0].x32);
    bf16_t* x0_16 = at<bf16_t>(saved, pl.layer[0].x16);
    {   // every weight of this forward -> bf16 (W for the forward, W^T for the input gradients), one launch
        WideCastBatch cb;
        for (int i = 0; i < pl.nseg; ++i)
            if (segs[i].proj_w && wide_cast_add(cb, segs[i].proj_w, d, segs[i].d_in, segs[i].d_in, at<bf16_t>(saved, pl.seg_w[i]), nullptr, st)) return 1;
        for (int l = 0; l < pl.L; ++l) {
            const WLayer& o = pl.layer[l];
            const egx_layer& w = layers[l];
            if (wide_cast_add(cb, w.in_proj_w, 3 * d, d, d, at<bf16_t>(saved, o.w_in), at<bf16_t>(saved, o.w_in_t), st)) return 1;
            if (wide_cast_add(cb, w.out_proj_w, d, d, d, at<bf16_t>(saved, o.w_o), at<bf16_t>(saved, o.w_o_t), st)) return 1;
            if (wide_cast_add(cb, w.lin1_w, dff, d, d, at<bf16_t>(saved, o.w1), at<bf16_t>(saved, o.w1_t), st)) return 1;
            if (wide_cast_add(cb, w.lin2_w, d, dff, dff, at<bf16_t>(saved, o.w2), at<bf16_t>(saved, o.w2_t), st)) return 1;
        }
        if (wide_cast_flush(cb, st)) return 1;
    }
    for (int i = 0; i < pl.nseg; ++i) {
        const egx_segment& sg = segs[i];
        const int rows = B * sg.T;
        const float* pre = sg.feat;
        if (sg.proj_w) {
            bf16_t* w16 = at<bf16_t>(saved, pl.seg_w[i]);
            const bool in_place = sg.feat_bf16 && sg.pool <= 1;
            const bf16_t* f16 = in_place ? reinterpret_cast<const bf16_t*>(sg.feat) : at<bf16_t>(saved, pl.seg_feat16[i]);
            float* po = at<float>(saved, pl.seg_pre[i]);
            if (!in_place && wide_pool_cast(sg.feat, sg.feat_bf16, rows, sg.pool > 1 ? sg.pool : 1, sg.d_in, at<bf16_t>(saved, pl.seg_feat16[i]), st)) return 1;
            WideGemmParams g;
            g.A = f16; g.B = w16; g.M = rows; g.N = d; g.K = sg.d_in; g.lda = sg.d_in; g.ldb = sg.d_in;
            g.Cf = po; g.ldc = d; g.bias = sg.proj_b; g.zero_page = zero;
            Drop df = mkdrop(training, cfg->p_feat, seed, (uint32_t)i, SITE_FEAT);
            g.drop_key = df.key; g.drop_thresh = df.thresh; g.drop_inv = df.inv;
            if (wide_gemm_nt(g, st)) return 1;
            pre = po;
        }
        WideLnFwdParams lp;
        lp.x = pre; lp.w = ln_w; lp.b = ln_b; lp.eps = cfg->ln_eps;
        lp.stats = at<float>(saved, pl.seg_stats[i]);
        lp.y32 = x0; lp.y16 = x0_16; lp.rows = rows; lp.d = d; lp.T = sg.T; lp.S = S; lp.off = pl.seg_off[i];
        lp.add_vec = sg.add_vec; lp.pos = sg.pos; lp.pos_stride = sg.pos_stride;
        Drop dp = mkdrop(training, cfg->p_pos, seed, 0, SITE_POS);
        lp.drop_key = dp.key; lp.drop_thresh = dp.thresh; lp.drop_inv = dp.inv;
        if (wide_ln_fwd(lp, st)) return 1;
    }

    for (int l = 0; l < pl.L; ++l) {
        const WLayer& o = pl.layer[l];
        const egx_layer& w = layers[l];
        bf16_t* w_in = at<bf16_t>(saved, o.w_in); bf16_t* w_o = at<bf16_t>(saved, o.w_o);
        bf16_t* w1 = at<bf16_t>(saved, o.w1); bf16_t* w2 = at<bf16_t>(saved, o.w2);
        const float* x32 = cat<float>(saved, o.x32);
        const bf16_t* x16 = cat<bf16_t>(saved, o.x16);
        const bool last = l + 1 == pl.L;
        float* xo32 = last ? tokens_out : at<float>(saved, pl.layer[l + 1].x32);
        bf16_t* xo16 = last ? nullptr : at<bf16_t>(saved, pl.layer[l + 1].x16);
        {   // packed in-projection
            WideGemmParams g;
            g.A = x16; g.B = w_in; g.M = N; g.N = 3 * d; g.K = d; g.lda = d; g.ldb = d;
            g.Cb = at<bf16_t>(saved, o.qkv); g.ldc = 3 * d; g.bias = w.in_proj_b; g.zero_page = zero;
            if (wide_gemm_nt(g, st)) return 1;
        }
        {
            WideAttnParams a;
            a.qkv = cat<bf16_t>(saved, o.qkv); a.out = at<bf16_t>(saved, o.attn); a.lse = at<float>(saved, o.lse);
            a.B = B; a.S = S; a.H = pl.H; a.d = d;
            Drop da = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_ATTN);
            a.drop_key = da.key; a.drop_thresh = da.thresh; a.drop_inv = da.inv;
            if (wide_attn_fwd(a, st)) return 1;
        }
        {   // out-projection + dropout1 + residual -> res1
            WideGemmParams g;
            g.A = cat<bf16_t>(saved, o.attn); g.B = w_o; g.M = N; g.N = d; g.K = d; g.lda = d; g.ldb = d;
            g.Cf = at<float>(saved, o.res1); g.ldc = d; g.bias = w.out_proj_b; g.residual = x32; g.ldr = d; g.zero_page = zero;
            Drop d1 = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES1);
            g.drop_key = d1.key; g.drop_thresh = d1.thresh; g.drop_inv = d1.inv;
            if (wide_gemm_nt(g, st)) return 1;
        }
        {
            WideLnFwdParams lp;
            lp.x = cat<float>(saved, o.res1); lp.w = w.norm1_w; lp.b = w.norm1_b; lp.eps = cfg->ln_eps;
            lp.stats = at<float>(saved, o.stats1); lp.y32 = at<float>(saved, o.x1_32); lp.y16 = at<bf16_t>(saved, o.x1_16);
            lp.rows = N; lp.d = d;
            if (wide_ln_fwd(lp, st)) return 1;
        }
        {   // linear1 + ReLU + dropout -> hidden (bf16)
            WideGemmParams g;
            g.A = cat<bf16_t>(saved, o.x1_16); g.B = w1; g.M = N; g.N = dff; g.K = d; g.lda = d; g.ldb = d;
            g.Cb = at<bf16_t>(saved, o.hid); g.ldc = dff; g.bias = w.lin1_b; g.relu = 1; g.zero_page = zero;
            Drop dh = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_FFN);
            g.drop_key = dh.key; g.drop_thresh = dh.thresh; g.drop_inv = dh.inv;
            if (wide_gemm_nt(g, st)) return 1;
        }
        {   // linear2 + dropout2 + residual -> res2
            WideGemmParams g;
            g.A = cat<bf16_t>(saved, o.hid); g.B = w2; g.M = N; g.N = d; g.K = dff; g.lda = dff; g.ldb = dff;
            g.Cf = at<float>(saved, o.res2); g.ldc = d; g.bias = w.lin2_b; g.residual = cat<float>(saved, o.x1_32); g.ldr = d;
            g.zero_page = zero;
            Drop d2 = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES2);
            g.drop_key = d2.key; g.drop_thresh = d2.thresh; g.drop_inv = d2.inv;
            if (wide_gemm_nt(g, st)) return 1;
        }
        {
            WideLnFwdParams lp;
            lp.x = cat<float>(saved, o.res2); lp.w = w.norm2_w; lp.b = w.norm2_b; lp.eps = cfg->ln_eps;
            lp.stats = at<float>(saved, o.stats2); lp.y32 = xo32; lp.y16 = xo16; lp.rows = N; lp.d = d;
            if (wide_ln_fwd(lp, st)) return 1;
        }
    }
    return 0;
}

int wide_encoder_bwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const egx_layer* layers, int B,
                     const float* d_tokens, const void* saved, void* scratch, const egx_segment_grads* seg_grads, float* d_ln_w,
                     float* d_ln_b, const egx_layer_grads* layer_grads, int training, uint64_t seed, hipStream_t st) {
    WPlan pl;
    make_wplan(cfg, segs, B, pl);
    const int d = pl.d, S = pl.S, dff = pl.dff, N = (int)pl.N;
    const void* zero = cat<char>(saved, pl.zero);
    // device-resident seed: the forward left this step's keys in `saved`
    const bool dev_keys = cfg->seed_ptr && training && (cfg->p_drop > 0.f || cfg->p_pos > 0.f || cfg->p_feat > 0.f);
    KeyScope key_scope(dev_keys ? cat<uint64_t>(saved, pl.keys) : nullptr);
    if (cfg->zero_buf && cfg->zero_bytes) EGX_HIP(hipMemsetAsync(cfg->zero_buf, 0, cfg->zero_bytes, st));
    float* gA = at<float>(scratch, pl.gA);
    float* gB = at<float>(scratch, pl.gB);
    float* dres = at<float>(scratch, pl.dres);
    bf16_t* dy16 = at<bf16_t>(scratch, pl.dy16);
    bf16_t* dhid16 = at<bf16_t>(scratch, pl.dhid16);
    bf16_t* dattn16 = at<bf16_t>(scratch, pl.dattn16);
    bf16_t* dqkv16 = at<bf16_t>(scratch, pl.dqkv16);
    void* slabs = at<char>(scratch, pl.slabs);
    void* lnpart = at<char>(scratch, pl.lnpart);
    float* cspart = at<float>(scratch, pl.cspart);
    const float* g = d_tokens;

    // weight gradients: when the split-K slabs of the whole backward are small (the d = 256 EgoT2-g encoder: 283 MB, 15
    // reductions of 8 us in 1.6 ms of work) every gradient gets a slab region of its own and ALL slab reductions run as one
    // launch at the end (-2.5 %); for larger problems that reduction reads its slabs back from HBM instead of the Infinity
    // Cache (d = 512: 764 MB, +1.5 %; C4: 1.1 GB, +0.8 %), so they keep the reduction right behind each GEMM.
    WideReduceBatch rb;
    size_t slab_cur = 0;
    const bool defer_ok = pl.slab_all_bytes > 0 && !cfg->bucket_cb;     // bucketed exchange: a layer's gradients must be final when its bucket is announced
    // second stages of the two-stage column sums (LayerNorm / bias gradients): queued, one launch at the end (wide.h WideRowReduceBatch);
    // each gets a partial buffer of its own out of `rowpart` (none left, or a bucketed exchange: the shared buffer and its own launch)
    WideRowReduceBatch rrb;
    size_t row_cur = 0;
    static int row_env = -2;
    if (row_env == -2) { const char* e = getenv("EGX_ROW_DEFER"); row_env = e ? atoi(e) : 1; }      // 0: every second stage as its own launch (A/B aid)
    const bool row_defer = row_env != 0 && pl.rowpart_bytes > 0 && !cfg->bucket_cb;
    auto row_region = [&](size_t need) -> void* {
        need = (need + 255) / 256 * 256;
        if (!row_defer || row_cur + need > pl.rowpart_bytes) return nullptr;
        void* r = at<char>(scratch, pl.rowpart) + row_cur;
        row_cur += need;
        return r;
    };
    auto ln_bwd_q = [&](WideLnBwdParams& b) -> int {
        void* reg = row_region(wide_ln_bwd_scratch(b.rows, b.d));
        return reg ? wide_ln_bwd(b, reg, st, &rrb) : wide_ln_bwd(b, lnpart, st);
    };
    auto dw_tn = [&](const bf16_t* dy, int ldy, const bf16_t* x, int ldx, float* dW, int n_out, int k_in, int tokens) -> int {
        if (!dW) return 0;
        WideGemmParams t;
        t.A = dy; t.B = x; t.M = n_out; t.N = k_in; t.K = tokens; t.lda = ldy; t.ldb = ldx;
        t.Cf = dW; t.ldc = k_in; t.accumulate = 1; t.zero_page = zero;
        const size_t need = (wide_gemm_tn_scratch(n_out, k_in, tokens) + 255) / 256 * 256;
        if (!defer_ok || slab_cur + need > pl.slab_all_bytes) return wide_gemm_tn(t, slabs, st);
        void* region = at<char>(scratch, pl.slab_all) + slab_cur;
        slab_cur += need;
        return wide_gemm_tn(t, region, st, &rb);
    };

    for (int l = pl.L - 1; l >= 0; --l) {
        const WLayer& o = pl.layer[l];
        const egx_layer& w = layers[l];
        const egx_layer_grads& gw = layer_grads[l];
        {   // LayerNorm2 backward: dres = d(res2), dy16 = dropout2-mask .* d(res2)
            WideLnBwdParams b;
            b.dy = g; b.pre = cat<float>(saved, o.res2); b.stats = cat<float>(saved, o.stats2); b.w = w.norm2_w;
            b.dx32 = dres; b.dx16 = dy16; b.rows = N; b.d = d;
            Drop d2 = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES2);
            b.out_key = d2.key; b.out_thresh = d2.thresh; b.out_inv = d2.inv;
            b.dw = gw.norm2_w; b.db = gw.norm2_b; b.dbias = gw.lin2_b;
            if (ln_bwd_q(b)) return 1;
        }
        if (dw_tn(dy16, d, cat<bf16_t>(saved, o.hid), dff, gw.lin2_w, d, dff, N)) return 1;
        {   // d(hidden) = (dy W2) .* alive / keep, column sums -> d(lin1_b)
            WideGemmParams q;
            q.A = dy16; q.B = cat<bf16_t>(saved, o.w2_t); q.M = N; q.N = dff; q.K = d; q.lda = d; q.ldb = d;
            q.Cb = dhid16; q.ldc = dff; q.mask = cat<bf16_t>(saved, o.hid); q.ldm = dff; q.zero_page = zero;
            q.mask_scale = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_FFN).inv;
            float* cs_reg = gw.lin1_b ? (float*)row_region((size_t)wide_gemm_nt_colsum_rows(N, dff) * dff * 4) : nullptr;
            q.colsum = gw.lin1_b ? (cs_reg ? cs_reg : cspart) : nullptr;
            if (wide_gemm_nt(q, st)) return 1;
            if (gw.lin1_b && wide_reduce_rows(q.colsum, wide_gemm_nt_colsum_rows(N, dff), dff, gw.lin1_b, st, cs_reg ? &rrb : nullptr)) return 1;
        }
        if (dw_tn(dhid16, dff, cat<bf16_t>(saved, o.x1_16), d, gw.lin1_w, dff, d, N)) return 1;
        float* g1 = (g == gA) ? gB : gA;
        {   // d(x1) = d(hidden) W1 + d(res2)
            WideGemmParams q;
            q.A = dhid16; q.B = cat<bf16_t>(saved, o.w1_t); q.M = N; q.N = d; q.K = dff; q.lda = dff; q.ldb = dff;
            q.Cf = g1; q.ldc = d; q.residual = dres; q.ldr = d; q.zero_page = zero;
            if (wide_gemm_nt(q, st)) return 1;
        }
        {   // LayerNorm1 backward: dres = d(res1), dy16 = dropout1-mask .* d(res1)
            WideLnBwdParams b;
            b.dy = g1; b.pre = cat<float>(saved, o.res1); b.stats = cat<float>(saved, o.stats1); b.w = w.norm1_w;
            b.dx32 = dres; b.dx16 = dy16; b.rows = N; b.d = d;
            Drop d1 = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_RES1);
            b.out_key = d1.key; b.out_thresh = d1.thresh; b.out_inv = d1.inv;
            b.dw = gw.norm1_w; b.db = gw.norm1_b; b.dbias = gw.out_proj_b;
            if (ln_bwd_q(b)) return 1;
        }
        if (dw_tn(dy16, d, cat<bf16_t>(saved, o.attn), d, gw.out_proj_w, d, d, N)) return 1;
        {   // d(attention output) = dy W_o
            WideGemmParams q;
            q.A = dy16; q.B = cat<bf16_t>(saved, o.w_o_t); q.M = N; q.N = d; q.K = d; q.lda = d; q.ldb = d;
            q.Cb = dattn16; q.ldc = d; q.zero_page = zero;
            if (wide_gemm_nt(q, st)) return 1;
        }
        {
            WideAttnParams a;
            a.qkv = cat<bf16_t>(saved, o.qkv); a.lse = const_cast<float*>(cat<float>(saved, o.lse));
            a.out = const_cast<bf16_t*>(cat<bf16_t>(saved, o.attn)); a.delta = at<float>(scratch, pl.adelta);
            a.d_out = dattn16; a.d_qkv = dqkv16; a.B = B; a.S = S; a.H = pl.H; a.d = d;
            Drop da = mkdrop(training, cfg->p_drop, seed, (uint32_t)l, SITE_ATTN);
            a.drop_key = da.key; a.drop_thresh = da.thresh; a.drop_inv = da.inv;
            if (wide_attn_bwd(a, st)) return 1;
        }
        if (gw.in_proj_b) {
            void* reg = row_region(wide_colsum_scratch(N, 3 * d));
            if (wide_colsum_bf16(dqkv16, N, 3 * d, 3 * d, gw.in_proj_b, reg ? reg : (void*)cspart, st, reg ? &rrb : nullptr)) return 1;
        }
        if (dw_tn(dqkv16, 3 * d, cat<bf16_t>(saved, o.x16), d, gw.in_proj_w, 3 * d, d, N)) return 1;
        float* g0 = (g1 == gA) ? gB : gA;
        {   // d(layer input) = dqkv W_in + d(res1)
            WideGemmParams q;
            q.A = dqkv16; q.B = cat<bf16_t>(saved, o.w_in_t); q.M = N; q.N = d; q.K = 3 * d; q.lda = 3 * d; q.ldb = 3 * d;
            q.Cf = g0; q.ldc = d; q.residual = dres; q.ldr = d; q.zero_page = zero;
            if (wide_gemm_nt(q, st)) return 1;
        }
        g = g0;
        if (cfg->bucket_cb) cfg->bucket_cb(cfg->bucket_user, pl.L - 1 - l);     // every gradient of layer l is enqueued
    }

    // token preparation backward
    Drop dp = mkdrop(training, cfg->p_pos, seed, 0, SITE_POS);
    bf16_t* dseg16 = at<bf16_t>(scratch, pl.dseg16);
    for (int i = 0; i < pl.nseg; ++i) {
        const egx_segment& sg = segs[i];
        egx_segment_grads sgr;
        memset(&sgr, 0, sizeof(sgr));
        if (seg_grads) sgr = seg_grads[i];
        // d(features): an identity segment's feature gradient IS the LayerNorm backward's fp32 input gradient (the trainable
        // SlowFast head of the LTA translators feeds such a segment); projected features stay frozen on this path
        EGX_CHECK(!sgr.feat || !sg.proj_w, "wide path: gradients into PROJECTED features are not supported (use impl = generic)");
        const int rows = B * sg.T;
        if (sgr.pos && wide_pos_grad(g, B, S, pl.seg_off[i], sg.T, d, sgr.pos, sg.pos_stride, dp.key, dp.thresh, dp.inv, cspart, st)) return 1;
        const bool need = (sg.proj_w && (sgr.proj_w || sgr.proj_b)) || d_ln_w || d_ln_b || sgr.add_vec || sgr.feat;
        if (!need) continue;
        WideLnBwdParams b;
        b.dy = g; b.pre = sg.proj_w ? cat<float>(saved, pl.seg_pre[i]) : sg.feat; b.stats = cat<float>(saved, pl.seg_stats[i]);
        b.w = ln_w; b.dx16 = sg.proj_w ? dseg16 : nullptr; b.dx32 = sg.proj_w ? nullptr : sgr.feat; b.rows = rows; b.d = d; b.T = sg.T; b.S = S; b.off = pl.seg_off[i];
        b.drop_key = dp.key; b.drop_thresh = dp.thresh; b.drop_inv = dp.inv;
        if (sg.proj_w) {
            Drop df = mkdrop(training, cfg->p_feat, seed, (uint32_t)i, SITE_FEAT);
            b.out_key = df.key; b.out_thresh = df.thresh; b.out_inv = df.inv;
        }
        b.dw = d_ln_w; b.db = d_ln_b; b.dadd = sgr.add_vec; b.dbias = sg.proj_w ? sgr.proj_b : nullptr;
        // (not queued: every segment adds into the SAME shared-LayerNorm gradient; queued reductions run concurrently and must have targets of their own)
        if (wide_ln_bwd(b, lnpart, st)) return 1;
        const bf16_t* f16 = (sg.feat_bf16 && sg.pool <= 1) ? reinterpret_cast<const bf16_t*>(sg.feat) : cat<bf16_t>(saved, pl.seg_feat16[i]);
        if (sg.proj_w && dw_tn(dseg16, d, f16, sg.d_in, sgr.proj_w, d, sg.d_in, rows)) return 1;
    }
    if (wide_row_reduce_flush(rrb, st)) return 1;
    return wide_reduce_flush(rb, st);
}

}  // namespace egx
