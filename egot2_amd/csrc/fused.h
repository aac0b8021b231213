// Parameter blocks of the fused per-clip kernels (fused.hip), passed by value as kernel arguments.
#pragma once
#include "common.h"

namespace egx {

constexpr int FUSED_MAX_SEG = 4;
constexpr int FUSED_MAX_LAYERS = 6;      // the shipped PNR / OSCC recipe stacks 6 (HOI/configs/pnr/ts_pnr.yaml:28-34)

// Weight-stream prefetch for a LATER launch (round 6). The packed weights of a launch are read by every workgroup; when they are not in
// the Infinity Cache (a step moves ~0.9 GB through its 256 MiB, so a weight read a step ago is gone) the first readers fetch from HBM under
// load and the weight rings run dry: without the packing launch in front of the forward — whose stores left the copies cache-resident —
// ffn_fwd_kernel ran 6 us and ffn_bwd_kernel 5 us longer (profiles/r06_weight_cache.txt). So every launch touches ONE dword of every
// 128-byte line of the streams its successor reads first, spread over its workgroups (a few hundred lines each: one load per thread),
// as the oldest loads of its own prologue.
constexpr int TOUCH_MAX = 4;
struct TouchList { const void* base[TOUCH_MAX]; unsigned lines[TOUCH_MAX]; int n; };
static inline void touch_add(TouchList& t, const void* base, size_t bytes) {
    if (base && bytes >= 128 && t.n < TOUCH_MAX) { t.base[t.n] = base; t.lines[t.n] = (unsigned)(bytes >> 7); ++t.n; }
}

struct FusedSeg {
    const float* feat;      // (B, T, d_in)
    const void* proj_wp;    // [128, d_in] packed in fragment order (pack_weights)
    const float* proj_b;    // [128]
    const float* add_vec;   // [128] or null
    const float* pos;       // rows of 128 with stride pos_stride, or null
    int T, d_in, off, pos_stride;
    int row0, Tfull;        // tiled mode (S > 48): this descriptor covers frames [row0, row0 + T) of a segment of Tfull frames (else 0, T)
    int seg_id, pad_;       // index of the segment in the caller's list (key of its feature dropout)
};

struct FusedLayer {
    const void* in_proj_wp; const float* in_proj_b;     // *_wp: packed in fragment order (pack_weights)
    const void* out_proj_wp; const float* out_proj_b;
    const void* lin1_wp; const float* lin1_b;
    const void* lin2_wp; const float* lin2_b;
    const float* norm1_w; const float* norm1_b;
    const float* norm2_w; const float* norm2_b;
    uint64_t attn_key, res1_key, ffn_key, res2_key;
    uint32_t attn_thresh, res_thresh, ffn_thresh;
    float drop_inv;
};

// Optional pooled task head fused into the per-clip kernels: logits = Linear(LN(mean_s tokens)).
struct FusedHead {
    const float* ln_w; const float* ln_b;   // head LayerNorm
    const float* W; const float* b;         // [n_out, 128], [n_out]
    int n_out;                               // 0 = no head
};
constexpr int FUSED_HEAD_MAX_OUT = 64;
// head section of the per-clip partial row: ln_w, ln_b (128 each), b (64 slots), W (n_out * 128)
static inline int fused_head_partial_len(int n_out) { return n_out > 0 ? 256 + FUSED_HEAD_MAX_OUT + n_out * 128 : 0; }

struct FusedFwdParams {
    FusedSeg seg[FUSED_MAX_SEG];
    FusedLayer layer[FUSED_MAX_LAYERS];
    const float* ln_w; const float* ln_b;
    float eps;
    int nseg, n_layers, B, S, d_ff;
    int rot_mode;           // hidden-block walk order of the FFN loops (ffn_rot_mode(): tuning aid EGX_FFN_ROT)
    float* tokens_out;      // (B, out_T, 128) or null when only the head output is wanted
    int out_T;              // tokens of every clip that leave the kernel (S, or egx_config.out_tokens)
    FusedHead head; float* logits_out;   // (B, n_out) when head.n_out > 0
    float* saved_pre;       // (B, S, 128)   projected features before the shared LN (token order)
    float* saved_res;       // (2L, B, S, 128) pre-LN residual sums: [2l] = res1, [2l+1] = res2
    uint32_t* relu_bits;    // (L, B, d_ff/32, 64) "alive" bits of the FFN hidden: ReLU active AND kept by dropout (24 per lane)
    void* hid_out;          // optional (L, B, 3, d_ff/16) tiles of the FFN hidden activation (after ReLU and dropout) in
                            // the layout store_hid_tile writes (fp32: ffn_dw's token-along-K operand order; bf16 / split:
                            // accumulator order, transposed by the weight-gradient kernel's LDS reads); fp32 or bf16 elements
    unsigned short* x1p_out;    // optional, CM_SPLIT: (L, 3, B*48, 128) the FFN input x1 (clip-padded token grid) as the three bf16 parts the FFN loop multiplies
                                // (the backward then skips its LayerNorm1 recompute; ffn_dw reads these planes)
    float* xin_out;         // (L, B*S, 128) the input of every layer: an operand of the in-projection weight gradient (small_dw)
    float* qkv_out;         // (L, B, 48, 384) Q | K | V rows with bias on the clip-padded token grid: the backward loads them instead of recomputing the layer
                            // input and its in-projection (22k + 17k of its 300k cycles in split mode; 24 KB + 69 KB per clip)
    uint64_t pos_key; uint32_t pos_thresh; float pos_inv;
    const uint64_t* seed_ptr;   // when non-null the dropout keys are derived in-kernel from *seed_ptr (hipGraph replay)
    // ---- tiled mode (d = 128 with 48 < S <= 512, fused_tiled.hip drives it): a workgroup is one 48-token TILE of a clip ("virtual
    // clip" v = clip * tpc + j holds tokens [48 j, 48 j + 48) of the clip); B = number of tiles; the dense (N, .) arrays are
    // addressed by the global token index, the 48-row grids by the tile. Attention runs between the launches (tiled_attn.hip).
    // feature dropout `self.dp(proj_k(feat_k))` (HOI/models/pnr/video_model_transfer_3task.py:249-252): on the projection output,
    // before the shared LayerNorm; keyed like the generic kernels (layer = segment index, row = b * T_k + t)
    uint64_t feat_key[FUSED_MAX_SEG]; uint32_t feat_thresh; float feat_inv;
    int n_heads;            // 4 (heads of 32) or 8 (heads of 16)
    int tpc, S_clip;        // tiles per clip, tokens per clip (full mode: 1, S)
    size_t Ntok;            // real tokens B_clips * S_clip: the layer stride of the dense saved arrays
    int mode, l0;           // FUSED_MODE_*; POST: the layer whose attention output `attn_in` holds
    const float* attn_in;   // (L, Ntok, 128) attention outputs (before the out-projection), written by tiled_attn_fwd
    // ---- sliced mode (small batches: B * n_slices <= CUs): n_slices workgroups per clip. Each runs the whole clip (identical
    // arithmetic, identical stores) except the FFN, of which it walks 1 / n_slices of the hidden blocks; the partial FFN outputs
    // are exchanged through `xchg` behind one "published" word per (layer, clip, slice) in `xflags` (zeroed before the launch) and
    // summed by every slice in slice order; a block that does not arrive in time is computed by the waiting workgroup (fused_dev.h).
    int n_slices;           // 1 = one workgroup per clip
    float* xchg;            // (L, B, n_slices, 48, 128) fp32
    unsigned* xflags;       // (L, B, 8) "published" words
    int slice_drop;         // testing aid (EGX_SLICE_DROP): bit s set = the workgroups of slice s leave at once, as if they never became resident
    // ---- cut mode (round 5, ffn_cut.hip): the clip kernel is CUT at the FFN. FUSED_MODE_ATTN runs [token preparation (layer 0) or the
    // saved layer input (xin_out, layer l0 > 0)] .. LayerNorm1 of layer l0 and leaves x1 (fp32 rows in x1f_out + the bf16 operand planes
    // in x1p_out); ffn_fwd_kernel (512 threads: two waves per SIMD) runs the FFN, the second residual, LayerNorm2 and the pooled head.
    float* x1f_out;         // (L, Ntok, 128) fp32: the FFN input x1 = LayerNorm1 output of every layer
    // ---- fused weighted cross entropy on the pooled head's logits (egx_ce, round 6; fused_dev.h FusedCe): evaluated by the launch that writes the logits
    const int64_t* ce_target; const float* ce_weight; float* ce_loss; float* ce_dlogits; int ce_B;
    float* zero_word;       // optional: one float the FIRST launch of a cut-mode forward zeroes (ce_loss: the FFN launch then adds into it)
    // per-token classifier + weighted cross entropy on the returned tokens (egx_token_ce): the launch that normalises the last layer's tokens
    const float* tce_W; const float* tce_b; const int64_t* tce_target; const float* tce_cw; int tce_C;
    float* tce_logits; float* tce_probs; float* tce_pred; float* tce_loss; float* tce_correct; float* tce_dlogits;
    unsigned* tce_ticket;   // {arrival counter, loss accumulator, correct-frame accumulator}, zero between launches (see ce_ticket)
    unsigned* ce_ticket;    // optional: {arrival counter, float accumulator} in the weight cache's control block, both zero between launches:
                            // the clips add their loss terms into the accumulator, the LAST one moves the sum to *ce_loss and leaves zeros
                            // (a one-launch forward with a valid weight cache has no earlier launch that could zero *ce_loss)
    TouchList touch;        // weight streams of a later launch to bring into the Infinity Cache (see TouchList)
};
enum { FUSED_MODE_FULL = 0, FUSED_MODE_PRE = 1, FUSED_MODE_POST = 2, FUSED_MODE_ATTN = 3 };

// one matrix to rewrite into MFMA-fragment order (A operand, rows = M dimension); transpose reads src[k][row]
struct PackDesc {
    const float* src; void* dst;
    int R, K, ld, transpose, first_block;
    float scale;        // the packed copy holds scale * W (0 = 1): the FFN dropout keep-scale 1 / (1 - p) rides on W1 and W2^T
};
constexpr int PACK_MAX = 56;          // 4 projections + 8 per layer (both orientations) x 6 layers
struct PackParams {
    PackDesc d[PACK_MAX];
    int n, mode;                // CM_F32 / CM_BF16 / CM_SPLIT (fused_dev.h): element format of the packed fragments
    uint64_t* seed_advance;     // optional: *seed = lcg(*seed) by the first thread (egx_config.advance_seed)
    unsigned* zero_words; int n_zero;   // optional: words to zero (the arrival counters of the sliced mode)
    float* zero_word2;                  // optional: one more word to zero (the fused cross entropy's loss accumulator)
    unsigned* zero_ctl;                 // optional: the weight cache's control block (ce_ticket / tce_ticket below, eight words), zeroed by a launch that packs
};
int pack_weights(PackParams& pp, hipStream_t st);
static inline size_t packed_bytes(int R, int K, int mode) { return (size_t)R * K * (mode == 1 ? 2 : mode == 2 ? 6 : 4); }

struct FfnDwParams {
    const float* x1;      // [N][128] FFN input (post-LN1)
    const float* g;       // [N][128] gradient w.r.t. the FFN output (dropout2 mask already applied)
    const void* w1p;      // packed W1   (R = d_ff, K = 128)
    const void* w2tp;     // packed W2^T (R = d_ff, K = 128)
    const float* b1;
    int N, S, d_ff;
    uint64_t drop_key; uint32_t drop_thresh; float drop_inv;   // FFN hidden dropout (same keying as the forward)
    const uint64_t* seed_ptr; int layer;                       // device-resident seed (see FusedFwdParams)
    // stored-operand variant: H and dH tiles written by the clip-parallel kernels (no recompute, no weights needed)
    const void* hs; const void* dhs; int B;
    int xg_planes;        // x1 / g hold three bf16 planes [part][B*48][128] on the clip-padded token grid (the CM_SPLIT operands, split by the backward kernel)
    float* slab_w1; float* slab_w2t; float* slab_b1;           // set by ffn_dw()
    int splits, kb_per_split;
};
constexpr int FUSED_TOK_TILES = 3;   // 16-token tiles per clip in the fused kernels (S <= 48)
constexpr int FUSED_TOK_PAD = FUSED_TOK_TILES * 16;   // rows per clip of the clip-padded token grid (rows >= S are zero)
static inline size_t fused_hid_bytes(int B, int d_ff, int bf16) {   // one layer of hidden tiles
    return (size_t)B * FUSED_TOK_TILES * (d_ff / 16) * 256 * (bf16 ? 2 : 4);
}
size_t ffn_dw_scratch_bytes(int N, int d_ff, int* splits_out);
bool ffn_dw_bf16_planes();
int ffn_rot_mode();         // EGX_FFN_ROT: 4 (default): clips of an XCD start 0..3 hidden blocks apart; 0: every clip at its own block; 1: all in step;
                            // 2 / 3: 4 / 2 phase groups per XCD; 5: 0..7 blocks apart   // bf16 mode hands x1 / g2 over as bf16 planes (the LDS-ring weight-gradient kernel)
struct ReducePartialsParams;
// out[k][i] += sum_z slab[k][z * n[k] + i] for up to SLAB_REDUCE_MAX arrays (three per layer: dW1, dW2^T, db1) in ONE launch
constexpr int SLAB_REDUCE_MAX = 3 * FUSED_MAX_LAYERS;
struct SlabReduce { const float* slab[SLAB_REDUCE_MAX]; float* out[SLAB_REDUCE_MAX]; size_t n[SLAB_REDUCE_MAX]; int nslab, narr; };
// `rp` (optional): per-clip partial sums to reduce in the same launch as the slab reduction.
// `defer` (optional): the slab reduction is not launched; its arrays are appended to *defer and ffn_dw_reduce() sums the
// slabs of every layer (each layer then needs its own `slabs`) and `rp` in one launch.
int ffn_dw(FfnDwParams p, int compute, float* dW1, float* db1, float* dW2, void* slabs, hipStream_t st,
           const ReducePartialsParams* rp = nullptr, bool deterministic = false, SlabReduce* defer = nullptr);
int ffn_dw_reduce(const SlabReduce& a, const ReducePartialsParams* rp, bool deterministic, hipStream_t st);

struct FusedBwdLayer {
    const void* in_proj_wp;    // packed W_in   (R = 384, K = 128): QKV recompute
    const void* in_proj_wtp;   // packed W_in^T (R = 128, K = 384): input gradient
    const void* out_proj_wtp;  // packed W_o^T  (R = 128, K = 128)
    const void* lin1_wp;       // packed W1     (R = d_ff, K = 128): H recompute
    const void* lin2_wtp;      // packed W2^T   (R = d_ff, K = 128): dH
    const void* lin1_wtp;      // packed W1^T   (R = 128, K = d_ff): d x1
    const float* in_proj_b; const float* lin1_b;
    const float* norm1_w; const float* norm1_b; const float* norm2_w; const float* norm2_b;
    uint64_t attn_key, res1_key, ffn_key, res2_key;
    uint32_t attn_thresh, res_thresh, ffn_thresh;
    float drop_inv;
    // per-token tensors handed to the weight-gradient kernels, all (B*S, .) token-major fp32 (g2_out: three bf16 planes
    // [part][B*48][128] when FusedBwdParams::xg_planes is set). x_in_out is NOT written by the backward: it points at the
    // layer input the forward saved (FusedFwdParams::xin_out) and is only read by the weight-gradient launch.
    float* x1_out; float* g2_out; float* attn_o_out; float* g1_out; float* x_in_out; float* dqkv_out;
};

// Per-clip partial sums of the small parameter gradients, laid out as `partials[clip][P]`:
//   layer l at l * 1152: norm2_w, norm2_b, lin2_b, norm1_w, norm1_b, out_proj_b (128 each), in_proj_b (384)
//   then ln_w, ln_b (128 each), then per segment: add_vec, proj_b (128 each)
constexpr int FUSED_P_LAYER = 1152;
static inline int fused_partial_len(int L, int nseg) { return L * FUSED_P_LAYER + 256 + nseg * 256; }

struct FusedBwdParams {
    FusedSeg seg[FUSED_MAX_SEG];
    float* dseg_out[FUSED_MAX_SEG];   // (B, T_k, 128) gradient w.r.t. the projected (pre-LN) features of segment k
    FusedBwdLayer layer[FUSED_MAX_LAYERS];
    const float* ln_w; const float* ln_b;
    float eps;
    int nseg, n_layers, B, S, d_ff;
    int rot_mode;           // hidden-block walk order of the FFN loops (ffn_rot_mode(): tuning aid EGX_FFN_ROT)
    const float* d_tokens;     // (B, out_T, 128), or null when the head is fused (then d_logits drives the backward)
    int out_T;                 // tokens of every clip that carry an upstream gradient (S, or egx_config.out_tokens)
    FusedHead head; const float* d_logits; int head_off;   // head_off: offset of the head section in the partial row
    const float* pooled;       // tiled mode with the fused head: (B, 128) token means saved by the forward's pool_head_fwd
    const float* saved_pre;    // from the forward
    const float* saved_res;
    const float* saved_qkv;    // (L, B, 48, 384) from the forward (FusedFwdParams::qkv_out)
    const uint32_t* relu_bits;
    int xg_planes;          // CM_SPLIT: g2_out leaves pre-split (what ffn_dw_stored_kernel<CM_SPLIT> reads) and x1_out is not
                            // written at all: the forward has saved x1 in that form (FusedFwdParams::x1p_out)
    void* dhid_out;         // optional gradient of the FFN pre-activation, same tile layout as FusedFwdParams::hid_out
    float* zero_buf; size_t zero_n;   // optional: floats the kernel zero-fills first (the caller's flat gradient buffer)
    float* partials; int P;
    uint64_t pos_key; uint32_t pos_thresh; float pos_inv;
    const uint64_t* seed_ptr;
    // ---- tiled mode (see FusedFwdParams): one launch runs [the in-projection input gradient of layer l_front] + [LayerNorm2 ..
    // out-projection input gradient of layer l_back] or, with l_back < 0, the token-preparation backward
    uint64_t feat_key[FUSED_MAX_SEG]; uint32_t feat_thresh; float feat_inv;    // feature dropout (see FusedFwdParams)
    int n_heads;
    float* dx0_out;                 // optional (Ntok, 128): gradient w.r.t. the token-preparation output BEHIND its dropout mask = the
                                    // per-token gradient of a learned positional table (summed over the clips by pos_grad_accum)
    int tpc, S_clip; size_t Ntok;
    int tiled, l_front, l_back;     // l_front < 0: the launch starts from d_tokens; l_back < 0: it ends with the token preparation
    float* datt;                    // (Ntok, 128) gradient w.r.t. the attention output of layer l_back (written) 
    float* dres;                    // (Ntok, 128) gradient reaching the layer input through the residual (written for l_back, read for l_front)
    int n_slices; float* xchg; unsigned* xflags; int slice_drop;     // sliced mode (see FusedFwdParams): the partial FFN input gradients are exchanged
    // ---- cut mode (see FusedFwdParams): per layer l = cut_layer, ffn_bwd_kernel (512 threads) runs [head backward] + LayerNorm2 backward +
    // the FFN input gradient and leaves dy1 = dX1 + d_res2; fused_bwd_kernel<CUT> picks it up at LayerNorm1's backward and runs on to
    // the in-projection input gradient, which it leaves in dxin for the next (lower) layer's ffn_bwd_kernel — or to the token preparation.
    int cut, cut_layer;
    float* dy1;             // (Ntok, 128) gradient reaching LayerNorm1's output (FFN path + residual path)
    float* dxin;            // (Ntok, 128) gradient w.r.t. the input of layer cut_layer (= LayerNorm2 output of the layer below)
    const float* d_logits_scale;    // optional device scalar multiplied into d_logits (egx_config.d_logits_scale)
    // egx_token_ce: d tokens = g * d_logits W rebuilt per clip, the clip's partial d W / d b rows in the head section of the partial row (head_off)
    const float* tce_W; const float* tce_dlogits; int tce_C;
    TouchList touch;        // weight streams of a later launch to bring into the Infinity Cache (see TouchList)
};
int fused_backward(const FusedBwdParams& p, int compute, hipStream_t st);
// cut mode (ffn_cut.hip): the FFN of layer l as launches of their own, eight waves per clip
bool ffn_cut_supported(int d_ff);
size_t ffn_cut_lds_bytes();
int ffn_cut_forward(const FusedFwdParams& p, int l, int compute, hipStream_t st);
int ffn_cut_backward(const FusedBwdParams& p, int l, int compute, hipStream_t st);

// One "dW[R][C] += G^T X" problem of the grouped small-weight-gradient kernel: G (K, R) and X (K, C) token-major.
struct SmallDwProblem {
    const float* G; const float* X; float* out;
    int R, C, K, ldg, ldx, first_block, splits;   // first_block / splits are set by small_dw()
};
constexpr int SMALL_DW_MAX = 8;
struct SmallDwParams {
    SmallDwProblem pr[SMALL_DW_MAX];
    int n, per;           // per = K-blocks (32 tokens) per workgroup, the same for every problem (balanced grid)
    float* slabs;         // deterministic mode: one dense [64][128] fp32 tile per workgroup, summed in split order afterwards
};
// slabs / slab_bytes: optional scratch for the deterministic (atomic-free) variant; null = atomic accumulation
struct SmallDwTail;
// `tail` (optional): the FFN slab reduction and the partial-row reduction, done by this launch's workgroups before their own
// work (each takes 1 / grid of the units) instead of by a launch of their own
// reduce_here = false (with slabs): the tiles are left for tail_reduce()
int small_dw(SmallDwParams& p, int compute, hipStream_t st, void* slabs = nullptr, size_t slab_bytes = 0, const SmallDwTail* tail = nullptr, bool reduce_here = true);
int seed_advance(uint64_t* seed, hipStream_t st);

struct PartialDst { float* dst; int off, len; };
constexpr int PARTIAL_MAX_DST = 64;
struct ReducePartialsParams {
    PartialDst d[PARTIAL_MAX_DST];
    int n, B, P;
    const float* partials;
};
int reduce_partials(const ReducePartialsParams& rp, hipStream_t st, bool deterministic = false);
struct SmallDwTail { SlabReduce red; ReducePartialsParams rp; unsigned slab_blocks; int rp_units, chunks; uint64_t* seed_advance; /* optional: *seed = lcg(*seed) by the first thread (egx_config.advance_seed == 2) */ TouchList touch; /* the next forward's first weight streams */ };
void small_dw_tail_init(SmallDwTail& t, const SlabReduce& red, const ReducePartialsParams* rp);
// Round 6: every cross-workgroup sum of the per-clip backward in ONE fixed-order launch (fused_bwd.hip tail_reduce_kernel): the tiles small_dw(sp, ..., slabs)
// wrote, the FFN slabs, the per-clip partial rows; also the seed advance and the next forward's weight prefetch. Null / empty parts are skipped.
int tail_reduce(const SmallDwParams* sp, const SlabReduce* red, const ReducePartialsParams* rp, uint64_t* seed_advance_ptr, const TouchList* touch, hipStream_t st);

// Optional per-kernel device timing (hipEvents on the launch stream) for bench.py's roofline block.
enum { TIMER_FUSED_FWD = 0, TIMER_FUSED_BWD = 1, TIMER_FFN_DW = 2, TIMER_FFN_FWD = 3, TIMER_FFN_BWD = 4, TIMER_WIDE_GEMM = 5,
       TIMER_WIDE_ATTN_FWD = 6, TIMER_WIDE_ATTN_BWD = 7, TIMER_COUNT = 8 };
void timing_enable(int on);
void timing_begin(int which, hipStream_t st);
void timing_end(int which, hipStream_t st);
int timing_read(int which, double* total_ms, int* count);

// ---- attention of the tiled mode (tiled_attn.hip): one workgroup per (clip, head, query / key range), K | V (forward, dQ pass) or
// Q | dO (dK / dV pass) of the whole clip in LDS, every score of a 16-query tile in accumulator registers (S <= 512)
struct TiledAttnParams {
    const float* qkv;       // (B * tpc * 48, 384) Q | K | V rows on the 48-row tile grid = clip c's tokens at rows c * tpc * 48 + s
    float* attn_o;          // (Ntok, 128) forward: written; backward: read (delta = rowsum(dO . O))
    float* lse;             // (B, 4, S) log-sum-exp of the scaled scores
    const float* d_o;       // backward: (Ntok, 128) gradient w.r.t. attn_o
    float* delta;           // backward: (B, 4, S) sum_k P dP per query (written by the dQ pass, read by the dK / dV pass)
    float* dqkv;            // backward: (Ntok, 384) dense
    int B, S, tpc;
    uint64_t drop_key; uint32_t drop_thresh; float drop_inv;
    const uint64_t* seed_ptr; int layer;
};
constexpr int TILED_MAX_S = 512;
int tiled_attn_fwd(const TiledAttnParams& p, int compute, hipStream_t st);
int tiled_attn_bwd(const TiledAttnParams& p, int compute, hipStream_t st);

bool fused_supported(int d_model, int n_heads, int d_ff, int S, int nseg, const int* d_in, const int* T, const bool* has_proj);      // n_heads 4 or 8
size_t fused_lds_bytes(int NT);
long long slices_stolen_fwd(int reset);     // sliced-mode diagnostic (fused_dev.h slice_stolen_note); synchronous device reads
long long slices_stolen_bwd(int reset);
int debug_read_stamps(unsigned long long* out, int n);
int debug_read_bstamps(unsigned long long* out, int n);
int fused_forward(const FusedFwdParams& p, int compute, hipStream_t st);

}  // namespace egx
