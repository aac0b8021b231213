/*
 * egot2x.h — C ABI of libegot2x.so, the MI355X (gfx950) native task-translation
 * transformer ("translator") for EgoT2's Stage-II fusion path.
 *
 * The reference has no native code: its translator is a chain of torch.nn calls.
 * Each entry point below replaces the torch.nn call sites named next to it
 * (paths relative to the reference checkout):
 *
 *   egx_encoder_fwd / egx_encoder_bwd
 *       HHI/models/ttm/model_taskspecific.py:222-226,238-242 (proj_* -> encode_prepare -> cat
 *       -> nn.TransformerEncoder), HHI/models/asd/model_taskspecific.py:133-155,
 *       HHI/models/multitask/task_prompt_model.py:224-250,
 *       HOI/models/lta/lta_models_lta_transfer.py:355-361 (proj_* -> cat -> ln + pe -> transformer),
 *       HOI/models/pnr/video_model_transfer_3task.py:249-255,
 *       HOI/models/multitask/video_model_builder.py:331-346; and their autograd backward.
 *   egx_pool_head_fwd / egx_pool_head_bwd
 *       HHI/models/ttm/model_taskspecific.py:243-244 (mean over tokens -> linear_head = LN + Linear),
 *       HOI/models/pnr/video_model_transfer_3task.py:256-257; HOI LTA mean-pool
 *       (lta_models_lta_transfer.py:362) with ln == NULL and W == NULL.
 *   egx_linear_fwd / egx_linear_bwd
 *       any nn.Linear on the path that is not inside the encoder (HOI MultiTaskHead projections,
 *       HOI/models/lta/head_helper.py:245-248,281-283).
 *   egx_gemm / egx_layernorm_* / egx_attention_*
 *       the individual ATen ops (addmm, layer_norm, multi_head_attention_forward core) for unit parity tests.
 *
 * Conventions
 *   - All tensors are dense fp32, row-major, device memory owned by the caller (PyTorch's caching
 *     allocator). The library never allocates device memory; `saved` and `scratch` are caller-provided
 *     workspaces sized by egx_encoder_workspace().
 *   - Token layout is batch-first packed (B, S, d): one clip's S*d block is contiguous.
 *   - Weights are torch-style [out, in] row-major.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no host sync and
 *     is hipGraph-capturable.
 *   - Return value 0 = success; non-zero = error, message via egx_last_error() (thread-local).
 *   - `compute`: EGX_F32 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32), EGX_BF16 = bf16 MFMA operands with
 *     fp32 accumulation, fp32 LayerNorm/softmax statistics, fp32 storage. EGX_F32_SPLIT = fp32 operands split exactly into
 *     three bf16 parts and multiplied by six v_mfma_f32_16x16x32_bf16 per K-block (the dropped cross terms are below fp32's
 *     own product rounding): fp32-grade results at 2.7x the matrix rate of the fp32 MFMA. Implemented by the fused d = 128
 *     kernels; everywhere else it means EGX_F32.
 */
#ifndef EGOT2X_H
#define EGOT2X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGX_ABI_VERSION 16
#define EGX_MAX_SEGMENTS 8

enum { EGX_F32 = 0, EGX_BF16 = 1, EGX_F32_SPLIT = 2 };
enum { EGX_IMPL_AUTO = 0, EGX_IMPL_GENERIC = 1, EGX_IMPL_FUSED = 2, EGX_IMPL_WIDE = 3, EGX_IMPL_TILED = 4 };

/* One contiguous run of tokens of the packed sequence, produced from one frozen-backbone feature
 * tensor: tokens[b, off + t, :] = LN(feat[b, t, :] @ proj_w^T + proj_b) + add_vec + pos[t * pos_stride + :]
 * (LN is the encoder's shared `ln`; proj_w == NULL means the feature is already d_model wide). */
typedef struct egx_segment {
    const float* feat;    /* (B, T, d_in) */
    int T;
    int d_in;
    const float* proj_w;  /* [d_model, d_in] or NULL */
    const float* proj_b;  /* [d_model] or NULL */
    const float* add_vec; /* [d_model] task-embedding row, or NULL */
    const float* pos;     /* first positional row for this segment, or NULL */
    int pos_stride;       /* floats between consecutive positional rows */
    /* Feature hand-off from the frozen backbones (wide bf16 path only, projected segments only):
     * feat_bf16 != 0: `feat` points at bf16 data (the backbone wrote its `middle=True` features in the packed bf16 layout
     *                 the projection GEMM reads; no fp32 copy, no cast pass).
     * pool > 1      : `feat` holds (B, T * pool, d_in) per-FRAME features and token t is the mean of frames
     *                 [t * pool, (t + 1) * pool): the temporal mean of HOI encode_clips_pnr
     *                 (HOI/models/lta/lta_models_lta_transfer.py:335-343, `tmp.mean(dim=1)`) fused with the cast into the
     *                 projection operand, so the pooled fp32 tensor is never written. */
    int feat_bf16;
    int pool;
} egx_segment;

/* Gradient sinks matching egx_segment; any pointer may be NULL (gradient not wanted).
 * All sinks are ACCUMULATED into (+=): the caller zero-fills them once per backward. */
typedef struct egx_segment_grads {
    float* proj_w;
    float* proj_b;
    float* add_vec;
    float* pos;       /* [T, d_model] rows with stride pos_stride (learned pe), or NULL */
    float* feat;      /* (B, T, d_in) gradient into the feature, or NULL (frozen backbones) */
} egx_segment_grads;

/* nn.TransformerEncoderLayer parameters (post-LN, ReLU). */
typedef struct egx_layer {
    const float* in_proj_w;  /* [3d, d] rows = [Wq; Wk; Wv] */
    const float* in_proj_b;  /* [3d] */
    const float* out_proj_w; /* [d, d] */
    const float* out_proj_b; /* [d] */
    const float* lin1_w;     /* [d_ff, d] */
    const float* lin1_b;     /* [d_ff] */
    const float* lin2_w;     /* [d, d_ff] */
    const float* lin2_b;     /* [d] */
    const float* norm1_w;
    const float* norm1_b;
    const float* norm2_w;
    const float* norm2_b;
} egx_layer;

typedef struct egx_layer_grads {
    float* in_proj_w;
    float* in_proj_b;
    float* out_proj_w;
    float* out_proj_b;
    float* lin1_w;
    float* lin1_b;
    float* lin2_w;
    float* lin2_b;
    float* norm1_w;
    float* norm1_b;
    float* norm2_w;
    float* norm2_b;
} egx_layer_grads;

struct egx_ce;
struct egx_token_ce;
typedef struct egx_config {
    int d_model;
    int n_heads;
    int d_ff;
    int n_layers;
    int n_segments;
    float ln_eps;
    int compute;      /* EGX_F32 | EGX_BF16 | EGX_F32_SPLIT */
    int impl;         /* EGX_IMPL_* */
    float p_drop;     /* encoder-layer dropout (attention probs, dropout1, FFN hidden, dropout2) */
    float p_pos;      /* dropout on the token-prep output (PositionalEncoding.dropout, fixed 0.1 in HHI) */
    float p_feat;     /* dropout on projected features before LN (HOI `dp`) */
    const uint64_t* seed_ptr; /* optional DEVICE pointer: when non-NULL the per-clip, tiled and wide bf16 kernels derive their dropout keys
                                 from *seed_ptr instead of the host `seed` argument, so a captured hipGraph draws fresh
                                 masks on every replay (advance it with egx_seed_advance inside the graph, or set
                                 advance_seed). */
    int advance_seed;         /* 1 with seed_ptr: a training-mode forward advances *seed_ptr by one LCG step before
                                 using it (folded into its first kernel), the backward of that forward reads the
                                 advanced value. 2 (ABI v15): the forward uses *seed_ptr as it is and the BACKWARD advances it
                                 behind its last reader (folded into its last launch): a forward + backward step needs no launch
                                 in front of the forward even when the packing launch is skipped (weight_cache_valid); a training
                                 forward that is never followed by its backward repeats the masks of the previous one. */
    void* zero_buf;           /* optional: a device buffer the BACKWARD zero-fills before any gradient is accumulated */
    size_t zero_bytes;        /* (the caller's flat gradient buffer: saves a separate fill launch); multiple of 16 */
    int bwd_stage;            /* backward only: 0 = everything; 1 = all but the grouped small weight gradients (dW_proj,
                                 dW_in, dW_o); 2 = only those (same arguments, zero_buf ignored). Lets the caller start the
                                 all-reduce of every other gradient while stage 2 still runs (fused path; the generic
                                 path does all its work in stage 1). */
    int deterministic;        /* != 0: every cross-workgroup sum of the BACKWARD runs in a fixed order (split-K slabs and
                                 per-clip partial rows reduced by one workgroup per output instead of fp32 atomics), so the
                                 same inputs and seed give bit-identical gradients run to run. Honoured by the fused per-clip
                                 kernels and the shape-generic kernels (split-K slabs, LayerNorm / bias / pooled-head partial
                                 buffers with ordered sums; pass the same flag to the workspace query: it sizes their scratch);
                                 the wide bf16 path is always deterministic. */
    int out_tokens;           /* 0 = the whole sequence. T > 0: only the first T tokens of every clip leave the encoder
                                 (tokens_out is (B, T, d)) and only they receive an upstream gradient (d_tokens is (B, T, d)) —
                                 the ASD translator returns its first segment, HHI/models/asd/model_taskspecific.py:156-158;
                                 without this the caller slices (copy) and autograd zero-fills and scatters the gradient.
                                 Fused per-clip kernels only (egx_encoder_impl() == EGX_IMPL_FUSED); an error elsewhere. */
    void (*bucket_cb)(void* user, int bucket);   /* BACKWARD, optional (wide bf16 path): called on the host thread right after the
                                 kernels that complete gradient bucket `bucket` have been enqueued on the stream. Bucket b in
                                 [0, n_layers) = every parameter gradient of encoder layer n_layers - 1 - b (the order the backward
                                 finishes them); everything else (projections, shared LayerNorm, embeddings) is complete when the
                                 call returns. The caller lays its flat gradient buffer out last-layer-first and starts the
                                 all-reduce of a layer's slice from the callback, so the exchange of layer l overlaps the backward
                                 of layers l - 1 .. 0 (the reference gets this from DDP's bucketed reducer,
                                 HOI/scripts/multitask/run.py:41-50). With a callback every split-K slab reduction runs right
                                 behind its GEMM (no deferred batch). Other implementations ignore it. */
    void* bucket_user;
    /* ---- ABI v15 (round 6) ---- */
    void* weight_cache;       /* optional PERSISTENT device buffer of egx_weight_cache_bytes() bytes (per-clip / tiled kernels): the
                                 MFMA-fragment-packed copies of the weights live there instead of in `saved`, so that a forward whose
                                 weights did not change since the forward that filled it can skip the packing launch
                                 (weight_cache_valid). The backward must be given the same buffer. NULL: packed into `saved` every forward.
                                 Its last 256 bytes are a control block of the library (arrival counter + accumulator of the fused
                                 cross entropy); a forward that packs resets it, so the buffer needs no initialisation. One cache
                                 serves one stream at a time. */
    int weight_cache_valid;   /* forward, with weight_cache: != 0 = the cache holds the packed copies of EXACTLY these weights in this
                                 compute mode with this FFN keep-scale (training, p_drop): nothing is packed. The caller owns that
                                 promise (egot2_amd/translator.py keys it on the parameters' storage and version counters). */
    const float* d_logits_scale; /* backward, optional DEVICE scalar: the pooled head's backward multiplies d_logits by it (the upstream
                                 gradient of a loss the forward computed itself, egx_ce; lets loss.backward() hand its ones / loss-scale
                                 tensor over without a launch). Per-clip kernels only; NULL = 1. */
    const struct egx_ce* ce;  /* forward, optional (egx_translator_fwd with a head): weighted cross entropy on the logits, see egx_ce */
    /* ---- ABI v16 (round 6) ---- */
    const struct egx_token_ce* token_ce;  /* forward + backward, optional (egx_encoder_fwd / _bwd with out_tokens): per-token classifier + weighted
                                 cross entropy on the returned tokens, see egx_token_ce. Only where egx_encoder_token_ce_ok() says so. */
} egx_config;

/* Weighted cross entropy ON the pooled head's logits, evaluated by the translator forward itself
 * (nn.CrossEntropyLoss(weight=[0.266, 0.734])(model(...), target), HHI/tasks/ttm/video_task_2loader.py:21-22,34 — the same arithmetic as
 * egx_weighted_ce, labels outside [0, n_out) contribute neither loss, weight nor gradient). On the per-clip kernels it runs in the epilogue of
 * the launch that produces the logits (the normaliser sum_i w[y_i] depends on the labels only: every workgroup sums it itself), so
 * d loss / d logits is in place when the backward starts and the step has one launch less; elsewhere (tiled / wide / generic kernels,
 * deterministic mode) the library appends the egx_weighted_ce launch. Pass `d_logits` as egx_translator_bwd's d_logits (scaled by
 * egx_config.d_logits_scale when the loss itself has an upstream gradient other than 1). */
typedef struct egx_ce {
    const int64_t* target;      /* (B) class indices */
    const float* class_weight;  /* (n_out) or NULL (= 1) */
    float* loss;                /* scalar, written */
    float* d_logits;            /* (B, n_out), written */
} egx_ce;

/* Per-token classifier + weighted cross entropy ON the tokens the encoder returns, evaluated by the encoder launches themselves: the ASD task's
 * lossAV on the translator's per-frame output (HHI/tasks/asd/video_task_taskspecific.py:24,33 -> HHI/tasks/asd/loss.py:11-30: x = FC(x);
 * nloss = CrossEntropyLoss(weight=[1, 4])(x, labels); softmax scores, rounded labels, number of correct frames). The arithmetic of
 * egx_linear_ce_fwd / _bwd; as two launches of their own they are 27 us of the 0.41 ms ASD step. Forward: the launch that normalises the last layer's
 * tokens also writes logits / probs / pred / d_logits of the clip's out_tokens rows and adds the clip's loss and correct-frame terms (the
 * normaliser sum_i w[y_i] depends on the labels only: every workgroup sums it itself). Backward (egx_encoder_bwd with d_tokens = NULL): the first
 * launch rebuilds d tokens = g * d_logits W per clip and leaves the clip's partial d W / d b rows to the step's fixed-order reduction; g =
 * *egx_config.d_logits_scale (NULL = 1). The tokens themselves are still written (tokens_out). */
typedef struct egx_token_ce {
    const float* W;             /* (C, d) classifier weight */
    const float* b;             /* (C) or NULL */
    const int64_t* target;      /* (B * out_tokens) class indices; outside [0, C): no loss, no weight, no gradient */
    const float* class_weight;  /* (C) or NULL (= 1) */
    int C;                      /* 1 <= C <= 8 */
    float* logits;              /* (B * out_tokens, C), written by the forward */
    float* probs;               /* same shape, optional: softmax(logits) */
    float* pred;                /* (B * out_tokens), optional: round(probs[:, 1]) */
    float* loss;                /* scalar, written by the forward */
    float* correct;             /* scalar, optional: rows with pred == target */
    float* d_logits;            /* (B * out_tokens, C): written by the forward, read by the backward */
    float* d_W; float* d_b;     /* backward: (C, d) and (C), each optional: the gradients are ADDED (as every parameter gradient of
                                   egx_encoder_bwd: into the caller's zero_buf-covered flat buffer, or pre-zeroed memory) */
} egx_token_ce;

int egx_abi_version(void);
/* != 0: this configuration and batch run on kernels that evaluate egx_config.token_ce themselves (one clip per workgroup, one launch per
 * direction, a persistent weight cache, no pooled head). Elsewhere the caller composes egx_linear_ce_fwd / _bwd with the encoder calls. */
int egx_encoder_token_ce_ok(const egx_config* cfg, const egx_segment* segs, int B);
/* The kernel-selection switches EGX_FFN_CUT / EGX_FFN_SLICES / EGX_SLICE_DROP (development and test aids) are read from the environment once,
 * at first use; this re-reads them (the parity tests compare the modes inside one process). */
void egx_tuning_reload(void);
/* Bytes of egx_config.weight_cache for this configuration (0: this configuration does not run on kernels that pack weights). Depends on
 * the model dimensions and the compute mode, not on the batch. */
size_t egx_weight_cache_bytes(const egx_config* cfg, const egx_segment* segs);
/* Kernel launches (and memsets) the library has enqueued on this thread's behalf since the last reset: the bench reports
 * launches per step with it. reset != 0 zeroes the counter after reading. Not thread-safe (a diagnostic). */
long long egx_launch_count(int reset);
const char* egx_last_error(void);

/* Workspace sizes in bytes for a batch of B clips of S tokens (S = sum of segment T). */
int egx_encoder_workspace(const egx_config* cfg, const egx_segment* segs, int B,
                          size_t* saved_bytes, size_t* scratch_bytes);

/* 1 when this configuration runs on the fused per-clip kernels (which leave d_tokens untouched in backward),
 * 0 for the shape-generic kernels (which consume d_tokens). */
int egx_encoder_uses_fused(const egx_config* cfg, const egx_segment* segs, int B);
/* Which kernels this configuration runs on: EGX_IMPL_FUSED (per-clip kernels: d = 128, h = 4, S <= 48), EGX_IMPL_TILED (the same
 * kernels over 48-token tiles with the attention of the whole clip between the launches: d = 128, h = 4, 48 < S <= 512, compute bf16
 * or f32s - the reference's real TTM / ASD batches of 15 .. 150 frames per task, HHI/dataset/ttm/data_loader_2task.py:119,150-162), EGX_IMPL_WIDE
 * (compute = bf16 with d_model >= 256, d_model / d_ff / projected d_in multiples of 128, S <= 128 with head dim 32 / 64 / 96 / 128 or S <= 512 with head dim 32 / 64: all B*S tokens
 * through bf16-storage MFMA GEMMs and MFMA attention - BASELINE.json configs[3], configs[4]) or EGX_IMPL_GENERIC; -1 on an
 * invalid configuration. */
int egx_encoder_impl(const egx_config* cfg, const egx_segment* segs, int B);
/* Workgroups per clip of the per-clip kernels (EGX_IMPL_FUSED) for this batch on the current device: 1, or 2 / 4 / 8 for small
 * batches (round_up(B, 8) * n <= compute units; the reference's own TTM batches are ~26 clips, HHI/dataset/ttm/sampler.py:41, and
 * strong scaling leaves 32 clips per GPU): every workgroup of a clip runs the clip except the FFN, of which it walks 1 / n of the
 * hidden blocks; the partial sums are exchanged through device memory. A partial sum that does not arrive within 100 us (its
 * workgroup is not resident: the device is shared) is computed by the waiting workgroup itself, so the result never depends on
 * the scheduling. EGX_FFN_SLICES=1 in the environment turns it off, =2 / 4 / 8 caps n. -1: invalid configuration. */
int egx_encoder_slices(const egx_config* cfg, const egx_segment* segs, int B);

/* tokens_out: (B, S, d). `saved` is written in forward and read in backward.
 * training != 0 applies dropout with masks derived from (seed, site, element). */
int egx_encoder_fwd(const egx_config* cfg, const egx_segment* segs,
                    const float* ln_w, const float* ln_b,
                    const egx_layer* layers, int B,
                    float* tokens_out, void* saved, void* scratch,
                    int training, uint64_t seed, void* stream);

/* d_tokens: (B, S, d) gradient w.r.t. tokens_out; it is consumed (overwritten). */
int egx_encoder_bwd(const egx_config* cfg, const egx_segment* segs,
                    const float* ln_w, const float* ln_b,
                    const egx_layer* layers, int B,
                    float* d_tokens, const void* saved, void* scratch,
                    const egx_segment_grads* seg_grads, float* d_ln_w, float* d_ln_b,
                    const egx_layer_grads* layer_grads,
                    int training, uint64_t seed, void* stream);

/* Encoder + pooled task head in one call: logits = Linear(LN(mean_s tokens)), the TTM / PNR head
 * (HHI/models/ttm/model_taskspecific.py:243-244). On the fused per-clip path the head runs inside the encoder
 * kernels (tokens never leave the chip unless tokens_out != NULL); on the generic path it is appended.
 * Workspaces: egx_translator_workspace(). n_out <= 64. */
typedef struct egx_head {
    const float* ln_w; const float* ln_b;   /* head LayerNorm [d] */
    const float* W; const float* b;         /* [n_out, d], [n_out] */
    int n_out;
} egx_head;
typedef struct egx_head_grads { float* ln_w; float* ln_b; float* W; float* b; } egx_head_grads;   /* accumulated (+=) */
int egx_translator_workspace(const egx_config* cfg, const egx_segment* segs, int B,
                             size_t* saved_bytes, size_t* scratch_bytes);
int egx_translator_fwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                       const egx_layer* layers, const egx_head* head, int B, float* logits_out, float* tokens_out,
                       void* saved, void* scratch, int training, uint64_t seed, void* stream);
int egx_translator_bwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b,
                       const egx_layer* layers, const egx_head* head, int B, const float* d_logits, const void* saved,
                       void* scratch, const egx_segment_grads* seg_grads, float* d_ln_w, float* d_ln_b,
                       const egx_layer_grads* layer_grads, const egx_head_grads* head_grads, int training,
                       uint64_t seed, void* stream);

/* pooled = mean_s tokens[b, s, :]; y = ln_w ? LN(pooled) : pooled; out = W ? y W^T + b : y.
 * `pooled_saved` (B, d) is kept for backward. n_out <= 64 when W != NULL. */
int egx_pool_head_fwd(const float* tokens, int B, int S, int d,
                      const float* ln_w, const float* ln_b, float ln_eps,
                      const float* W, const float* b, int n_out,
                      float* pooled_saved, float* out, void* stream);
int egx_pool_head_bwd(const float* d_out, const float* pooled_saved, int B, int S, int d,
                      const float* ln_w, const float* ln_b, float ln_eps,
                      const float* W, int n_out,
                      float* d_tokens, float* d_ln_w, float* d_ln_b, float* d_W, float* d_b,
                      void* stream);

/* y[M,N] = x[M,K] W[N,K]^T + b (+ReLU). */
int egx_linear_fwd(const float* x, const float* W, const float* b, float* y,
                   int M, int N, int K, int relu, int compute, void* stream);
/* y = x W^T + b + residual[M,N]: a projection with the residual connection in its epilogue (pre-LN blocks of
 * HOI/models/pnr/simple_vit.py:104-106: `x = attn(x) + x`, `x = ff(x) + x`). b may be NULL. */
int egx_linear_residual_fwd(const float* x, const float* W, const float* b, const float* residual, float* y, int M, int N, int K,
                            int compute, void* stream);
/* Exact (erf) GELU, nn.GELU() of simple_vit.FeedForward (HOI/models/pnr/simple_vit.py:55-65): h = z Phi(z);
 * backward dz = dh (Phi(z) + z phi(z)). n %% 4 == 0. */
int egx_gelu_fwd(const float* z, float* h, size_t n, void* stream);
int egx_gelu_bwd(const float* z, const float* dh, float* dz, size_t n, void* stream);
/* dx[M,K] = dy W ; dW[N,K] += dy^T x ; db[N] += colsum(dy). Any output may be NULL.
 * scratch must hold egx_linear_bwd_scratch(M,N,K) bytes. */
size_t egx_linear_bwd_scratch(int M, int N, int K);
int egx_linear_bwd(const float* dy, const float* x, const float* W,
                   float* dx, float* dW, float* db, int M, int N, int K,
                   int compute, void* scratch, void* stream);

/* --- single-op entry points (unit parity tests) --- */
/* layout: 0 = NT (C = A[M,K] B[N,K]^T), 1 = NN (C = A[M,K] B[K,N]), 2 = TN (C = A[K,M]^T B[K,N]). */
int egx_gemm(int layout, const float* A, const float* B, float* C, int M, int N, int K,
             const float* bias, int relu, int compute, void* scratch, size_t scratch_bytes, void* stream);
int egx_layernorm_fwd(const float* x, const float* res, const float* w, const float* b, float eps,
                      float* pre, float* stats, float* y, int rows, int d, void* stream);
int egx_layernorm_bwd(const float* dy, const float* pre, const float* stats, const float* w,
                      float* dx, float* dw, float* db, int rows, int d, void* stream);
/* qkv: (B, S, 3d) packed in-proj output; out: (B, S, d); lse: (B, H, S). */
int egx_attention_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int d,
                      float p_drop, uint64_t seed, void* stream);
int egx_attention_bwd(const float* qkv, const float* out, const float* lse, const float* d_out,
                      float* d_qkv, int B, int S, int H, int d,
                      float p_drop, uint64_t seed, void* stream);

/* --- unit hooks of the wide bf16 path (operands are bf16 in device memory, passed as void*) ---
 * layout 0 (NT): C[M,N] = A[M,K] B[N,K]^T (+ bias) (ReLU) (+ residual[M,N] fp32): the nn.Linear forward / input gradient.
 * layout 2 (TN): C[M,N] = A[K,M]^T B[K,N]: the nn.Linear weight gradient over K = B*S tokens (M, N multiples of 128).
 * Cf (fp32) and / or Cb (bf16) receive the result (TN: Cf only). K %% 64 == 0 for NT. scratch: egx_wide_gemm_scratch() bytes. */
size_t egx_wide_gemm_scratch(int layout, int M, int N, int K);
int egx_wide_gemm(int layout, const void* A, const void* B, float* Cf, void* Cb, int M, int N, int K, const float* bias,
                  int relu, const float* residual, void* scratch, void* stream);
/* qkv: (B*S, 3d) packed bf16 in-projection rows; out: (B*S, d) bf16; lse: (B, H, S) fp32. S <= 128 with head dim 32/64/96/128,
 * or 128 < S <= 480 with head dim 32/64 (the EgoT2-g encoders on sequences of up to 3 x 150 tokens: online softmax).
 * Backward: d_out (B*S, d) bf16 -> d_qkv (B*S, 3d) bf16 (probabilities recomputed from lse; every element written once).
 * S > 128 also reads `out` (the forward's output) and uses `delta` ((B, H, S) fp32 scratch); both may be NULL for S <= 128. */
int egx_wide_attention_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int d, float p_drop, uint64_t seed,
                           void* stream);
int egx_wide_attention_bwd(const void* qkv, const void* out, const float* lse, const void* d_out, void* d_qkv, float* delta,
                           int B, int S, int H, int d, float p_drop, uint64_t seed, void* stream);

/* Fused FFN weight gradients (d_model = 128): dW1 += dH^T x1, db1 += colsum(dH), dW2 += g^T H where
 * H = dropout(relu(x1 W1^T + b1)) and dH = (g W2) .* mask are recomputed on chip. x1, g: (N, 128); S = tokens per
 * clip (dropout keying). scratch: egx_ffn_dw_scratch() bytes. Replaces the autograd of linear1/ReLU/linear2 weight
 * gradients (torch.nn.TransformerEncoderLayer._ff_block). */
size_t egx_ffn_dw_scratch(int N, int d_ff, int compute);
int egx_ffn_dw(const float* x1, const float* g, const float* W1, const float* b1, const float* W2, int N, int S,
               int d_ff, float p_drop, uint64_t seed, float* dW1, float* db1, float* dW2, int compute,
               void* scratch, void* stream);
/* *seed = lcg(*seed) on the stream (one tiny kernel): the per-step dropout seed of a replayed hipGraph. */
int egx_seed_advance(uint64_t* seed, void* stream);

/* ---- training step around the translator ------------------------------------------------
 * loss = sum_i w[y_i] * nll_i / sum_i w[y_i] with nll_i = -log_softmax(logits_i)[y_i]  (weight NULL: w = 1), and
 * d_logits = d loss / d logits when d_logits != NULL - in one launch. logits (B, C) fp32, target (B) int64, weight (C).
 * Replaces nn.CrossEntropyLoss(weight=[0.266, 0.734]) and its backward, HHI/tasks/ttm/video_task_2loader.py:21-22,34. */
int egx_weighted_ce(const float* logits, const int64_t* target, const float* weight, int B, int C, float* loss,
                    float* d_logits, void* stream);
/* Classifier head of the ASD task in one launch each way: logits = x W^T + b; loss = weighted CE(logits, target) as above;
 * probs = softmax(logits) (optional), pred_label = round(probs[:, 1]) (M, optional), *correct = number of rows with
 * pred_label == target (optional), d_logits = d loss / d logits (optional). x (M, K) fp32, W (C, K), b (C) or NULL, C <= 8, K a multiple of 64. Replaces lossAV.forward,
 * HHI/tasks/asd/loss.py:11-30 (nn.Linear(dim, 2) + nn.CrossEntropyLoss(weight=[1, 4]) + softmax / round / count), whose
 * separate launches cost 15 % of the ASD translator step. `scratch`: egx_linear_ce_scratch(M, K, C) bytes, ZERO before the
 * first use (arrival counters; every launch leaves them zero again). Sums run in a fixed order (deterministic).
 * Backward: dx = g * d_logits W (optional), dW = g * d_logits^T x, db = g * colsum(d_logits) (db optional), g = *grad_scale
 * (device scalar, NULL = 1). K in {64, 128, 256}. */
size_t egx_linear_ce_scratch(int M, int K, int C);
int egx_linear_ce_fwd(const float* x, const float* W, const float* b, const int64_t* target, const float* weight, int M, int K,
                      int C, float* logits, float* probs, float* d_logits, float* loss, float* correct, float* pred_label,
                      void* scratch, void* stream);
int egx_linear_ce_bwd(const float* x, const float* W, const float* d_logits, const float* grad_scale, int M, int K, int C,
                      float* dx, float* dW, float* db, void* scratch, void* stream);
/* *counter += inc on the stream (device-resident step count of the optimizer). */
int egx_counter_add(int64_t* counter, int64_t inc, void* stream);
/* One Adam (decoupled = 0) / AdamW (decoupled = 1) update of n fp32 elements with torch.optim semantics:
 *   g = grad * grad_scale (+ weight_decay * p if !decoupled);  p *= 1 - lr * weight_decay if decoupled;
 *   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * with t = *step (1-based, device memory: advance it with egx_counter_add first, which also makes the update
 * replayable inside a hipGraph). Replaces torch.optim.Adam(lr=5e-4) HHI/tasks/ttm/video_task_2loader.py:62-64 and
 * AdamW(lr=1e-4, weight_decay=1e-4) HOI/tasks/multitask/video_task.py:624-626 over a flat parameter buffer. */
int egx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, const int64_t* step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                  void* stream);
/* ---- EgoT2-g sequence decoder + vocabulary head --------------------------------------------
 * decode() of HHI/models/multitask/task_prompt_model.py:260-269 and HOI/models/multitask/video_model_builder.py:150-159:
 * embedding * sqrt(d) + positional encoding -> nn.TransformerDecoder of CustomDecoderLayer (causal self-attention over
 * the 2..5 target tokens, cross-attention onto the encoder memory, FFN; post-LN) -> fc. Projections, LayerNorms and the
 * FFN reuse egx_linear_* / egx_layernorm_*; the entry points below are the decoder-specific pieces.
 *
 * Attention of a few queries against a short key set, one (batch element, head) per wave: rows are tokens (row index
 * b * S + s, row stride ld* floats), head h owns columns [h * dh, (h + 1) * dh). Sq <= 8, Sk <= 1024 (one wave per (b, h) up to 64 keys, four waves and chunked K / V beyond), dh <= 128.
 * causal != 0 (needs Sq == Sk) applies the reference's lower-triangular target mask. Self-attention passes the packed
 * qkv rows three times (q, q + d, q + 2d with ld = 3d), cross-attention q and the packed kv rows of the memory.
 * p_drop > 0: dropout on the probabilities, keyed by (seed, site); the backward regenerates the same mask and
 * recomputes the probabilities (nothing is saved). */
int egx_small_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                            int B, int Sq, int Sk, int H, int dh, int causal, float p_drop, uint64_t seed, uint32_t site,
                            void* stream);
/* dq / dk / dv use the strides of q / k / v and are overwritten (=), every (row, head column) exactly once. */
int egx_small_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* d_o,
                            int ldo, float* dq, float* dk, float* dv, int B, int Sq, int Sk, int H, int dh, int causal,
                            float p_drop, uint64_t seed, uint32_t site, void* stream);
/* out[b, t, :] = dropout(emb[tokens[b, t]] * scale + pe[t * pe_stride ...]); tokens (B, sy) int64, emb (V, d).
 * Backward: d_emb[tokens[b, t]] += scale * mask * dy[b, t]  (atomic accumulation into a zero-filled or live buffer). */
int egx_embed_pos_fwd(const int64_t* tokens, const float* emb, const float* pe, int pe_stride, float scale, float* out,
                      int B, int sy, int d, int V, float p_drop, uint64_t seed, void* stream);
int egx_embed_pos_bwd(const int64_t* tokens, const float* dy, float* d_emb, float scale, int B, int sy, int d, int V,
                      float p_drop, uint64_t seed, void* stream);
/* ---- the whole decoder as ONE call per direction (round 3) ----------------------------------------------------------
 * Same reference lines as above (task_prompt_model.py:260-269, video_model_builder.py:150-159; CustomDecoderLayer
 * task_prompt_model.py:163-172). compute = EGX_BF16 only: the B * sy target rows and the B * S memory rows run through the
 * bf16 MFMA GEMMs of the wide path with fused epilogues, the LayerNorms through its row kernels, the two attentions through
 * a register-resident kernel (one wave per (clip, head)). d_model a multiple of 128 in [256, 1024], head dim 32 or 64,
 * sy <= 8 target tokens, S <= 1024 memory tokens per clip; other configurations return an error (the caller composes the
 * decoder from the entry points above instead). Parameters are torch-layout fp32 ([out, in] row-major); `ca_in_w` /
 * `ca_in_b` are the packed in-projection of the cross-attention (rows [0, d): query, rows [d, 3d): key / value). */
typedef struct egx_dec_config {
    int d_model, n_heads, d_ff, n_layers;
    int vocab;          /* |V|: rows of the embedding, outputs of fc */
    int sy, S;          /* target tokens per clip, memory tokens per clip */
    float ln_eps;
    int compute;        /* EGX_BF16 */
    float p_drop;       /* dropout of the decoder layers (attention probabilities, the three residual branches, FFN hidden) */
    float p_pos;        /* dropout of the positional encoding */
    const uint64_t* seed_ptr;   /* optional DEVICE pointer (the translator's egx_config.seed_ptr): dropout keys are derived on the stream
                                   from *seed_ptr instead of the host `seed`, so a captured hipGraph draws fresh masks per replay */
} egx_dec_config;
typedef struct egx_dec_layer {
    const float* sa_in_w; const float* sa_in_b; const float* sa_out_w; const float* sa_out_b; const float* norm1_w; const float* norm1_b;
    const float* ca_in_w; const float* ca_in_b; const float* ca_out_w; const float* ca_out_b; const float* norm2_w; const float* norm2_b;
    const float* lin1_w; const float* lin1_b; const float* lin2_w; const float* lin2_b; const float* norm3_w; const float* norm3_b;
} egx_dec_layer;
typedef struct egx_dec_layer_grads {     /* accumulated (+=); any may be NULL */
    float* sa_in_w; float* sa_in_b; float* sa_out_w; float* sa_out_b; float* norm1_w; float* norm1_b;
    float* ca_in_w; float* ca_in_b; float* ca_out_w; float* ca_out_b; float* norm2_w; float* norm2_b;
    float* lin1_w; float* lin1_b; float* lin2_w; float* lin2_b; float* norm3_w; float* norm3_b;
} egx_dec_layer_grads;
int egx_decoder_workspace(const egx_dec_config* cfg, int B, size_t* saved_bytes, size_t* scratch_bytes);
/* tokens (B, sy) int64; memory (B * S, d) fp32, rows b * S + s (the encoder output, batch-first); emb (vocab, d); pe rows at
 * pe + t * pe_stride; logits (B * sy, vocab) fp32 out. `saved` is read by the backward; `scratch` is free afterwards. */
int egx_decoder_fwd(const egx_dec_config* cfg, const int64_t* tokens, const float* memory, const float* emb, const float* pe, int pe_stride,
                    const egx_dec_layer* layers, const float* fc_w, const float* fc_b, int B, float* logits, void* saved, void* scratch,
                    int training, uint64_t seed, void* stream);
/* d_memory (B * S, d) is overwritten (=); every parameter gradient is accumulated (+=) — `zero_buf` / `zero_bytes` (optional):
 * a buffer the first operation of the call zero-fills (the caller's flat gradient buffer holding all += targets).
 * Both calls are asynchronous on `stream`; internally the K | V projections (forward) and the weight gradients (backward) run on one
 * library-owned side stream that is forked off and joined back inside the call (also under stream capture). */
int egx_decoder_bwd(const egx_dec_config* cfg, const int64_t* tokens, const egx_dec_layer* layers, const float* fc_w, int B, const float* d_logits,
                    const void* saved, void* scratch, float* d_memory, float* d_emb, const egx_dec_layer_grads* grads, float* d_fc_w,
                    float* d_fc_b, void* zero_buf, size_t zero_bytes, int training, uint64_t seed, void* stream);
/* dy[i] = y[i] > 0 ? dy[i] : 0 in place: backward of the ReLU fused into egx_linear_fwd(relu = 1). */
int egx_relu_mask(float* dy, const float* y, size_t n, void* stream);
/* Producer side of the feature hand-off: the `middle=True` head of the frozen PNR / OSCC backbones,
 * ResNetKeyframeLocalizationHead.forward (HOI/models/pnr/head_helper.py:353-373) = AvgPool3d((kt, kh, kw), stride 1) of the res5
 * map fmap (N, C, T, H, W) [fp32, or bf16 with fmap_bf16], permute to (N, T', H', W', C), reshape to rows of H' W' C columns
 * (8192 = 2 * 2 * 2048) - optionally followed by the per-clip temporal mean of encode_clips_pnr
 * (HOI/models/lta/lta_models_lta_transfer.py:335-345; frames_mean != 0, or kt == T) - written as packed fp32 / bf16 rows:
 * out[n * out_map_stride + t' * (H' W' C) + (h' W' + w') * C + c]. With out_map_stride = n_clips * H' W' C and `out` offset by
 * i * H' W' C, call i fills token i of every sample of a (B, n_clips, 8192) translator input in place. kt must be 1 or T. */
int egx_pool_pack(const void* fmap, int fmap_bf16, int N, int C, int T, int H, int W, int kt, int kh, int kw, int frames_mean,
                  void* out, int out_bf16, long long out_map_stride, void* stream);

/* ---- gradient exchange -----------------------------------------------------------------------------
 * RCCL (ncclAllReduce over xGMI) behind plain pointers, for callers that do not use torch.distributed. Replaces what the reference
 * gets from Lightning DDP's reducer (HOI/scripts/multitask/run.py:41-50: strategy DDP, `find_unused_parameters`; HHI: the
 * `gpus=-1, accelerator='ddp'` trainers): ONE all-reduce, sum then 1 / n, over the flat gradient buffer every backward of this
 * library writes its gradients into (2.77 MB for the TTM translator, 181 MB for the LTA 4-task one).
 *   egx_comm_unique_id : rank 0 draws the 128-byte RCCL id; the caller carries it to the other ranks.
 *   egx_comm_create    : one communicator per process, on the calling thread's current HIP device (one rank per GPU).
 *   egx_allreduce      : in place over `n` elements of `buf` (dtype 0 = fp32, 1 = bf16), sum or average, asynchronous on `stream`
 *                        (hipGraph-capturable: RCCL kernels are ordinary stream work). Every rank must call it with the same n.
 *   egx_comm_size      : the rank count RCCL itself reports for the communicator (-1 on error).
 *   egx_comm_library   : the RCCL library that was resolved ("" when none): the copy already loaded in the process if there is
 *                        one (PyTorch's), else librccl.so(.1) / $EGX_RCCL_LIB. The library has no link-time dependency on RCCL.
 * egot2_amd/ddp.py keeps torch.distributed (backend "nccl" = RCCL) for the nn.Module path — Lightning's DDP hands it a process group —
 * and offers `EgxComm` over these calls for harnesses without one. */
typedef struct egx_comm egx_comm;
int egx_comm_unique_id(void* id128);
int egx_comm_create(const void* id128, int rank, int world, egx_comm** out);
int egx_comm_size(const egx_comm* comm);
int egx_allreduce(egx_comm* comm, void* buf, size_t n, int dtype, int average, void* stream);
int egx_comm_destroy(egx_comm* comm);
const char* egx_comm_library(void);

/* x[r, c] *= keep(seed, site, r, c) / (1 - p) in place (inverted dropout); calling it on the gradient with the same
 * (seed, site) is its backward. */
int egx_dropout(float* x, int rows, int cols, float p_drop, uint64_t seed, uint32_t site, void* stream);

/* Per-kernel device timing for bench.py's roofline block: hipEvents recorded on the launch stream around the
 * fused kernels while enabled (which: 0 = fused forward, 1 = fused per-clip backward, 2 = FFN weight gradients).
 * egx_timing_read synchronises on the recorded events; never call it inside a timed or captured region. */
void egx_timing_enable(int on);
int egx_timing_read(int which, double* total_ms, int* count);
/* Sliced mode (small batches, egx_encoder_slices() > 1) diagnostic: how many FFN slices waiting workgroups have computed themselves since the
 * last reset because a partner workgroup's partial sum did not arrive within 100 us (a busy GPU shared with another process — or a stray
 * EGX_SLICE_DROP in the environment). 0 on a quiet GPU; every count is ~100 us of waiting plus a recomputed slice. Synchronous (reads device
 * memory): never inside a timed or captured region. -1 on error. */
long long egx_slices_stolen(int reset);
/* development aid: phase timestamps of the fused forward (only meaningful in -DEGX_STAMPS builds) */
int egx_debug_stamps(unsigned long long* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* EGOT2X_H */
