// Internal launch interfaces shared by the translation units of libegot2x.
#pragma once
#include "common.h"

namespace egx {

struct GemmParams {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0;
    const float* bias = nullptr;       // [N]
    const float* residual = nullptr;   // [M, ldr], added last
    int ldr = 0;
    const float* mask = nullptr;       // [M, ldm]; C = mask > 0 ? C * mask_scale : 0   (ReLU / dropout backward)
    int ldm = 0;
    float mask_scale = 1.f;
    int relu = 0;
    uint64_t drop_key = 0;             // inverted dropout on (acc + bias [relu]) keyed by (row, col)
    uint32_t drop_thresh = 0;
    float drop_inv_keep = 1.f;
    int atomic = 0;                    // split-K partials are atomically added into C (small weight gradients)
    int k_chunk = 0;                   // set by gemm()
    size_t slab_stride = 0;            // set by gemm()
};

// Deterministic mode of the shape-generic kernels (egx_config.deterministic): while a DetScope is alive on this thread, the
// cross-workgroup sums that normally use fp32 atomics (split-K weight gradients, LayerNorm / bias / head parameter gradients)
// go through partial buffers in `buf` and fixed-order reductions instead. `buf` must hold generic_det_scratch_bytes().
struct DetScope {
    DetScope(void* buf, size_t bytes);
    ~DetScope();
};
bool det_on();
size_t generic_det_scratch_bytes(int B, int d, int d_ff);

// layout: 0 NT, 1 NN, 2 TN (see gemm.hip). compute: 0 fp32 MFMA, 1 bf16 MFMA.
int gemm(int layout, GemmParams p, int compute, int accumulate, void* scratch, size_t scratch_bytes, hipStream_t st);
size_t gemm_scratch_bytes(int layout, int M, int N, int K);

// y[orow] = dropout(LN(x[row] (+ res[row])) * w + b) (+ add_vec) (+ pos[t]); row -> orow = (row / T) * S + off + row % T.
struct LnFwdParams {
    const float* x = nullptr;     // [rows, d]
    const float* res = nullptr;   // [rows, d] or null
    const float* w = nullptr;
    const float* b = nullptr;
    float eps = 1e-5f;
    float* pre = nullptr;         // [rows, d] x + res written when non-null
    float* stats = nullptr;       // [rows, 2] (mean, rstd) when non-null
    float* y = nullptr;
    int rows = 0, d = 0;
    int T = 1, S = 1, off = 0;    // output row remap (T == S == 1... identity when T == S and off == 0)
    const float* add_vec = nullptr;
    const float* pos = nullptr;
    int pos_stride = 0;
    uint64_t drop_key = 0;
    uint32_t drop_thresh = 0;
    float drop_inv_keep = 1.f;
};
int layernorm_fwd(const LnFwdParams& p, hipStream_t st);

// dx[row] = LN backward of dy[orow] (dropout mask re-applied to dy first); dw/db accumulated (+=).
struct LnBwdParams {
    const float* dy = nullptr;    // rows addressed through the remap
    const float* pre = nullptr;   // [rows, d]
    const float* stats = nullptr; // [rows, 2]
    const float* w = nullptr;
    float* dx = nullptr;          // [rows, d]
    float* dw = nullptr;          // [d] +=
    float* db = nullptr;          // [d] +=
    float* dadd = nullptr;        // [d] += sum of (masked) dy rows (task-embedding gradient) or null
    int rows = 0, d = 0;
    int T = 1, S = 1, off = 0;
    uint64_t drop_key = 0;        // mask on dy (dropout applied after LN in forward)
    uint32_t drop_thresh = 0;
    float drop_inv_keep = 1.f;
    uint64_t out_drop_key = 0;    // mask on dx (dropout applied before LN in forward: HOI feature dropout)
    uint32_t out_drop_thresh = 0;
    float out_drop_inv_keep = 1.f;
};
int layernorm_bwd(const LnBwdParams& p, hipStream_t st);

// out[c] += sum_r x[r * ld + c], r < rows, c < cols
int colsum_accum(const float* x, int rows, int cols, int ld, float* out, hipStream_t st);
// learned positional gradient: dpos[t * pos_stride + c] += sum_b dtok[(b * S + off + t) * d + c]  (dropout mask re-applied)
int pos_grad_accum(const float* dtok, int B, int S, int off, int T, int d, float* dpos, int pos_stride,
                   uint64_t drop_key, uint32_t drop_thresh, float drop_inv_keep, hipStream_t st);
// x[i] *= dropmask(row, col) for a dense [rows, d] tensor
int apply_dropout_mask(float* x, int rows, int d, uint64_t key, uint32_t thresh, float inv_keep, hipStream_t st);

int attention_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int d,
                  uint64_t drop_key, uint32_t drop_thresh, float drop_inv_keep, hipStream_t st);
int attention_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, float* d_qkv,
                  int B, int S, int H, int d,
                  uint64_t drop_key, uint32_t drop_thresh, float drop_inv_keep, hipStream_t st);

int pool_head_fwd(const float* tokens, int B, int S, int d, const float* ln_w, const float* ln_b, float eps,
                  const float* W, const float* b, int n_out, float* pooled, float* out, hipStream_t st);
int pool_head_bwd(const float* d_out, const float* pooled, int B, int S, int d, const float* ln_w,
                  const float* ln_b, float eps, const float* W, int n_out, float* d_tokens, float* d_ln_w,
                  float* d_ln_b, float* d_W, float* d_b, hipStream_t st);

// train.hip
int weighted_ce(const float* logits, const int64_t* target, const float* weight, int B, int C, float* loss,
                float* dlogits, hipStream_t st);
int counter_add(int64_t* c, int64_t inc, hipStream_t st);
size_t linear_ce_scratch_bytes(int M, int K, int C);
int linear_ce_fwd(const float* x, const float* W, const float* b, const int64_t* target, const float* weight, int M, int K, int C,
                  float* logits, float* probs, float* dlogits, float* loss, float* correct, float* pred, void* scratch, hipStream_t st);
int linear_ce_bwd(const float* x, const float* W, const float* dlogits, const float* gscale, int M, int K, int C, float* dx,
                  float* dW, float* db, void* scratch, hipStream_t st);
int adam_step(float* p, const float* g, float* m, float* v, size_t n, const int64_t* step, float lr, float b1, float b2,
              float eps, float wd, int decoupled, float grad_scale, hipStream_t st);

// decoder.hip
struct SmallAttnParams {
    const float* q; const float* k; const float* v;
    int ldq, ldk, ldv;
    float* o; int ldo;                  // forward output
    const float* d_o;                   // backward input (same ld as o)
    float* dq; float* dk; float* dv;    // backward outputs (same ld as q / k / v)
    int B, Sq, Sk, H, dh, causal;
    float scale;
    uint64_t drop_key; uint32_t drop_thresh; float drop_inv;
};
int small_attention_fwd(SmallAttnParams p, hipStream_t st);
int small_attention_bwd(SmallAttnParams p, hipStream_t st);
int embed_pos_fwd(const int64_t* tok, const float* emb, const float* pe, int pe_stride, float scale, float* out, int B, int sy,
                  int d, int V, uint64_t key, uint32_t thresh, float inv, hipStream_t st);
int embed_pos_bwd(const int64_t* tok, const float* dy, float* d_emb, float scale, int B, int sy, int d, int V, uint64_t key,
                  uint32_t thresh, float inv, hipStream_t st);
int relu_mask(float* dy, const float* y, size_t n, hipStream_t st);
int gelu_fwd(const float* z, float* h, size_t n, hipStream_t st);
int gelu_bwd(const float* z, const float* dh, float* dz, size_t n, hipStream_t st);

// Device-resident dropout seed outside the per-clip kernels: table[(layer - layer0) * 8 + site] = site_key(*seed, layer, site) for
// `nlayer` layers, computed on the stream by one small launch (advance != 0: *seed = lcg(*seed) first - the training forward).
// Kernels then receive the ADDRESS of their key (common.h resolve_key), so a captured hipGraph draws fresh masks per replay.
constexpr int DROP_KEY_SLOTS = 8;
int derive_keys(uint64_t* seed, uint64_t* table, uint32_t layer0, int nlayer, int advance, hipStream_t st);
static inline uint64_t key_slot(const uint64_t* table, uint32_t layer_rel, uint32_t site) {
    return (uint64_t)(uintptr_t)(table + (size_t)layer_rel * DROP_KEY_SLOTS + site);
}

}  // namespace egx
