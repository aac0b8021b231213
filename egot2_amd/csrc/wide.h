// "Wide" bf16 path of the translator: d_model >= 256 configurations (BASELINE.json configs[3] HOI LTA 4-task d = 768,
// configs[4] EgoT2-g encoders d = 256 / 512) where the dense layers are large GEMMs over all B*S tokens.
//
// Storage: GEMM operands (layer inputs, qkv, attention output, FFN hidden, upstream gradients) live in HBM as bf16;
// the residual stream, LayerNorm inputs and statistics, biases and all parameter gradients stay fp32. Weights are cast
// to bf16 once per forward (W for the forward, W^T for the input gradients).
#pragma once
#include "common.h"

namespace egx {

typedef unsigned short bf16_t;

struct WideGemmParams {
    // NT:  C[m][n] = sum_k A[m][k] * B[n][k]      A: activations (M, lda), B: weights (N, ldb), both k-contiguous
    // TN:  C[m][n] = sum_k A[k][m] * B[k][n]      A: (K, lda) upstream gradient dY, B: (K, ldb) layer input (K = tokens)
    const bf16_t* A = nullptr; const bf16_t* B = nullptr;
    int M = 0, N = 0, K = 0, lda = 0, ldb = 0;
    float* Cf = nullptr; bf16_t* Cb = nullptr; int ldc = 0;      // fp32 and / or bf16 output
    const float* bias = nullptr;                                 // [N]
    int relu = 0;
    uint64_t drop_key = 0; uint32_t drop_thresh = 0; float drop_inv = 1.f;   // inverted dropout keyed by (m, n)
    const float* residual = nullptr; int ldr = 0;                // fp32 (M, ldr), added last
    const bf16_t* mask = nullptr; int ldm = 0; float mask_scale = 1.f;       // C = mask[m][n] != 0 ? C * mask_scale : 0
    float* colsum = nullptr;      // optional [wide_gemm_nt_colsum_rows(M, N)][N]: column sums per 64 output rows (bias gradients)
    int accumulate = 0;           // TN: C += result (parameter gradients)
    int tn_max_splits = 0;        // TN: upper bound on the split-K count (0: the cost model's choice). 1 with accumulate == 0 and a dense C
                                  // (ldc == N) writes C directly: no slab, no reduction (launches that need no parallelism: side stream)
    const void* zero_page = nullptr;   // >= 256 zero bytes in device memory (source of out-of-range operand rows)
    int epi_lds = 0;              // NT: set by wide_gemm_nt (EGX_WIDE_EPI): epilogue through LDS
};

// scratch for the TN split-K slabs
size_t wide_gemm_tn_scratch(int M, int N, int K);
int wide_gemm_nt(const WideGemmParams& p, hipStream_t st);
int wide_gemm_nt_colsum_rows(int M, int N);     // rows of WideGemmParams::colsum written for an (M, N) output
// `defer`: do not launch the slab reduction, queue it (then `scratch` must be a region of its own, wide_gemm_tn_scratch bytes,
// untouched until wide_reduce_flush() has run)
struct WideReduceDesc { const float* slabs; float* C; size_t stride; int splits, ldc, M, N, accumulate, first_block, blocks; };
constexpr int WIDE_REDUCE_MAX = 40;
struct WideReduceBatch { WideReduceDesc d[WIDE_REDUCE_MAX]; int n = 0, total_blocks = 0; };
int wide_gemm_tn(const WideGemmParams& p, void* scratch, hipStream_t st, WideReduceBatch* defer = nullptr);
// grouped weight-gradient launches (wide_gemm.hip): independent TN problems queued by tile variant, one grid per variant at the flush
struct WideTnDesc { const bf16_t* A; const bf16_t* B; float* slabs; size_t slab_stride; int M, N, K, lda, ldb, ntM, ntN, kps, splits, first_block; };
constexpr int WIDE_TN_GROUP_MAX = 40;
struct WideTnGroup { WideTnDesc d[WIDE_TN_GROUP_MAX]; const void* zero_page = nullptr; int n = 0, epi_lds = 0, by_slice = 0; };
struct WideTnQueue { WideTnGroup g[3]; int blocks[3] = {0, 0, 0}; };
int wide_tn_queue_add(WideTnQueue& Q, const WideGemmParams& p, void* scratch, hipStream_t st, WideReduceBatch* defer);
int wide_tn_queue_flush(WideTnQueue& Q, hipStream_t st);
int wide_reduce_flush(WideReduceBatch& b, hipStream_t st);

// dst[r][c] = bf16(src[r][c]) and / or dst_t[c][r] = bf16(src[r][c]); src (R, ld) fp32
int wide_cast(const float* src, int R, int C, int ld, bf16_t* dst, bf16_t* dst_t, hipStream_t st);
// the same for many matrices in one launch: wide_cast_add() queues (and flushes a full batch), wide_cast_flush() launches
struct WideCastDesc { const float* src; bf16_t* dst; bf16_t* dst_t; int R, C, ld, first_block; };
constexpr int WIDE_CAST_MAX = 40;
struct WideCastBatch { WideCastDesc d[WIDE_CAST_MAX]; int n = 0, blocks = 0; };
int wide_cast_add(WideCastBatch& b, const float* src, int R, int C, int ld, bf16_t* dst, bf16_t* dst_t, hipStream_t st);
int wide_cast_flush(WideCastBatch& b, hipStream_t st);

// dst[r][c] = bf16(mean_{j < pool} src[r * pool + j][c]); src fp32 or bf16 (R * pool, C), dst (R, C) bf16. C % 8 == 0.
int wide_pool_cast(const void* src, int src_bf16, int R, int pool, int C, bf16_t* dst, hipStream_t st);

// Row kernels with bf16 side outputs --------------------------------------------------------------------------------
// y[orow] = dropout(LN(x[row]) * w + b + add_vec + pos[t]); y32 and / or y16 written; stats (mean, rstd) saved.
struct WideLnFwdParams {
    const float* x = nullptr; const float* w = nullptr; const float* b = nullptr; float eps = 1e-5f;
    float* stats = nullptr; float* y32 = nullptr; bf16_t* y16 = nullptr;
    int rows = 0, d = 0, T = 1, S = 1, off = 0;
    const float* add_vec = nullptr; const float* pos = nullptr; int pos_stride = 0;
    uint64_t drop_key = 0; uint32_t drop_thresh = 0; float drop_inv = 1.f;
};
int wide_ln_fwd(const WideLnFwdParams& p, hipStream_t st);

// dx[row] = LN backward of (mask .* dy[orow]); dx32 = d(pre-LN sum) fp32; dx16 = out-mask .* dx as bf16 (the upstream
// gradient of the GEMM that produced the branch; out-mask = the dropout applied to that branch in the forward, keyed by
// (row, col)). Per-block partial sums of d(gamma), d(beta) [and of dx16's columns = the branch's bias gradient] go to
// `partials` ([blocks][3][d]) and are reduced in fixed order by wide_reduce_partials (deterministic).
struct WideLnBwdParams {
    const float* dy = nullptr; const float* pre = nullptr; const float* stats = nullptr; const float* w = nullptr;
    float* dx32 = nullptr; bf16_t* dx16 = nullptr;
    int rows = 0, d = 0, T = 1, S = 1, off = 0;
    uint64_t drop_key = 0; uint32_t drop_thresh = 0; float drop_inv = 1.f;           // mask on dy (dropout after LN)
    uint64_t out_key = 0; uint32_t out_thresh = 0; float out_inv = 1.f;              // mask on dx16 / dx32_masked
    int mask_dx32 = 0;            // 1: dx32 also gets the out-mask (feature dropout before the shared LN)
    float* partials = nullptr; int blocks = 0;   // set by wide_ln_bwd
    float* dw = nullptr; float* db = nullptr; float* dbias = nullptr; float* dadd = nullptr;   // += targets (any may be null)
};
size_t wide_ln_bwd_scratch(int rows, int d);
// Deferred row reductions (round 5): the second stage of every two-stage column sum of a backward — LayerNorm parameter gradients,
// bias gradients from GEMM epilogues and from bf16 arrays — was a launch of its own (4.6-5.1 us each, 35 of them per EgoT2-g HHI
// step: 7 % of it). With a batch they are queued and summed by ONE launch (wide_row_reduce_flush); every queued reduction needs a
// partial buffer of its own that stays untouched until the flush.
struct WideRowReduceDesc {
    const float* part; float* o0; float* o1; float* o2; float* o3;      // kind 0: out o0 (ld / row_len); kind 1 (LayerNorm): dw, db, dbias, dadd
    int nt, cols, d, kind, first_block, out_ld, row_len, pad_;
};
constexpr int WIDE_ROWRED_MAX = 48;
struct WideRowReduceBatch { WideRowReduceDesc d[WIDE_ROWRED_MAX]; int n = 0, total_blocks = 0; };
int wide_row_reduce_flush(WideRowReduceBatch& b, hipStream_t st);
int wide_ln_bwd(WideLnBwdParams p, void* scratch, hipStream_t st, WideRowReduceBatch* defer = nullptr);

// out[c] += sum_r x[r][c] over a bf16 (rows, ld) matrix, deterministic two-stage reduction; scratch >= wide_colsum_scratch
size_t wide_colsum_scratch(int rows, int cols);
int wide_colsum_bf16(const bf16_t* x, int rows, int cols, int ld, float* out, void* scratch, hipStream_t st, WideRowReduceBatch* defer = nullptr);
// out[c] += sum_t part[t][c], t < nt (fixed order)
int wide_reduce_rows(const float* part, int nt, int cols, float* out, hipStream_t st, WideRowReduceBatch* defer = nullptr);

// dpos[t * pos_stride + c] += sum_b mask .* dtok[(b * S + off + t) * d + c] (learned positional table), fixed-order chunk sums
size_t wide_pos_grad_scratch(int B, int T, int d);
int wide_pos_grad(const float* dtok, int B, int S, int off, int T, int d, float* dpos, int pos_stride, uint64_t key, uint32_t thresh,
                  float inv, void* scratch, hipStream_t st);

// Attention over packed bf16 qkv rows (B*S, 3d): out (B*S, d) bf16, lse (B, H, S) fp32. S <= 128, head dim in {32, 64, 96, 128}.
struct WideAttnParams {
    const bf16_t* qkv = nullptr; bf16_t* out = nullptr; float* lse = nullptr;
    const bf16_t* d_out = nullptr; bf16_t* d_qkv = nullptr;
    int B = 0, S = 0, H = 0, d = 0;
    uint64_t drop_key = 0; uint32_t drop_thresh = 0; float drop_inv = 1.f;
    float* delta = nullptr;       // S > 128 backward: (B, H, S) scratch, delta = rowsum(dO . O); `out` must then hold the forward's output
};
bool wide_attn_supported(int S, int dh);      // S <= 128: head dim 32 / 64 / 96 / 128; 128 < S <= ~480: head dim 32 / 64 (wide_attn_long_*)
size_t wide_attn_delta_bytes(int B, int H, int S);
int wide_attn_fwd(const WideAttnParams& p, hipStream_t st);
int wide_attn_bwd(const WideAttnParams& p, hipStream_t st);

}  // namespace egx
