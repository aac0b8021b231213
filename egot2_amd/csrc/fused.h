// Parameter blocks of the fused per-clip kernels (fused.hip), passed by value as kernel arguments.
#pragma once
#include "common.h"

namespace egx {

constexpr int FUSED_MAX_SEG = 4;
constexpr int FUSED_MAX_LAYERS = 4;

struct FusedSeg {
    const float* feat;      // (B, T, d_in)
    const void* proj_wp;    // [128, d_in] packed in fragment order (pack_weights)
    const float* proj_b;    // [128]
    const float* add_vec;   // [128] or null
    const float* pos;       // rows of 128 with stride pos_stride, or null
    int T, d_in, off, pos_stride;
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

struct FusedFwdParams {
    FusedSeg seg[FUSED_MAX_SEG];
    FusedLayer layer[FUSED_MAX_LAYERS];
    const float* ln_w; const float* ln_b;
    float eps;
    int nseg, n_layers, B, S, d_ff;
    float* tokens_out;      // (B, S, 128)
    float* saved_pre;       // (B, S, 128)   projected features before the shared LN (token order)
    float* saved_res;       // (2L, B, S, 128) pre-LN residual sums: [2l] = res1, [2l+1] = res2
    uint64_t pos_key; uint32_t pos_thresh; float pos_inv;
};

// one matrix to rewrite into MFMA-fragment order (A operand, rows = M dimension); transpose reads src[k][row]
struct PackDesc {
    const float* src; void* dst;
    int R, K, ld, transpose, first_block;
};
constexpr int PACK_MAX = 40;
struct PackParams {
    PackDesc d[PACK_MAX];
    int n, bf16;
};
int pack_weights(PackParams& pp, hipStream_t st);
static inline size_t packed_bytes(int R, int K, int bf16) { return (size_t)R * K * (bf16 ? 2 : 4); }

bool fused_supported(int d_model, int n_heads, int d_ff, int S, int nseg, const int* d_in, const int* T, const bool* has_proj);
size_t fused_lds_bytes(int NT);
int debug_read_stamps(unsigned long long* out, int n);
int fused_forward(const FusedFwdParams& p, int compute, hipStream_t st);

}  // namespace egx
