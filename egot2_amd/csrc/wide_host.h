// Entry points of the wide bf16 path used by encoder.hip (see wide_host.hip).
#pragma once
#include "../../include/egot2x.h"
#include "common.h"

namespace egx {

// bf16 compute, d_model >= 256, d_model / d_ff / projected d_in multiples of 128, S <= 128, head dim 32 / 64 / 96 / 128
bool wide_ok(const egx_config* cfg, const egx_segment* segs, int B);
void wide_workspace(const egx_config* cfg, const egx_segment* segs, int B, size_t* saved, size_t* scratch);
int wide_encoder_fwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const float* ln_b, const egx_layer* layers,
                     int B, float* tokens_out, void* saved, int training, uint64_t seed, hipStream_t st);
int wide_encoder_bwd(const egx_config* cfg, const egx_segment* segs, const float* ln_w, const egx_layer* layers, int B,
                     const float* d_tokens, const void* saved, void* scratch, const egx_segment_grads* seg_grads, float* d_ln_w,
                     float* d_ln_b, const egx_layer_grads* layer_grads, int training, uint64_t seed, hipStream_t st);

}  // namespace egx
