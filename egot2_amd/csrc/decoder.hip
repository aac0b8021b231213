// Kernels of the EgoT2-g sequence decoder: the reference decodes a 2-5 token target against the
// encoder memory with a stock nn.TransformerDecoder (HHI/models/multitask/task_prompt_model.py:260-269,
// HOI/models/multitask/video_model_builder.py:150-159). Its GEMMs and LayerNorms reuse gemm.hip / norm.hip; what is
// specific to it lives here:
//   small_attention_fwd/bwd  attention of a FEW queries (Sq <= 8: the target tokens) against Sk <= 64 keys (one wave per (b, h);
//                            64 < Sk <= 1024: four waves, K / V chunked through LDS) with separate
//                            Q and K/V operands: causal self-attention over the target (packed qkv rows) and cross-
//                            attention onto the memory (Q from the target, packed kv rows from the memory projection).
//                            One wave per (batch element, head); probabilities are recomputed in the backward.
//   embed_pos_fwd/bwd        y = embedding[token] * sqrt(d) + pe[position] (+ dropout), and the scatter-add of its gradient.
//   relu_mask                dy <- dy where y > 0 (backward of the ReLU fused into the linear1 GEMM epilogue).
//   gelu_fwd/bwd             exact (erf) GELU of the pre-LN translator's FeedForward (HOI/models/pnr/simple_vit.py:55-65).
#include "common.h"
#include "kernels.h"

namespace egx {

constexpr int SA_MAXQ = 8, SA_MAXK = 64, SA_MAXDH = 128;


__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid = B * H, block = 64. LDS: K, V [Sk][dh + 1]; Q (and dO) [Sq][dh]; P / dS [Sq][64].
template <bool BWD>
__global__ __launch_bounds__(64) void small_attention_kernel(SmallAttnParams p) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int dh = p.dh, LDK = dh + 1;
    float* Ks = sm;
    float* Vs = Ks + SA_MAXK * LDK;
    float* Qs = Vs + SA_MAXK * LDK;
    float* Gs = Qs + SA_MAXQ * dh;        // dO (backward only)
    float* Ps = Gs + SA_MAXQ * dh;        // probabilities after dropout, [Sq][64]
    float* Ds = Ps + SA_MAXQ * 64;        // dS (backward only)
    const float* kb = p.k + (size_t)b * p.Sk * p.ldk + h * dh;
    const float* vb = p.v + (size_t)b * p.Sk * p.ldv + h * dh;
    const float* qb = p.q + (size_t)b * p.Sq * p.ldq + h * dh;
    for (int i = lane; i < p.Sk * dh; i += 64) {
        int j = i / dh, c = i - j * dh;
        Ks[j * LDK + c] = kb[(size_t)j * p.ldk + c];
        Vs[j * LDK + c] = vb[(size_t)j * p.ldv + c];
    }
    for (int i = lane; i < p.Sq * dh; i += 64) {
        int r = i / dh, c = i - r * dh;
        Qs[i] = qb[(size_t)r * p.ldq + c];
        if constexpr (BWD) Gs[i] = p.d_o[((size_t)b * p.Sq + r) * p.ldo + h * dh + c];
    }
    __syncthreads();
    // lane j = key j: scores, softmax across the wave, dropout; one query at a time
    for (int i = 0; i < p.Sq; ++i) {
        const bool live = lane < p.Sk && !(p.causal && lane > i);
        float s = -INFINITY;
        if (live) {
            s = 0.f;
            for (int c = 0; c < dh; ++c) s += Qs[i * dh + c] * Ks[lane * LDK + c];
            s *= p.scale;
        }
        float m = wave_max64(s);
        float e = live ? __expf(s - m) : 0.f;
        float prob = e / wave_sum64(e);
        float mask = 1.f;
        if (p.drop_thresh) mask = drop_scale(p.drop_key, (uint32_t)(blockIdx.x * SA_MAXQ + i), (uint32_t)lane, p.drop_thresh, p.drop_inv);
        Ps[i * 64 + lane] = prob * mask;
        if constexpr (BWD) {
            float dp = 0.f;
            if (live) {
                for (int c = 0; c < dh; ++c) dp += Gs[i * dh + c] * Vs[lane * LDK + c];
                dp *= mask;
            }
            float delta = wave_sum64(prob * dp);
            Ds[i * 64 + lane] = live ? prob * (dp - delta) * p.scale : 0.f;
        }
    }
    __syncthreads();
    // lane c = channel c (and c + 64 for dh > 64)
    for (int c = lane; c < dh; c += 64) {
        if constexpr (!BWD) {
            for (int i = 0; i < p.Sq; ++i) {
                float acc = 0.f;
                for (int j = 0; j < p.Sk; ++j) acc += Ps[i * 64 + j] * Vs[j * LDK + c];
                p.o[((size_t)b * p.Sq + i) * p.ldo + h * dh + c] = acc;
            }
        } else {
            for (int i = 0; i < p.Sq; ++i) {
                float acc = 0.f;
                for (int j = 0; j < p.Sk; ++j) acc += Ds[i * 64 + j] * Ks[j * LDK + c];
                p.dq[((size_t)b * p.Sq + i) * p.ldq + h * dh + c] = acc;
            }
            for (int j = 0; j < p.Sk; ++j) {
                float ak = 0.f, av = 0.f;
                for (int i = 0; i < p.Sq; ++i) {
                    ak += Ds[i * 64 + j] * Qs[i * dh + c];
                    av += Ps[i * 64 + j] * Gs[i * dh + c];
                }
                p.dk[((size_t)b * p.Sk + j) * p.ldk + h * dh + c] = ak;
                p.dv[((size_t)b * p.Sk + j) * p.ldv + h * dh + c] = av;
            }
        }
    }
}

// Memory longer than 64 tokens (EgoT2-g HHI on real TTM / ASD sequences: up to 3 x 150 memory tokens): one workgroup of four
// waves per (batch element, head); K and V pass through ONE 64-row LDS buffer in chunks, all Sq x Sk probabilities stay in LDS.
// Same arithmetic and the same dropout keying (row = block * 8 + query, column = key) as the one-wave kernel.
constexpr int SAL_MAXK = 1024, SAL_NTH = 256;
template <bool BWD>
__global__ __launch_bounds__(SAL_NTH) void long_memory_attention_kernel(SmallAttnParams p) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int dh = p.dh, LDK = dh + 1, Sq = p.Sq, Sk = p.Sk, SKP = (Sk + 63) & ~63;
    float* Cs = sm;                         // K or V chunk [64][dh + 1]
    float* Qs = Cs + 64 * LDK;
    float* Gs = Qs + SA_MAXQ * dh;
    float* Ps = Gs + SA_MAXQ * dh;          // [Sq][SKP]
    float* Ds = Ps + SA_MAXQ * SKP;         // backward only
    const float* kb = p.k + (size_t)b * Sk * p.ldk + h * dh;
    const float* vb = p.v + (size_t)b * Sk * p.ldv + h * dh;
    const float* qb = p.q + (size_t)b * Sq * p.ldq + h * dh;
    auto stage = [&](const float* src, int ld, int j0) {
        __syncthreads();
        for (int i = tid; i < 64 * dh; i += SAL_NTH) {
            const int j = i / dh, c = i - j * dh;
            Cs[j * LDK + c] = j0 + j < Sk ? src[(size_t)(j0 + j) * ld + c] : 0.f;
        }
        __syncthreads();
    };
    for (int i = tid; i < Sq * dh; i += SAL_NTH) {
        const int r = i / dh, c = i - r * dh;
        Qs[i] = qb[(size_t)r * p.ldq + c];
        if constexpr (BWD) Gs[i] = p.d_o[((size_t)b * Sq + r) * p.ldo + h * dh + c];
    }
    // scores: wave w owns queries w, w + 4; lane = key within the chunk
    for (int j0 = 0; j0 < Sk; j0 += 64) {
        stage(kb, p.ldk, j0);
        for (int i = wave; i < Sq; i += 4) {
            float sc = 0.f;
            for (int c = 0; c < dh; ++c) sc += Qs[i * dh + c] * Cs[lane * LDK + c];
            Ps[i * SKP + j0 + lane] = j0 + lane < Sk ? sc * p.scale : -INFINITY;
        }
    }
    __syncthreads();
    for (int i = wave; i < Sq; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < SKP; j += 64) m = fmaxf(m, Ps[i * SKP + j]);
        m = wave_max64(m);
        float sum = 0.f;
        for (int j = lane; j < SKP; j += 64) { const float e = __expf(Ps[i * SKP + j] - m); Ps[i * SKP + j] = e; sum += e; }
        sum = 1.f / wave_sum64(sum);
        for (int j = lane; j < SKP; j += 64) Ps[i * SKP + j] *= sum;        // the plain probabilities (0 beyond Sk)
    }
    if constexpr (BWD) {
        // dP = (dO V^T) .* mask into Ds
        for (int j0 = 0; j0 < Sk; j0 += 64) {
            stage(vb, p.ldv, j0);
            for (int i = wave; i < Sq; i += 4) {
                float dp = 0.f;
                for (int c = 0; c < dh; ++c) dp += Gs[i * dh + c] * Cs[lane * LDK + c];
                Ds[i * SKP + j0 + lane] = dp;
            }
        }
        __syncthreads();
        for (int i = wave; i < Sq; i += 4) {
            float delta = 0.f;
            for (int j = lane; j < SKP; j += 64) {
                float mask = 1.f;
                if (p.drop_thresh) mask = drop_scale(p.drop_key, (uint32_t)(blockIdx.x * SA_MAXQ + i), (uint32_t)j, p.drop_thresh, p.drop_inv);
                const float dp = Ds[i * SKP + j] * mask;
                Ds[i * SKP + j] = dp;
                delta += Ps[i * SKP + j] * dp;
            }
            delta = wave_sum64(delta);
            for (int j = lane; j < SKP; j += 64) {
                float mask = 1.f;
                if (p.drop_thresh) mask = drop_scale(p.drop_key, (uint32_t)(blockIdx.x * SA_MAXQ + i), (uint32_t)j, p.drop_thresh, p.drop_inv);
                const float pr = Ps[i * SKP + j];
                Ds[i * SKP + j] = pr * (Ds[i * SKP + j] - delta) * p.scale;
                Ps[i * SKP + j] = pr * mask;
            }
        }
        // dQ = dS K (accumulated over the chunks); dK = dS^T Q, dV = P^T dO per chunk
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < Sk; j0 += 64) {
            stage(kb, p.ldk, j0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + u * SAL_NTH;
                if (e < Sq * dh) {
                    const int i = e / dh, c = e - i * dh;
                    float a = 0.f;
                    for (int j = 0; j < 64; ++j) a += Ds[i * SKP + j0 + j] * Cs[j * LDK + c];
                    acc[u] += a;
                }
            }
            for (int e = tid; e < 64 * dh; e += SAL_NTH) {
                const int j = e / dh, c = e - j * dh;
                if (j0 + j < Sk) {
                    float ak = 0.f, av = 0.f;
                    for (int i = 0; i < Sq; ++i) {
                        ak += Ds[i * SKP + j0 + j] * Qs[i * dh + c];
                        av += Ps[i * SKP + j0 + j] * Gs[i * dh + c];
                    }
                    p.dk[((size_t)b * Sk + j0 + j) * p.ldk + h * dh + c] = ak;
                    p.dv[((size_t)b * Sk + j0 + j) * p.ldv + h * dh + c] = av;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * SAL_NTH;
            if (e < Sq * dh) { const int i = e / dh, c = e - i * dh; p.dq[((size_t)b * Sq + i) * p.ldq + h * dh + c] = acc[u]; }
        }
    } else {
        if (p.drop_thresh) {
            for (int i = wave; i < Sq; i += 4)
                for (int j = lane; j < SKP; j += 64)
                    Ps[i * SKP + j] *= drop_scale(p.drop_key, (uint32_t)(blockIdx.x * SA_MAXQ + i), (uint32_t)j, p.drop_thresh, p.drop_inv);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < Sk; j0 += 64) {
            stage(vb, p.ldv, j0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + u * SAL_NTH;
                if (e < Sq * dh) {
                    const int i = e / dh, c = e - i * dh;
                    float a = 0.f;
                    for (int j = 0; j < 64; ++j) a += Ps[i * SKP + j0 + j] * Cs[j * LDK + c];
                    acc[u] += a;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * SAL_NTH;
            if (e < Sq * dh) { const int i = e / dh, c = e - i * dh; p.o[((size_t)b * Sq + i) * p.ldo + h * dh + c] = acc[u]; }
        }
    }
}
static size_t long_attn_lds(int dh, int Sk) {
    return ((size_t)64 * (dh + 1) + (size_t)2 * SA_MAXQ * dh + (size_t)2 * SA_MAXQ * ((Sk + 63) & ~63)) * sizeof(float);
}
template <bool BWD>
static int launch_long_memory_attention(const SmallAttnParams& p, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&long_memory_attention_kernel<BWD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)long_attn_lds(SA_MAXDH, SAL_MAXK)));
        attr = true;
    }
    hipLaunchKernelGGL(long_memory_attention_kernel<BWD>, dim3(p.B * p.H), dim3(SAL_NTH), long_attn_lds(p.dh, p.Sk), st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

static size_t small_attn_lds(int dh) {
    return ((size_t)2 * SA_MAXK * (dh + 1) + (size_t)2 * SA_MAXQ * dh + (size_t)2 * SA_MAXQ * 64) * sizeof(float);
}

static int small_attention_check(const SmallAttnParams& p) {
    EGX_CHECK(p.B >= 1 && p.H >= 1, "small_attention: B=%d H=%d", p.B, p.H);
    EGX_CHECK(p.Sq >= 1 && p.Sq <= SA_MAXQ, "small_attention: Sq=%d outside 1..%d (target tokens)", p.Sq, SA_MAXQ);
    EGX_CHECK(p.Sk >= 1 && p.Sk <= SAL_MAXK, "small_attention: Sk=%d outside 1..%d", p.Sk, SAL_MAXK);
    EGX_CHECK(p.dh >= 1 && p.dh <= SA_MAXDH, "small_attention: head dim %d outside 1..%d", p.dh, SA_MAXDH);
    EGX_CHECK(!p.causal || p.Sq == p.Sk, "small_attention: the causal mask needs Sq == Sk");
    return 0;
}

int small_attention_fwd(SmallAttnParams p, hipStream_t st) {
    EGX_CHECK(p.q && p.k && p.v && p.o, "small_attention_fwd: null pointer argument");
    if (small_attention_check(p)) return 1;
    p.scale = 1.f / sqrtf((float)p.dh);
    if (p.Sk > SA_MAXK) return launch_long_memory_attention<false>(p, st);
    size_t lds = small_attn_lds(p.dh);
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_attention_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_attn_lds(SA_MAXDH)));
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_attention_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_attn_lds(SA_MAXDH)));
        attr = true;
    }
    hipLaunchKernelGGL(small_attention_kernel<false>, dim3(p.B * p.H), dim3(64), lds, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

int small_attention_bwd(SmallAttnParams p, hipStream_t st) {
    EGX_CHECK(p.q && p.k && p.v && p.d_o && p.dq && p.dk && p.dv, "small_attention_bwd: null pointer argument");
    if (small_attention_check(p)) return 1;
    p.scale = 1.f / sqrtf((float)p.dh);
    if (p.Sk > SA_MAXK) return launch_long_memory_attention<true>(p, st);
    size_t lds = small_attn_lds(p.dh);
    static bool attr = false;
    if (!attr) {
        EGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_attention_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_attn_lds(SA_MAXDH)));
        attr = true;
    }
    hipLaunchKernelGGL(small_attention_kernel<true>, dim3(p.B * p.H), dim3(64), lds, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- embedding * sqrt(d) + positional encoding ------------------------------------------------------------------
// tokens (B, sy) int64; emb (V, d); pe rows at pe + t * pe_stride; out (B, sy, d) batch-first.
__global__ __launch_bounds__(256) void embed_pos_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb,
                                                        const float* __restrict__ pe, int pe_stride, float scale,
                                                        float* __restrict__ out, int rows, int sy, int d, int V,
                                                        uint64_t key, uint32_t thresh, float inv) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * d) return;
    int row = (int)(i / d), c = (int)(i - (size_t)row * d);
    int64_t t = tok[row];
    float e = (t >= 0 && t < V) ? emb[(size_t)t * d + c] : 0.f;
    float y = e * scale + pe[(size_t)(row % sy) * pe_stride + c];
    if (thresh) y *= drop_scale(key, (uint32_t)row, (uint32_t)c, thresh, inv);
    out[i] = y;
}
__global__ __launch_bounds__(256) void embed_grad_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dy,
                                                         float* __restrict__ d_emb, float scale, int rows, int d, int V,
                                                         uint64_t key, uint32_t thresh, float inv) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * d) return;
    int row = (int)(i / d), c = (int)(i - (size_t)row * d);
    int64_t t = tok[row];
    if (t < 0 || t >= V) return;
    float g = dy[i] * scale;
    if (thresh) g *= drop_scale(key, (uint32_t)row, (uint32_t)c, thresh, inv);
    atomicAdd(d_emb + (size_t)t * d + c, g);
}

int embed_pos_fwd(const int64_t* tok, const float* emb, const float* pe, int pe_stride, float scale, float* out, int B, int sy,
                  int d, int V, uint64_t key, uint32_t thresh, float inv, hipStream_t st) {
    EGX_CHECK(tok && emb && pe && out, "embed_pos_fwd: null pointer argument");
    size_t n = (size_t)B * sy * d;
    hipLaunchKernelGGL(embed_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tok, emb, pe, pe_stride, scale, out,
                       B * sy, sy, d, V, key, thresh, inv);
    EGX_LAUNCH_CHECK();
    return 0;
}
int embed_pos_bwd(const int64_t* tok, const float* dy, float* d_emb, float scale, int B, int sy, int d, int V, uint64_t key,
                  uint32_t thresh, float inv, hipStream_t st) {
    EGX_CHECK(tok && dy && d_emb, "embed_pos_bwd: null pointer argument");
    size_t n = (size_t)B * sy * d;
    hipLaunchKernelGGL(embed_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tok, dy, d_emb, scale, B * sy, d, V,
                       key, thresh, inv);
    EGX_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ dy, const float* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) dy[i] = 0.f;
}
int relu_mask(float* dy, const float* y, size_t n, hipStream_t st) {
    EGX_CHECK(dy && y, "relu_mask: null pointer argument");
    if (!n) return 0;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, y, n);
    EGX_LAUNCH_CHECK();
    return 0;
}

// h = z * Phi(z); dz = dh * (Phi(z) + z * phi(z))   (nn.GELU() default: exact erf form)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ z, float* __restrict__ h, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4*>(z)[i];
    auto f = [](float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); };
    reinterpret_cast<float4*>(h)[i] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dh, float* __restrict__ dz, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4*>(z)[i], g = reinterpret_cast<const float4*>(dh)[i];
    auto f = [](float x, float d) {
        float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        return d * (cdf + x * pdf);
    };
    reinterpret_cast<float4*>(dz)[i] = make_float4(f(v.x, g.x), f(v.y, g.y), f(v.z, g.z), f(v.w, g.w));
}
int gelu_fwd(const float* z, float* h, size_t n, hipStream_t st) {
    EGX_CHECK(z && h && n % 4 == 0, "gelu_fwd: null pointer or n %% 4 != 0");
    if (!n) return 0;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, z, h, n / 4);
    EGX_LAUNCH_CHECK();
    return 0;
}
int gelu_bwd(const float* z, const float* dh, float* dz, size_t n, hipStream_t st) {
    EGX_CHECK(z && dh && dz && n % 4 == 0, "gelu_bwd: null pointer or n %% 4 != 0");
    if (!n) return 0;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, z, dh, dz, n / 4);
    EGX_LAUNCH_CHECK();
    return 0;
}

}  // namespace egx
