// Row-wise kernels of the translator: LayerNorm forward/backward with fused residual, task-embedding,
// positional encoding and dropout (token preparation), column sums for bias gradients, and the pooled
// task head. HBM-bound: one 64-lane wave per token row, wavefront (DPP/shuffle) reductions, no LDS for
// the row statistics.
//
// Reference math: HHI/models/ttm/model_taskspecific.py:222-226 (encode_prepare), :243-244 (mean + linear_head),
// torch.nn.TransformerEncoderLayer post-LN residual blocks (norm1/norm2).
#include "common.h"
#include "kernels.h"

namespace egx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int remap_row(int row, int T, int S, int off) { return (row / T) * S + off + (row % T); }

// ---- LayerNorm forward ------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int d = p.d;
    const float inv_d = 1.f / (float)d;
    for (int row = blockIdx.x * 4 + wave; row < p.rows; row += gridDim.x * 4) {
        const float* x = p.x + (size_t)row * d;
        const float* rs = p.res ? p.res + (size_t)row * d : nullptr;
        float v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane + 64 * i;
            float t = 0.f;
            if (c < d) {
                t = x[c];
                if (rs) t += rs[c];
            }
            v[i] = t;
            s += t;
        }
        float mean = wave_sum(s) * inv_d;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane + 64 * i;
            float t = (c < d) ? v[i] - mean : 0.f;
            ss += t * t;
        }
        float var = wave_sum(ss) * inv_d;
        float rstd = rsqrtf(var + p.eps);
        if (p.stats && lane == 0) {
            p.stats[2 * (size_t)row] = mean;
            p.stats[2 * (size_t)row + 1] = rstd;
        }
        int t_in = row % p.T;
        int orow = remap_row(row, p.T, p.S, p.off);
        float* y = p.y + (size_t)orow * d;
        float* pre = p.pre ? p.pre + (size_t)row * d : nullptr;
        const float* pos = p.pos ? p.pos + (size_t)t_in * p.pos_stride : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane + 64 * i;
            if (c < d) {
                if (pre) pre[c] = v[i];
                float o = (v[i] - mean) * rstd * p.w[c] + p.b[c];
                if (p.add_vec) o += p.add_vec[c];
                if (pos) o += pos[c];
                if (p.drop_thresh) o *= drop_scale(p.drop_key, (uint32_t)orow, (uint32_t)c, p.drop_thresh, p.drop_inv_keep);
                y[c] = o;
            }
        }
    }
}

int layernorm_fwd(const LnFwdParams& p, hipStream_t st) {
    EGX_CHECK(p.d > 0 && p.d <= 1024, "layernorm: d=%d unsupported (max 1024)", p.d);
    if (p.rows <= 0) return 0;
    int nv = cdiv(p.d, 64);
    int blocks = min(cdiv(p.rows, 4), 4096);
    dim3 g(blocks), b(256);
    if (nv <= 2) hipLaunchKernelGGL(ln_fwd_kernel<2>, g, b, 0, st, p);
    else if (nv <= 4) hipLaunchKernelGGL(ln_fwd_kernel<4>, g, b, 0, st, p);
    else if (nv <= 8) hipLaunchKernelGGL(ln_fwd_kernel<8>, g, b, 0, st, p);
    else if (nv <= 12) hipLaunchKernelGGL(ln_fwd_kernel<12>, g, b, 0, st, p);
    else hipLaunchKernelGGL(ln_fwd_kernel<16>, g, b, 0, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- LayerNorm backward -----------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdParams p, float* __restrict__ part) {
    __shared__ float red[3][4][64 * NV];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int d = p.d;
    const float inv_d = 1.f / (float)d;
    float aw[NV], ab[NV], aa[NV];
    float wv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        aw[i] = 0.f; ab[i] = 0.f; aa[i] = 0.f;
        int c = lane + 64 * i;
        wv[i] = (c < d) ? p.w[c] : 0.f;
    }
    for (int row = blockIdx.x * 4 + wave; row < p.rows; row += gridDim.x * 4) {
        int orow = remap_row(row, p.T, p.S, p.off);
        const float* dy = p.dy + (size_t)orow * d;
        const float* pre = p.pre + (size_t)row * d;
        float mean = p.stats[2 * (size_t)row];
        float rstd = p.stats[2 * (size_t)row + 1];
        float g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane + 64 * i;
            float dyv = 0.f, x = 0.f;
            if (c < d) {
                dyv = dy[c];
                if (p.drop_thresh) dyv *= drop_scale(p.drop_key, (uint32_t)orow, (uint32_t)c, p.drop_thresh, p.drop_inv_keep);
                x = (pre[c] - mean) * rstd;
            }
            xh[i] = x;
            g[i] = dyv * wv[i];
            s1 += g[i];
            s2 += g[i] * x;
            aw[i] += dyv * x;
            ab[i] += dyv;
        }
        s1 = wave_sum(s1) * inv_d;
        s2 = wave_sum(s2) * inv_d;
        float* dx = p.dx + (size_t)row * d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane + 64 * i;
            if (c < d) {
                float o = rstd * (g[i] - s1 - xh[i] * s2);
                if (p.out_drop_thresh) o *= drop_scale(p.out_drop_key, (uint32_t)row, (uint32_t)c, p.out_drop_thresh, p.out_drop_inv_keep);
                dx[c] = o;
            }
        }
    }
    // block reduction of the parameter-gradient partials, one atomic per column per block
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        red[0][wave][lane + 64 * i] = aw[i];
        red[1][wave][lane + 64 * i] = ab[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        float sw = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (part) {         // deterministic mode: [block][2][d] partials, summed in block order by det_reduce_rows
            part[((size_t)blockIdx.x * 2) * d + c] = sw;
            part[((size_t)blockIdx.x * 2 + 1) * d + c] = sb;
        } else {
            if (p.dw) atomicAdd(p.dw + c, sw);
            if (p.db) atomicAdd(p.db + c, sb);
            if (p.dadd) atomicAdd(p.dadd + c, sb);
        }
    }
    (void)aa;
}

// ---- deterministic mode plumbing ------------------------------------------------------------------
static thread_local void* g_det_buf = nullptr;
static thread_local size_t g_det_bytes = 0;
static thread_local int g_det_depth = 0;
DetScope::DetScope(void* buf, size_t bytes) { if (buf && bytes) { g_det_buf = buf; g_det_bytes = bytes; ++g_det_depth; } }
DetScope::~DetScope() { if (g_det_depth > 0 && --g_det_depth == 0) { g_det_buf = nullptr; g_det_bytes = 0; } }
bool det_on() { return g_det_depth > 0; }
constexpr int DET_LN_BLOCKS = 256, DET_COLSUM_ROWBLOCKS = 64;
size_t generic_det_scratch_bytes(int B, int d, int d_ff) {
    size_t ln = (size_t)DET_LN_BLOCKS * 2 * d;
    size_t cs = (size_t)DET_COLSUM_ROWBLOCKS * (size_t)(d_ff > 3 * d ? d_ff : 3 * d);
    size_t ph = (size_t)B * ((size_t)64 * d + 64 + 2 * (size_t)d);
    size_t m = ln > cs ? ln : cs;
    return (m > ph ? m : ph) * sizeof(float);
}
// out[c] += sum_t part[t * stride + c], t < nt, in order of t
__global__ __launch_bounds__(256) void det_reduce_rows_kernel(const float* __restrict__ part, size_t stride, int nt, int cols, float* __restrict__ out) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    int t = 0;
    for (; t + 8 <= nt; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(t + u) * stride + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; t < nt; ++t) s += part[(size_t)t * stride + c];
    out[c] += s;
}
static int det_reduce_rows(const float* part, size_t stride, int nt, int cols, float* out, hipStream_t st) {
    if (!out || cols <= 0 || nt <= 0) return 0;
    hipLaunchKernelGGL(det_reduce_rows_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, part, stride, nt, cols, out);
    EGX_LAUNCH_CHECK();
    return 0;
}

int layernorm_bwd(const LnBwdParams& p, hipStream_t st) {
    EGX_CHECK(p.d > 0 && p.d <= 1024, "layernorm_bwd: d=%d unsupported (max 1024)", p.d);
    if (p.rows <= 0) return 0;
    int nv = cdiv(p.d, 64);
    int blocks = min(cdiv(p.rows, 4 * 8), 256);
    if (blocks < 1) blocks = 1;
    float* part = nullptr;
    if (det_on() && (p.dw || p.db || p.dadd)) {
        EGX_CHECK((size_t)blocks * 2 * p.d * sizeof(float) <= g_det_bytes, "layernorm_bwd: deterministic scratch too small");
        part = (float*)g_det_buf;
    }
    dim3 g(blocks), b(256);
    if (nv <= 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, g, b, 0, st, p, part);
    else if (nv <= 4) hipLaunchKernelGGL(ln_bwd_kernel<4>, g, b, 0, st, p, part);
    else if (nv <= 8) hipLaunchKernelGGL(ln_bwd_kernel<8>, g, b, 0, st, p, part);
    else if (nv <= 12) hipLaunchKernelGGL(ln_bwd_kernel<12>, g, b, 0, st, p, part);
    else hipLaunchKernelGGL(ln_bwd_kernel<16>, g, b, 0, st, p, part);
    EGX_LAUNCH_CHECK();
    if (part) {
        if (det_reduce_rows(part, (size_t)2 * p.d, blocks, p.d, p.dw, st)) return 1;
        if (det_reduce_rows(part + p.d, (size_t)2 * p.d, blocks, p.d, p.db, st)) return 1;
        if (det_reduce_rows(part + p.d, (size_t)2 * p.d, blocks, p.d, p.dadd, st)) return 1;
    }
    return 0;
}

// ---- column sums (bias gradients) -------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                      float* __restrict__ out, int rows_per_block, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int c = blockIdx.x * 64 + lane;
    int r0 = blockIdx.y * rows_per_block;
    int r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + wave; r < r1; r += 4) s += x[(size_t)r * ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) {
        const float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (part) part[(size_t)blockIdx.y * cols + c] = v;      // deterministic mode: one partial row per row block
        else atomicAdd(out + c, v);
    }
}

int colsum_accum(const float* x, int rows, int cols, int ld, float* out, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return 0;
    int cb = cdiv(cols, 64);
    int want = max(1, 1024 / cb);
    int rpb = max(32, cdiv(rows, want));
    float* part = nullptr;
    if (det_on()) {
        rpb = max(rpb, cdiv(rows, DET_COLSUM_ROWBLOCKS));
        EGX_CHECK((size_t)cdiv(rows, rpb) * cols * sizeof(float) <= g_det_bytes, "colsum: deterministic scratch too small");
        part = (float*)g_det_buf;
    }
    dim3 g(cb, cdiv(rows, rpb));
    hipLaunchKernelGGL(colsum_kernel, g, dim3(256), 0, st, x, rows, cols, ld, out, rpb, part);
    EGX_LAUNCH_CHECK();
    if (part) return det_reduce_rows(part, (size_t)cols, (int)g.y, cols, out, st);
    return 0;
}

// ---- learned positional-embedding gradient -----------------------------------------------------
// dpos[t][:] += sum_b mask .* dtok[b * S + off + t][:]. One workgroup per position row t: d / 4 float4 columns x (256 / (d / 4)) clip
// lanes, a lane walks its clips four loads at a time, the lanes are added in lane order through LDS (deterministic). (Until round 4
// one THREAD walked all B clips of its element with dependent-latency loads: 64 us per segment at B = 256, 12 % of the PNR / OSCC step.)
__global__ __launch_bounds__(256) void pos_grad_kernel(const float* __restrict__ dtok, int B, int S, int off, int T, int d,
                                                       float* __restrict__ dpos, int pos_stride, uint64_t key, uint32_t thresh, float inv_keep) {
    __shared__ float4 red[256];
    const int t = blockIdx.x, nc = d >> 2;              // d % 4 == 0, d <= 1024
    const int c4 = threadIdx.x % nc, lane = threadIdx.x / nc, nl = 256 / nc;
    float4 s = make_float4(0, 0, 0, 0);
    if (lane < nl) {
        auto fetch = [&](int b) {
            const int orow = b * S + off + t;
            float4 v = *reinterpret_cast<const float4*>(dtok + (size_t)orow * d + 4 * c4);
            if (thresh) {
                v.x *= drop_scale(key, (uint32_t)orow, (uint32_t)(4 * c4), thresh, inv_keep);
                v.y *= drop_scale(key, (uint32_t)orow, (uint32_t)(4 * c4 + 1), thresh, inv_keep);
                v.z *= drop_scale(key, (uint32_t)orow, (uint32_t)(4 * c4 + 2), thresh, inv_keep);
                v.w *= drop_scale(key, (uint32_t)orow, (uint32_t)(4 * c4 + 3), thresh, inv_keep);
            }
            return v;
        };
        int b = lane;
        for (; b + 3 * nl < B; b += 4 * nl) {
            const float4 v0 = fetch(b), v1 = fetch(b + nl), v2 = fetch(b + 2 * nl), v3 = fetch(b + 3 * nl);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; b < B; b += nl) { const float4 v = fetch(b); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0) {
        for (int j = 1; j < nl; ++j) { const float4 v = red[j * nc + c4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        float4* o = reinterpret_cast<float4*>(dpos + (size_t)t * pos_stride + 4 * c4);
        float4 cur = *o;
        *o = make_float4(cur.x + s.x, cur.y + s.y, cur.z + s.z, cur.w + s.w);
    }
}

int pos_grad_accum(const float* dtok, int B, int S, int off, int T, int d, float* dpos, int pos_stride,
                   uint64_t key, uint32_t thresh, float inv_keep, hipStream_t st) {
    if (T <= 0 || d <= 0) return 0;
    EGX_CHECK(d % 4 == 0 && d <= 1024 && pos_stride % 4 == 0, "pos_grad: d = %d (multiple of 4, <= 1024), pos_stride = %d", d, pos_stride);
    EGX_CHECK((((uintptr_t)dtok) & 15) == 0 && (((uintptr_t)dpos) & 15) == 0, "pos_grad: d_tokens and the positional gradient must be 16-byte aligned (float4 accesses)");
    hipLaunchKernelGGL(pos_grad_kernel, dim3(T), dim3(256), 0, st, dtok, B, S, off, T, d, dpos, pos_stride,
                       key, thresh, inv_keep);
    EGX_LAUNCH_CHECK();
    return 0;
}

__global__ void dropout_mask_kernel(float* __restrict__ x, int rows, int d, uint64_t key, uint32_t thresh, float inv_keep) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)rows * d;
    if (i >= n) return;
    uint32_t row = (uint32_t)(i / d), col = (uint32_t)(i % d);
    x[i] *= drop_scale(key, row, col, thresh, inv_keep);
}

int apply_dropout_mask(float* x, int rows, int d, uint64_t key, uint32_t thresh, float inv_keep, hipStream_t st) {
    size_t n = (size_t)rows * d;
    if (!n || !thresh) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, rows, d, key, thresh, inv_keep);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- pooled head: mean over tokens -> (LN) -> (Linear) ----------------------------------------
constexpr int PH_MAXD = 1024;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void pool_head_fwd_kernel(const float* __restrict__ tokens, int S, int d,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                                                             const float* __restrict__ W, const float* __restrict__ bias, int n_out,
                                                             float* __restrict__ pooled, float* __restrict__ out) {
    __shared__ float y[PH_MAXD];
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* tk = tokens + (size_t)b * S * d;
    float inv_s = 1.f / (float)S;
    float loc = 0.f;
    if (d <= 128 && d % 4 == 0) {
        // small d: 8 row groups x 32 float4 columns, four loads in flight per thread (a clip of 450 tokens took 33 us with two
        // threads per column walking 225 dependent loads each: the tiled path's pooled head, tests/test_gpu_tiled.py)
        __shared__ float4 ps4[8][32];
        const int c4 = threadIdx.x & 31, g = threadIdx.x >> 5;
        float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
        if (c4 * 4 < d) {
            const float4* t4 = reinterpret_cast<const float4*>(tk) + c4;
            const int ld4 = d / 4;
            int t = g;
            for (; t + 24 < S; t += 32) {
                const float4 v0 = t4[(size_t)t * ld4], v1 = t4[(size_t)(t + 8) * ld4], v2 = t4[(size_t)(t + 16) * ld4], v3 = t4[(size_t)(t + 24) * ld4];
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w; a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (; t < S; t += 8) { const float4 v0 = t4[(size_t)t * ld4]; a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
        }
        ps4[g][c4] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
        __syncthreads();
        if (threadIdx.x < d) {
            const float* pf = reinterpret_cast<const float*>(&ps4[0][0]);
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sacc += pf[k * 128 + threadIdx.x];
            const float s = sacc * inv_s;
            y[threadIdx.x] = s;
            pooled[(size_t)b * d + threadIdx.x] = s;
            loc = s;
        }
    } else if (d <= 128) {
        // small d: split the token loop over the 256 threads (2 threads per column) and combine through LDS
        __shared__ float ps[2][128];
        int c = threadIdx.x & 127, g = threadIdx.x >> 7;
        float s0 = 0.f, s1 = 0.f;
        if (c < d) {
            int t = g;
            for (; t + 2 < S; t += 4) { s0 += tk[(size_t)t * d + c]; s1 += tk[(size_t)(t + 2) * d + c]; }
            for (; t < S; t += 2) s0 += tk[(size_t)t * d + c];
        }
        ps[g][c] = s0 + s1;
        __syncthreads();
        if (threadIdx.x < d) {
            float s = (ps[0][threadIdx.x] + ps[1][threadIdx.x]) * inv_s;
            y[threadIdx.x] = s;
            pooled[(size_t)b * d + threadIdx.x] = s;
            loc = s;
        }
    } else {
        for (int c = threadIdx.x; c < d; c += 256) {
            // four rows in flight per thread (one dependent load per token took 93 us per launch at C4's S = 128, d = 768)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int t = 0;
            for (; t + 3 < S; t += 4) {
                s0 += tk[(size_t)t * d + c]; s1 += tk[(size_t)(t + 1) * d + c]; s2 += tk[(size_t)(t + 2) * d + c]; s3 += tk[(size_t)(t + 3) * d + c];
            }
            for (; t < S; ++t) s0 += tk[(size_t)t * d + c];
            float s = (s0 + s1) + (s2 + s3);
            s *= inv_s;
            y[c] = s;
            pooled[(size_t)b * d + c] = s;
            loc += s;
        }
    }
    if (ln_w) {
        float mean = block_sum(loc, red) / (float)d;
        float v = 0.f;
        for (int c = threadIdx.x; c < d; c += 256) { float t = y[c] - mean; v += t * t; }
        float var = block_sum(v, red) / (float)d;
        float rstd = rsqrtf(var + eps);
        for (int c = threadIdx.x; c < d; c += 256) y[c] = (y[c] - mean) * rstd * ln_w[c] + ln_b[c];
    }
    __syncthreads();
    if (W) {
        int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int o = wave; o < n_out; o += 4) {
            float s = 0.f;
            for (int c = lane; c < d; c += 64) s += y[c] * W[(size_t)o * d + c];
            s = wave_sum(s);
            if (lane == 0) out[(size_t)b * n_out + o] = s + (bias ? bias[o] : 0.f);
        }
    } else {
        for (int c = threadIdx.x; c < d; c += 256) out[(size_t)b * d + c] = y[c];
    }
}

__global__ __launch_bounds__(256) void pool_head_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ pooled,
                                                             int S, int d, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             float eps, const float* __restrict__ W, int n_out,
                                                             float* __restrict__ d_tokens, float* __restrict__ d_ln_w,
                                                             float* __restrict__ d_ln_b, float* __restrict__ d_W, float* __restrict__ d_b,
                                                             float* __restrict__ part) {
    // deterministic mode: this clip's contributions go to part[b][n_out * d | n_out | d (ln_w) | d (ln_b)]
    const bool first = blockIdx.y == 0;     // gridDim.y workgroups share a clip's d(tokens) rows; the parameter gradients leave once
    float* pw = (part && first) ? part + (size_t)blockIdx.x * ((size_t)n_out * d + n_out + 2 * (size_t)d) : nullptr;
    __shared__ float xh[PH_MAXD];   // normalised pooled (or pooled when no LN)
    __shared__ __attribute__((aligned(16))) float dy[PH_MAXD];   // gradient w.r.t. head input y
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* pl = pooled + (size_t)b * d;
    float mean = 0.f, rstd = 1.f;
    if (ln_w) {
        float loc = 0.f;
        for (int c = threadIdx.x; c < d; c += 256) loc += pl[c];
        mean = block_sum(loc, red) / (float)d;
        float v = 0.f;
        for (int c = threadIdx.x; c < d; c += 256) { float t = pl[c] - mean; v += t * t; }
        float var = block_sum(v, red) / (float)d;
        rstd = rsqrtf(var + eps);
    }
    const float* go = d_out + (size_t)b * (W ? n_out : d);
    for (int c = threadIdx.x; c < d; c += 256) {
        float x = ln_w ? (pl[c] - mean) * rstd : pl[c];
        xh[c] = x;
        float g;
        if (W) {
            g = 0.f;
            for (int o = 0; o < n_out; ++o) g += go[o] * W[(size_t)o * d + c];
            float yv = ln_w ? x * ln_w[c] + ln_b[c] : x;
            if (d_W && first)
                for (int o = 0; o < n_out; ++o) {
                    if (pw) pw[(size_t)o * d + c] = go[o] * yv;
                    else atomicAdd(d_W + (size_t)o * d + c, go[o] * yv);
                }
        } else {
            g = go[c];
        }
        dy[c] = g;
    }
    if (W && d_b && first && threadIdx.x < n_out) {
        if (pw) pw[(size_t)n_out * d + threadIdx.x] = go[threadIdx.x];
        else atomicAdd(d_b + threadIdx.x, go[threadIdx.x]);
    }
    __syncthreads();
    float inv_s = 1.f / (float)S;
    if (ln_w) {
        float s1 = 0.f, s2 = 0.f;
        for (int c = threadIdx.x; c < d; c += 256) {
            float g = dy[c] * ln_w[c];
            s1 += g;
            s2 += g * xh[c];
            if (pw) {
                pw[(size_t)n_out * d + n_out + c] = dy[c] * xh[c];
                pw[(size_t)n_out * d + n_out + d + c] = dy[c];
            } else if (first) {
                if (d_ln_w) atomicAdd(d_ln_w + c, dy[c] * xh[c]);
                if (d_ln_b) atomicAdd(d_ln_b + c, dy[c]);
            }
        }
        s1 = block_sum(s1, red) / (float)d;
        s2 = block_sum(s2, red) / (float)d;
        for (int c = threadIdx.x; c < d; c += 256) {
            float g = dy[c] * ln_w[c];
            dy[c] = rstd * (g - s1 - xh[c] * s2) * inv_s;
        }
    } else {
        for (int c = threadIdx.x; c < d; c += 256) dy[c] *= inv_s;
    }
    __syncthreads();
    float* dt = d_tokens + (size_t)b * S * d;
    const int rows_per = (S + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * rows_per, r1 = min(S, r0 + rows_per);
    if (d % 4 == 0) {
        const int ld4 = d / 4;
        for (int i = r0 * ld4 + threadIdx.x; i < r1 * ld4; i += 256)
            reinterpret_cast<float4*>(dt)[i] = *reinterpret_cast<const float4*>(dy + (i % ld4) * 4);
    } else {
        for (int i = r0 * d + threadIdx.x; i < r1 * d; i += 256) dt[i] = dy[i % d];
    }
}

int pool_head_fwd(const float* tokens, int B, int S, int d, const float* ln_w, const float* ln_b, float eps,
                  const float* W, const float* b, int n_out, float* pooled, float* out, hipStream_t st) {
    EGX_CHECK(d <= PH_MAXD, "pool_head: d=%d exceeds %d", d, PH_MAXD);
    EGX_CHECK(!W || (n_out >= 1 && n_out <= 64), "pool_head: n_out=%d out of range 1..64", n_out);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pool_head_fwd_kernel, dim3(B), dim3(256), 0, st, tokens, S, d, ln_w, ln_b, eps, W, b, n_out, pooled, out);
    EGX_LAUNCH_CHECK();
    return 0;
}

int pool_head_bwd(const float* d_out, const float* pooled, int B, int S, int d, const float* ln_w,
                  const float* ln_b, float eps, const float* W, int n_out, float* d_tokens, float* d_ln_w,
                  float* d_ln_b, float* d_W, float* d_b, hipStream_t st) {
    EGX_CHECK(d <= PH_MAXD, "pool_head: d=%d exceeds %d", d, PH_MAXD);
    EGX_CHECK(!W || (n_out >= 1 && n_out <= 64), "pool_head: n_out=%d out of range 1..64", n_out);
    if (B <= 0) return 0;
    float* part = nullptr;
    const size_t prow = (size_t)n_out * d + n_out + 2 * (size_t)d;
    if (det_on() && W && ln_w) {
        EGX_CHECK((size_t)B * prow * sizeof(float) <= g_det_bytes, "pool_head_bwd: deterministic scratch too small");
        part = (float*)g_det_buf;
    }
    // a clip's d(tokens) rows are shared out over several workgroups when the batch is small and the sequence long
    int ysplit = 1;
    if (B < 256) { ysplit = 256 / B; const int maxs = cdiv(S, 16); ysplit = ysplit > maxs ? maxs : ysplit; ysplit = ysplit < 1 ? 1 : ysplit; }
    hipLaunchKernelGGL(pool_head_bwd_kernel, dim3(B, ysplit), dim3(256), 0, st, d_out, pooled, S, d, ln_w, ln_b, eps, W, n_out,
                       d_tokens, d_ln_w, d_ln_b, d_W, d_b, part);
    EGX_LAUNCH_CHECK();
    if (part) {
        if (det_reduce_rows(part, prow, B, n_out * d, d_W, st)) return 1;
        if (det_reduce_rows(part + (size_t)n_out * d, prow, B, n_out, d_b, st)) return 1;
        if (det_reduce_rows(part + (size_t)n_out * d + n_out, prow, B, d, d_ln_w, st)) return 1;
        if (det_reduce_rows(part + (size_t)n_out * d + n_out + d, prow, B, d, d_ln_b, st)) return 1;
    }
    return 0;
}

}  // namespace egx
