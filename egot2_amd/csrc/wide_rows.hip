// Row-wise kernels of the wide bf16 path: LayerNorm forward / backward with a bf16 side output (the operand of the next
// GEMM), bf16 column sums (bias gradients) and fixed-order partial-sum reductions. HBM-bound: one wave per token row,
// 16 bytes per lane per access, wavefront reductions; parameter-gradient partials are summed in a fixed order
// (bitwise reproducible, no atomics).
//
// Reference math: torch.nn.LayerNorm inside nn.TransformerEncoderLayer (norm1 / norm2, post-LN) and the shared token-prep
// LayerNorm `self.ln` + positional table (HOI/models/lta/lta_models_lta_transfer.py:359-360,
// HOI/models/multitask/video_model_builder.py:331-346).
#include "common.h"
#include "wide.h"

namespace egx {

namespace {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t pk2(float a, float b) { return pack_bf16x2(a, b); }
__device__ __forceinline__ float bff(uint32_t lo16) { return __builtin_bit_cast(float, lo16 << 16); }
__device__ __forceinline__ int remap(int row, int T, int S, int off) { return (row / T) * S + off + (row % T); }

constexpr int LN_MAXV = 4;      // float4 per lane: d <= 1024

}  // namespace

__global__ __launch_bounds__(256) void wide_ln_fwd_kernel(WideLnFwdParams p) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = p.d, nv = (d + 255) / 256;
    const float inv_d = 1.f / (float)d;
    for (int row = blockIdx.x * 4 + wave; row < p.rows; row += gridDim.x * 4) {
        const float* x = p.x + (size_t)row * d;
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = 4 * lane + 256 * i;
            v[i] = make_float4(0, 0, 0, 0);
            if (i < nv && c < d) v[i] = *reinterpret_cast<const float4*>(x + c);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wsum64(s) * inv_d;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = 4 * lane + 256 * i;
            if (i < nv && c < d) {
                float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
                ss += (a * a + b * b) + (cc * cc + dd * dd);
            }
        }
        const float rstd = rsqrtf(wsum64(ss) * inv_d + p.eps);
        if (p.stats && lane == 0) *reinterpret_cast<float2*>(p.stats + 2 * (size_t)row) = make_float2(mean, rstd);
        const int t_in = row % p.T, orow = remap(row, p.T, p.S, p.off);
        const float* pos = p.pos ? p.pos + (size_t)t_in * p.pos_stride : nullptr;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = 4 * lane + 256 * i;
            if (i < nv && c < d) {
                float4 w = *reinterpret_cast<const float4*>(p.w + c), b = *reinterpret_cast<const float4*>(p.b + c);
                float o[4] = {(v[i].x - mean) * rstd * w.x + b.x, (v[i].y - mean) * rstd * w.y + b.y,
                              (v[i].z - mean) * rstd * w.z + b.z, (v[i].w - mean) * rstd * w.w + b.w};
                if (p.add_vec) { float4 a = *reinterpret_cast<const float4*>(p.add_vec + c); o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; }
                if (pos) { float4 a = *reinterpret_cast<const float4*>(pos + c); o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; }
                if (p.drop_thresh) {
                    float ds[4];
                    drop_scale4(dkey, (uint32_t)orow, (uint32_t)c, p.drop_thresh, p.drop_inv, ds);
                    o[0] *= ds[0]; o[1] *= ds[1]; o[2] *= ds[2]; o[3] *= ds[3];
                }
                if (p.y32) *reinterpret_cast<float4*>(p.y32 + (size_t)orow * d + c) = make_float4(o[0], o[1], o[2], o[3]);
                if (p.y16) *reinterpret_cast<uint2*>(p.y16 + (size_t)orow * d + c) = make_uint2(pk2(o[0], o[1]), pk2(o[2], o[3]));
            }
        }
    }
}

int wide_ln_fwd(const WideLnFwdParams& p, hipStream_t st) {
    EGX_CHECK(p.x && p.w && p.b && (p.y32 || p.y16), "wide_ln_fwd: null pointer");
    EGX_CHECK(p.d > 0 && p.d <= 1024 && p.d % 4 == 0, "wide_ln_fwd: d=%d unsupported", p.d);
    if (p.rows <= 0) return 0;
    int blocks = cdiv(p.rows, 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wide_ln_fwd_kernel, dim3(blocks), dim3(256), 0, st, p);
    EGX_LAUNCH_CHECK();
    return 0;
}

// Each block owns a CONTIGUOUS range of rows; its four waves interleave over it and keep per-column partial sums in
// registers, combined through LDS at the end: partials[block][k][d], k = 0 d(gamma), 1 d(beta), 2 column sums of dx16.
__global__ __launch_bounds__(256) void wide_ln_bwd_kernel(WideLnBwdParams p, int rows_per_block) {
    const uint64_t dkey = p.drop_thresh ? resolve_key(p.drop_key) : 0ull, okey = p.out_thresh ? resolve_key(p.out_key) : 0ull;
    __shared__ float red[3][4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = p.d, nv = (d + 255) / 256;
    const float inv_d = 1.f / (float)d;
    float4 aw[LN_MAXV], ab[LN_MAXV], ac[LN_MAXV], wv[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        aw[i] = ab[i] = ac[i] = wv[i] = make_float4(0, 0, 0, 0);
        int c = 4 * lane + 256 * i;
        if (i < nv && c < d) wv[i] = *reinterpret_cast<const float4*>(p.w + c);
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(p.rows, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += 4) {
        const int orow = remap(row, p.T, p.S, p.off);
        const float* dy = p.dy + (size_t)orow * d;
        const float* pre = p.pre + (size_t)row * d;
        const float2 st2 = *reinterpret_cast<const float2*>(p.stats + 2 * (size_t)row);
        const float mean = st2.x, rstd = st2.y;
        float4 gq[LN_MAXV], xh[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = 4 * lane + 256 * i;
            gq[i] = xh[i] = make_float4(0, 0, 0, 0);
            if (i < nv && c < d) {
                float4 g = *reinterpret_cast<const float4*>(dy + c);
                if (p.drop_thresh) {
                    float ds[4];
                    drop_scale4(dkey, (uint32_t)orow, (uint32_t)c, p.drop_thresh, p.drop_inv, ds);
                    g.x *= ds[0]; g.y *= ds[1]; g.z *= ds[2]; g.w *= ds[3];
                }
                float4 x = *reinterpret_cast<const float4*>(pre + c);
                x.x = (x.x - mean) * rstd; x.y = (x.y - mean) * rstd; x.z = (x.z - mean) * rstd; x.w = (x.w - mean) * rstd;
                aw[i].x += g.x * x.x; aw[i].y += g.y * x.y; aw[i].z += g.z * x.z; aw[i].w += g.w * x.w;
                ab[i].x += g.x; ab[i].y += g.y; ab[i].z += g.z; ab[i].w += g.w;
                g.x *= wv[i].x; g.y *= wv[i].y; g.z *= wv[i].z; g.w *= wv[i].w;
                s1 += (g.x + g.y) + (g.z + g.w);
                s2 += (g.x * x.x + g.y * x.y) + (g.z * x.z + g.w * x.w);
                gq[i] = g; xh[i] = x;
            }
        }
        s1 = wsum64(s1) * inv_d;
        s2 = wsum64(s2) * inv_d;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = 4 * lane + 256 * i;
            if (i < nv && c < d) {
                float o[4] = {rstd * (gq[i].x - s1 - xh[i].x * s2), rstd * (gq[i].y - s1 - xh[i].y * s2),
                              rstd * (gq[i].z - s1 - xh[i].z * s2), rstd * (gq[i].w - s1 - xh[i].w * s2)};
                float m[4] = {o[0], o[1], o[2], o[3]};
                if (p.out_thresh) {
                    float ds[4];
                    drop_scale4(okey, (uint32_t)row, (uint32_t)c, p.out_thresh, p.out_inv, ds);
                    m[0] *= ds[0]; m[1] *= ds[1]; m[2] *= ds[2]; m[3] *= ds[3];
                }
                if (p.dx32) {
                    const float* q = p.mask_dx32 ? m : o;
                    *reinterpret_cast<float4*>(p.dx32 + (size_t)row * d + c) = make_float4(q[0], q[1], q[2], q[3]);
                }
                if (p.dx16) {
                    uint2 u = make_uint2(pk2(m[0], m[1]), pk2(m[2], m[3]));
                    *reinterpret_cast<uint2*>(p.dx16 + (size_t)row * d + c) = u;
                    // the bias gradient is the column sum of the operand AS STORED (bf16-rounded)
                    ac[i].x += bff(u.x & 0xffffu); ac[i].y += bff(u.x >> 16); ac[i].z += bff(u.y & 0xffffu); ac[i].w += bff(u.y >> 16);
                } else {
                    ac[i].x += m[0]; ac[i].y += m[1]; ac[i].z += m[2]; ac[i].w += m[3];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int c = 4 * lane + 256 * i;
        if (i < nv && c < d) {
            *reinterpret_cast<float4*>(&red[0][wave][c]) = aw[i];
            *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[i];
            *reinterpret_cast<float4*>(&red[2][wave][c]) = ac[i];
        }
    }
    __syncthreads();
    float* out = p.partials + (size_t)blockIdx.x * 3 * d;
    for (int i = threadIdx.x; i < 3 * d; i += 256) {
        int k = i / d, c = i % d;
        out[i] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
    }
}

// Fixed-order sum over the rows of a partial buffer: 1024 threads = 16 waves per 64 columns; wave w adds rows w, w + 16, ...
// (four independent accumulators), the 16 wave sums are combined in wave order through LDS. Bitwise reproducible.
__device__ __forceinline__ float reduce_rows_1024(const float* __restrict__ part, int nt, size_t ld, int col, bool valid, float (*red)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int t = wave;
        for (; t + 48 < nt; t += 64) {
            s0 += part[(size_t)(t + 0) * ld + col];  s1 += part[(size_t)(t + 16) * ld + col];
            s2 += part[(size_t)(t + 32) * ld + col]; s3 += part[(size_t)(t + 48) * ld + col];
        }
        for (; t < nt; t += 16) s0 += part[(size_t)t * ld + col];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float s = 0.f;
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][lane];
    }
    return s;
}

// dst_k[c] += sum_b partials[b][k][c], k = 0 d(gamma), 1 d(beta) (+ task embedding), 2 branch bias
__global__ __launch_bounds__(1024) void wide_ln_bwd_reduce_kernel(const float* __restrict__ partials, int blocks, int d,
                                                                  float* dw, float* db, float* dbias, float* dadd) {
    __shared__ float red[16][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = reduce_rows_1024(partials, blocks, (size_t)3 * d, i, i < 3 * d, red);
    if (threadIdx.x >= 64 || i >= 3 * d) return;
    const int k = i / d, c = i % d;
    if (k == 0) { if (dw) dw[c] += s; }
    else if (k == 1) { if (db) db[c] += s; if (dadd) dadd[c] += s; }
    else { if (dbias) dbias[c] += s; }
}

// every queued reduction in one launch: workgroup -> (reduction, 64-column group) by prefix sums
__global__ __launch_bounds__(1024) void wide_row_reduce_batch_kernel(WideRowReduceBatch b) {
    __shared__ float red[16][64];
    int di = 0;
    while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].first_block) ++di;
    const WideRowReduceDesc& q = b.d[di];
    const int c = ((int)blockIdx.x - q.first_block) * 64 + (threadIdx.x & 63);
    const float s = reduce_rows_1024(q.part, q.nt, (size_t)q.cols, c, c < q.cols, red);
    if (threadIdx.x >= 64 || c >= q.cols) return;
    if (q.kind == 0) {
        q.o0[(size_t)(c / q.row_len) * q.out_ld + c % q.row_len] += s;
    } else {
        const int k = c / q.d, cc = c % q.d;
        if (k == 0) { if (q.o0) q.o0[cc] += s; }
        else if (k == 1) { if (q.o1) q.o1[cc] += s; if (q.o3) q.o3[cc] += s; }
        else { if (q.o2) q.o2[cc] += s; }
    }
}
int wide_row_reduce_flush(WideRowReduceBatch& b, hipStream_t st) {
    if (!b.n) return 0;
    hipLaunchKernelGGL(wide_row_reduce_batch_kernel, dim3(b.total_blocks), dim3(1024), 0, st, b);
    EGX_LAUNCH_CHECK();
    b.n = 0; b.total_blocks = 0;
    return 0;
}
// false: the batch is full — the caller then runs this reduction at once, on the stream that produced its partials. (rounds 4-5 flushed the
// whole batch on the ADDING call's stream; the decoder backward queues from two streams, so a flush triggered from one could read partials the other had
// not written yet — reachable from 7 decoder layers on. A queued entry is only ever summed by the flush at the end, behind the join of both streams.)
static bool row_reduce_queue(WideRowReduceBatch& b, const WideRowReduceDesc& in) {
    if (b.n == WIDE_ROWRED_MAX) return false;
    WideRowReduceDesc& q = b.d[b.n++];
    q = in;
    q.first_block = b.total_blocks;
    b.total_blocks += cdiv(q.cols, 64);
    return true;
}

static int ln_bwd_blocks(int rows) {
    int blocks = cdiv(rows, 16);          // >= 4 rows per wave; at most 1024 partial rows to reduce
    if (blocks > 1024) blocks = 1024;
    return blocks < 1 ? 1 : blocks;
}
size_t wide_ln_bwd_scratch(int rows, int d) { return (size_t)ln_bwd_blocks(rows) * 3 * d * sizeof(float); }

int wide_ln_bwd(WideLnBwdParams p, void* scratch, hipStream_t st, WideRowReduceBatch* defer) {
    EGX_CHECK(p.dy && p.pre && p.stats && p.w && scratch, "wide_ln_bwd: null pointer");
    EGX_CHECK(p.d > 0 && p.d <= 1024 && p.d % 4 == 0, "wide_ln_bwd: d=%d unsupported", p.d);
    if (p.rows <= 0) return 0;
    p.blocks = ln_bwd_blocks(p.rows);
    p.partials = (float*)scratch;
    const int rpb = cdiv(p.rows, p.blocks);
    p.blocks = cdiv(p.rows, rpb);
    hipLaunchKernelGGL(wide_ln_bwd_kernel, dim3(p.blocks), dim3(256), 0, st, p, rpb);
    if (defer && (p.dw || p.db || p.dbias || p.dadd)) {
        EGX_LAUNCH_CHECK();
        WideRowReduceDesc q;
        q.part = p.partials; q.o0 = p.dw; q.o1 = p.db; q.o2 = p.dbias; q.o3 = p.dadd; q.nt = p.blocks; q.cols = 3 * p.d; q.d = p.d; q.kind = 1;
        q.first_block = 0; q.out_ld = 0; q.row_len = 1; q.pad_ = 0;
        if (row_reduce_queue(*defer, q)) return 0;
    }
    if (p.dw || p.db || p.dbias || p.dadd)
        hipLaunchKernelGGL(wide_ln_bwd_reduce_kernel, dim3(cdiv(3 * p.d, 64)), dim3(1024), 0, st, (const float*)p.partials, p.blocks,
                           p.d, p.dw, p.db, p.dbias, p.dadd);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- bf16 column sums ---------------------------------------------------------------------------------------------------
// block (bx, by): columns [bx * 512, +512) (8 per lane), rows [by * rpb, +rpb); partial[by][c]
__global__ __launch_bounds__(256) void wide_colsum_kernel(const bf16_t* __restrict__ x, int rows, int cols, int ld, int rpb,
                                                          float* __restrict__ partial) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    const int r0 = blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (c < cols)
        for (int row = r0 + wave; row < r1; row += 4) {
            uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)row * ld + c);
            s[0] += bff(v.x & 0xffffu); s[1] += bff(v.x >> 16); s[2] += bff(v.y & 0xffffu); s[3] += bff(v.y >> 16);
            s[4] += bff(v.z & 0xffffu); s[5] += bff(v.z >> 16); s[6] += bff(v.w & 0xffffu); s[7] += bff(v.w >> 16);
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = s[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        int cc = blockIdx.x * 512 + i;
        if (cc < cols) partial[(size_t)blockIdx.y * cols + cc] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

__global__ __launch_bounds__(1024) void wide_reduce_rows_kernel(const float* __restrict__ part, int nt, int cols, float* __restrict__ out, int out_ld, int row_len) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = reduce_rows_1024(part, nt, (size_t)cols, c, c < cols, red);
    if (threadIdx.x < 64 && c < cols) out[(size_t)(c / row_len) * out_ld + c % row_len] += s;
}

int wide_reduce_rows(const float* part, int nt, int cols, float* out, hipStream_t st, WideRowReduceBatch* defer) {
    if (nt <= 0 || cols <= 0) return 0;
    if (defer) {
        WideRowReduceDesc q;
        q.part = part; q.o0 = out; q.o1 = q.o2 = q.o3 = nullptr; q.nt = nt; q.cols = cols; q.d = cols; q.kind = 0;
        q.first_block = 0; q.out_ld = cols; q.row_len = cols; q.pad_ = 0;
        if (row_reduce_queue(*defer, q)) return 0;
    }
    hipLaunchKernelGGL(wide_reduce_rows_kernel, dim3(cdiv(cols, 64)), dim3(1024), 0, st, part, nt, cols, out, cols, cols);
    EGX_LAUNCH_CHECK();
    return 0;
}

// ---- learned positional-table gradient -------------------------------------------------------------------------------
// dpos[t][c] += sum_b mask .* dtok[(b * S + off + t)][c]: clips are split into chunks summed by separate workgroups, the chunk
// sums are added in fixed order (the generic path's one-thread-per-element loop over all B clips was latency-bound).
__global__ __launch_bounds__(256) void wide_pos_grad_kernel(const float* __restrict__ dtok, int B, int S, int off, int T, int d, int bchunk,
                                                            float* __restrict__ partial, uint64_t key, uint32_t thresh, float inv) {
    const int i4 = blockIdx.x * 256 + threadIdx.x;           // float4 index into (T, d)
    if (i4 * 4 >= T * d) return;
    const int t = (i4 * 4) / d, c = (i4 * 4) % d;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    float4 s = make_float4(0, 0, 0, 0);
    for (int b = b0; b < b1; ++b) {
        const int orow = b * S + off + t;
        float4 v = *reinterpret_cast<const float4*>(dtok + (size_t)orow * d + c);
        if (thresh) {
            float ds[4];
            drop_scale4(resolve_key(key), (uint32_t)orow, (uint32_t)c, thresh, inv, ds);
            v.x *= ds[0]; v.y *= ds[1]; v.z *= ds[2]; v.w *= ds[3];
        }
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * T * d + (size_t)i4 * 4) = s;
}

// clip chunks summed by separate workgroups. (Until round 5 the scratch query said cdiv(B, 8) chunks while the launch used the count
// below: for B < 256 the partial sums ran past the buffer — found when the deferred row reductions put their first partial buffer behind it.)
static int pos_grad_chunks(int B, int* bchunk_out = nullptr) {
    const int want = B >= 64 ? 32 : cdiv(B, 2);
    const int bchunk = cdiv(B, want);
    if (bchunk_out) *bchunk_out = bchunk;
    return cdiv(B, bchunk);
}
size_t wide_pos_grad_scratch(int B, int T, int d) { return (size_t)pos_grad_chunks(B) * T * d * sizeof(float); }

int wide_pos_grad(const float* dtok, int B, int S, int off, int T, int d, float* dpos, int pos_stride, uint64_t key, uint32_t thresh,
                  float inv, void* scratch, hipStream_t st) {
    EGX_CHECK(d % 4 == 0 && scratch, "wide_pos_grad: bad arguments");
    int bchunk = 1;
    const int chunks = pos_grad_chunks(B, &bchunk);
    hipLaunchKernelGGL(wide_pos_grad_kernel, dim3(cdiv(T * d / 4, 256), chunks), dim3(256), 0, st, dtok, B, S, off, T, d, bchunk,
                       (float*)scratch, key, thresh, inv);
    hipLaunchKernelGGL(wide_reduce_rows_kernel, dim3(cdiv(T * d, 64)), dim3(1024), 0, st, (const float*)scratch, chunks, T * d, dpos,
                       pos_stride, d);
    EGX_LAUNCH_CHECK();
    return 0;
}

static int colsum_row_blocks(int rows, int cols) {
    int cb = cdiv(cols, 512);
    int want = cdiv(1024, cb);
    int rb = cdiv(rows, 64);
    if (rb > want) rb = want;
    return rb < 1 ? 1 : rb;
}
size_t wide_colsum_scratch(int rows, int cols) { return (size_t)colsum_row_blocks(rows, cols) * cols * sizeof(float); }

int wide_colsum_bf16(const bf16_t* x, int rows, int cols, int ld, float* out, void* scratch, hipStream_t st, WideRowReduceBatch* defer) {
    EGX_CHECK(x && out && scratch && cols % 8 == 0 && ld % 8 == 0, "wide_colsum: bad arguments");
    if (rows <= 0) return 0;
    int rb = colsum_row_blocks(rows, cols);
    const int rpb = cdiv(rows, rb);
    rb = cdiv(rows, rpb);
    hipLaunchKernelGGL(wide_colsum_kernel, dim3(cdiv(cols, 512), rb), dim3(256), 0, st, x, rows, cols, ld, rpb, (float*)scratch);
    EGX_LAUNCH_CHECK();
    return wide_reduce_rows((const float*)scratch, rb, cols, out, st, defer);
}

}  // namespace egx
