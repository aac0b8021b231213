// Generic scaled-dot-product self-attention (forward + backward) on the packed in-projection output.
//
// Replaces the per-head bmm / softmax / dropout / bmm chain inside F.multi_head_attention_forward as used by
// nn.TransformerEncoderLayer (reference call site HHI/models/ttm/model_taskspecific.py:242).
//
// This is the shape-generic path (any S, d_h in {32,64,96,128}); attention is 2-3 % of the translator's FLOPs.
// One workgroup = 64 query (or key) rows of one (clip, head); every row is owned by 4 adjacent lanes that each
// hold a quarter of the head channels, so a score is d_h/4 FMAs + a 2-step quad reduction (DPP), and the
// running softmax lives in registers. K/V (or Q/dO) rows are staged through LDS in chunks of <= 64 KiB and
// are broadcast-read (all 16 row groups of a wave read the same address). No S x S matrix touches HBM: the
// backward recomputes probabilities from the saved per-row log-sum-exp.
#include "common.h"
#include "kernels.h"

namespace egx {

__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}

template <int DQ>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int row_stride, int nrows) {
    // dst[r][DH + 4] <- src[r * row_stride + 0..DH)
    constexpr int DH = DQ * 4;
    constexpr int LD = DH + 4;
    constexpr int V4 = DH / 4;
    for (int f = threadIdx.x; f < nrows * V4; f += blockDim.x) {
        int r = f / V4, c = (f % V4) * 4;
        float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * row_stride + c);
        *reinterpret_cast<float4*>(dst + r * LD + c) = v;
    }
}

template <int DQ>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        float* __restrict__ lse, int S, int H, int D, float scale, int SC,
                                                        uint64_t dkey, uint32_t dthresh, float dinv) {
    constexpr int DH = DQ * 4;
    constexpr int LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (size_t)SC * LD;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int sub = threadIdx.x & 3;
    const int qi = blockIdx.y * 64 + (threadIdx.x >> 2);
    const bool active = qi < S;
    const float* base = qkv + (size_t)b * S * 3 * D;
    float q[DQ], acc[DQ];
#pragma unroll
    for (int c = 0; c < DQ; ++c) {
        q[c] = active ? base[(size_t)qi * 3 * D + h * DH + sub * DQ + c] * scale : 0.f;
        acc[c] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int c0 = 0; c0 < S; c0 += SC) {
        int nc = min(SC, S - c0);
        __syncthreads();
        stage_rows<DQ>(Ks, base + (size_t)c0 * 3 * D + D + h * DH, 3 * D, nc);
        stage_rows<DQ>(Vs, base + (size_t)c0 * 3 * D + 2 * D + h * DH, 3 * D, nc);
        __syncthreads();
        if (active) {
            for (int j = 0; j < nc; ++j) {
                const float* kr = Ks + j * LD + sub * DQ;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < DQ; ++c) s += q[c] * kr[c];
                s = quad_sum(s);
                float mn = fmaxf(m, s);
                float corr = __expf(m - mn);
                float p = __expf(s - mn);
                l = l * corr + p;
                m = mn;
                if (dthresh) p *= drop_scale(dkey, (uint32_t)(bh * S + qi), (uint32_t)(c0 + j), dthresh, dinv);
                const float* vr = Vs + j * LD + sub * DQ;
#pragma unroll
                for (int c = 0; c < DQ; ++c) acc[c] = acc[c] * corr + p * vr[c];
            }
        }
    }
    if (active) {
        float inv_l = 1.f / l;
        float* o = out + ((size_t)b * S + qi) * D + h * DH + sub * DQ;
#pragma unroll
        for (int c = 0; c < DQ; ++c) o[c] = acc[c] * inv_l;
        if (sub == 0) lse[(size_t)bh * S + qi] = m + __logf(l);
    }
}

// dQ: one query row per lane quad; streams K/V chunks.
template <int DQ>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                           const float* __restrict__ lse, const float* __restrict__ d_out,
                                                           float* __restrict__ d_qkv, int S, int H, int D, float scale, int SC,
                                                           uint64_t dkey, uint32_t dthresh, float dinv) {
    constexpr int DH = DQ * 4;
    constexpr int LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (size_t)SC * LD;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int sub = threadIdx.x & 3;
    const int qi = blockIdx.y * 64 + (threadIdx.x >> 2);
    const bool active = qi < S;
    const float* base = qkv + (size_t)b * S * 3 * D;
    float q[DQ], go[DQ], dq[DQ];
    float delta = 0.f, li = 0.f;
#pragma unroll
    for (int c = 0; c < DQ; ++c) {
        size_t oc = ((size_t)b * S + qi) * D + h * DH + sub * DQ + c;
        q[c] = active ? base[(size_t)qi * 3 * D + h * DH + sub * DQ + c] : 0.f;
        go[c] = active ? d_out[oc] : 0.f;
        float ov = active ? out[oc] : 0.f;
        delta += go[c] * ov;
        dq[c] = 0.f;
    }
    delta = quad_sum(delta);
    if (active) li = lse[(size_t)bh * S + qi];
    for (int c0 = 0; c0 < S; c0 += SC) {
        int nc = min(SC, S - c0);
        __syncthreads();
        stage_rows<DQ>(Ks, base + (size_t)c0 * 3 * D + D + h * DH, 3 * D, nc);
        stage_rows<DQ>(Vs, base + (size_t)c0 * 3 * D + 2 * D + h * DH, 3 * D, nc);
        __syncthreads();
        if (active) {
            for (int j = 0; j < nc; ++j) {
                const float* kr = Ks + j * LD + sub * DQ;
                const float* vr = Vs + j * LD + sub * DQ;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < DQ; ++c) { s += q[c] * kr[c]; dp += go[c] * vr[c]; }
                s = quad_sum(s);
                dp = quad_sum(dp);
                float p = __expf(s * scale - li);
                if (dthresh) dp *= drop_scale(dkey, (uint32_t)(bh * S + qi), (uint32_t)(c0 + j), dthresh, dinv);
                float ds = p * (dp - delta) * scale;
#pragma unroll
                for (int c = 0; c < DQ; ++c) dq[c] += ds * kr[c];
            }
        }
    }
    if (active) {
        float* o = d_qkv + ((size_t)b * S + qi) * 3 * D + h * DH + sub * DQ;
#pragma unroll
        for (int c = 0; c < DQ; ++c) o[c] = dq[c];
    }
}

// dK, dV: one key row per lane quad; streams Q/dO chunks (+ per-row lse and delta).
template <int DQ>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                            const float* __restrict__ lse, const float* __restrict__ d_out,
                                                            float* __restrict__ d_qkv, int S, int H, int D, float scale, int SC,
                                                            uint64_t dkey, uint32_t dthresh, float dinv) {
    constexpr int DH = DQ * 4;
    constexpr int LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Gs = smem + (size_t)SC * LD;
    float* Ls = smem + (size_t)2 * SC * LD;   // lse per staged row
    float* Ds = Ls + SC;                      // delta per staged row
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int sub = threadIdx.x & 3;
    const int kj = blockIdx.y * 64 + (threadIdx.x >> 2);
    const bool active = kj < S;
    const float* base = qkv + (size_t)b * S * 3 * D;
    float k[DQ], v[DQ], dk[DQ], dv[DQ];
#pragma unroll
    for (int c = 0; c < DQ; ++c) {
        k[c] = active ? base[(size_t)kj * 3 * D + D + h * DH + sub * DQ + c] : 0.f;
        v[c] = active ? base[(size_t)kj * 3 * D + 2 * D + h * DH + sub * DQ + c] : 0.f;
        dk[c] = 0.f;
        dv[c] = 0.f;
    }
    for (int c0 = 0; c0 < S; c0 += SC) {
        int nc = min(SC, S - c0);
        __syncthreads();
        stage_rows<DQ>(Qs, base + (size_t)c0 * 3 * D + h * DH, 3 * D, nc);
        stage_rows<DQ>(Gs, d_out + ((size_t)b * S + c0) * D + h * DH, D, nc);
        for (int r = threadIdx.x; r < nc; r += blockDim.x) {
            const float* gr = d_out + ((size_t)b * S + c0 + r) * D + h * DH;
            const float* orow = out + ((size_t)b * S + c0 + r) * D + h * DH;
            float dl = 0.f;
            for (int c = 0; c < DH; ++c) dl += gr[c] * orow[c];
            Ds[r] = dl;
            Ls[r] = lse[(size_t)bh * S + c0 + r];
        }
        __syncthreads();
        if (active) {
            for (int i = 0; i < nc; ++i) {
                const float* qr = Qs + i * LD + sub * DQ;
                const float* gr = Gs + i * LD + sub * DQ;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < DQ; ++c) { s += qr[c] * k[c]; dp += gr[c] * v[c]; }
                s = quad_sum(s);
                dp = quad_sum(dp);
                float p = __expf(s * scale - Ls[i]);
                float msk = 1.f;
                if (dthresh) msk = drop_scale(dkey, (uint32_t)(bh * S + c0 + i), (uint32_t)kj, dthresh, dinv);
                float pd = p * msk;
                float ds = p * (dp * msk - Ds[i]) * scale;
#pragma unroll
                for (int c = 0; c < DQ; ++c) { dv[c] += pd * gr[c]; dk[c] += ds * qr[c]; }
            }
        }
    }
    if (active) {
        float* ok = d_qkv + ((size_t)b * S + kj) * 3 * D + D + h * DH + sub * DQ;
        float* ov = d_qkv + ((size_t)b * S + kj) * 3 * D + 2 * D + h * DH + sub * DQ;
#pragma unroll
        for (int c = 0; c < DQ; ++c) { ok[c] = dk[c]; ov[c] = dv[c]; }
    }
}

static int chunk_rows(int S, int DH, size_t extra_per_row) {
    size_t per_row = (size_t)2 * (DH + 4) * sizeof(float) + extra_per_row;
    int sc = (int)((size_t)60 * 1024 / per_row);
    return sc < S ? sc : S;
}

#define EGX_ATTN_DISPATCH(KERNEL, ...)                                                                   \
    switch (DH) {                                                                                        \
        case 16: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), lds, st, __VA_ARGS__); break;            \
        case 32: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), lds, st, __VA_ARGS__); break;            \
        case 64: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(256), lds, st, __VA_ARGS__); break;           \
        case 96: hipLaunchKernelGGL(KERNEL<24>, grid, dim3(256), lds, st, __VA_ARGS__); break;           \
        case 128: hipLaunchKernelGGL(KERNEL<32>, grid, dim3(256), lds, st, __VA_ARGS__); break;          \
        default: EGX_CHECK(false, "attention: head dim %d unsupported (16/32/64/96/128)", DH);              \
    }

int attention_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int d,
                  uint64_t dkey, uint32_t dthresh, float dinv, hipStream_t st) {
    EGX_CHECK(H > 0 && d % H == 0, "attention: d=%d not divisible by heads=%d", d, H);
    EGX_CHECK((((uintptr_t)qkv) & 15) == 0, "attention: qkv must be 16-byte aligned");
    if (B <= 0 || S <= 0) return 0;
    int DH = d / H;
    float scale = 1.f / sqrtf((float)DH);
    int SC = chunk_rows(S, DH, 0);
    size_t lds = (size_t)2 * SC * (DH + 4) * sizeof(float);
    dim3 grid(B * H, cdiv(S, 64));
    EGX_ATTN_DISPATCH(attn_fwd_kernel, qkv, out, lse, S, H, d, scale, SC, dkey, dthresh, dinv)
    EGX_LAUNCH_CHECK();
    return 0;
}

int attention_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, float* d_qkv,
                  int B, int S, int H, int d, uint64_t dkey, uint32_t dthresh, float dinv, hipStream_t st) {
    EGX_CHECK(H > 0 && d % H == 0, "attention: d=%d not divisible by heads=%d", d, H);
    EGX_CHECK((((uintptr_t)qkv) & 15) == 0 && (((uintptr_t)d_out) & 15) == 0, "attention: inputs must be 16-byte aligned");
    if (B <= 0 || S <= 0) return 0;
    int DH = d / H;
    float scale = 1.f / sqrtf((float)DH);
    dim3 grid(B * H, cdiv(S, 64));
    {
        int SC = chunk_rows(S, DH, 0);
        size_t lds = (size_t)2 * SC * (DH + 4) * sizeof(float);
        EGX_ATTN_DISPATCH(attn_bwd_dq_kernel, qkv, out, lse, d_out, d_qkv, S, H, d, scale, SC, dkey, dthresh, dinv)
        EGX_LAUNCH_CHECK();
    }
    {
        int SC = chunk_rows(S, DH, 2 * sizeof(float));
        size_t lds = (size_t)2 * SC * (DH + 4) * sizeof(float) + (size_t)2 * SC * sizeof(float);
        EGX_ATTN_DISPATCH(attn_bwd_dkv_kernel, qkv, out, lse, d_out, d_qkv, S, H, d, scale, SC, dkey, dthresh, dinv)
        EGX_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace egx
