// Training-step kernels around the translator: the weighted cross-entropy of the TTM task and
// the Adam / AdamW update over flat parameter buffers.
//
//   weighted_ce   HHI/tasks/ttm/video_task_2loader.py:21-22,34: nn.CrossEntropyLoss(weight=[0.266, 0.734]) =
//                 sum_i w[y_i] * nll_i / sum_i w[y_i]; one launch produces the loss AND d loss / d logits (torch needs
//                 log_softmax + nll_loss forward and two backward kernels for the same).
//   adam_step     HHI/tasks/ttm/video_task_2loader.py:62-64 (Adam, lr 5e-4, wd 0) and
//                 HOI/tasks/multitask/video_task.py:624-626 (AdamW, lr 1e-4, wd 1e-4); semantics of torch.optim.Adam /
//                 AdamW (bias-corrected, eps added outside the square root). The step counter lives in device memory so
//                 that a captured hipGraph (forward + loss + backward + update) replays with the right bias correction.
#include "common.h"
#include "kernels.h"

namespace egx {

// One workgroup; thread i owns samples i, i + 1024, ... (B is a few hundred to a few thousand, C a handful).
__global__ __launch_bounds__(1024) void weighted_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                           const float* __restrict__ weight, int B, int C,
                                                           float* __restrict__ loss, float* __restrict__ dlogits) {
    __shared__ float red[2][16];
    __shared__ float tot[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sw = 0.f, sl = 0.f;
    for (int i = tid; i < B; i += 1024) {
        const float* z = logits + (size_t)i * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(z[c] - m);
        const int64_t y = target[i];
        if (y < 0 || y >= C) continue;      // ignore_index (-100) / out-of-range label: no loss, no weight, no gradient
        float wy = weight ? weight[y] : 1.f;
        sw += wy;
        sl += wy * (m + __logf(s) - z[y]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sw += __shfl_xor(sw, o, 64); sl += __shfl_xor(sl, o, 64); }
    if (lane == 0) { red[0][wave] = sw; red[1][wave] = sl; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
        tot[0] = a; tot[1] = b;
        *loss = b / a;
    }
    __syncthreads();
    if (!dlogits) return;
    const float inv = 1.f / tot[0];
    for (int i = tid; i < B; i += 1024) {
        const float* z = logits + (size_t)i * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(z[c] - m);
        const int64_t y = target[i];
        if (y < 0 || y >= C) {
            for (int c = 0; c < C; ++c) dlogits[(size_t)i * C + c] = 0.f;
            continue;
        }
        float k = (weight ? weight[y] : 1.f) * inv, rs = 1.f / s;
        for (int c = 0; c < C; ++c) dlogits[(size_t)i * C + c] = k * (__expf(z[c] - m) * rs - (c == (int)y ? 1.f : 0.f));
    }
}

int weighted_ce(const float* logits, const int64_t* target, const float* weight, int B, int C, float* loss,
                float* dlogits, hipStream_t st) {
    EGX_CHECK(logits && target && loss, "weighted_ce: null pointer argument");
    EGX_CHECK(B >= 1 && C >= 1, "weighted_ce: B=%d C=%d", B, C);
    hipLaunchKernelGGL(weighted_ce_kernel, dim3(1), dim3(1024), 0, st, logits, target, weight, B, C, loss, dlogits);
    EGX_LAUNCH_CHECK();
    return 0;
}

// p, g, m, v: flat fp32 buffers of n elements (16-byte aligned). *step is the 1-based step count; the caller bumps it
// with counter_add on the same stream before the update (race-free and replayable inside a hipGraph).
template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const int64_t* __restrict__ step,
                                                   float lr, float b1, float b2, float eps, float wd, int decoupled,
                                                   float grad_scale) {
    const float t = (float)*step;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (VEC && i + 4 <= n) {
        float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i);
        float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
        float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
        float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gg = ga[e] * grad_scale;
            if (decoupled) pa[e] *= 1.f - lr * wd; else gg += wd * pa[e];
            ma[e] = b1 * ma[e] + (1.f - b1) * gg;
            va[e] = b2 * va[e] + (1.f - b2) * gg * gg;
            pa[e] -= step_size * ma[e] / (sqrtf(va[e]) * inv_sqrt_bc2 + eps);
        }
        *reinterpret_cast<float4*>(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
        *reinterpret_cast<float4*>(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(va[0], va[1], va[2], va[3]);
    } else {
        for (size_t e = i + 4 < n ? i + 4 : n; i < e; ++i) {
            float gg = g[i] * grad_scale, pp = p[i];
            if (decoupled) pp *= 1.f - lr * wd; else gg += wd * pp;
            float mm = b1 * m[i] + (1.f - b1) * gg, vv = b2 * v[i] + (1.f - b2) * gg * gg;
            m[i] = mm; v[i] = vv;
            p[i] = pp - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
        }
    }
}

__global__ void counter_add_kernel(int64_t* c, int64_t inc) { *c += inc; }

int counter_add(int64_t* c, int64_t inc, hipStream_t st) {
    EGX_CHECK(c, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, st, c, inc);
    EGX_LAUNCH_CHECK();
    return 0;
}

int adam_step(float* p, const float* g, float* m, float* v, size_t n, const int64_t* step, float lr, float b1, float b2,
              float eps, float wd, int decoupled, float grad_scale, hipStream_t st) {
    EGX_CHECK(p && g && m && v && step, "adam_step: null pointer argument");
    if (n == 0) return 0;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    size_t blocks = (n / 4 + 1 + 255) / 256;
    if (vec) hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, step, lr, b1, b2, eps, wd, decoupled, grad_scale);
    else hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, step, lr, b1, b2, eps, wd, decoupled, grad_scale);
    EGX_LAUNCH_CHECK();
    return 0;
}

}  // namespace egx
