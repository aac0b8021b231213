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

// ---- classifier head: Linear(K -> C) + weighted cross-entropy in one launch each way ------------------------------------
// The ASD task's lossAV (HHI/tasks/asd/loss.py:11-30: x = FC(x); nloss = CrossEntropyLoss(weight=[1, 4])(x, labels); softmax
// scores, rounded labels and the number of correct frames for logging) is a (B*T, 128) x (2, 128) product followed by a
// per-row softmax. As separate launches (GEMM, bias, CE, three backward GEMMs with split-K reductions, fills) it cost 80 us
// of the 0.53 ms ASD step; here the forward is ONE launch and the backward ONE launch.
//   forward : 16 lanes per row (K / 16 features each), C <= 8 dot products, quad-butterfly reductions; every workgroup first
//             sums the class weights of ALL targets itself (the normaliser depends on labels only), so the loss term and
//             d loss / d logits of its rows are final in the same pass. Loss and correct-frame count: per-workgroup partials,
//             summed in workgroup order by the last workgroup to finish (deterministic; the arrival counter resets itself).
//   backward: dx = g * dlogits W (row-parallel) and per-workgroup partial dW / db (registers -> LDS -> scratch), summed in
//             workgroup order by the last workgroup.
constexpr int LCE_MAXC = 8;
struct LinearCeParams {
    const float* x; const float* W; const float* b; const int64_t* target; const float* weight;
    int M, K, C;
    float* logits; float* probs; float* dlogits;      // (M, C); probs / dlogits optional
    float* loss; float* correct;                      // scalars; correct optional: frames with round(softmax)[:, 1] == label
    float* pred;                                      // (M) optional: round(softmax)[:, 1]
    float* partials;                                  // [blocks][2] scratch
    unsigned* counter;                                // arrival counter, zero before the first launch
};
// K = 64 * KPL (4 * KPL features per lane); CMAX = 2 or 8 classes held in registers; LCE_UNR rows per 16-lane group have their
// loads in flight together (the kernels are chains of memory round trips: every sequential pass costs 2 - 3 us)
template <int KPL, int CMAX, int LCE_UNR>
__global__ __launch_bounds__(256) void linear_ce_fwd_kernel(LinearCeParams p) {
    constexpr int LCE_MAXC = CMAX;
    __shared__ float red[4];
    __shared__ float red2[2][4];
    __shared__ float wsum_s;
    __shared__ bool last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, grp = tid >> 4;     // 16 row groups per block
    float cw[LCE_MAXC];
#pragma unroll
    for (int c = 0; c < LCE_MAXC; ++c) cw[c] = c < p.C ? (p.weight ? p.weight[c] : 1.f) : 0.f;
    auto wof = [&](int64_t y) {         // class weight by compare chain: no load that depends on the label
        float w = 0.f;
#pragma unroll
        for (int c = 0; c < LCE_MAXC; ++c) w = y == c ? cw[c] : w;
        return w;
    };
    // normaliser: sum of the class weights of every valid target (every block computes it: it depends on the labels only)
    float sw = 0.f;
    for (int i0 = tid; i0 < p.M; i0 += 256 * 16) {
        int64_t y[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) y[u] = i0 + u * 256 < p.M ? p.target[i0 + u * 256] : -1;
#pragma unroll
        for (int u = 0; u < 16; ++u) sw += wof(y[u]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sw += __shfl_xor(sw, o, 64);
    if (lane == 0) red[wave] = sw;
    // this lane's slice of W (and the bias) stays in registers
    float wr[LCE_MAXC][4 * KPL], bias[LCE_MAXC];
#pragma unroll
    for (int c = 0; c < LCE_MAXC; ++c) {
        bias[c] = (c < p.C && p.b) ? p.b[c] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KPL; ++kk) {
            float4 wv = c < p.C ? *reinterpret_cast<const float4*>(p.W + (size_t)c * p.K + sub * 4 + kk * 64) : make_float4(0, 0, 0, 0);
            wr[c][4 * kk] = wv.x; wr[c][4 * kk + 1] = wv.y; wr[c][4 * kk + 2] = wv.z; wr[c][4 * kk + 3] = wv.w;
        }
    }
    __syncthreads();
    if (tid == 0) wsum_s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float inv = 1.f / wsum_s;
    float lpart = 0.f, cpart = 0.f;
    const int stride = gridDim.x * 16;
    for (int row0 = blockIdx.x * 16 + grp; row0 < p.M; row0 += stride * LCE_UNR) {
        float4 xv[LCE_UNR][KPL];
        int64_t yv[LCE_UNR];
#pragma unroll
        for (int u = 0; u < LCE_UNR; ++u) {
            const int row = row0 + u * stride;
            const bool in = row < p.M;
            yv[u] = in ? p.target[row] : -1;
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk)
                xv[u][kk] = in ? *reinterpret_cast<const float4*>(p.x + (size_t)row * p.K + sub * 4 + kk * 64) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < LCE_UNR; ++u) {
            const int row = row0 + u * stride;
            if (row >= p.M) break;
            float acc[LCE_MAXC];
#pragma unroll
            for (int c = 0; c < LCE_MAXC; ++c) {
                float a = 0.f;
#pragma unroll
                for (int kk = 0; kk < KPL; ++kk)
                    a += (xv[u][kk].x * wr[c][4 * kk] + xv[u][kk].y * wr[c][4 * kk + 1]) + (xv[u][kk].z * wr[c][4 * kk + 2] + xv[u][kk].w * wr[c][4 * kk + 3]);
                acc[c] = a;
            }
#pragma unroll
            for (int c = 0; c < LCE_MAXC; ++c)
                if (c < p.C) {
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
                    acc[c] += bias[c];
                }
            if (sub == 0) {
                float m = acc[0];
#pragma unroll
                for (int c = 1; c < LCE_MAXC; ++c) if (c < p.C) m = fmaxf(m, acc[c]);
                float e[LCE_MAXC], ssum = 0.f;
#pragma unroll
                for (int c = 0; c < LCE_MAXC; ++c) { e[c] = c < p.C ? __expf(acc[c] - m) : 0.f; ssum += e[c]; }
                const float rs = 1.f / ssum;
                const int64_t y = yv[u];
                const bool ok = y >= 0 && y < p.C;
                const float wy = ok ? wof(y) : 0.f;
#pragma unroll
                for (int c = 0; c < LCE_MAXC; ++c)
                    if (c < p.C) {
                        p.logits[(size_t)row * p.C + c] = acc[c];
                        if (p.probs) p.probs[(size_t)row * p.C + c] = e[c] * rs;
                        if (p.dlogits) p.dlogits[(size_t)row * p.C + c] = ok ? wy * inv * (e[c] * rs - (c == (int)y ? 1.f : 0.f)) : 0.f;
                        if (ok && c == (int)y) lpart += wy * (m + __logf(ssum) - acc[c]);
                    }
                const float pl = p.C > 1 ? rintf(e[1] * rs) : 0.f;
                if (p.pred) p.pred[row] = pl;
                if (p.C > 1 && pl == (float)y) cpart += 1.f;
            }
        }
    }
    // block partials -> scratch; the last block to arrive sums them in block order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lpart += __shfl_xor(lpart, o, 64); cpart += __shfl_xor(cpart, o, 64); }
    if (lane == 0) { red2[0][wave] = lpart; red2[1][wave] = cpart; }
    __syncthreads();
    if (tid == 0) {
        p.partials[2 * blockIdx.x] = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
        p.partials[2 * blockIdx.x + 1] = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
        __threadfence();
        last_s = atomicAdd(p.counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last_s && wave == 0) {
        __threadfence();
        float l = 0.f, cnum = 0.f;                     // <= 64 blocks: lane i takes block i, fixed-order butterfly
        if (lane < (int)gridDim.x) { l = __builtin_nontemporal_load(p.partials + 2 * lane); cnum = __builtin_nontemporal_load(p.partials + 2 * lane + 1); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o, 64); cnum += __shfl_xor(cnum, o, 64); }
        if (lane == 0) {
            *p.loss = l * inv;
            if (p.correct) *p.correct = cnum;
            *p.counter = 0u;
        }
    }
}

struct LinearCeBwdParams {
    const float* x; const float* W; const float* dlogits; const float* gscale;   // gscale: optional device scalar (upstream gradient)
    int M, K, C;
    float* dx; float* dW; float* db;      // dx (M, K) optional; dW (C, K), db (C) written (not accumulated)
    float* partials;                      // [blocks][C * K + C]
    unsigned* counter;
};
template <int KPL, int CMAX, int LCE_UNR>
__global__ __launch_bounds__(256) void linear_ce_bwd_kernel(LinearCeBwdParams p) {
    constexpr int LCE_MAXC = CMAX;
    extern __shared__ float sm[];          // [4 waves][C * K + C]
    __shared__ bool last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid & 15, grp = tid >> 4;
    const float g = p.gscale ? *p.gscale : 1.f;
    float wr[LCE_MAXC][4 * KPL];
    float dwacc[LCE_MAXC][4 * KPL], dbacc[LCE_MAXC];
#pragma unroll
    for (int c = 0; c < LCE_MAXC; ++c) {
        dbacc[c] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KPL; ++kk) {
            float4 wv = c < p.C ? *reinterpret_cast<const float4*>(p.W + (size_t)c * p.K + sub * 4 + kk * 64) : make_float4(0, 0, 0, 0);
            wr[c][4 * kk] = wv.x; wr[c][4 * kk + 1] = wv.y; wr[c][4 * kk + 2] = wv.z; wr[c][4 * kk + 3] = wv.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) dwacc[c][4 * kk + e] = 0.f;
        }
    }
    const int stride = gridDim.x * 16;
    for (int row0 = blockIdx.x * 16 + grp; row0 < p.M; row0 += stride * LCE_UNR) {
        float4 xv[LCE_UNR][KPL];
        float dl[LCE_UNR][LCE_MAXC];
#pragma unroll
        for (int u = 0; u < LCE_UNR; ++u) {
            const int row = row0 + u * stride;
            const bool in = row < p.M;
#pragma unroll
            for (int c = 0; c < LCE_MAXC; ++c) dl[u][c] = (in && c < p.C) ? p.dlogits[(size_t)row * p.C + c] : 0.f;
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk)
                xv[u][kk] = in ? *reinterpret_cast<const float4*>(p.x + (size_t)row * p.K + sub * 4 + kk * 64) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < LCE_UNR; ++u) {
            const int row = row0 + u * stride;
            if (row >= p.M) break;
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) {
                float d[4] = {0.f, 0.f, 0.f, 0.f};
                const float xs[4] = {xv[u][kk].x, xv[u][kk].y, xv[u][kk].z, xv[u][kk].w};
#pragma unroll
                for (int c = 0; c < LCE_MAXC; ++c)
                    if (c < p.C) {
                        const float dlc = dl[u][c] * g;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { d[e] += dlc * wr[c][4 * kk + e]; dwacc[c][4 * kk + e] += dlc * xs[e]; }
                    }
                if (p.dx) *reinterpret_cast<float4*>(p.dx + (size_t)row * p.K + sub * 4 + kk * 64) = make_float4(d[0], d[1], d[2], d[3]);
            }
            if (sub == 0)
#pragma unroll
                for (int c = 0; c < LCE_MAXC; ++c) dbacc[c] += dl[u][c] * g;
        }
    }
    // the four row groups of a wave by butterfly (same feature slice per `sub`), the four waves through LDS in wave order
    const int CK = p.C * p.K, PN = CK + p.C;
#pragma unroll
    for (int c = 0; c < LCE_MAXC; ++c)
        if (c < p.C) {
#pragma unroll
            for (int j = 0; j < 4 * KPL; ++j) { dwacc[c][j] += __shfl_xor(dwacc[c][j], 16, 64); dwacc[c][j] += __shfl_xor(dwacc[c][j], 32, 64); }
            dbacc[c] += __shfl_xor(dbacc[c], 16, 64); dbacc[c] += __shfl_xor(dbacc[c], 32, 64);
            if (lane < 16) {
#pragma unroll
                for (int kk = 0; kk < KPL; ++kk)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sm[wave * PN + c * p.K + sub * 4 + kk * 64 + e] = dwacc[c][4 * kk + e];
                if (sub == 0) sm[wave * PN + CK + c] = dbacc[c];
            }
        }
    __syncthreads();
    float* part = p.partials + (size_t)blockIdx.x * PN;
    for (int i = tid; i < PN; i += 256) part[i] = (sm[i] + sm[PN + i]) + (sm[2 * PN + i] + sm[3 * PN + i]);
    __threadfence();
    __syncthreads();
    if (tid == 0) last_s = atomicAdd(p.counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last_s) {
        __threadfence();
        const int nb = (int)gridDim.x;
        for (int i = tid; i < PN; i += 256) {
            float s = 0.f;
            int b0 = 0;
            for (; b0 + 8 <= nb; b0 += 8) {           // eight loads in flight, block order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p.partials + (size_t)(b0 + u) * PN + i);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; b0 < nb; ++b0) s += __builtin_nontemporal_load(p.partials + (size_t)b0 * PN + i);
            if (i < CK) p.dW[i] = s; else if (p.db) p.db[i - CK] = s;
        }
        if (tid == 0) *p.counter = 0u;
    }
}

// forward: a block per 64 rows (one pass of 4 rows per group), at most 64 (a lane of the last block per partial);
// backward: a block per 256 rows, at most 64 (the last block sums C * K + C partials of every block)
static int lce_blocks_fwd(int M) { int b = (M + 63) / 64; return b < 1 ? 1 : (b > 64 ? 64 : b); }
static int lce_blocks(int M) { int b = (M + 255) / 256; return b < 1 ? 1 : (b > 64 ? 64 : b); }
size_t linear_ce_scratch_bytes(int M, int K, int C) {
    // [arrival counters: 2 x 64 B][forward partials][backward partials]
    return 128 + (size_t)64 * 2 * sizeof(float) + (size_t)lce_blocks(M) * ((size_t)C * K + C) * sizeof(float);
}
int linear_ce_fwd(const float* x, const float* W, const float* b, const int64_t* target, const float* weight, int M, int K, int C,
                  float* logits, float* probs, float* dlogits, float* loss, float* correct, float* pred, void* scratch, hipStream_t st) {
    EGX_CHECK(x && W && target && logits && loss && scratch, "linear_ce_fwd: null pointer argument");
    EGX_CHECK(M >= 1 && C >= 1 && C <= 8 && (K == 64 || K == 128 || K == 256), "linear_ce_fwd: M=%d K=%d C=%d (needs C <= 8, K in {64, 128, 256})", M, K, C);
    LinearCeParams p;
    p.x = x; p.W = W; p.b = b; p.target = target; p.weight = weight; p.M = M; p.K = K; p.C = C;
    p.logits = logits; p.probs = probs; p.dlogits = dlogits; p.loss = loss; p.correct = correct; p.pred = pred;
    p.counter = (unsigned*)scratch;
    p.partials = (float*)((char*)scratch + 128);
    const dim3 grid(lce_blocks_fwd(M));
#define EGX_LCE_FWD(KPL)                                                                                   \
    do {                                                                                                   \
        if (C <= 2) hipLaunchKernelGGL((linear_ce_fwd_kernel<KPL, 2, 4>), grid, dim3(256), 0, st, p);      \
        else hipLaunchKernelGGL((linear_ce_fwd_kernel<KPL, 8, 2>), grid, dim3(256), 0, st, p);             \
    } while (0)
    if (K == 64) EGX_LCE_FWD(1); else if (K == 128) EGX_LCE_FWD(2); else EGX_LCE_FWD(4);
#undef EGX_LCE_FWD
    EGX_LAUNCH_CHECK();
    return 0;
}
int linear_ce_bwd(const float* x, const float* W, const float* dlogits, const float* gscale, int M, int K, int C, float* dx,
                  float* dW, float* db, void* scratch, hipStream_t st) {
    EGX_CHECK(x && W && dlogits && dW && scratch, "linear_ce_bwd: null pointer argument");
    EGX_CHECK(M >= 1 && C >= 1 && C <= 8 && (K == 64 || K == 128 || K == 256), "linear_ce_bwd: M=%d K=%d C=%d (needs C <= 8, K in {64, 128, 256})", M, K, C);
    LinearCeBwdParams p;
    p.x = x; p.W = W; p.dlogits = dlogits; p.gscale = gscale; p.M = M; p.K = K; p.C = C; p.dx = dx; p.dW = dW; p.db = db;
    p.counter = (unsigned*)((char*)scratch + 64);
    p.partials = (float*)((char*)scratch + 128 + (size_t)64 * 2 * sizeof(float));
    const size_t lds = 4 * ((size_t)C * K + C) * sizeof(float);
    const dim3 grid(lce_blocks(M));
#define EGX_LCE_BWD(KPL)                                                                                   \
    do {                                                                                                   \
        if (C <= 2) hipLaunchKernelGGL((linear_ce_bwd_kernel<KPL, 2, 8>), grid, dim3(256), lds, st, p);    \
        else hipLaunchKernelGGL((linear_ce_bwd_kernel<KPL, 8, 2>), grid, dim3(256), lds, st, p);           \
    } while (0)
    if (K == 64) EGX_LCE_BWD(1); else if (K == 128) EGX_LCE_BWD(2); else EGX_LCE_BWD(4);
#undef EGX_LCE_BWD
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
