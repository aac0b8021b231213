// Microbenchmark: v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_bf16 issue rate, one wave per SIMD,
// with NCHAIN independent accumulator chains issued round-robin (NCHAIN=1: every MFMA depends on the previous).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_dep mfma_dep.hip ; run: ./mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NCHAIN, bool BF16, int VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    f32x4 acc[NCHAIN];
    for (int c = 0; c < NCHAIN; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (short)(threadIdx.x + i); bh[i] = (short)(threadIdx.x * 3 + i); }
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 48 / NCHAIN; ++j) {
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) {
                if constexpr (BF16) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
                if constexpr (VALU < 0) {
#pragma unroll
                    for (int u = 0; u < -VALU; ++u) {
                        v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
                        v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (VALU > 0) {
#pragma unroll
                    for (int u = 0; u < VALU; ++u) {
                        v0 = __builtin_fmaf(v0, 1.0001f, v1); v1 = __builtin_fmaf(v1, 0.9999f, v2);
                        v2 = __builtin_fmaf(v2, 1.0002f, v3); v3 = __builtin_fmaf(v3, 0.9998f, v0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3;
    for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NCHAIN, bool BF16, int VALU>
void run(const char* name, float* out, long long* cyc, int grid = 256) {
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCHAIN, BF16, VALU>), dim3(256), dim3(256), 0, 0, out, 10, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCHAIN, BF16, VALU>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 48 * (grid / 256);
    printf("%-34s %8.2f ns/MFMA  %7.2f memtime-ticks/MFMA  (%.1f us)\n", name, ms * 1e6 / n, (double)c / n, ms * 1e3);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    run<1, false, 0>("f32 16x16x4  chains=1", out, cyc);
    run<2, false, 0>("f32 16x16x4  chains=2", out, cyc);
    run<3, false, 0>("f32 16x16x4  chains=3", out, cyc);
    run<6, false, 0>("f32 16x16x4  chains=6", out, cyc);
    run<1, false, 1>("f32 chains=1 + 4 VALU/MFMA", out, cyc);
    run<3, false, 1>("f32 chains=3 + 4 VALU/MFMA", out, cyc);
    run<3, false, 2>("f32 chains=3 + 8 VALU/MFMA", out, cyc);
    run<3, false, 3>("f32 chains=3 + 12 VALU/MFMA", out, cyc);
    run<3, false, -1>("f32 chains=3 + 4 indep VALU/MFMA", out, cyc);
    run<3, false, -2>("f32 chains=3 + 8 indep VALU/MFMA", out, cyc);
    run<3, false, 0>("f32 chains=3, 2 WG/CU", out, cyc, 512);
    run<3, false, 1>("f32 chains=3 + 4 VALU, 2 WG/CU", out, cyc, 512);
    run<3, false, 2>("f32 chains=3 + 8 VALU, 2 WG/CU", out, cyc, 512);
    run<3, false, -2>("f32 chains=3 + 8 indep VALU, 2 WG/CU", out, cyc, 512);
    run<3, false, -4>("f32 chains=3 + 16 indep VALU, 2 WG/CU", out, cyc, 512);
    run<1, true, 0>("bf16 16x16x32 chains=1", out, cyc);
    run<2, true, 0>("bf16 16x16x32 chains=2", out, cyc);
    run<3, true, 0>("bf16 16x16x32 chains=3", out, cyc);
    run<6, true, 0>("bf16 16x16x32 chains=6", out, cyc);
    run<3, true, 1>("bf16 chains=3 + 4 VALU/MFMA", out, cyc);
    run<3, true, 2>("bf16 chains=3 + 8 VALU/MFMA", out, cyc);
    run<3, true, 0>("bf16 chains=3, 2 WG/CU", out, cyc, 512);
    run<3, true, -2>("bf16 chains=3 + 8 indep VALU, 2 WG/CU", out, cyc, 512);
    run<3, true, -2>("bf16 chains=3 + 8 indep VALU, 4 WG/CU", out, cyc, 1024);
    return 0;
}
