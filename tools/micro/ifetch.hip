// Microbenchmark: does straight-line code (executed once, as the non-loop phases of the clip kernels are) run at the issue
// rate of looped code?  N VOP3 instructions (8 bytes each), four independent chains, one wave per SIMD (256 threads / WG,
// one WG per CU), either fully unrolled (REPS = 1: every instruction line is a cold instruction-cache miss) or as a loop of
// REPS passes over a 1/REPS-sized body. Also with an s_barrier every 256 instructions (the clip kernels synchronise often).
// build: hipcc --offload-arch=gfx950 -O3 -o ifetch ifetch.hip ; run: ./ifetch
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N, int REPS, bool BAR>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float m) {
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < N / REPS / 4; ++i) {
            asm volatile("v_fma_f32 %0, %0, %4, %1\n\tv_fma_f32 %1, %1, %4, %2\n\tv_fma_f32 %2, %2, %4, %3\n\tv_fma_f32 %3, %3, %4, %0"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(m));
            if (BAR && (i & 63) == 63) __builtin_amdgcn_s_barrier();
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int N, int REPS, bool BAR>
void run(float* out, long long* cyc) {
    hipLaunchKernelGGL((k<N, REPS, BAR>), dim3(256), dim3(256), 0, 0, out, cyc, 1.0001f);
    hipDeviceSynchronize();
    long long c1; hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<N, REPS, BAR>), dim3(256), dim3(256), 0, 0, out, cyc, 1.0001f);
    hipDeviceSynchronize();
    long long c2; hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost);
    printf("N=%6d instr (%4d KB of code / %d passes)%s: first launch %8lld cycles (%.2f / instr), second %8lld (%.2f / instr)\n", N, N * 8 / 1024 / REPS, REPS,
           BAR ? " + barrier / 256" : "", c1, (double)c1 / N, c2, (double)c2 / N);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<16384, 64, false>(out, cyc);
    run<16384, 8, false>(out, cyc);
    run<16384, 2, false>(out, cyc);
    run<16384, 1, false>(out, cyc);
    run<32768, 1, false>(out, cyc);
    run<16384, 64, true>(out, cyc);
    run<16384, 1, true>(out, cyc);
    run<32768, 1, true>(out, cyc);
    return 0;
}
