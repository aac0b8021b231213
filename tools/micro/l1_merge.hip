// Does the CU's vector L1 merge two waves' requests for the same weight fragment? (round 6 micro-benchmark)
// Every CU (256 workgroups of 512 threads, 8 waves) streams packed-weight-like fragments (3 x 1 KB wave-loads each, ring of 4 in flight per wave)
// from ONE buffer shared by all workgroups (the per-clip FFN launches' situation: every CU reads every weight, the XCD's L2 delivers).
//   mode 0: the 8 waves read 8 disjoint eighths of the buffer            (today: 3.1 MB distinct per CU, 3.1 MB consumed)
//   mode 1: waves w and w + 4 read the SAME quarter, in step             (1.55 MB distinct per CU, 3.1 MB consumed)
//   mode 2: as mode 1, the second wave of a pair half a ring behind
// usage: l1_merge [MB of weights, default 3] [repeats] [s_sleep per fragment: 0 | 1 | 2 | 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int SLEEP>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4* __restrict__ w, int nfrag_total, int mode, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int first, count;       // fragments [first, first + count) of 192 u32x4 rows x 64 lanes each (3 KB)
    if (mode == 0) { count = nfrag_total / 8; first = wave * count; }
    else { count = nfrag_total / 8; first = (wave & 3) * count; }        // modes 1 / 2: a quarter of HALF the buffer per pair: same consumed bytes per CU
    const int rot = (blockIdx.x >> 3) & 3;      // clips of an XCD start 0..3 fragments-of-16 apart (as the product kernels)
    u32x4 ring[4][3];
    auto idx = [&](int i) { int j = i + rot * 16 + ((mode == 2 && wave >= 4) ? 2 : 0); j %= count; return first + j; };
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ring[k][pl] = w[((size_t)idx(k) * 3 + pl) * 64 + lane];
    for (int i = 0; i < count; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) acc ^= ring[k][pl];
            // ~the MFMA work of a fragment in the product (18 MFMAs = 288 cycles at one wave; two waves share the SIMD)
            if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            const int nx = i + k + 4 < count ? i + k + 4 : i + k;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ring[k][pl] = w[((size_t)idx(nx) * 3 + pl) * 64 + lane];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}
int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 3.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nfrag = ((int)(mb * 1024 * 1024 / 3072) / 64) * 64;
    u32x4* w; unsigned* sink;
    hipMalloc(&w, (size_t)nfrag * 3072); hipMalloc(&sink, 4);
    hipMemset(w, 1, (size_t)nfrag * 3072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int slp = argc > 3 ? atoi(argv[3]) : 2;
    auto launch = [&](int mode) {
        if (slp == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(256), dim3(512), 0, 0, w, nfrag, mode, sink);
        else if (slp == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(256), dim3(512), 0, 0, w, nfrag, mode, sink);
        else if (slp == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(256), dim3(512), 0, 0, w, nfrag, mode, sink);
        else hipLaunchKernelGGL(stream_kernel<2>, dim3(256), dim3(512), 0, 0, w, nfrag, mode, sink);
    };
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 3; ++i) launch(mode);
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) launch(mode);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / reps, consumed = (double)nfrag * 3072;     // bytes delivered to registers per CU (same in every mode)
        printf("mode %d: %.1f us per launch, %.1f GB/s per CU consumed (%.2f MB per CU; distinct per CU: %.2f MB)\n", mode, us, consumed / us / 1e3,
               consumed / 1e6, (mode == 0 ? consumed : consumed / 2) / 1e6);
    }
    return 0;
}
