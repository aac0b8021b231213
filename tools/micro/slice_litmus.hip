// Message-passing litmus for the sliced-mode exchange of the per-clip kernels (egot2_amd/csrc/fused_dev.h: slice_publish / slice_wait /
// slice_gather), built from the product's OWN device functions.
//
//   hipcc --offload-arch=gfx950 -O3 -I egot2_amd/csrc tools/micro/slice_litmus.hip -o slice_litmus            (the shipped protocol)
//   hipcc ... -DEGX_LITMUS_NO_WAITCNT ... -o slice_litmus_nowait     (slice_publish WITHOUT its `s_waitcnt vmcnt(0)`)
//
// P producer / consumer workgroup pairs on neighbouring XCDs (block ids 2i, 2i + 1 under round-robin dispatch) run R rounds each:
// the producer fills four LDS partial blocks with a round-dependent pattern and publishes their sum (48 x 128 words behind one flag)
// exactly as a slice of a clip does; the consumer waits for the flag (bounded, retried) and gathers the block; every word that is not
// the round's pattern is a STALE read = the flag overtook the data. Every round has a buffer and a flag of its own (zeroed by the
// host), so a stale word reads 0. All 2P workgroups run together: the fabric is as busy as in a sliced launch.
// usage: slice_litmus [pairs 128] [rounds 100] [launches 5] [extra stores per round 8] [noise workgroups 0] [noise passes 4]
// Output: one line `variant V noise K pairs P rounds R reps L stale_words N stale_rounds M timeouts T`. Exit code 0 always (the caller judges).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "fused_dev.h"

using namespace egx;

constexpr int S = 48, WORDS = 48 * 128;

__device__ __forceinline__ float pattern(int round, int pair, int e) { return (float)(1 + ((round * 131 + pair * 17 + e) & 1023)); }

__global__ __launch_bounds__(256) void litmus_kernel(float* xchg, unsigned* flags, int rounds, unsigned long long* stale_words,
                                                     unsigned long long* stale_rounds, unsigned long long* timeouts, int extra_stores, float* sink,
                                                     int pairs, float4* noise, size_t noise_f4, int noise_iters) {
    __shared__ float blk[4][48 * LDX];
    if ((int)blockIdx.x >= 2 * pairs) {
        // NOISE workgroups: streaming stores and loads over a large buffer for the whole run, so that the fabric and the memory channels the
        // exchanged words cross are busy and unevenly loaded (a flag can only overtake its data where the data's path is slower)
        const size_t nb = gridDim.x - 2 * pairs, me = blockIdx.x - 2 * pairs;
        float4 acc = make_float4(0, 0, 0, 0);
        for (int it = 0; it < noise_iters; ++it)
            for (size_t i = me * 256 + threadIdx.x; i < noise_f4; i += nb * 256) {
                const float4 v = noise[i];
                acc.x += v.x;
                noise[(i * 7 + it) % noise_f4] = make_float4(acc.x, (float)it, v.z, v.w);
            }
        if (acc.x == 12345.678f) sink[0] = acc.x;
        return;
    }
    const int pair = blockIdx.x >> 1, producer = !(blockIdx.x & 1), tid = threadIdx.x;
    for (int rd = 0; rd < rounds; ++rd) {
        float* xs = xchg + ((size_t)pair * rounds + rd) * WORDS;
        unsigned* fl = flags + (size_t)pair * rounds + rd;
        if (producer) {
            // the value is split over the four "wave partials" the way the FFN loop leaves it: q, q, q, v - 3q
            for (int e = tid; e < WORDS; e += 256) {
                const int o = (e >> 7) * LDX + (e & 127);
                const float v = pattern(rd, pair, e), q = 0.25f * v;
                blk[0][o] = q; blk[1][o] = q; blk[2][o] = q; blk[3][o] = v - 3.f * q;
            }
            __syncthreads();
            // unrelated write-through traffic of the same wave in front of the block (the kernels store H tiles, saves, planes)
            for (int k = 0; k < extra_stores; ++k) xchg_store(sink + ((size_t)blockIdx.x * extra_stores + k) * 256 + tid, (float)k);
            slice_publish(blk[0], blk[1], blk[2], blk[3], LDX, S, xs, fl);
            __syncthreads();
        } else {
            int tries = 0;
            while (!slice_wait(fl) && ++tries < 2000) { }
            if (tries >= 2000) { if (tid == 0) atomicAdd(timeouts, 1ull); continue; }
            slice_gather(xs, 1, blk[0], LDX, S);
            unsigned bad = 0;
            for (int e = tid; e < WORDS; e += 256) bad += blk[0][(e >> 7) * LDX + (e & 127)] != pattern(rd, pair, e);
            __shared__ unsigned bad_total;
            if (tid == 0) bad_total = 0;
            __syncthreads();
            if (bad) atomicAdd(&bad_total, bad);
            __syncthreads();
            if (tid == 0 && bad_total) { atomicAdd(stale_words, (unsigned long long)bad_total); atomicAdd(stale_rounds, 1ull); }
            __syncthreads();
        }
    }
}

int main(int argc, char** argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 128, rounds = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 5;
    const int extra = argc > 4 ? atoi(argv[4]) : 8;
    const int noise_blocks = argc > 5 ? atoi(argv[5]) : 0, noise_iters = argc > 6 ? atoi(argv[6]) : 4;
    const size_t noise_f4 = (size_t)64 << 20;      // 1 GiB
    float4* noise = nullptr;
    if (noise_blocks > 0 && hipMalloc(&noise, noise_f4 * 16) != hipSuccess) { printf("noise alloc failed\n"); return 0; }
    float* xchg; unsigned* flags; unsigned long long* ctr; float* sink;
    const size_t xb = (size_t)pairs * rounds * WORDS * 4, fb = (size_t)pairs * rounds * 4, sb = (size_t)2 * pairs * (extra > 0 ? extra : 1) * 256 * 4;
    if (hipMalloc(&xchg, xb) != hipSuccess || hipMalloc(&flags, fb) != hipSuccess || hipMalloc(&ctr, 24) != hipSuccess || hipMalloc(&sink, sb) != hipSuccess) { printf("alloc failed\n"); return 0; }
    hipMemset(ctr, 0, 24);
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(xchg, 0, xb);
        hipMemset(flags, 0, fb);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(litmus_kernel, dim3(2 * pairs + noise_blocks), dim3(256), 0, 0, xchg, flags, rounds, ctr, ctr + 1, ctr + 2, extra, sink, pairs, noise, noise_f4, noise_iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 0; }
    }
    unsigned long long h[3];
    hipMemcpy(h, ctr, 24, hipMemcpyDeviceToHost);
#ifdef EGX_LITMUS_NO_WAITCNT
    const char* variant = "no_waitcnt";
#else
    const char* variant = "shipped";
#endif
    printf("variant %s noise %d pairs %d rounds %d reps %d stale_words %llu stale_rounds %llu timeouts %llu\n", variant, noise_blocks, pairs, rounds, reps, h[0], h[1], h[2]);
    return 0;
}
