// Shared device/host helpers for libegot2x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace egx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

void set_error(const char* fmt, ...);

#define EGX_CHECK(cond, ...)                         \
    do {                                             \
        if (!(cond)) {                               \
            ::egx::set_error(__VA_ARGS__);           \
            return 1;                                \
        }                                            \
    } while (0)

#define EGX_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::egx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                             __FILE__, __LINE__);                                        \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

#define EGX_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            ::egx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),  \
                             __FILE__, __LINE__);                                        \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// fp32 -> bf16 round-to-nearest-even (plain cast keeps NaNs, v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}

// Counter-based dropout RNG: a 64-bit (seed, site) key and a 32-bit element index -> uniform u32.
// Forward and backward regenerate identical masks; nothing is stored.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rand_u32(uint64_t key, uint32_t idx_hi, uint32_t idx_lo) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    uint32_t h = mix32(idx_lo ^ k0);
    h = mix32(h + idx_hi * 0x9E3779B9U + k1);
    return mix32(h ^ (k0 * 0x85ebca6bU));
}
// keep-scale for inverted dropout: returns 0 or 1/(1-p)
__device__ __forceinline__ float drop_scale(uint64_t key, uint32_t idx_hi, uint32_t idx_lo,
                                            uint32_t thresh, float inv_keep) {
    return rand_u32(key, idx_hi, idx_lo) >= thresh ? inv_keep : 0.f;
}
static inline uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0) return 0u;
    if (t >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)t;
}
__host__ __device__ static inline uint64_t site_key(uint64_t seed, uint32_t layer, uint32_t site) {
    uint64_t k = seed * 0x9E3779B97F4A7C15ull + ((uint64_t)layer << 8 | site) * 0xD1B54A32D192ED03ull;
    k ^= k >> 29;
    return k | 1ull;
}

enum DropSite { SITE_FEAT = 1, SITE_POS = 2, SITE_ATTN = 3, SITE_RES1 = 4, SITE_FFN = 5, SITE_RES2 = 6 };

}  // namespace egx
