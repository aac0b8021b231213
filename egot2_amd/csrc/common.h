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
void count_launch();        // kernel launches issued by the library since egx_launch_count_reset (egx_launch_count)

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
        ::egx::count_launch();                                                           \
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

// two fp32 -> one packed bf16 pair (a in the low half) by ONE v_cvt_pk_bf16_f32: the scalar casts above compile to a
// conversion per element plus a merge
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    f32x2_ v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}

// Counter-based dropout RNG: a 64-bit (seed, site) key and a (row, column) element index -> 16 uniform bits.
// Forward and backward regenerate identical masks; nothing is stored. ONE hash serves the four columns of a quad
// (4c .. 4c+3): a multiply-fold round (32 x 32 -> 64, high ^ low: one v_mad_u64_u32) and one multiply-xorshift round give
// 64 bits = 4 x 16, i.e. half a quarter-rate integer multiply per element (round 2: one 32-bit hash per column pair, one
// multiply per element — the multiplies were a quarter of the FFN epilogue of the clip kernels). Statistics of the four
// lanes (uniformity, pairwise / adjacent-row / adjacent-column correlation, joint keep patterns): tools/rng_quality.py.
__device__ __forceinline__ uint2 rand_quad(uint64_t key, uint32_t row, uint32_t colquad) {
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    uint32_t x = (row * 0x9E3779B1U + k1) ^ (colquad * 0x85EBCA77U + k0);
    x ^= x >> 16;
    const uint64_t p = (uint64_t)x * 0x7feb352dU;
    uint32_t y = (uint32_t)p ^ (uint32_t)(p >> 32);
    y ^= y >> 15;
    uint32_t z = y * 0x846ca68bU;
    z ^= z >> 16;
    return make_uint2(z, y);        // columns 4c, 4c+1 <- low / high half of z; 4c+2, 4c+3 <- low / high half of y
}
// keep tests on the 16-bit halves of a hash word without extracting them: the high half by a full-word compare against
// thresh << 16, the low half by a 16-bit compare. Only for callers whose keep-scale is applied elsewhere and is 0 at p >= 1
// (thresh = 65536 wraps to "keep everything" here; the scale 1 / (1 - p) := 0 folded into the weights still zeroes the output).
__device__ __forceinline__ bool keep_hi(uint32_t w, uint32_t thresh) { return w >= (thresh << 16); }
__device__ __forceinline__ bool keep_lo(uint32_t w, uint32_t thresh) { return (uint16_t)w >= (uint16_t)thresh; }
// keep-scale for inverted dropout: returns 0 or 1/(1-p). `thresh` is on the 16-bit scale (drop_threshold).
__device__ __forceinline__ float drop_scale(uint64_t key, uint32_t row, uint32_t col, uint32_t thresh, float inv_keep) {
    const uint2 h = rand_quad(key, row, col >> 2);
    const uint32_t w = (col & 2u) ? h.y : h.x;
    const uint32_t v = (col & 1u) ? (w >> 16) : (w & 0xffffu);
    return v >= thresh ? inv_keep : 0.f;
}
// four adjacent columns col0..col0+3 (col0 a multiple of 4) of one row: one hash
__device__ __forceinline__ void drop_scale4(uint64_t key, uint32_t row, uint32_t col0, uint32_t thresh, float inv_keep, float (&out)[4]) {
    const uint2 h = rand_quad(key, row, col0 >> 2);
    out[0] = (h.x & 0xffffu) >= thresh ? inv_keep : 0.f;
    out[1] = (h.x >> 16) >= thresh ? inv_keep : 0.f;
    out[2] = (h.y & 0xffffu) >= thresh ? inv_keep : 0.f;
    out[3] = (h.y >> 16) >= thresh ? inv_keep : 0.f;
}
// Keep bits of a 16 x 16 tile held in MFMA accumulator layout with the mask ROWS on the registers: lane (r = lane & 15, g = lane >> 4)
// holds rows row0 + 4g + e (e = 0 .. 3) of ONE column 4 cq0 + r — four different hashes of which one 16-bit field each would be used
// (the transposed passes of the attention backwards: dK / dV tiles, mask keyed (query, key)). The tile needs 16 rows x 4 column quads =
// 64 hashes in all: every lane draws ONE, that of (row row0 + 4g + (r & 3), column quad cq0 + (r >> 2)), and the four lanes of a quad
// exchange their 4-bit results by DPP quad broadcasts. Element e is kept iff bit (r & 3) of m4[e] is set.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {       // quad_perm [e, e, e, e]: every lane of a quad reads quad-lane e
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ void tile_keep_rows(uint64_t key, uint32_t row0, uint32_t cq0, int r, int g, uint32_t thresh, uint32_t (&m4)[4]) {
    const uint2 h = rand_quad(key, row0 + (uint32_t)(4 * g + (r & 3)), cq0 + (uint32_t)(r >> 2));
    const uint32_t mw = (uint32_t)((h.x & 0xffffu) >= thresh) | ((uint32_t)((h.x >> 16) >= thresh) << 1) |
                        ((uint32_t)((h.y & 0xffffu) >= thresh) << 2) | ((uint32_t)((h.y >> 16) >= thresh) << 3);
    m4[0] = quad_bcast<0x00>(mw); m4[1] = quad_bcast<0x55>(mw); m4[2] = quad_bcast<0xAA>(mw); m4[3] = quad_bcast<0xFF>(mw);
}
static inline uint32_t drop_threshold(float p) {
    double t = (double)p * 65536.0;
    if (t <= 0) return 0u;
    if (t >= 65536.0) return 65536u;
    uint32_t u = (uint32_t)(t + 0.5);
    return u ? u : 1u;
}
// A dropout key handed to a kernel is either the key itself (site_key() results are ODD) or the device address (8-byte aligned:
// EVEN) of a key that a kernel ahead of it on the stream derives from a device-resident seed (derive_keys: hipGraph replays then
// draw fresh masks). Resolved once per kernel, under the `thresh != 0` test.
__device__ __forceinline__ uint64_t resolve_key(uint64_t k) {
    return (k & 1ull) ? k : *reinterpret_cast<const uint64_t*>(k);
}
__host__ __device__ static inline uint64_t site_key(uint64_t seed, uint32_t layer, uint32_t site) {
    uint64_t k = seed * 0x9E3779B97F4A7C15ull + ((uint64_t)layer << 8 | site) * 0xD1B54A32D192ED03ull;
    k ^= k >> 29;
    return k | 1ull;
}

enum DropSite { SITE_FEAT = 1, SITE_POS = 2, SITE_ATTN = 3, SITE_RES1 = 4, SITE_FFN = 5, SITE_RES2 = 6 };

}  // namespace egx
