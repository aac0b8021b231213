// Producer side of the feature hand-off: the head of the frozen PNR / OSCC backbones as ONE kernel that reads
// the res5 feature map once and writes the packed token rows the translator's projection GEMM consumes in place.
//
// Replaces, for `middle=True`, ResNetKeyframeLocalizationHead.forward (HOI/models/pnr/head_helper.py:353-373):
//     pathwayN_avgpool = AvgPool3d((kt, kh, kw), stride 1)   (N, C, T, H, W) -> (N, C, T', H', W')
//     permute (0, 2, 3, 4, 1), reshape                        -> (N, T', H' W' C)   (8192 = 2 * 2 * 2048 columns, channel fastest)
// and, fused behind it, the temporal mean of encode_clips_pnr (`model([x[:, i]], middle=True).mean(dim=1)`,
// HOI/models/lta/lta_models_lta_transfer.py:335-345) and the cast to the bf16 operand format of the wide path: the fp32
// (N, T', 8192) intermediate, its permuted copy and the stacked per-clip means are never materialised.
//
// One workgroup per (map n, block of 64 channels): per input frame the 64 channel planes (H W contiguous floats each) are staged
// through LDS with coalesced reads, thread (o, c) sums its kh x kw window of output position o = (h', w') and either emits the
// frame's value (kt = 1) or accumulates it over the frames (kt = T, or frames_mean). Rows leave with the channel on the lane:
// 128-byte (bf16) / 256-byte (fp32) segments.
#include "../../include/egot2x.h"
#include "common.h"

namespace egx {

namespace {
constexpr int PP_CB = 64;           // channels per workgroup
constexpr int PP_MAX_HW = 144;      // H * W of the feature map (res5 of a 224 .. 384 pixel crop: 7 x 7 .. 12 x 12)

struct PoolPackParams {
    const void* fmap; int fmap_bf16;
    int N, C, T, H, W, kt, kh, kw, frames_mean;
    void* out; int out_bf16;
    long long out_map_stride;       // elements between the first output rows of consecutive maps
};

__device__ __forceinline__ float ld_elem(const void* base, size_t i, int bf16) {
    if (bf16) return __uint_as_float((uint32_t)reinterpret_cast<const unsigned short*>(base)[i] << 16);
    return reinterpret_cast<const float*>(base)[i];
}

__global__ __launch_bounds__(256) void pool_pack_kernel(PoolPackParams p) {
    __shared__ float tile[PP_CB][PP_MAX_HW + 1];
    const int n = blockIdx.x, c0 = blockIdx.y * PP_CB;
    const int HW = p.H * p.W, Ho = p.H - p.kh + 1, Wo = p.W - p.kw + 1, No = Ho * Wo;
    const int To = p.T - p.kt + 1;                          // kt == 1 or kt == T (checked by the host)
    const bool reduce_t = p.kt > 1 || p.frames_mean;
    const int rows_out = reduce_t ? 1 : To;
    const size_t row_len = (size_t)No * p.C;
    const float inv_win = 1.f / (float)(p.kh * p.kw), inv_t = 1.f / (float)p.T;
    // thread -> (output position o, channel c): consecutive lanes = consecutive channels (coalesced stores)
    constexpr int MAXO = 4;                                 // output positions a thread may own (No <= 16 with 256 threads)
    float acc[MAXO];
#pragma unroll
    for (int k = 0; k < MAXO; ++k) acc[k] = 0.f;
    for (int t = 0; t < p.T; ++t) {
        __syncthreads();
        for (int i = threadIdx.x; i < PP_CB * HW; i += 256) {
            const int c = i / HW, s = i - c * HW;
            float v = 0.f;
            if (c0 + c < p.C) v = ld_elem(p.fmap, (((size_t)n * p.C + c0 + c) * p.T + t) * HW + s, p.fmap_bf16);
            tile[c][s] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXO; ++k) {
            const int idx = threadIdx.x + k * 256, o = idx / PP_CB, c = idx - o * PP_CB;
            if (o >= No) break;
            const int ho = o / Wo, wo = o - ho * Wo;
            float s = 0.f;
            for (int dh = 0; dh < p.kh; ++dh)
                for (int dw = 0; dw < p.kw; ++dw) s += tile[c][(ho + dh) * p.W + wo + dw];
            s *= inv_win;
            if (reduce_t) {
                acc[k] += s;
            } else if (c0 + c < p.C) {
                const size_t dst = (size_t)n * p.out_map_stride + (size_t)t * row_len + (size_t)o * p.C + c0 + c;
                if (p.out_bf16) reinterpret_cast<unsigned short*>(p.out)[dst] = f2bf(s);
                else reinterpret_cast<float*>(p.out)[dst] = s;
            }
        }
    }
    if (reduce_t) {
#pragma unroll
        for (int k = 0; k < MAXO; ++k) {
            const int idx = threadIdx.x + k * 256, o = idx / PP_CB, c = idx - o * PP_CB;
            if (o >= No || c0 + c >= p.C) break;
            const size_t dst = (size_t)n * p.out_map_stride + (size_t)o * p.C + c0 + c;
            const float v = acc[k] * inv_t;
            if (p.out_bf16) reinterpret_cast<unsigned short*>(p.out)[dst] = f2bf(v);
            else reinterpret_cast<float*>(p.out)[dst] = v;
        }
    }
    (void)rows_out;
}
}  // namespace

}  // namespace egx

using namespace egx;

extern "C" int egx_pool_pack(const void* fmap, int fmap_bf16, int N, int C, int T, int H, int W, int kt, int kh, int kw,
                             int frames_mean, void* out, int out_bf16, long long out_map_stride, void* stream) {
    EGX_CHECK(fmap && out, "egx_pool_pack: null pointer argument");
    EGX_CHECK(N > 0 && C > 0 && T > 0 && H > 0 && W > 0, "egx_pool_pack: empty feature map (%d, %d, %d, %d, %d)", N, C, T, H, W);
    EGX_CHECK(kh >= 1 && kh <= H && kw >= 1 && kw <= W, "egx_pool_pack: spatial window %d x %d outside the %d x %d map", kh, kw, H, W);
    EGX_CHECK(kt == 1 || kt == T, "egx_pool_pack: temporal window %d (supported: 1, or all %d frames)", kt, T);
    EGX_CHECK(H * W <= PP_MAX_HW, "egx_pool_pack: H * W = %d > %d", H * W, PP_MAX_HW);
    const int No = (H - kh + 1) * (W - kw + 1);
    EGX_CHECK(No * PP_CB <= 4 * 256, "egx_pool_pack: %d output positions per frame (at most 16)", No);
    const long long rows = (kt > 1 || frames_mean) ? 1 : T;
    EGX_CHECK(out_map_stride >= rows * (long long)No * C, "egx_pool_pack: out_map_stride %lld smaller than a map's %lld output elements",
              out_map_stride, rows * (long long)No * C);
    PoolPackParams p;
    p.fmap = fmap; p.fmap_bf16 = fmap_bf16; p.N = N; p.C = C; p.T = T; p.H = H; p.W = W; p.kt = kt; p.kh = kh; p.kw = kw;
    p.frames_mean = frames_mean; p.out = out; p.out_bf16 = out_bf16; p.out_map_stride = out_map_stride;
    hipLaunchKernelGGL(pool_pack_kernel, dim3(N, cdiv(C, PP_CB)), dim3(256), 0, (hipStream_t)stream, p);
    EGX_LAUNCH_CHECK();
    return 0;
}
