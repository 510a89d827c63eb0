// Per-frame scale of the initial depth maps against COLMAP's dense depth -- the stage that writes the `metadata_scaled.npz` the hot
// path reads.
//
// Replaces (reference, /root/reference): scale_calibration.py:253-278 -- per frame
//     ix     = isfinite(inv_cmp_depth)
//     scales = (inv_src_depth / inv_cmp_depth)[ix]          float32 division
//     scale  = np.median(scales)                            mean of the two middle values for an even count, computed in float32
//     scaled_inv_src_depth = inv_src_depth / scale
// EXACT selection, bit for bit np.median's number (tests/test_scale_gpu.py): one workgroup per frame, MSB-first radix select over
// the order-preserving integer image of the float32 ratios (4 passes of an 8-bit LDS histogram), one more pass for the upper of the
// two middle values.  A NaN among the selected ratios (0 / 0: a finite COLMAP value of 0 under a 0 initial value) makes the median
// NaN, like numpy's.  HBM-bound in principle (8 B per pixel, re-read 5-6 times from L2: a frame is 0.7 MB); an offline stage.
#include "cd_common.h"

namespace cd {

constexpr int kScaleThreads = 1024;

__device__ __forceinline__ unsigned order_key(float x) {        // monotonic: a < b  <=>  key(a) < key(b)  (-0 < +0: both are the value 0)
    const unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(kScaleThreads) void frame_median_scale_kernel(const float* __restrict__ inv_src, const float* __restrict__ inv_cmp,
                                                                           int HW, float* __restrict__ scale_out, int* __restrict__ n_out,
                                                                           float* __restrict__ scaled_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_k, s_n, s_nan, s_cle, s_mingt;
    const int f = blockIdx.x, t = threadIdx.x;
    const float* a = inv_src + (size_t)f * HW;
    const float* c = inv_cmp + (size_t)f * HW;
    if (t == 0) { s_n = 0u; s_nan = 0u; s_prefix = 0u; s_cle = 0u; s_mingt = 0xffffffffu; }
    __syncthreads();
    {   // how many ratios, and is one of them NaN
        unsigned n = 0u, nan = 0u;
        for (int p = t; p < HW; p += kScaleThreads) {
            const float cv = c[p];
            if (isfinite(cv)) {
                ++n;
                const float r = __fdiv_rn(a[p], cv);
                nan += r != r ? 1u : 0u;
            }
        }
        atomicAdd(&s_n, n);
        if (nan) atomicAdd(&s_nan, nan);
    }
    __syncthreads();
    const unsigned n = s_n;
    if (t == 0) { n_out[f] = (int)n; s_k = n ? (n - 1u) / 2u : 0u; }
    float scale = __uint_as_float(0x7fc00000u);          // NaN: no valid pixel (np.median of an empty array), or a NaN ratio
    if (n > 0u && s_nan == 0u) {          // (workgroup-uniform)
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = t; i < 256; i += kScaleThreads) hist[i] = 0u;
            __syncthreads();
            const unsigned prefix = s_prefix;
            for (int p = t; p < HW; p += kScaleThreads) {
                const float cv = c[p];
                if (isfinite(cv)) {
                    const unsigned key = order_key(__fdiv_rn(a[p], cv));
                    if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            if (t == 0) {
                unsigned k = s_k, cum = 0u;
                int bin = 0;
                for (; bin < 255; ++bin) {
                    if (cum + hist[bin] > k) break;
                    cum += hist[bin];
                }
                s_k = k - cum;
                s_prefix = (prefix << 8) | (unsigned)bin;
            }
            __syncthreads();
        }
        const unsigned key_lo = s_prefix;            // the ((n - 1) / 2)-th smallest ratio (0-based)
        unsigned key_hi = key_lo;
        if ((n & 1u) == 0u) {                        // even count: the (n / 2)-th as well
            unsigned cle = 0u, mingt = 0xffffffffu;
            for (int p = t; p < HW; p += kScaleThreads) {
                const float cv = c[p];
                if (isfinite(cv)) {
                    const unsigned key = order_key(__fdiv_rn(a[p], cv));
                    if (key <= key_lo) ++cle;
                    else mingt = key < mingt ? key : mingt;
                }
            }
            atomicAdd(&s_cle, cle);
            atomicMin(&s_mingt, mingt);
            __syncthreads();
            key_hi = s_cle >= n / 2u + 1u ? key_lo : s_mingt;
        }
        const float lo = key_value(key_lo), hi = key_value(key_hi);
        scale = (n & 1u) ? lo : __fmul_rn(__fadd_rn(lo, hi), 0.5f);     // np.mean of two float32 values: float32 sum, halved
    }
    if (t == 0) scale_out[f] = scale;
    if (scaled_out != nullptr) {
        float* o = scaled_out + (size_t)f * HW;
        for (int p = t; p < HW; p += kScaleThreads) o[p] = __fdiv_rn(a[p], scale);
    }
}

}  // namespace cd

extern "C" int cd_frame_median_scales(const float* inv_src, const float* inv_cmp, int N, int H, int W, float* scales_out, int* n_valid_out,
                                      float* scaled_out, void* stream) {
    if (!inv_src || !inv_cmp || !scales_out || !n_valid_out || N <= 0 || H <= 0 || W <= 0 || (long long)H * W > (1ll << 30)) return CD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cd::frame_median_scale_kernel, dim3(N), dim3(cd::kScaleThreads), 0, (hipStream_t)stream, inv_src, inv_cmp, H * W, scales_out,
                       n_valid_out, scaled_out);
    CD_CHECK_LAUNCH();
    return CD_OK;
}
