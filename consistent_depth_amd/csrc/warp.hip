// Bilinear backward-warp `sample` for gfx950.
// Replaces /root/reference/utils/geometry.py:201-208 (grid_sample bilinear, border padding,
// align_corners=False applied to an align_corners=True style normalisation:
//   ix = clamp(u*W/(W-1) - 0.5, 0, W-1)).  Used outside the fused loss by the callers the
// reference has for `sample` (flow.py:30-32 masks, scale_calibration.py:105 warp_image).
#include "cd_common.h"

namespace cd {

__global__ __launch_bounds__(kBlock) void sample_kernel(const float* __restrict__ data, const float* __restrict__ uv,
                                                        int C, int H, int W, float* __restrict__ out) {
    const int HW = H * W;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float u = uv[(size_t)b * 2 * HW + p], v = uv[(size_t)b * 2 * HW + HW + p];
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
    const float ix = fminf(fmaxf(u * sx - 0.5f, 0.f), (float)(W - 1));
    const float iy = fminf(fmaxf(v * sy - 0.5f, 0.f), (float)(H - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float tx = ix - fx0, ty = iy - fy0;
    const int xa = (int)fx0, ya = (int)fy0, xb = min(xa + 1, W - 1), yb = min(ya + 1, H - 1);
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    for (int c = 0; c < C; ++c) {
        const float* src = data + ((size_t)b * C + c) * HW;
        out[((size_t)b * C + c) * HW + p] =
            src[ya * W + xa] * w00 + src[ya * W + xb] * w01 + src[yb * W + xa] * w10 + src[yb * W + xb] * w11;
    }
}

}  // namespace cd

extern "C" int cd_sample_bilinear_border(const float* data, const float* uv, int B, int C, int H, int W, float* out,
                                         void* stream) {
    if (!data || !uv || !out || B <= 0 || C <= 0 || H < 2 || W < 2 || B > 65535) return CD_ERR_INVALID_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(cd::sample_kernel, dim3((HW + cd::kBlock - 1) / cd::kBlock, B), dim3(cd::kBlock), 0,
                       (hipStream_t)stream, data, uv, C, H, W, out);
    CD_CHECK_LAUNCH();
    return CD_OK;
}
