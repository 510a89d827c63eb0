// Bilinear backward-warp `sample` for gfx950.
// Replaces /root/reference/utils/geometry.py:201-208 (grid_sample bilinear, border padding,
// align_corners=False applied to an align_corners=True style normalisation:
//   ix = clamp(u*W/(W-1) - 0.5, 0, W-1)).  Used outside the fused loss by the callers the
// reference has for `sample` (flow.py:30-32 masks, scale_calibration.py:105 warp_image).
#include "cd_common.h"

namespace cd {

// two x-adjacent taps = ONE 8-byte load at a 4-byte-aligned address (see the mask kernel below)
struct __attribute__((packed, aligned(4))) FloatPair { float a, b; };
// the four taps of the geometry.sample mapping (xb = min(xa + 1, W - 1), yb = min(ya + 1, H - 1)) of one channel, weighted: the same
// four products in the same order as four single loads -- the pair starts at min(xa, W - 2); at xa = W - 1 both taps of a row are its
// second float (xb = xa there)
__device__ __forceinline__ float sample4(const float* __restrict__ src, int ya, int yb, int xa, int W, float w00, float w01, float w10, float w11) {
    const int xp = xa < W - 2 ? xa : W - 2;
    const bool hi = xa > W - 2;
    const FloatPair n = *reinterpret_cast<const FloatPair*>(src + ya * W + xp), s = *reinterpret_cast<const FloatPair*>(src + yb * W + xp);
    return (hi ? n.b : n.a) * w00 + n.b * w01 + (hi ? s.b : s.a) * w10 + s.b * w11;
}

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
    const int xa = (int)fx0, ya = (int)fy0, yb = min(ya + 1, H - 1);
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    for (int c = 0; c < C; ++c) {
        const float* src = data + ((size_t)b * C + c) * HW;
        out[((size_t)b * C + c) * HW + p] = sample4(src, ya, yb, xa, W, w00, w01, w10, w11);
    }
}

// ---------------------------------------------------------------- flow-consistency masks
// Replaces /root/reference/utils/consistency.py:32-67 (called from flow.py:199-228): for direction k of a pair
//   inside_k = 0 <= x+u <= W-1 and 0 <= y+v <= H-1
//   mask_k   = inside_k  and  |flow_k - (-flow_{1-k} warped by flow_k)|^2 < flow_thresh^2
//                        and  sum_c (color_k - color_{1-k} warped by flow_k)^2 < C * color_thresh^2
// The warp is the reference's OTHER sampler (consistency.py:8-23): grid = 2*uv/(W,H) - 1 evaluated in fp64 and cast to
// fp32, then grid_sample(border, align_corners=False): ix = ((g+1)*W - 1)/2 clipped to [0, W-1] -- i.e. u - 0.5, not
// geometry.sample's u*W/(W-1) - 0.5.  Every rounding step of the reference is reproduced (explicit _rn intrinsics, no
// contraction), so the masks are bit-identical to the oracle.  One thread = one pixel of one direction; the four taps
// are shared by the flow and the colour test.
struct TapsB { int x0, y0, x1, y1; float wnw, wne, wsw, wse; bool in_x1, in_y1; };

__device__ __forceinline__ TapsB taps_border_grid(double idx_x, double idx_y, int W, int H) {
    const float gx = (float)__dsub_rn(__ddiv_rn(__dmul_rn(2.0, idx_x), (double)W), 1.0);
    const float gy = (float)__dsub_rn(__ddiv_rn(__dmul_rn(2.0, idx_y), (double)H), 1.0);
    float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 0.5f);
    float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 0.5f);
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy), x1f = __fadd_rn(x0f, 1.f), y1f = __fadd_rn(y0f, 1.f);
    TapsB t;
    t.x0 = (int)x0f; t.y0 = (int)y0f; t.x1 = t.x0 + 1; t.y1 = t.y0 + 1;
    t.in_x1 = t.x1 <= W - 1; t.in_y1 = t.y1 <= H - 1;
    t.wnw = __fmul_rn(__fsub_rn(x1f, ix), __fsub_rn(y1f, iy));
    t.wne = __fmul_rn(__fsub_rn(ix, x0f), __fsub_rn(y1f, iy));
    t.wsw = __fmul_rn(__fsub_rn(x1f, ix), __fsub_rn(iy, y0f));
    t.wse = __fmul_rn(__fsub_rn(ix, x0f), __fsub_rn(iy, y0f));
    return t;
}

// The four tap offsets of a pixel, shared by every channel sampled at that position.  Taps outside the image are CLAMPED to a
// valid address (and their product replaced by an exact 0 afterwards): every gather of a pixel is an unconditional load, so the
// 10 pair loads (2 flow + 3 colour channels x 2 rows: round 5, below) are all in flight before the first wait.  Round 1's version loaded the optional
// taps under their conditions -- a load under a divergent branch is followed by s_waitcnt vmcnt(0): twenty serialised round trips
// per pixel, 8 % of the HBM rate.
// Round 5: the two taps of a row are adjacent floats -- ONE 8-byte load at a 4-byte-aligned address (gfx950 global loads take it:
// hipcc emits global_load_dwordx2 for an align-4 pair) instead of two gathers: 10 gather instructions per pixel instead of 20 on a
// kernel bound by the texture-address path.  The pair starts at min(x0, W - 2): for x0 = W - 1 (only the clamped right border) the
// west tap is the pair's SECOND float and the east tap carries an exact 0.
struct TapIdx { int n, s; bool hi; };       // offsets of the two pairs; hi: x0 = W - 1
__device__ __forceinline__ TapIdx tap_offsets(const TapsB& t, int W) {
    const int xp = t.x0 < W - 2 ? t.x0 : W - 2, ys = t.in_y1 ? t.y1 : t.y0;
    return TapIdx{t.y0 * W + xp, ys * W + xp, t.x0 > W - 2};
}
struct TapVals { float nw, ne, sw, se; };
__device__ __forceinline__ TapVals tap_load(const float* __restrict__ src, const TapIdx& i) {
    const FloatPair n = *reinterpret_cast<const FloatPair*>(src + i.n), s = *reinterpret_cast<const FloatPair*>(src + i.s);
    return TapVals{i.hi ? n.b : n.a, n.b, i.hi ? s.b : s.a, s.b};
}
__device__ __forceinline__ float tap_sum(const TapVals& v, const TapsB& t) {
    // ((nw + ne) + sw) + se, each term rounded, taps outside the image contribute nothing
    float o = __fmul_rn(v.nw, t.wnw);
    o = __fadd_rn(o, t.in_x1 ? __fmul_rn(v.ne, t.wne) : 0.f);
    o = __fadd_rn(o, t.in_y1 ? __fmul_rn(v.sw, t.wsw) : 0.f);
    o = __fadd_rn(o, (t.in_x1 && t.in_y1) ? __fmul_rn(v.se, t.wse) : 0.f);
    return o;
}

constexpr int kMaskMaxC = 3;   // colour channels held in registers at once (RGB: one batch)

__global__ __launch_bounds__(kBlock) void flow_consistency_mask_kernel(
    const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd, const float* __restrict__ color0,
    const float* __restrict__ color1, int C, float thr_flow, float thr_color, int H, int W, float* __restrict__ mask_fwd,
    float* __restrict__ mask_bwd) {
    const int HW = H * W, b = blockIdx.z, k = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* fl = (k == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
    const float* fo = (k == 0 ? flow_bwd : flow_fwd) + (size_t)b * 2 * HW;
    const float* cr = (k == 0 ? color0 : color1) + (size_t)b * C * HW;
    const float* ct = (k == 0 ? color1 : color0) + (size_t)b * C * HW;
    const float u = fl[p], v = fl[HW + p];
    // the reference pixel's own colours do not depend on the flow: requested before the tap arithmetic (first kMaskMaxC channels)
    float own[kMaskMaxC];
#pragma unroll
    for (int c = 0; c < kMaskMaxC; ++c) own[c] = cr[(size_t)(c < C ? c : C - 1) * HW + p];
    const double idx_x = (double)u + (double)x, idx_y = (double)v + (double)y;
    const bool inside = idx_x >= 0.0 && idx_x <= (double)(W - 1) && idx_y >= 0.0 && idx_y <= (double)(H - 1);
    const TapsB t = taps_border_grid(idx_x, idx_y, W, H);
    const TapIdx ti = tap_offsets(t, W);
    const TapVals fu = tap_load(fo, ti), fv = tap_load(fo + HW, ti);
    TapVals cv[kMaskMaxC];
#pragma unroll
    for (int c = 0; c < kMaskMaxC; ++c) cv[c] = tap_load(ct + (size_t)(c < C ? c : C - 1) * HW, ti);
    // flow test: flow_k - (-(flow_{1-k} warped)) = flow_k + warped   (negation commutes exactly with the weighted sum)
    const float du = __fadd_rn(u, tap_sum(fu, t)), dv = __fadd_rn(v, tap_sum(fv, t));
    const float sse_f = __fadd_rn(__fmul_rn(du, du), __fmul_rn(dv, dv));
    float sse_c = 0.f;
#pragma unroll
    for (int c = 0; c < kMaskMaxC; ++c)
        if (c < C) {
            const float d = __fsub_rn(own[c], tap_sum(cv[c], t));
            sse_c = c == 0 ? __fmul_rn(d, d) : __fadd_rn(sse_c, __fmul_rn(d, d));
        }
    for (int c = kMaskMaxC; c < C; ++c) {   // (more than three colour channels: one at a time, same order of the sum)
        const float d = __fsub_rn(cr[(size_t)c * HW + p], tap_sum(tap_load(ct + (size_t)c * HW, ti), t));
        sse_c = __fadd_rn(sse_c, __fmul_rn(d, d));
    }
    const bool m = inside && sse_f < thr_flow && sse_c < thr_color;
    (k == 0 ? mask_fwd : mask_bwd)[(size_t)b * HW + p] = m ? 1.f : 0.f;
}

// ---------------------------------------------------------------- depth-based warp (offline stages around the hot path)
// Replaces /root/reference/utils/geometry.py:130-139 depth_to_points, :179-200 warping_field, :213-227 warp_image
// (used by scale_calibration.py:84-120) and the point-cloud means of :142-176 calibrate_scale.  Conventions as in the
// fused loss: ray = ((x-cx)/fx, -(y-cy)/fy, -1), x_world = R p + t, camera looks along -z.
//   uv[i]     = project_t( R_t^T (R_i (d * ray) + t_i - t_t) ),  t = tgt_ids[i]
//   warped[i] = sample(images[t], uv[i])          (the geometry.sample mapping of sample_kernel above)
__global__ __launch_bounds__(kBlock) void warp_image_kernel(const float* __restrict__ images, const float* __restrict__ depths,
                                                            const float* __restrict__ intr, const float* __restrict__ extr,
                                                            const int* __restrict__ tgt_ids, int C, int H, int W,
                                                            float* __restrict__ uv_out, float* __restrict__ warped) {
    const int HW = H * W, i = blockIdx.y, t = tgt_ids[i];
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float* Ei = extr + i * 12;
    const float* Et = extr + t * 12;
    // M = R_t^T R_i, c = R_t^T (t_i - t_t)        (rows of E: [R | t], R[r][c] = E[r*4 + c])
    float M[9], c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int q = 0; q < 3; ++q) M[r * 3 + q] = Et[0 + r] * Ei[0 + q] + Et[4 + r] * Ei[4 + q] + Et[8 + r] * Ei[8 + q];
        c[r] = Et[0 + r] * (Ei[3] - Et[3]) + Et[4 + r] * (Ei[7] - Et[7]) + Et[8 + r] * (Ei[11] - Et[11]);
    }
    const float fx = intr[i * 4], fy = intr[i * 4 + 1], cx = intr[i * 4 + 2], cy = intr[i * 4 + 3];
    const float fxt = intr[t * 4], fyt = intr[t * 4 + 1], cxt = intr[t * 4 + 2], cyt = intr[t * 4 + 3];
    const int y = p / W, x = p - y * W;
    const float d = depths[(size_t)i * HW + p];
    const float r0 = ((float)x - cx) / fx, r1 = -((float)y - cy) / fy;
    const float px = d * r0, py = d * r1, pz = -d;
    const float X = M[0] * px + M[1] * py + M[2] * pz + c[0];
    const float Y = M[3] * px + M[4] * py + M[5] * pz + c[1];
    const float Z = M[6] * px + M[7] * py + M[8] * pz + c[2];
    const float u = (X / -Z) * fxt + cxt, v = -((Y / -Z) * fyt) + cyt;
    if (uv_out) {
        uv_out[(size_t)i * 2 * HW + p] = u;
        uv_out[(size_t)i * 2 * HW + HW + p] = v;
    }
    if (warped) {
        const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
        const float ix = fminf(fmaxf(u * sx - 0.5f, 0.f), (float)(W - 1));
        const float iy = fminf(fmaxf(v * sy - 0.5f, 0.f), (float)(H - 1));
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float tx = ix - fx0, ty = iy - fy0;
        const int xa = (int)fx0, ya = (int)fy0, yb = min(ya + 1, H - 1);
        const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
        for (int ch = 0; ch < C; ++ch) {
            const float* src = images + ((size_t)t * C + ch) * HW;
            warped[((size_t)i * C + ch) * HW + p] = sample4(src, ya, yb, xa, W, w00, w01, w10, w11);
        }
    }
}

// points[n] = depth * ray (3, H, W); optionally only their per-frame sums (fp64 partials) for calibrate_scale
__global__ __launch_bounds__(kBlock) void depth_points_kernel(const float* __restrict__ depths, const float* __restrict__ intr,
                                                              int H, int W, float* __restrict__ points,
                                                              double* __restrict__ sums) {
    __shared__ double red[3][kBlock / kWave];
    const int HW = H * W, n = blockIdx.y;
    const float fx = intr[n * 4], fy = intr[n * 4 + 1], cx = intr[n * 4 + 2], cy = intr[n * 4 + 3];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < HW; p += gridDim.x * kBlock) {
        const int y = p / W, x = p - y * W;
        const float d = depths[(size_t)n * HW + p];
        const float px = d * (((float)x - cx) / fx), py = d * (-((float)y - cy) / fy), pz = -d;
        if (points) {
            points[((size_t)n * 3 + 0) * HW + p] = px;
            points[((size_t)n * 3 + 1) * HW + p] = py;
            points[((size_t)n * 3 + 2) * HW + p] = pz;
        }
        a0 += (double)px; a1 += (double)py; a2 += (double)pz;
    }
    if (!sums) return;   // block-uniform
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off, kWave); a1 += __shfl_down(a1, off, kWave); a2 += __shfl_down(a2, off, kWave);
    }
    if (lane == 0) { red[0][wid] = a0; red[1][wid] = a1; red[2][wid] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int w = 0; w < kBlock / kWave; ++w) s += red[threadIdx.x][w];
        atomicAdd(&sums[n * 3 + threadIdx.x], s);
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

extern "C" int cd_flow_consistency_masks(const float* flow_fwd, const float* flow_bwd, const float* color0, const float* color1,
                                         int C, double flow_thresh, double color_thresh, int B, int H, int W, float* mask_fwd,
                                         float* mask_bwd, void* stream) {
    if (!flow_fwd || !flow_bwd || !color0 || !color1 || !mask_fwd || !mask_bwd) return CD_ERR_INVALID_ARG;
    if (B <= 0 || B > 65535 || C <= 0 || H < 2 || W < 2 || !(flow_thresh > 0.0) || !(color_thresh > 0.0)) return CD_ERR_INVALID_ARG;
    // the reference compares fp32 sums with python floats under NumPy's weak-scalar rule: the thresholds are rounded to fp32
    const float thr_f = (float)(flow_thresh * flow_thresh), thr_c = (float)((double)C * (color_thresh * color_thresh));
    const int HW = H * W;
    hipLaunchKernelGGL(cd::flow_consistency_mask_kernel, dim3((HW + cd::kBlock - 1) / cd::kBlock, 2, B), dim3(cd::kBlock), 0,
                       (hipStream_t)stream, flow_fwd, flow_bwd, color0, color1, C, thr_f, thr_c, H, W, mask_fwd, mask_bwd);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

extern "C" int cd_warp_image(const float* images, const float* depths, const float* intrinsics, const float* extrinsics,
                             const int* tgt_ids, int N, int C, int H, int W, float* uv_out, float* warped_out, void* stream) {
    if (!depths || !intrinsics || !extrinsics || !tgt_ids || (!uv_out && !warped_out) || (warped_out && (!images || C <= 0))) return CD_ERR_INVALID_ARG;
    if (N <= 0 || N > 65535 || H < 2 || W < 2) return CD_ERR_INVALID_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(cd::warp_image_kernel, dim3((HW + cd::kBlock - 1) / cd::kBlock, N), dim3(cd::kBlock), 0, (hipStream_t)stream,
                       images, depths, intrinsics, extrinsics, tgt_ids, C, H, W, uv_out, warped_out);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

extern "C" int cd_depth_to_points(const float* depths, const float* intrinsics, int N, int H, int W, float* points_out,
                                  double* sums_out, void* stream) {
    if (!depths || !intrinsics || (!points_out && !sums_out) || N <= 0 || N > 65535 || H <= 0 || W <= 0) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (sums_out && hipMemsetAsync(sums_out, 0, sizeof(double) * 3 * N, s) != hipSuccess) return CD_ERR_LAUNCH;
    const int HW = H * W;
    int bx = (HW + cd::kBlock * 8 - 1) / (cd::kBlock * 8);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(cd::depth_points_kernel, dim3(bx, N), dim3(cd::kBlock), 0, s, depths, intrinsics, H, W, points_out, sums_out);
    CD_CHECK_LAUNCH();
    return CD_OK;
}
