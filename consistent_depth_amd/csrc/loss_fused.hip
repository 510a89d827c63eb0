// Fused geometric-consistency loss for gfx950, v1: one pass per (pair, direction, pixel); the
// scatter part of the depth gradient goes through global fp32 atomics.
//
// Replaces (reference, /root/reference): loss/consistency_loss.py:98-253 and the
// utils/geometry.py chain pixel_grid :9-19 -> pixels_to_rays :38-61 -> pixels_to_points
// :86-100 -> reproject_points :103-128 -> project :64-83 -> sample :201-208 ->
// weighted_mean_loss consistency_loss.py:73-89, plus the autograd backward of all of it.
// Per-pixel closed form: SURVEY.md appendix A.1 / DESIGN.md section 3.
//
// Role since the tiled gradient kernels exist (loss_slab.hip, loss_sweep.hip):
//   (a) the FORWARD-ONLY path (validation sweep, GRAD = false): no gradient, hence no atomics --
//       a pure coalesced streaming kernel (16 B per lane when W % 4 == 0);
//   (b) the device-side FALLBACK of v2 (GRAD = true): needs a zeroed gradient and issues up to 5
//       global atomics per valid pixel, which run memory-side on MI355X at <= 267 G atomics/s
//       (profiles/atomics_exp_r01.txt) -- ~4 % of the HBM roofline.  It is always enqueued after v2
//       but its body only runs when v2's overflow list overflowed (*run_flag != 0).
#include "loss_common.h"
#include "loss_v1_pixel.h"

namespace cd {

int v1_blocks_per_plane(int HW, int vec) { return (HW + kBlock * vec - 1) / (kBlock * vec); }

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

// 1-D grid, grid-stride over the work items (blk, k, b) so the guarded fallback can be a small
// launch that exits at once; the unguarded launch uses one block per item.
template <int VEC, bool GRAD, int MODE, bool REPROJ, bool DISP>
__global__ __launch_bounds__(kBlock) void loss_main_kernel(
    const float* __restrict__ depth, const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd,
    const float* __restrict__ mask_fwd, const float* __restrict__ mask_bwd,
    const PairCam* __restrict__ cams, int H, int W, int nblk, int B,
    float* __restrict__ partial, float* __restrict__ grad, const int* __restrict__ run_flag) {
    __shared__ float lds[kBlock / kWave];
    if (run_flag != nullptr && *run_flag == 0) return;
    const int HW = H * W;
    const int items = nblk * 2 * B;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int blk = item % nblk, pk = item / nblk;  // pk = b*2 + k
        const int b = pk >> 1, k = pk & 1;
        const PairCam& cam = cams[pk];  // block-uniform -> scalar loads
        const float* __restrict__ v_ref = depth + (size_t)pk * HW;
        const float* __restrict__ v_tgt = depth + (size_t)(b * 2 + (1 - k)) * HW;
        const float* __restrict__ fl = (k == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
        const float* __restrict__ mk = (k == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
        float* g_ref = GRAD ? grad + (size_t)pk * HW : nullptr;
        float* g_tgt = GRAD ? grad + (size_t)(b * 2 + (1 - k)) * HW : nullptr;

        float acc_r = 0.f, acc_d = 0.f;
        const int p0 = (blk * kBlock + threadIdx.x) * VEC;
        if (p0 < HW) {
            float vin[VEC], fx[VEC], fy[VEC], m[VEC];
            using V = typename VecT<VEC>::type;
            *reinterpret_cast<V*>(vin) = *reinterpret_cast<const V*>(v_ref + p0);
            // flow / mask are read once per call: non-temporal (round 6: as in the row sweep, loss_sweep_core.h); the depth plane is
            // also the other direction's gather target: default policy
            typedef float nt_t __attribute__((ext_vector_type(VEC)));
#ifndef CD_V1_NT
#define CD_V1_NT 1
#endif
            if (CD_V1_NT) {
                const nt_t a = __builtin_nontemporal_load(reinterpret_cast<const nt_t*>(fl + p0));
                const nt_t b2 = __builtin_nontemporal_load(reinterpret_cast<const nt_t*>(fl + HW + p0));
                const nt_t c = __builtin_nontemporal_load(reinterpret_cast<const nt_t*>(mk + p0));
#pragma unroll
                for (int i = 0; i < VEC; ++i) { fx[i] = a[i]; fy[i] = b2[i]; m[i] = c[i]; }
            } else {
                *reinterpret_cast<V*>(fx) = *reinterpret_cast<const V*>(fl + p0);
                *reinterpret_cast<V*>(fy) = *reinterpret_cast<const V*>(fl + HW + p0);
                *reinterpret_cast<V*>(m) = *reinterpret_cast<const V*>(mk + p0);
            }
            const int y = p0 / W;  // the VEC pixels share a row (W % VEC == 0)
            const int x0 = p0 - y * W;
            const float yf = (float)y;
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                v1_pixel<GRAD, MODE, REPROJ, DISP>(cam, v_tgt, vin[i], fx[i], fy[i], m[i], (float)(x0 + i), yf, p0 + i, H, W, g_ref, g_tgt, acc_r, acc_d);
        }
        acc_r = block_sum(acc_r, lds);
        acc_d = block_sum(acc_d, lds);
        if (threadIdx.x == 0) {
            float* o = partial + ((size_t)pk * nblk + blk) * 2;
            o[0] = acc_r;
            o[1] = acc_d;
        }
    }
}

__global__ __launch_bounds__(kBlock) void zero_guarded_kernel(float* __restrict__ buf, size_t n,
                                                              const int* __restrict__ run_flag) {
    if (run_flag != nullptr && *run_flag == 0) return;
    const size_t n4 = n / 4;  // buf comes from the caller's allocator: 16-byte aligned
    float4* b4 = reinterpret_cast<float4*>(buf);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock)
        b4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) buf[n4 * 4 + threadIdx.x] = 0.f;
}

int launch_zero_guarded(float* buf, size_t n, const int* run_flag, hipStream_t s) {
    if (reinterpret_cast<uintptr_t>(buf) % 16 != 0) return CD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(zero_guarded_kernel, dim3(1024), dim3(kBlock), 0, s, buf, n, run_flag);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

template <int VEC, bool GRAD, int MODE>
static void launch_rd(bool r, bool d, dim3 grid, hipStream_t s, const float* depth, const float* ff, const float* fb,
                      const float* mf, const float* mb, const PairCam* cams, int H, int W, int nblk, int B,
                      float* partial, float* grad, const int* flag) {
#define CD_LAUNCH(R, D)                                                                                          \
    hipLaunchKernelGGL((loss_main_kernel<VEC, GRAD, MODE, R, D>), grid, dim3(kBlock), 0, s, depth, ff, fb, mf, mb, \
                       cams, H, W, nblk, B, partial, grad, flag)
    if (r && d) CD_LAUNCH(true, true);
    else if (r) CD_LAUNCH(true, false);
    else if (d) CD_LAUNCH(false, true);
    else CD_LAUNCH(false, false);
#undef CD_LAUNCH
}

template <int VEC, bool GRAD>
static void launch_mode(int mode, bool r, bool d, dim3 grid, hipStream_t s, const float* depth, const float* ff,
                        const float* fb, const float* mf, const float* mb, const PairCam* cams, int H, int W,
                        int nblk, int B, float* partial, float* grad, const int* flag) {
    if (mode == CD_DEPTH_EXP) launch_rd<VEC, GRAD, CD_DEPTH_EXP>(r, d, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, flag);
    else if (mode == CD_DEPTH_RECIPROCAL) launch_rd<VEC, GRAD, CD_DEPTH_RECIPROCAL>(r, d, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, flag);
    else launch_rd<VEC, GRAD, CD_DEPTH_IDENTITY>(r, d, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, flag);
}

int launch_v1(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, const void* cams_,
              int mode, bool reproj, bool disp, bool vec4, int B, int H, int W, float* partial, float* grad,
              const int* run_flag, hipStream_t s) {
    const PairCam* cams = (const PairCam*)cams_;
    const int nblk = v1_blocks_per_plane(H * W, vec4 ? 4 : 1);
    const long long items = (long long)nblk * 2 * B;
    // guarded (fallback) launches stay small; they grid-stride if they ever have to run
    const dim3 grid((unsigned)(run_flag ? (items < 2048 ? items : 2048) : items));
    if (vec4) {
        if (grad) launch_mode<4, true>(mode, reproj, disp, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, run_flag);
        else launch_mode<4, false>(mode, reproj, disp, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, run_flag);
    } else {
        if (grad) launch_mode<1, true>(mode, reproj, disp, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, run_flag);
        else launch_mode<1, false>(mode, reproj, disp, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, B, partial, grad, run_flag);
    }
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // namespace cd
