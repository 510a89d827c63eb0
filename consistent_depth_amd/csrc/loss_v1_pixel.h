// One source pixel of the v1 formulation (scatter through global fp32 atomics): the body shared by loss_fused.hip's kernel (forward-only
// path, guarded fallback of the tile kernels) and the row sweep's per-pair exact mode (loss_sweep.hip).  Closed form: SURVEY.md appendix
// A.1 (= oracle/cd_oracle_body.inc).  Reference: loss/consistency_loss.py:98-253 over utils/geometry.py:38-128,201-208.
#pragma once
#include "loss_common.h"

namespace cd {

// pixel p = y * W + x (xf, yf its coordinates as floats) of direction k of one pair: cam = that direction's constants, v_ref / v_tgt the
// raw depth planes of the direction's own frame and of the other one, fx / fy / m its flow and mask; the loss partial sums go to
// acc_r / acc_d, with GRAD the gradient to g_ref[p] and the four taps of g_tgt (atomics: the planes must have been zeroed).
template <bool GRAD, int MODE, bool REPROJ, bool DISP>
__device__ __forceinline__ void v1_pixel(const PairCam& cam, const float* __restrict__ v_tgt, float vin, float fx, float fy, float m,
                                         float xf, float yf, int p, int H, int W, float* g_ref, float* g_tgt, float& acc_r, float& acc_d) {
    const float r1 = -(yf - cam.cy_r) * cam.ify_r;
    const float d = to_depth<MODE>(vin);
    const float r0 = (xf - cam.cx_r) * cam.ifx_r;
    // a = M (r0, r1, -1);  P = d a + c
    const float a0 = cam.M[0] * r0 + cam.M[1] * r1 - cam.M[2];
    const float a1 = cam.M[3] * r0 + cam.M[4] * r1 - cam.M[5];
    const float a2 = cam.M[6] * r0 + cam.M[7] * r1 - cam.M[8];
    const float X = d * a0 + cam.c[0], Y = d * a1 + cam.c[1], Z = d * a2 + cam.c[2];
    const float iZ = __builtin_amdgcn_rcpf(Z);
    float g = 0.f;  // d total / d depth_ref at this pixel
    if (REPROJ) {
        // project (geometry.py:73-83): px = fx X/(-Z) + cx ; py = -(fy Y/(-Z)) + cy
        const float mx = xf + fx, my = yf + fy;
        const float ex = (cam.cx_t - cam.fx_t * X * iZ) - mx, ey = (cam.cy_t + cam.fy_t * Y * iZ) - my;
        const float e = __builtin_amdgcn_sqrtf(ex * ex + ey * ey);
        acc_r += m * e;  // multiply, not select: 0*inf = NaN exactly like the reference
        if (GRAD) {
            const float dpx = cam.fx_t * iZ * (X * a2 * iZ - a0);
            const float dpy = cam.fy_t * iZ * (a1 - Y * a2 * iZ);
            const float ie = e > 0.f ? __builtin_amdgcn_rcpf(e) : 0.f;  // subgradient 0 at e = 0
            g += cam.gr * m * (ex * dpx + ey * dpy) * ie;
        }
    }
    if (DISP) {
        const Taps t = tap_coords(xf, yf, fx, fy, cam.sx, cam.sy, W, H);
        const int i00 = t.ya * W + t.xa, i01 = t.ya * W + t.xb, i10 = t.yb * W + t.xa, i11 = t.yb * W + t.xb;
        const float d00 = to_depth<MODE>(v_tgt[i00]), d01 = to_depth<MODE>(v_tgt[i01]);
        const float d10 = to_depth<MODE>(v_tgt[i10]), d11 = to_depth<MODE>(v_tgt[i11]);
        const float zs = -(d00 * t.w00 + d01 * t.w01 + d10 * t.w10 + d11 * t.w11);  // z = -depth
        const float izs = __builtin_amdgcn_rcpf(zs);
        const float dd = iZ - izs;
        acc_d += m * fabsf(dd);
        if (GRAD) {
            const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
            const float gm = cam.gb * m * sg;
            g -= gm * a2 * iZ * iZ;           // d(1/Z)/dd = -a2/Z^2
            const float gz = gm * izs * izs;  // d(-1/zs)/dzs, zs = -sum w d
            if (gz != 0.f) {
                atomic_add_f32(g_tgt + i00, -gz * t.w00 * depth_jac<MODE>(d00));
                atomic_add_f32(g_tgt + i01, -gz * t.w01 * depth_jac<MODE>(d01));
                atomic_add_f32(g_tgt + i10, -gz * t.w10 * depth_jac<MODE>(d10));
                atomic_add_f32(g_tgt + i11, -gz * t.w11 * depth_jac<MODE>(d11));
            }
        }
    }
    if (GRAD) {
        const float gv = g * depth_jac<MODE>(d);
        if (gv != 0.f) atomic_add_f32(g_ref + p, gv);
    }
}

}  // namespace cd
