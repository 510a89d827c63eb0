// Host/device-neutral pieces of the consistency-loss kernels: the per-(pair, direction) constants, the sampling
// coordinate, the depth parametrisations and the fixed-point format of the row-sweep kernel.
// Compiled by hipcc into the kernels and by g++ into tests/emul (a sequential execution of the sweep kernel's phase
// functions on the host, used only by the CPU tests to check plans, ring indexing and flush logic before a GPU run).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CD_HD __host__ __device__ __forceinline__
#else
#define CD_HD inline
#endif

namespace cd {

struct __attribute__((aligned(16))) PairCam {  // 32 floats, one per (pair, direction)
    float M[9];   // R_tgt^T R_ref            (geometry.py:119-127 folded)
    float c[3];   // R_tgt^T (t_ref - t_tgt)
    float ifx_r, ify_r, cx_r, cy_r;  // ref intrinsics (1/fx, 1/fy, cx, cy)
    float fx_t, fy_t, cx_t, cy_t;    // tgt intrinsics
    float gr;     // lambda_r / (2 B S_k)           d total / d (mask-weighted reprojection term)
    float gb;     // lambda_b fbar_k / (2 B S_k)    same for the disparity term
    float invS;   // 1 / max(S_k, 1e-6)             consistency_loss.py:85-87
    float fbar;   // mean over the batch of (fx,fy) of the ref frames   :178
    float sx, sy; // W/(W-1), H/(H-1): geometry.py:205-207 + align_corners=False un-normalise
    // row-sweep kernel (loss_sweep.hip): its gradient accumulator of frame j counts in units of U_j, an O(1) scale
    // (normally the OTHER direction's gb, whose scatter lands there), so the fixed-point range does not depend on
    // batch size, lambda or mask coverage.  For direction j = this entry, k = 1 - j:
    float unit;   // U_j: one accumulator unit of ring j in gradient units (flush multiplies by it)
    float dr;     // gr_j / U_j   direct reprojection term  -> ring j
    float db;     // gb_j / U_j   direct disparity term     -> ring j
    float sc;     // gb_j / U_k   scatter term              -> ring k
    float pad[2];
};
static_assert(sizeof(PairCam) == 128, "PairCam must be 128 bytes");

enum { kDepthIdentity = 0, kDepthExp = 1, kDepthReciprocal = 2 };   // = CD_DEPTH_* of include/consistent_depth_amd.h

#if defined(__HIP_DEVICE_COMPILE__)
CD_HD float cd_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CD_HD float cd_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
CD_HD float cd_exp(float x) { return __expf(x); }
CD_HD float cd_fadd(float a, float b) { return __fadd_rn(a, b); }
CD_HD float cd_fsub(float a, float b) { return __fsub_rn(a, b); }
CD_HD float cd_fma(float a, float b, float c) { return __fmaf_rn(a, b, c); }
CD_HD float cd_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }   // v_med3_f32
CD_HD float cd_fract(float x) { return __builtin_amdgcn_fractf(x); }                               // v_fract_f32
#else
CD_HD float cd_rcp(float x) { return 1.0f / x; }
CD_HD float cd_rsq(float x) { return 1.0f / sqrtf(x); }
CD_HD float cd_exp(float x) { return expf(x); }
CD_HD float cd_fadd(float a, float b) { volatile float r = a + b; return r; }
CD_HD float cd_fsub(float a, float b) { volatile float r = a - b; return r; }
CD_HD float cd_fma(float a, float b, float c) { return fmaf(a, b, c); }
CD_HD float cd_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
CD_HD float cd_fract(float x) { return x - floorf(x); }    // exact, like v_fract_f32, for the coordinates it sees (0 <= x < 32768)
#endif

// depth parametrisation fused into the loss (cd_depth_mode)
template <int MODE> CD_HD float to_depth(float v) {
    // __expf = v_exp_f32(x*log2e): <= 2e-7 relative for |x| <= 3, far inside the loss tolerance
    if (MODE == kDepthExp) return cd_exp(v);
    if (MODE == kDepthReciprocal) return cd_rcp(v);
    return v;
}
// d depth / d v expressed through the depth value
template <int MODE> CD_HD float depth_jac(float d) {
    if (MODE == kDepthExp) return d;
    if (MODE == kDepthReciprocal) return -d * d;
    return 1.f;
}

struct Taps {
    int xa, ya, xb, yb;
    float w00, w01, w10, w11;
};

// geometry.py:205-208 + grid_sample(border, align_corners=False): ix = clamp(u*W/(W-1) - 0.5, 0, W-1).
// ONE definition with explicit fma/add/sub roundings: every kernel that predicts where a source samples (tile windows,
// sweep plans, the owner scan) must agree bit for bit with the kernels that sample, whatever the compiler contracts.
// (One fma instead of the reference's five roundings ((2u/(W-1) - 1 + 1) W - 1)/2: measured and emulated distances to
// the fp64 truth are the same, tools/exp/loss_emul.py / profiles/parity_loss_r02*.txt.)
CD_HD Taps tap_coords(float xf, float yf, float fx, float fy, float sx, float sy, int W, int H) {
    const float mx = cd_fadd(xf, fx), my = cd_fadd(yf, fy);
    const float ix = cd_clamp(cd_fma(mx, sx, -0.5f), 0.f, (float)(W - 1));
    const float iy = cd_clamp(cd_fma(my, sy, -0.5f), 0.f, (float)(H - 1));
    const float tx = cd_fract(ix), ty = cd_fract(iy);          // ix >= 0: fract = ix - floor(ix), exact
    Taps t;
    t.xa = (int)ix; t.ya = (int)iy;                            // truncation = floor for ix >= 0
    t.xb = t.xa + 1 < W - 1 ? t.xa + 1 : W - 1; t.yb = t.ya + 1 < H - 1 ? t.ya + 1 : H - 1;  // the clipped tap carries weight 0
    t.w00 = (1.f - tx) * (1.f - ty); t.w01 = tx * (1.f - ty);
    t.w10 = (1.f - tx) * ty;         t.w11 = tx * ty;
    return t;
}

// Per-pair constants of both directions (the body of prep_kernel, loss_api.hip).  intr_p: [2][4] fx,fy,cx,cy of the pair's
// two frames; extr_p: [2][3][4]; msum_p: [2] mask sums; fbar: [2] batch means of (fx+fy)/2 of the ref frames.
CD_HD void prep_pair(const float* intr_p, const float* extr_p, const float* msum_p, const float* fbar, float lambda_r,
                     float lambda_b, int B, int H, int W, PairCam* out /* [2] */) {
    for (int k = 0; k < 2; ++k) {
        const float* ir = intr_p + k * 4;
        const float* it = intr_p + (1 - k) * 4;
        const float* er = extr_p + k * 12;
        const float* et = extr_p + (1 - k) * 12;
        PairCam c;
        for (int j = 0; j < 3; ++j) {
            for (int l = 0; l < 3; ++l)
                c.M[j * 3 + l] = et[0 * 4 + j] * er[0 * 4 + l] + et[1 * 4 + j] * er[1 * 4 + l] + et[2 * 4 + j] * er[2 * 4 + l];
            c.c[j] = et[0 * 4 + j] * (er[3] - et[3]) + et[1 * 4 + j] * (er[7] - et[7]) + et[2 * 4 + j] * (er[11] - et[11]);
        }
        c.ifx_r = 1.f / ir[0]; c.ify_r = 1.f / ir[1]; c.cx_r = ir[2]; c.cy_r = ir[3];
        c.fx_t = it[0]; c.fy_t = it[1]; c.cx_t = it[2]; c.cy_t = it[3];
        const float S = fmaxf(msum_p[k], 1e-6f);
        c.invS = 1.f / S;
        c.fbar = fbar[k];
        c.gr = lambda_r > 0.f ? lambda_r / (2.f * (float)B * S) : 0.f;
        c.gb = lambda_b > 0.f ? lambda_b * fbar[k] / (2.f * (float)B * S) : 0.f;
        c.sx = (float)W / (float)(W - 1);
        c.sy = (float)H / (float)(H - 1);
        c.unit = c.dr = c.db = c.sc = 0.f;
        c.pad[0] = c.pad[1] = 0.f;
        out[k] = c;
    }
    // accumulator units of the row-sweep kernel: ring j receives the scatter of direction k (scale gb_k) and the direct
    // term of direction j.  U_j = gb_k unless direction k has an empty mask (its gb is then the 1e-6-clamp value and it
    // contributes nothing): then U_j = gb_j.
    for (int j = 0; j < 2; ++j) {
        const int k = 1 - j;
        float U = msum_p[k] >= 0.5f ? out[k].gb : out[j].gb;
        if (!(U > 0.f)) U = 1.f;   // lambda_b <= 0: the sweep kernel is not used
        out[j].unit = U;
    }
    for (int j = 0; j < 2; ++j) {
        const int k = 1 - j;
        out[j].dr = out[j].gr / out[j].unit;
        out[j].db = out[j].gb / out[j].unit;
        out[j].sc = out[j].gb / out[k].unit;
    }
}

// ---------------------------------------------------------------- fixed point of the row-sweep accumulator
// Round 3: a 32-BIT integer per ring element (round 2: 64-bit, 2^34 per unit).  The ring of frame j counts in units U_j with
// SWEEP_FX_BITS fractional bits: a contribution c (gradient units) is added as round(c / U_j * 2^20).  Why 32 bits are enough:
//   * resolution: rounding is +-1/2 * 2^-20 U per contribution, ~5 contributions per element -> a mean absolute error of
//     ~5e-7 U; with U_j <= the mean |gradient| of the plane (sweep_units below: a data-dependent estimate, a power of two)
//     that is <= 1e-6 relative L1, the class of the reference's own fp32 arithmetic (and integer sums are order independent:
//     the gradient stays bit-reproducible);
//   * range: a source is accepted while the |.| sum of its 5 contributions is <= LIMIT; larger ones (and NaN / inf) take the
//     overflow list like before.  An element cannot wrap: the plan counts, from the flows (a dataset constant), how many sources
//     reach each target pixel; with a fan-in of at most F the pair's LIMIT is the largest power of two with (F + 1) LIMIT < 2^31,
//     capped at 2^27 = 128 U (F <= 15: every consistent flow field; zooming 2x gives ~16) and never below 2^25 = 32 U -- pairs
//     with more than 62 sources on one pixel get no plan and take the exact fallback path.
// What it buys (the kernel is VALU- and register-bound): one v_cvt_rpi_i32_f32 + one ds_add_u32 per contribution
// instead of cvt_f64_f32 + add_f64 + sub + ds_add_u64 with a register pair, a 2-instruction flush conversion instead of the
// i64 -> f64 -> f32 chain, and 8 instead of 12 bytes of LDS per ring element (a 45-row ring at W = 224 instead of 30).
constexpr int SWEEP_FX_BITS = 20;
constexpr float SWEEP_FX_ONE_F = 1048576.f;                // 2^20: folded into the wave-uniform scale factors (exact)
constexpr int SWEEP_MAX_FAN_IN = 62;                       // sources per target pixel a plan accepts: (62 + 1 direct) * 2^25 < 2^31
// bound of the |.| sum of one source's pre-scaled contributions for a pair whose target pixels receive at most `fan_in` sources
CD_HD float sweep_limit_scaled(int fan_in) {
    if (fan_in < 16) return 134217728.f;    // 2^27
    if (fan_in < 32) return 67108864.f;     // 2^26
    return 33554432.f;                      // 2^25 (fan_in <= SWEEP_MAX_FAN_IN)
}

// pre-scaled value -> accumulator integer, round to nearest (ties up): ONE instruction
CD_HD int sweep_scaled_to_fixed(float cs) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(cs));   // floor(cs + 0.5) computed exactly
    return r;
#else
    return (int)floor((double)cs + 0.5);
#endif
}

}  // namespace cd
