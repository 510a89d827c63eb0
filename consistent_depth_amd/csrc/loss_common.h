// Shared by the consistency-loss kernels (loss_fused.hip: v1 scatter-by-atomics, now the
// forward-only path and the device-side fallback; loss_owner.hip: v2 owner-computes; loss_api.hip:
// C-ABI entry points and launch sequence).
#pragma once
#include "cd_common.h"

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
    float pad[6];
};
static_assert(sizeof(PairCam) == 128, "PairCam must be 128 bytes");

// depth parametrisation fused into the loss (cd_depth_mode)
template <int MODE> __device__ __forceinline__ float to_depth(float v) {
    // __expf = v_exp_f32(x*log2e): <= 2e-7 relative for |x| <= 3, far inside the 1e-5 loss tolerance
    if (MODE == CD_DEPTH_EXP) return __expf(v);
    if (MODE == CD_DEPTH_RECIPROCAL) return __builtin_amdgcn_rcpf(v);
    return v;
}
// d depth / d v expressed through the depth value
template <int MODE> __device__ __forceinline__ float depth_jac(float d) {
    if (MODE == CD_DEPTH_EXP) return d;
    if (MODE == CD_DEPTH_RECIPROCAL) return -d * d;
    return 1.f;
}

struct Taps {
    int xa, ya, xb, yb;
    float w00, w01, w10, w11;
};

// geometry.py:205-208 + grid_sample(border, align_corners=False): ix = clamp(u*W/(W-1) - 0.5, 0, W-1).
// ONE definition with explicit fma/add/sub intrinsics: v2's "will the owner see me" test and the owner's
// own scan must agree bit for bit, whatever the compiler contracts elsewhere.
__device__ __forceinline__ Taps tap_coords(float xf, float yf, float fx, float fy, float sx, float sy, int W, int H) {
    const float mx = __fadd_rn(xf, fx), my = __fadd_rn(yf, fy);
    const float ix = fminf(fmaxf(__fmaf_rn(mx, sx, -0.5f), 0.f), (float)(W - 1));
    const float iy = fminf(fmaxf(__fmaf_rn(my, sy, -0.5f), 0.f), (float)(H - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float tx = __fsub_rn(ix, fx0), ty = __fsub_rn(iy, fy0);
    Taps t;
    t.xa = (int)fx0; t.ya = (int)fy0;
    t.xb = min(t.xa + 1, W - 1); t.yb = min(t.ya + 1, H - 1);  // the clipped tap carries weight 0
    t.w00 = (1.f - tx) * (1.f - ty); t.w01 = tx * (1.f - ty);
    t.w10 = (1.f - tx) * ty;         t.w11 = tx * ty;
    return t;
}

// ---- v1 (loss_fused.hip)
int v1_blocks_per_plane(int HW, int vec);
// run_flag: nullptr = always run; else the kernel body runs only if *run_flag != 0 (device-side fallback).
int launch_v1(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, const void* cams,
              int mode, bool reproj, bool disp, bool vec4, int B, int H, int W, float* partial, float* grad,
              const int* run_flag, hipStream_t s);
int launch_zero_guarded(float* buf, size_t n, const int* run_flag, hipStream_t s);

// ---- v2 (loss_owner.hip)
int owner_tiles_x(int W);
int owner_ntiles(int H, int W);
size_t owner_windows_bytes(int B, int H, int W);
int launch_tile_windows(const float* ff, const float* fb, const float* mf, const float* mb, int B, int H, int W,
                        void* wins, hipStream_t s);
// Enqueues: overflow header reset, [before_main] owner kernel [after_main], overflow apply.
// ovf_mem = 256-byte header + idx[cap] + val[cap].
int launch_owner(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                 const void* cams, const void* wins, int mode, bool reproj, int B, int H, int W, float* partial,
                 float* grad, void* ovf_mem, int ovf_cap, hipStream_t s, void (*before_main)(hipStream_t),
                 void (*after_main)(hipStream_t));
const int* owner_fallback_flag(void* ovf_mem);
int launch_overflow_apply(void* ovf_mem, int ovf_cap, float* grad, hipStream_t s);

// ---- v3 (loss_slab.hip): source pass + gather pass; slabs = slab_floats(B,H,W) floats of scratch
size_t slab_floats(int B, int H, int W);
int slab_chunk_pairs(int H, int W);     // pairs per source+gather launch pair (slabs sized to stay cache resident)
void set_slab_chunk_pairs(int pairs);   // test/measurement hook; 0 restores the default
int launch_slab(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                const void* cams, const void* wins, int mode, bool reproj, int B, int H, int W, float* partial,
                float* grad, float* slabs, void* ovf_mem, int ovf_cap, hipStream_t s, void (*before_main)(hipStream_t),
                void (*after_main)(hipStream_t));

}  // namespace cd
