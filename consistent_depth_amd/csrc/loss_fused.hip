// Fused geometric-consistency loss for gfx950: forward + analytic backward in ONE pass
// over the frame pairs.
//
// Replaces (reference, /root/reference): loss/consistency_loss.py:98-253 and the
// utils/geometry.py chain pixel_grid :9-19 -> pixels_to_rays :38-61 -> pixels_to_points
// :86-100 -> reproject_points :103-128 -> project :64-83 -> sample :201-208 ->
// weighted_mean_loss consistency_loss.py:73-89, plus the autograd backward of all of it.
// Per-pixel closed form: SURVEY.md appendix A.1 / DESIGN.md section 3.
//
// HBM-bound (~110 flop per 20 algorithmic bytes).  Per (pair b, direction k, pixel) the
// kernel reads  ref depth, flow (dx,dy), mask   once, coalesced (16 B/lane when W%4==0),
// gathers the 4 bilinear taps of the tgt depth (neighbouring lanes hit the same lines in
// L1/L2), and emits  d total/d depth  for the ref pixel and the 4 tgt taps.  The sampling
// position and the mask normaliser do not depend on depth,
// which is what makes the single pass possible.
//
// Launch sequence of one call (all on the caller's stream):
//   [mask_sum_kernel]   only when the caller did not pass cached mask sums
//   prep_kernel         per-(b,k) camera constants + gradient scales -> workspace (128 B each)
//   loss_main_kernel    the fused pass; per-block partial sums -> workspace
//   finalize_pairs / finalize_total   fixed-order reduction -> reproj[B], disp[B], total[1]
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

struct Workspace {
    PairCam* cams;      // [B*2]
    float* mask_sum;    // [B*2]  (only used when the caller passes none)
    float* partial;     // [B*2][nblk][2]
    int nblk;
};

static inline int blocks_per_plane(int HW, int vec) { return (HW + kBlock * vec - 1) / (kBlock * vec); }

static inline size_t ws_bytes(int B, int H, int W) {
    const int HW = H * W;
    size_t s = align_up(sizeof(PairCam) * (size_t)B * 2, 256);
    s += align_up(sizeof(float) * (size_t)B * 2, 256);
    s += align_up(sizeof(float) * (size_t)B * 2 * blocks_per_plane(HW, 1) * 2, 256);
    return s;
}

static inline Workspace carve(void* ws, int B, int H, int W, int vec) {
    Workspace w;
    char* p = (char*)ws;
    w.cams = (PairCam*)p;
    p += align_up(sizeof(PairCam) * (size_t)B * 2, 256);
    w.mask_sum = (float*)p;
    p += align_up(sizeof(float) * (size_t)B * 2, 256);
    w.partial = (float*)p;
    w.nblk = blocks_per_plane(H * W, vec);
    return w;
}

// ---------------------------------------------------------------- mask sums
// S[b,k] = sum(mask_k[b]).  Masks are {0,1}, so fp32 atomics are exact and order-free.
__global__ __launch_bounds__(kBlock) void mask_sum_kernel(const float* __restrict__ mask_fwd,
                                                          const float* __restrict__ mask_bwd,
                                                          int HW, float* __restrict__ mask_sum) {
    __shared__ float lds[kBlock / kWave];
    const int k = blockIdx.y, b = blockIdx.z;
    const float* m = (k == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
    float acc = 0.f;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < HW; p += gridDim.x * kBlock) acc += m[p];
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) atomic_add_f32(&mask_sum[b * 2 + k], acc);
}

// ---------------------------------------------------------------- per-(b,k) constants
__global__ __launch_bounds__(kBlock) void prep_kernel(const float* __restrict__ intr,
                                                      const float* __restrict__ extr,
                                                      const float* __restrict__ mask_sum,
                                                      float lambda_r, float lambda_b, int B, int H, int W,
                                                      PairCam* __restrict__ cams) {
    __shared__ float lds[kBlock / kWave];
    __shared__ float fbar_s[2];
    // fbar_k = mean_b (fx + fy)/2 of the ref frame (frame k) -- batch-coupled scalar
    for (int k = 0; k < 2; ++k) {
        float acc = 0.f;
        for (int b = threadIdx.x; b < B; b += kBlock) acc += intr[(b * 2 + k) * 4 + 0] + intr[(b * 2 + k) * 4 + 1];
        acc = block_sum(acc, lds);
        if (threadIdx.x == 0) fbar_s[k] = acc / (2.f * (float)B);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < B * 2; i += kBlock) {
        const int b = i >> 1, k = i & 1;
        const float* ir = intr + (b * 2 + k) * 4;
        const float* it = intr + (b * 2 + (1 - k)) * 4;
        const float* er = extr + (b * 2 + k) * 12;
        const float* et = extr + (b * 2 + (1 - k)) * 12;
        PairCam c;
        for (int j = 0; j < 3; ++j) {
            for (int l = 0; l < 3; ++l)
                c.M[j * 3 + l] = et[0 * 4 + j] * er[0 * 4 + l] + et[1 * 4 + j] * er[1 * 4 + l] + et[2 * 4 + j] * er[2 * 4 + l];
            c.c[j] = et[0 * 4 + j] * (er[3] - et[3]) + et[1 * 4 + j] * (er[7] - et[7]) + et[2 * 4 + j] * (er[11] - et[11]);
        }
        c.ifx_r = 1.f / ir[0]; c.ify_r = 1.f / ir[1]; c.cx_r = ir[2]; c.cy_r = ir[3];
        c.fx_t = it[0]; c.fy_t = it[1]; c.cx_t = it[2]; c.cy_t = it[3];
        const float S = fmaxf(mask_sum[i], 1e-6f);
        c.invS = 1.f / S;
        c.fbar = fbar_s[k];
        c.gr = lambda_r > 0.f ? lambda_r / (2.f * (float)B * S) : 0.f;
        c.gb = lambda_b > 0.f ? lambda_b * fbar_s[k] / (2.f * (float)B * S) : 0.f;
        c.sx = (float)W / (float)(W - 1);
        c.sy = (float)H / (float)(H - 1);
        for (int j = 0; j < 6; ++j) c.pad[j] = 0.f;
        cams[i] = c;
    }
}

// ---------------------------------------------------------------- depth parametrisation
template <int MODE> __device__ __forceinline__ float to_depth(float v) {
    if (MODE == CD_DEPTH_EXP) return expf(v);
    if (MODE == CD_DEPTH_RECIPROCAL) return __builtin_amdgcn_rcpf(v);
    return v;
}
// d depth / d v expressed through the depth value
template <int MODE> __device__ __forceinline__ float depth_jac(float d) {
    if (MODE == CD_DEPTH_EXP) return d;
    if (MODE == CD_DEPTH_RECIPROCAL) return -d * d;
    return 1.f;
}

// ---------------------------------------------------------------- the fused pass
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

template <int VEC, bool GRAD, int MODE, bool REPROJ, bool DISP>
__global__ __launch_bounds__(kBlock) void loss_main_kernel(
    const float* __restrict__ depth, const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd,
    const float* __restrict__ mask_fwd, const float* __restrict__ mask_bwd,
    const PairCam* __restrict__ cams, int H, int W, int nblk,
    float* __restrict__ partial, float* __restrict__ grad) {
    __shared__ float lds[kBlock / kWave];
    const int k = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const int pk = b * 2 + k;
    const PairCam& cam = cams[pk];  // block-uniform -> scalar loads
    const float* __restrict__ v_ref = depth + (size_t)pk * HW;
    const float* __restrict__ v_tgt = depth + (size_t)(b * 2 + (1 - k)) * HW;
    const float* __restrict__ fl = (k == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
    const float* __restrict__ mk = (k == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
    float* g_ref = GRAD ? grad + (size_t)pk * HW : nullptr;
    float* g_tgt = GRAD ? grad + (size_t)(b * 2 + (1 - k)) * HW : nullptr;

    float acc_r = 0.f, acc_d = 0.f;
    const int p0 = (blockIdx.x * kBlock + threadIdx.x) * VEC;
    if (p0 < HW) {
        float vin[VEC], fx[VEC], fy[VEC], m[VEC];
        using V = typename VecT<VEC>::type;
        *reinterpret_cast<V*>(vin) = *reinterpret_cast<const V*>(v_ref + p0);
        *reinterpret_cast<V*>(fx) = *reinterpret_cast<const V*>(fl + p0);
        *reinterpret_cast<V*>(fy) = *reinterpret_cast<const V*>(fl + HW + p0);
        *reinterpret_cast<V*>(m) = *reinterpret_cast<const V*>(mk + p0);
        const int y = p0 / W;          // the VEC pixels share a row (W % VEC == 0)
        const int x0 = p0 - y * W;
        const float r1 = -((float)y - cam.cy_r) * cam.ify_r;
        const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xf = (float)(x0 + i), yf = (float)y;
            const float d = to_depth<MODE>(vin[i]);
            const float r0 = (xf - cam.cx_r) * cam.ifx_r;
            // a = M (r0, r1, -1);  P = d a + c
            const float a0 = cam.M[0] * r0 + cam.M[1] * r1 - cam.M[2];
            const float a1 = cam.M[3] * r0 + cam.M[4] * r1 - cam.M[5];
            const float a2 = cam.M[6] * r0 + cam.M[7] * r1 - cam.M[8];
            const float X = d * a0 + cam.c[0], Y = d * a1 + cam.c[1], Z = d * a2 + cam.c[2];
            const float iZ = __builtin_amdgcn_rcpf(Z);
            const float mx = xf + fx[i], my = yf + fy[i];
            float g = 0.f;  // d total / d depth_ref at this pixel
            if (REPROJ) {
                // project (geometry.py:73-83): px = fx X/(-Z) + cx ; py = -(fy Y/(-Z)) + cy
                const float px = cam.cx_t - cam.fx_t * X * iZ;
                const float py = cam.cy_t + cam.fy_t * Y * iZ;
                const float ex = px - mx, ey = py - my;
                const float e2 = ex * ex + ey * ey;
                const float e = __builtin_amdgcn_sqrtf(e2);
                acc_r += m[i] * e;  // multiply, not select: 0*inf = NaN exactly like the reference
                if (GRAD) {
                    const float dpx = cam.fx_t * iZ * (X * a2 * iZ - a0);
                    const float dpy = cam.fy_t * iZ * (a1 - Y * a2 * iZ);
                    const float ie = e > 0.f ? __builtin_amdgcn_rcpf(e) : 0.f;  // subgradient 0 at e = 0
                    g += cam.gr * m[i] * (ex * dpx + ey * dpy) * ie;
                }
            }
            if (DISP) {
                // sample (geometry.py:201-208, grid_sample border, align_corners=False)
                float ix = fminf(fmaxf(mx * cam.sx - 0.5f, 0.f), Wm1);
                float iy = fminf(fmaxf(my * cam.sy - 0.5f, 0.f), Hm1);
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const float tx = ix - fx0, ty = iy - fy0;
                const int xa = (int)fx0, ya = (int)fy0;
                const int xb = min(xa + 1, W - 1), yb = min(ya + 1, H - 1);  // clipped tap has weight 0
                const int i00 = ya * W + xa, i01 = ya * W + xb, i10 = yb * W + xa, i11 = yb * W + xb;
                const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty);
                const float w10 = (1.f - tx) * ty, w11 = tx * ty;
                const float d00 = to_depth<MODE>(v_tgt[i00]), d01 = to_depth<MODE>(v_tgt[i01]);
                const float d10 = to_depth<MODE>(v_tgt[i10]), d11 = to_depth<MODE>(v_tgt[i11]);
                const float zs = -(d00 * w00 + d01 * w01 + d10 * w10 + d11 * w11);  // z = -depth
                const float izs = __builtin_amdgcn_rcpf(zs);
                const float dd = iZ - izs;
                acc_d += m[i] * fabsf(dd);
                if (GRAD) {
                    const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
                    const float gm = cam.gb * m[i] * sg;
                    g -= gm * a2 * iZ * iZ;          // d(1/Z)/dd = -a2/Z^2
                    const float gz = gm * izs * izs;  // -(d(-1/zs)/dzs) folded with zs = -sum w d
                    if (gz != 0.f) {
                        atomic_add_f32(g_tgt + i00, -gz * w00 * depth_jac<MODE>(d00));
                        atomic_add_f32(g_tgt + i01, -gz * w01 * depth_jac<MODE>(d01));
                        atomic_add_f32(g_tgt + i10, -gz * w10 * depth_jac<MODE>(d10));
                        atomic_add_f32(g_tgt + i11, -gz * w11 * depth_jac<MODE>(d11));
                    }
                }
            }
            if (GRAD) {
                const float gv = g * depth_jac<MODE>(d);
                if (gv != 0.f) atomic_add_f32(g_ref + p0 + i, gv);
            }
        }
    }
    acc_r = block_sum(acc_r, lds);
    acc_d = block_sum(acc_d, lds);
    if (threadIdx.x == 0) {
        float* o = partial + ((size_t)pk * nblk + blockIdx.x) * 2;
        o[0] = acc_r;
        o[1] = acc_d;
    }
}

// ---------------------------------------------------------------- fixed-order reductions
__global__ __launch_bounds__(kWave) void finalize_pairs_kernel(const float* __restrict__ partial,
                                                               const PairCam* __restrict__ cams, int nblk,
                                                               float lambda_r, float lambda_b,
                                                               float* __restrict__ reproj, float* __restrict__ disp) {
    const int b = blockIdx.x;
    double r[2], q[2];
    for (int k = 0; k < 2; ++k) {
        double ar = 0.0, ad = 0.0;
        const float* src = partial + (size_t)(b * 2 + k) * nblk * 2;
        for (int i = threadIdx.x; i < nblk; i += kWave) { ar += (double)src[i * 2]; ad += (double)src[i * 2 + 1]; }
        for (int off = kWave / 2; off > 0; off >>= 1) { ar += __shfl_down(ar, off, kWave); ad += __shfl_down(ad, off, kWave); }
        r[k] = ar * (double)cams[b * 2 + k].invS;
        q[k] = (double)cams[b * 2 + k].fbar * (ad * (double)cams[b * 2 + k].invS);
    }
    if (threadIdx.x == 0) {
        reproj[b] = lambda_r > 0.f ? (float)((double)lambda_r * (r[0] + r[1]) * 0.5) : 0.f;
        disp[b] = lambda_b > 0.f ? (float)((double)lambda_b * (q[0] + q[1]) * 0.5) : 0.f;
    }
}

__global__ __launch_bounds__(kBlock) void finalize_total_kernel(const float* __restrict__ reproj,
                                                                const float* __restrict__ disp, int B,
                                                                float* __restrict__ total) {
    __shared__ double lds[kBlock];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += kBlock) acc += (double)reproj[b] + (double)disp[b];
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = (float)(lds[0] / (double)B);
}

// ---------------------------------------------------------------- profiling hook
// bench.py brackets the fused pass (loss_main_kernel only) with HIP events recorded on the
// stream the kernel is launched on; events are pre-created by cd_profile_begin so recording
// costs ~1 us and never synchronises.  Not thread-safe; meant for one profiling thread.
struct Profiler {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    int* batch = nullptr;
};
static Profiler g_prof;

// ---------------------------------------------------------------- launch plumbing
template <int VEC, bool GRAD, int MODE>
static void launch_main(bool reproj_on, bool disp_on, dim3 grid, hipStream_t s, const float* depth,
                        const float* ff, const float* fb, const float* mf, const float* mb, const PairCam* cams,
                        int H, int W, int nblk, float* partial, float* grad) {
#define CD_LAUNCH(R, D) \
    hipLaunchKernelGGL((loss_main_kernel<VEC, GRAD, MODE, R, D>), grid, dim3(kBlock), 0, s, depth, ff, fb, mf, mb, \
                       cams, H, W, nblk, partial, grad)
    if (reproj_on && disp_on) CD_LAUNCH(true, true);
    else if (reproj_on) CD_LAUNCH(true, false);
    else if (disp_on) CD_LAUNCH(false, true);
    else CD_LAUNCH(false, false);
#undef CD_LAUNCH
}

template <int VEC, bool GRAD>
static void launch_mode(int mode, bool r, bool d, dim3 grid, hipStream_t s, const float* depth, const float* ff,
                        const float* fb, const float* mf, const float* mb, const PairCam* cams, int H, int W,
                        int nblk, float* partial, float* grad) {
    if (mode == CD_DEPTH_EXP) launch_main<VEC, GRAD, CD_DEPTH_EXP>(r, d, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, partial, grad);
    else if (mode == CD_DEPTH_RECIPROCAL) launch_main<VEC, GRAD, CD_DEPTH_RECIPROCAL>(r, d, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, partial, grad);
    else launch_main<VEC, GRAD, CD_DEPTH_IDENTITY>(r, d, grid, s, depth, ff, fb, mf, mb, cams, H, W, nblk, partial, grad);
}

static int run_loss(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                    const float* mask_sum, const float* intr, const float* extr, float lambda_r, float lambda_b,
                    int depth_mode, int B, int H, int W, float* reproj, float* disp, float* total, float* grad,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!depth || !ff || !fb || !mf || !mb || !intr || !extr || !reproj || !disp || !total || !workspace)
        return CD_ERR_INVALID_ARG;
    if (B <= 0 || H < 2 || W < 2 || depth_mode < 0 || depth_mode > 2) return CD_ERR_INVALID_ARG;
    if ((long long)H * W > (1ll << 30) || B > 65535) return CD_ERR_UNSUPPORTED;
    if (workspace_bytes < ws_bytes(B, H, W)) return CD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    const bool vec4 = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(depth) | reinterpret_cast<uintptr_t>(ff) |
                                        reinterpret_cast<uintptr_t>(fb) | reinterpret_cast<uintptr_t>(mf) |
                                        reinterpret_cast<uintptr_t>(mb)) % 16 == 0);
    Workspace w = carve(workspace, B, H, W, vec4 ? 4 : 1);

    if (!mask_sum) {
        if (hipMemsetAsync(w.mask_sum, 0, sizeof(float) * B * 2, s) != hipSuccess) return CD_ERR_LAUNCH;
        const int chunks = (HW + kBlock * 8 - 1) / (kBlock * 8);
        hipLaunchKernelGGL(mask_sum_kernel, dim3(chunks, 2, B), dim3(kBlock), 0, s, mf, mb, HW, w.mask_sum);
        CD_CHECK_LAUNCH();
        mask_sum = w.mask_sum;
    }
    hipLaunchKernelGGL(prep_kernel, dim3(1), dim3(kBlock), 0, s, intr, extr, mask_sum, lambda_r, lambda_b, B, H, W, w.cams);
    CD_CHECK_LAUNCH();
    if (grad && hipMemsetAsync(grad, 0, sizeof(float) * (size_t)B * 2 * HW, s) != hipSuccess) return CD_ERR_LAUNCH;

    const dim3 grid(w.nblk, 2, B);
    const bool r_on = lambda_r > 0.f, d_on = lambda_b > 0.f;
    const bool rec = g_prof.on && g_prof.n < g_prof.cap;
    if (rec) (void)hipEventRecord(g_prof.start[g_prof.n], s);
    if (vec4) {
        if (grad) launch_mode<4, true>(depth_mode, r_on, d_on, grid, s, depth, ff, fb, mf, mb, w.cams, H, W, w.nblk, w.partial, grad);
        else launch_mode<4, false>(depth_mode, r_on, d_on, grid, s, depth, ff, fb, mf, mb, w.cams, H, W, w.nblk, w.partial, grad);
    } else {
        if (grad) launch_mode<1, true>(depth_mode, r_on, d_on, grid, s, depth, ff, fb, mf, mb, w.cams, H, W, w.nblk, w.partial, grad);
        else launch_mode<1, false>(depth_mode, r_on, d_on, grid, s, depth, ff, fb, mf, mb, w.cams, H, W, w.nblk, w.partial, grad);
    }
    if (rec) {
        (void)hipEventRecord(g_prof.stop[g_prof.n], s);
        g_prof.batch[g_prof.n] = grad ? B : -B;  // negative = forward-only launch
        ++g_prof.n;
    }
    CD_CHECK_LAUNCH();
    hipLaunchKernelGGL(finalize_pairs_kernel, dim3(B), dim3(kWave), 0, s, w.partial, w.cams, w.nblk, lambda_r, lambda_b, reproj, disp);
    CD_CHECK_LAUNCH();
    hipLaunchKernelGGL(finalize_total_kernel, dim3(1), dim3(kBlock), 0, s, reproj, disp, B, total);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

}  // namespace cd

extern "C" {

int cd_profile_begin(int max_records) {
    using cd::g_prof;
    if (max_records <= 0 || g_prof.on) return CD_ERR_INVALID_ARG;
    g_prof.start = new hipEvent_t[max_records];
    g_prof.stop = new hipEvent_t[max_records];
    g_prof.batch = new int[max_records];
    for (int i = 0; i < max_records; ++i) {
        if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess)
            return CD_ERR_LAUNCH;
    }
    g_prof.cap = max_records;
    g_prof.n = 0;
    g_prof.on = true;
    return CD_OK;
}

int cd_profile_end(float* ms_out, int* batch_out, int capacity, int* n_out) {
    using cd::g_prof;
    if (!g_prof.on || !n_out) return CD_ERR_INVALID_ARG;
    int n = g_prof.n < capacity ? g_prof.n : capacity;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(g_prof.stop[i]);
        (void)hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]);
        if (ms_out) ms_out[i] = ms;
        if (batch_out) batch_out[i] = g_prof.batch[i];
    }
    *n_out = n;
    for (int i = 0; i < g_prof.cap; ++i) { (void)hipEventDestroy(g_prof.start[i]); (void)hipEventDestroy(g_prof.stop[i]); }
    delete[] g_prof.start; delete[] g_prof.stop; delete[] g_prof.batch;
    g_prof = cd::Profiler();
    return CD_OK;
}

size_t cd_consistency_loss_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return cd::ws_bytes(B, H, W);
}

int cd_mask_sums(const float* mask_fwd, const float* mask_bwd, int B, int H, int W, float* mask_sum, void* stream) {
    if (!mask_fwd || !mask_bwd || !mask_sum || B <= 0 || H <= 0 || W <= 0) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    if (hipMemsetAsync(mask_sum, 0, sizeof(float) * B * 2, s) != hipSuccess) return CD_ERR_LAUNCH;
    const int chunks = (HW + cd::kBlock * 8 - 1) / (cd::kBlock * 8);
    hipLaunchKernelGGL(cd::mask_sum_kernel, dim3(chunks, 2, B), dim3(cd::kBlock), 0, s, mask_fwd, mask_bwd, HW, mask_sum);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_consistency_loss_fwd_bwd(const float* depth, const float* flow_fwd, const float* flow_bwd,
                                const float* mask_fwd, const float* mask_bwd, const float* mask_sum,
                                const float* intr, const float* extr, float lambda_r, float lambda_b,
                                int depth_mode, int B, int H, int W, float* reproj, float* disp, float* total,
                                float* grad_in, void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad_in) return CD_ERR_INVALID_ARG;
    return cd::run_loss(depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, mask_sum, intr, extr, lambda_r, lambda_b,
                        depth_mode, B, H, W, reproj, disp, total, grad_in, workspace, workspace_bytes, stream);
}

int cd_consistency_loss_fwd(const float* depth, const float* flow_fwd, const float* flow_bwd,
                            const float* mask_fwd, const float* mask_bwd, const float* mask_sum,
                            const float* intr, const float* extr, float lambda_r, float lambda_b, int depth_mode,
                            int B, int H, int W, float* reproj, float* disp, float* total, void* workspace,
                            size_t workspace_bytes, void* stream) {
    return cd::run_loss(depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, mask_sum, intr, extr, lambda_r, lambda_b,
                        depth_mode, B, H, W, reproj, disp, total, nullptr, workspace, workspace_bytes, stream);
}

}  // extern "C"
