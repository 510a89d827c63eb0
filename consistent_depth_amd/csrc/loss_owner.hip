// Fused geometric-consistency loss, "owner-computes" formulation for gfx950 (v2).
//
// Why: the depth gradient has a scatter component (the 4 bilinear taps of the OTHER frame's
// depth, /root/reference/utils/geometry.py:201-208 under autograd).  fp32 global atomics on a
// multi-XCD MI355X execute memory-side at <= 267 G atomics/s (measured, profiles/), which caps a
// scatter-by-atomics kernel (loss_fused.hip, v1) at ~4 % of the HBM roofline.  This kernel has NO
// global atomics and needs no zero-initialised gradient:
//
//   one workgroup OWNS one TH x TW tile T of one gradient plane grad[b, j] and
//     phase 0  stages exp/reciprocal'ed depth in LDS: the tile of frame j (+1 halo) and a window
//              of the other frame k = 1-j,
//     phase 1  runs direction j over the pixels of T (they are its sources): reprojection +
//              disparity residuals, loss partial sums, and the DIRECT gradient term (registers);
//              bilinear taps come from the LDS window,
//     phase 2  PULLS the scatter term: re-evaluates direction k for the sources p inside the
//              window (depth from LDS, flow/mask from L2/HBM) and adds the tap contributions
//              that land inside T into an LDS accumulator.  The accumulator is 64-bit FIXED POINT
//              (2^-40 units, |value| < 2048, ds_add_u64): on gfx950 ds_add_f32 retires ~1 lane
//              per 3 clocks (185 CU-cycles per wave instruction, profiles/lds_atomics_exp_r01.txt)
//              while integer LDS atomics run at full rate (5 cycles) -- and integer addition is
//              associative, so the scatter sum is bit-reproducible,
//     phase 3  writes grad[b, j, T] = direct + scatter with plain coalesced stores.
//
// The window is a per-tile prediction: the bounding box of where T's own (valid) pixels sample
// frame k, grown by a margin -- for forward/backward-consistent flow (which is what the masks
// certify) that is exactly where the sources that sample T live.  Exactness does not depend on the
// prediction: in phase 1 every source checks whether the owner of each tap's tile will see it
// (integer test against that owner's window, from the same table); if not, the contribution goes to
// a small global overflow list that a follow-up kernel applies with atomics.  If even that list
// overflows, a device-side flag makes the (guarded, normally idle) v1 path recompute the gradient.
//
// Sampling positions are computed by ONE inline function with explicit fma/add intrinsics so the
// "will the owner see me" test and the owner's own scan agree bit for bit.
#include "loss_tiles.h"

namespace cd {

// ---------------------------------------------------------------- the owner kernel
__device__ __forceinline__ int imin_wave(int v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, kWave));
    return v;
}
__device__ __forceinline__ int imax_wave(int v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = max(v, __shfl_down(v, off, kWave));
    return v;
}

// ---------------------------------------------------------------- per-tile source windows
// wins[(b*2 + j)*ntiles + tile] = where the VALID pixels of tile T of frame j sample frame k = 1-j
// (tight tap bounding box, centre-cropped to WMAXW x WMAXH; v2 widens it on the fly with expand_win).
// Depends only on flows and masks, i.e. on the dataset: callers cache it per pair.
__global__ __launch_bounds__(kBlock) void tile_window_kernel(const float* __restrict__ flow_fwd,
                                                             const float* __restrict__ flow_bwd,
                                                             const float* __restrict__ mask_fwd,
                                                             const float* __restrict__ mask_bwd, int H, int W,
                                                             int tiles_x, int ntiles, int wstride, TileWin* __restrict__ wins) {
    __shared__ int red[4][kBlock / kWave];
    const int j = blockIdx.y, b = blockIdx.z, tile = blockIdx.x;
    const int HW = H * W;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * TW, Y0 = ty * TH;
    const float* fl = (j == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
    const float* mk = (j == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
    int x0 = 1 << 20, y0 = 1 << 20, x1 = -1, y1 = -1;
    const int lx = threadIdx.x & (TW - 1), ly0 = threadIdx.x / TW;
#pragma unroll
    for (int it = 0; it < TH / (kBlock / TW); ++it) {
        const int x = X0 + lx, y = Y0 + ly0 + it * (kBlock / TW);
        if (x < W && y < H) {
            const int p = y * W + x;
            if (mk[p] != 0.f) {
                const Taps t = tap_coords((float)x, (float)y, fl[p], fl[HW + p], sx, sy, W, H);
                x0 = min(x0, t.xa); y0 = min(y0, t.ya); x1 = max(x1, t.xb); y1 = max(y1, t.yb);
            }
        }
    }
    x0 = imin_wave(x0); y0 = imin_wave(y0); x1 = imax_wave(x1); y1 = imax_wave(y1);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) { red[0][wid] = x0; red[1][wid] = y0; red[2][wid] = x1; red[3][wid] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kBlock / kWave; ++i) {
            x0 = min(x0, red[0][i]); y0 = min(y0, red[1][i]); x1 = max(x1, red[2][i]); y1 = max(y1, red[3][i]);
        }
        TileWin w;
        if (x1 < 0) { w.x0 = 0; w.y0 = 0; w.w = 0; w.h = 0; }
        else {
            int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
            if (ww > WMAXW) { x0 += (ww - WMAXW) / 2; ww = WMAXW; }
            if (wh > WMAXH) { y0 += (wh - WMAXH) / 2; wh = WMAXH; }
            w.x0 = (short)x0; w.y0 = (short)y0; w.w = (short)ww; w.h = (short)wh;
        }
        wins[(size_t)b * wstride + (size_t)j * ntiles + tile] = w;
    }
}

template <int MODE, bool REPROJ>
__global__ __launch_bounds__(kBlock) void loss_owner_kernel(
    const float* __restrict__ depth, const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd,
    const float* __restrict__ mask_fwd, const float* __restrict__ mask_bwd, const PairCam* __restrict__ cams,
    const TileWin* __restrict__ wins, int H, int W, int tiles_x, int ntiles, int wstride, float* __restrict__ partial,
    float* __restrict__ grad, Overflow* ovf, unsigned* __restrict__ oidx, float* __restrict__ oval) {
    __shared__ float sA[WMAXH * WMAXW];   // depth of frame k over the window
    __shared__ float sB[SBH * SBW];       // depth of frame j over T + halo
    __shared__ unsigned long long sG[TH * TW];  // scatter accumulator for T, 2^-40 fixed point
    __shared__ TileWin sWin[MAXT_LDS];    // windows of plane (b,k)'s owners (in frame-j coordinates)
    __shared__ float red[kBlock / kWave];

    const int j = blockIdx.y, b = blockIdx.z, tile = blockIdx.x, k = 1 - j;
    const int HW = H * W;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int X0 = txi * TW, Y0 = tyi * TH;
    const PairCam& cj = cams[b * 2 + j];  // direction j: ref = frame j, tgt = frame k
    const PairCam& ck = cams[b * 2 + k];  // direction k: ref = frame k, tgt = frame j
    const TileWin win = expand_win(wins[(size_t)b * wstride + (size_t)j * ntiles + tile], EXPAND_V2, W, H);
    const TileWin* __restrict__ wins_k = wins + (size_t)b * wstride + (size_t)k * ntiles;
    const float* __restrict__ v_j = depth + (size_t)(b * 2 + j) * HW;
    const float* __restrict__ v_k = depth + (size_t)(b * 2 + k) * HW;
    const float* __restrict__ fl_j = (j == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
    const float* __restrict__ fl_k = (k == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
    const float* __restrict__ mk_j = (j == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
    const float* __restrict__ mk_k = (k == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
    float* __restrict__ g_j = grad + (size_t)(b * 2 + j) * HW;
    const bool table_in_lds = ntiles <= MAXT_LDS;

    // ---------------- phase 0: stage
    const int wn = (int)win.w * (int)win.h;
    const float inv_ww = win.w > 0 ? 1.f / (float)win.w : 0.f;
    for (int i = threadIdx.x; i < wn; i += kBlock) {
        const int r = (int)(((float)i + 0.5f) * inv_ww), c = i - r * win.w;
        sA[r * WMAXW + c] = to_depth<MODE>(v_k[(win.y0 + r) * W + win.x0 + c]);
    }
    for (int i = threadIdx.x; i < SBH * SBW; i += kBlock) {
        const int r = i / SBW, c = i - r * SBW;
        const int y = min(max(Y0 - 1 + r, 0), H - 1), x = min(max(X0 - 1 + c, 0), W - 1);
        sB[i] = to_depth<MODE>(v_j[y * W + x]);
    }
    for (int i = threadIdx.x; i < TH * TW; i += kBlock) sG[i] = 0ull;
    if (table_in_lds)
        for (int i = threadIdx.x; i < ntiles; i += kBlock) sWin[i] = expand_win(wins_k[i], EXPAND_V2, W, H);
    __syncthreads();

    // ---------------- phase 1: direction j over the pixels of T
    constexpr int ROWS_PER_IT = kBlock / TW, ITERS = TH / ROWS_PER_IT;
    const int lx = threadIdx.x & (TW - 1), ly0 = threadIdx.x / TW;
    float g_dir[ITERS];
    float acc_r = 0.f, acc_d = 0.f;
    const unsigned base_k = (unsigned)(b * 2 + k) * (unsigned)HW;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        g_dir[it] = 0.f;
        const int ly = ly0 + it * ROWS_PER_IT;
        const int x = X0 + lx, y = Y0 + ly;
        const bool valid = x < W && y < H;
        const int p = valid ? y * W + x : 0;
        const float d = sB[(ly + 1) * SBW + lx + 1];
        const float m = valid ? mk_j[p] : 0.f;
        const float fx = fl_j[p], fy = fl_j[HW + p];
        const float xf = (float)x, yf = (float)y;
        const float r0 = (xf - cj.cx_r) * cj.ifx_r, r1 = -(yf - cj.cy_r) * cj.ify_r;
        const float a0 = cj.M[0] * r0 + cj.M[1] * r1 - cj.M[2];
        const float a1 = cj.M[3] * r0 + cj.M[4] * r1 - cj.M[5];
        const float a2 = cj.M[6] * r0 + cj.M[7] * r1 - cj.M[8];
        const float X = d * a0 + cj.c[0], Y = d * a1 + cj.c[1], Z = d * a2 + cj.c[2];
        const float iZ = __builtin_amdgcn_rcpf(Z);
        float g = 0.f;
        if (REPROJ) {
            const float mx = xf + fx, my = yf + fy;
            const float ex = (cj.cx_t - cj.fx_t * X * iZ) - mx, ey = (cj.cy_t + cj.fy_t * Y * iZ) - my;
            const float e2 = ex * ex + ey * ey;
            // one transcendental for both e and 1/e; e2 == 0 -> e = 0 and subgradient 0 (like torch.norm's backward)
            const float ie = e2 > 0.f ? __builtin_amdgcn_rsqf(e2) : 0.f;
            const float e = e2 * ie;
            acc_r += valid ? m * e : 0.f;  // lanes outside a partial tile carry garbage
            const float dpx = cj.fx_t * iZ * (X * a2 * iZ - a0), dpy = cj.fy_t * iZ * (a1 - Y * a2 * iZ);
            g += cj.gr * m * (ex * dpx + ey * dpy) * ie;
        }
        const Taps t = tap_coords(xf, yf, fx, fy, cj.sx, cj.sy, W, H);
        // tap values: the LDS window when the whole wave's taps are inside it (the common case), else per tap
        const int ra = t.ya - win.y0, ca = t.xa - win.x0, dyb = t.yb - t.ya, dxb = t.xb - t.xa;
        const bool inside = (unsigned)ra < (unsigned)max(win.h - dyb, 0) && (unsigned)ca < (unsigned)max(win.w - dxb, 0);
        float d00, d01, d10, d11;
        if (__all(inside || !valid)) {
            const int i00 = inside ? ra * WMAXW + ca : 0;
            d00 = sA[i00]; d01 = sA[i00 + dxb]; d10 = sA[i00 + dyb * WMAXW]; d11 = sA[i00 + dyb * WMAXW + dxb];
        } else {
            const int rb = ra + dyb, cb = ca + dxb;
            const bool ina = (unsigned)ra < (unsigned)win.h, inb = (unsigned)rb < (unsigned)win.h;
            const bool inca = (unsigned)ca < (unsigned)win.w, incb = (unsigned)cb < (unsigned)win.w;
            d00 = (ina && inca) ? sA[ra * WMAXW + ca] : to_depth<MODE>(v_k[t.ya * W + t.xa]);
            d01 = (ina && incb) ? sA[ra * WMAXW + cb] : to_depth<MODE>(v_k[t.ya * W + t.xb]);
            d10 = (inb && inca) ? sA[rb * WMAXW + ca] : to_depth<MODE>(v_k[t.yb * W + t.xa]);
            d11 = (inb && incb) ? sA[rb * WMAXW + cb] : to_depth<MODE>(v_k[t.yb * W + t.xb]);
        }
        const float zs = -(d00 * t.w00 + d01 * t.w01 + d10 * t.w10 + d11 * t.w11);
        const float izs = __builtin_amdgcn_rcpf(zs);
        const float dd = iZ - izs;
        acc_d += valid ? m * fabsf(dd) : 0.f;
        const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
        const float gm = cj.gb * m * sg;
        g -= gm * a2 * iZ * iZ;
        const float gz = gm * izs * izs;
        if (valid) g_dir[it] = g * depth_jac<MODE>(d);
        // will the owners of the taps' tiles (plane (b,k)) see this source?  All four taps in one tile (the
        // common case) = one table lookup; anything else goes through the (wave-uniform) slow path.
        const int ta = (t.ya / TH) * tiles_x + t.xa / TW, tb = (t.yb / TH) * tiles_x + t.xb / TW;
        const bool seen = ta == tb && in_win(table_in_lds ? sWin[ta] : expand_win(wins_k[ta], EXPAND_V2, W, H), x, y);
        const bool check = m != 0.f && !seen;
        if (__any(check)) {
            const int xs[4] = {t.xa, t.xb, t.xa, t.xb}, ys[4] = {t.ya, t.ya, t.yb, t.yb};
            const float ws[4] = {t.w00, t.w01, t.w10, t.w11}, ds[4] = {d00, d01, d10, d11};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tq = (ys[q] / TH) * tiles_x + xs[q] / TW;
                const float cv = -gz * ws[q] * depth_jac<MODE>(ds[q]);
                const bool need = check && cv != 0.f && !in_win(table_in_lds ? sWin[tq] : expand_win(wins_k[tq], EXPAND_V2, W, H), x, y);
                ovf_push(need, ovf, oidx, oval, base_k + (unsigned)(ys[q] * W + xs[q]), cv);
            }
        }
    }

    // ---------------- phase 2: pull the scatter term from direction k sources inside the window
    if (ck.gb != 0.f)  // block-uniform: no disparity term (lambda_b <= 0) -> nothing to pull
    for (int i = threadIdx.x; i < wn; i += kBlock) {
        const int r = (int)(((float)i + 0.5f) * inv_ww), c = i - r * win.w;
        const int px = win.x0 + c, py = win.y0 + r, p = py * W + px;
        const float m = mk_k[p];
        if (m == 0.f) continue;
        const Taps t = tap_coords((float)px, (float)py, fl_k[p], fl_k[HW + p], ck.sx, ck.sy, W, H);
        if (t.xa > X0 + TW - 1 || t.xb < X0 || t.ya > Y0 + TH - 1 || t.yb < Y0) continue;
        const float d = sA[r * WMAXW + c];
        const float r0 = ((float)px - ck.cx_r) * ck.ifx_r, r1 = -((float)py - ck.cy_r) * ck.ify_r;
        const float a2 = ck.M[6] * r0 + ck.M[7] * r1 - ck.M[8];
        const float iZ = __builtin_amdgcn_rcpf(d * a2 + ck.c[2]);
        // all four taps lie inside T + 1 px halo
        const int ra = t.ya - Y0 + 1, rb = t.yb - Y0 + 1, ca = t.xa - X0 + 1, cb = t.xb - X0 + 1;
        const float d00 = sB[ra * SBW + ca], d01 = sB[ra * SBW + cb], d10 = sB[rb * SBW + ca], d11 = sB[rb * SBW + cb];
        const float zs = -(d00 * t.w00 + d01 * t.w01 + d10 * t.w10 + d11 * t.w11);
        const float izs = __builtin_amdgcn_rcpf(zs);
        const float dd = iZ - izs;
        const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
        const float gz = ck.gb * m * sg * izs * izs;
        const bool ya_in = (unsigned)(ra - 1) < (unsigned)TH, yb_in = (unsigned)(rb - 1) < (unsigned)TH;
        const bool xa_in = (unsigned)(ca - 1) < (unsigned)TW, xb_in = (unsigned)(cb - 1) < (unsigned)TW;
        if (ya_in && xa_in) atomicAdd(&sG[(ra - 1) * TW + ca - 1], to_fixed(-gz * t.w00 * depth_jac<MODE>(d00)));
        if (ya_in && xb_in) atomicAdd(&sG[(ra - 1) * TW + cb - 1], to_fixed(-gz * t.w01 * depth_jac<MODE>(d01)));
        if (yb_in && xa_in) atomicAdd(&sG[(rb - 1) * TW + ca - 1], to_fixed(-gz * t.w10 * depth_jac<MODE>(d10)));
        if (yb_in && xb_in) atomicAdd(&sG[(rb - 1) * TW + cb - 1], to_fixed(-gz * t.w11 * depth_jac<MODE>(d11)));
    }
    __syncthreads();

    // ---------------- phase 3: plain, coalesced store of the finished tile
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int ly = ly0 + it * ROWS_PER_IT;
        const int x = X0 + lx, y = Y0 + ly;
        if (x < W && y < H) g_j[y * W + x] = g_dir[it] + from_fixed(sG[ly * TW + lx]);
    }
    acc_r = block_sum(acc_r, red);
    acc_d = block_sum(acc_d, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((size_t)(b * 2 + j) * ntiles + tile) * 2;
        o[0] = acc_r;
        o[1] = acc_d;
    }
}

// ---------------------------------------------------------------- overflow list -> gradient
__global__ __launch_bounds__(kBlock) void overflow_apply_kernel(Overflow* ovf, const unsigned* __restrict__ oidx,
                                                                const float* __restrict__ oval,
                                                                float* __restrict__ grad) {
    const int count = ovf->count, cap = ovf->cap;
    if (count > cap || ovf->degenerate) {  // list overflowed (or the sweep met a degenerate depth): ask the v1 path to redo the gradient
        if (blockIdx.x == 0 && threadIdx.x == 0) ovf->fallback = 1;
        return;
    }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock)
        atomic_add_f32(grad + oidx[i], oval[i]);
}

}  // namespace cd

// ---------------------------------------------------------------- host side (used by loss_fused.hip's run_loss)
namespace cd {

int owner_tiles_x(int W) { return (W + TW - 1) / TW; }
int owner_ntiles(int H, int W) { return owner_tiles_x(W) * ((H + TH - 1) / TH); }
// the blob the callers cache per pair: one record per pair = tile windows [+ the row-sweep plan, loss_sweep.hip]
size_t owner_windows_bytes(int B, int H, int W) { return (size_t)B * pair_record_bytes(H, W); }

int launch_tile_windows(const float* ff, const float* fb, const float* mf, const float* mb, int B, int H, int W,
                        void* wins, hipStream_t s) {
    const int tx = owner_tiles_x(W), nt = owner_ntiles(H, W);
    const int wstride = (int)(pair_record_bytes(H, W) / sizeof(TileWin));
    hipLaunchKernelGGL(tile_window_kernel, dim3(nt, 2, B), dim3(kBlock), 0, s, ff, fb, mf, mb, H, W, tx, nt, wstride, (TileWin*)wins);
    if (hipGetLastError() != hipSuccess) return CD_ERR_LAUNCH;
    return launch_sweep_plan(ff, fb, mf, mb, B, H, W, wins, s);
}

template <int MODE>
static void launch_owner_mode(bool reproj, dim3 grid, hipStream_t s, const float* depth, const float* ff,
                              const float* fb, const float* mf, const float* mb, const PairCam* cams,
                              const TileWin* wins, int H, int W, int tx, int nt, int wstride, float* partial, float* grad,
                              Overflow* ovf, unsigned* oidx, float* oval) {
    if (reproj)
        hipLaunchKernelGGL((loss_owner_kernel<MODE, true>), grid, dim3(kBlock), 0, s, depth, ff, fb, mf, mb, cams, wins,
                           H, W, tx, nt, wstride, partial, grad, ovf, oidx, oval);
    else
        hipLaunchKernelGGL((loss_owner_kernel<MODE, false>), grid, dim3(kBlock), 0, s, depth, ff, fb, mf, mb, cams, wins,
                           H, W, tx, nt, wstride, partial, grad, ovf, oidx, oval);
}

// Enqueues: [overflow header reset] owner kernel, overflow apply.  `ovf_mem` holds the Overflow header
// followed by idx[cap] and val[cap].
int launch_owner(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                 const void* cams, const void* wins, int mode, bool reproj, int B, int H, int W, float* partial,
                 float* grad, void* ovf_mem, int ovf_cap, hipStream_t s, void (*before_main)(hipStream_t),
                 void (*after_main)(hipStream_t)) {
    const int tx = owner_tiles_x(W), nt = owner_ntiles(H, W);
    const int wstride = (int)(pair_record_bytes(H, W) / sizeof(TileWin));
    Overflow* ovf = (Overflow*)ovf_mem;
    unsigned* oidx = (unsigned*)((char*)ovf_mem + 256);
    float* oval = (float*)(oidx + ovf_cap);
    const Overflow init = {0, ovf_cap, 0, 0};
    // header reset: 16 bytes, stream ordered (hipMemcpyAsync from pageable host memory is staged by the runtime)
    if (hipMemsetAsync(ovf, 0, sizeof(Overflow), s) != hipSuccess) return CD_ERR_LAUNCH;
    if (hipMemsetD32Async((hipDeviceptr_t)&ovf->cap, init.cap, 1, s) != hipSuccess) return CD_ERR_LAUNCH;
    const dim3 grid(nt, 2, B);
    if (before_main) before_main(s);
    if (mode == CD_DEPTH_EXP)
        launch_owner_mode<CD_DEPTH_EXP>(reproj, grid, s, depth, ff, fb, mf, mb, (const PairCam*)cams, (const TileWin*)wins, H, W, tx, nt, wstride, partial, grad, ovf, oidx, oval);
    else if (mode == CD_DEPTH_RECIPROCAL)
        launch_owner_mode<CD_DEPTH_RECIPROCAL>(reproj, grid, s, depth, ff, fb, mf, mb, (const PairCam*)cams, (const TileWin*)wins, H, W, tx, nt, wstride, partial, grad, ovf, oidx, oval);
    else
        launch_owner_mode<CD_DEPTH_IDENTITY>(reproj, grid, s, depth, ff, fb, mf, mb, (const PairCam*)cams, (const TileWin*)wins, H, W, tx, nt, wstride, partial, grad, ovf, oidx, oval);
    if (after_main) after_main(s);
    if (hipGetLastError() != hipSuccess) return CD_ERR_LAUNCH;
    hipLaunchKernelGGL(overflow_apply_kernel, dim3(64), dim3(kBlock), 0, s, ovf, oidx, oval, grad);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

const int* owner_fallback_flag(void* ovf_mem) { return &((Overflow*)ovf_mem)->fallback; }

int launch_overflow_apply(void* ovf_mem, int ovf_cap, float* grad, hipStream_t s) {
    Overflow* ovf = (Overflow*)ovf_mem;
    unsigned* oidx = (unsigned*)((char*)ovf_mem + 256);
    float* oval = (float*)(oidx + ovf_cap);
    hipLaunchKernelGGL(overflow_apply_kernel, dim3(64), dim3(kBlock), 0, s, ovf, oidx, oval, grad);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // namespace cd
