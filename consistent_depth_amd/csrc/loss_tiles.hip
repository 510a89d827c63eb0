// Tile tables and overflow list shared by the gradient kernels of the fused loss (loss_slab.hip, loss_sweep.hip):
//   tile_window_kernel     per (pair, direction, 32x32 source tile): the window of the other frame its valid pixels sample
//                          (dataset constant; first part of the per-pair record, the row-sweep plan follows)
//   overflow_apply_kernel  the few contributions a gradient kernel could not place (tap outside the staged window /
//                          accumulator range) are pushed to a global list and applied here with atomics; if the list itself
//                          overflows, a device-side flag lets the guarded v1 pass (loss_fused.hip) recompute the gradient.
// (Round 1's "owner-computes" gradient kernel (v2) lived here; it was never the dispatch choice after v3 and was removed in
// round 4 -- its measurements stay in profiles/loss_bench_v2_owner_r01.txt and DESIGN.md section 4.1.)
#include "loss_tiles.h"

namespace cd {

// ---------------------------------------------------------------- wave reductions
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
// (tight tap bounding box, centre-cropped to WMAXW x WMAXH).
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

const int* owner_fallback_flag(void* ovf_mem) { return &((Overflow*)ovf_mem)->fallback; }

int launch_overflow_apply(void* ovf_mem, int ovf_cap, float* grad, hipStream_t s) {
    Overflow* ovf = (Overflow*)ovf_mem;
    unsigned* oidx = (unsigned*)((char*)ovf_mem + 256);
    float* oval = (float*)(oidx + ovf_cap);
    hipLaunchKernelGGL(overflow_apply_kernel, dim3(64), dim3(kBlock), 0, s, ovf, oidx, oval, grad);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // namespace cd
