// C-ABI entry points of the geometric-consistency loss and the launch sequence of one call:
//
//   [mask_sum_kernel]      only when the caller passes no cached mask sums
//   [tile_window_kernel]   only with gradients and when the caller passes no cached tile windows
//   prep_kernel            per-(pair, direction) camera constants + gradient scales (128 B each)
//   forward only :  loss_main_kernel<GRAD=false>          (v1, pure streaming, no atomics)
//   with gradient:  loss_source_kernel + loss_gather_kernel (v3: evaluate once, reduce slabs; no global atomics)
//                   -> overflow_apply_kernel
//                   -> zero_guarded + loss_main_kernel<GRAD=true> (idle unless the overflow list overflowed)
//   finalize_pairs / finalize_total   fixed-order fp64 reduction -> reproj[B], disp[B], total[1]
#include "loss_common.h"

namespace cd {

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

static int launch_mask_sums(const float* mf, const float* mb, int B, int HW, float* out, hipStream_t s) {
    if (hipMemsetAsync(out, 0, sizeof(float) * B * 2, s) != hipSuccess) return CD_ERR_LAUNCH;
    const int chunks = (HW + kBlock * 8 - 1) / (kBlock * 8);
    hipLaunchKernelGGL(mask_sum_kernel, dim3(chunks, 2, B), dim3(kBlock), 0, s, mf, mb, HW, out);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- per-(b,k) constants
__global__ __launch_bounds__(kBlock) void prep_kernel(const float* __restrict__ intr,
                                                      const float* __restrict__ extr,
                                                      const float* __restrict__ mask_sum,
                                                      float lambda_r, float lambda_b, int B, int H, int W,
                                                      PairCam* __restrict__ cams) {
    __shared__ float lds[kBlock / kWave];
    __shared__ float fbar_s[2];
    // fbar_k = mean_b (fx + fy)/2 of the ref frame (frame k) -- batch-coupled scalar (consistency_loss.py:178)
    for (int k = 0; k < 2; ++k) {
        float acc = 0.f;
        for (int b = threadIdx.x; b < B; b += kBlock) acc += intr[(b * 2 + k) * 4 + 0] + intr[(b * 2 + k) * 4 + 1];
        acc = block_sum(acc, lds);
        if (threadIdx.x == 0) fbar_s[k] = acc / (2.f * (float)B);
        __syncthreads();
    }
    for (int b = threadIdx.x; b < B; b += kBlock)
        prep_pair(intr + b * 8, extr + b * 24, mask_sum + b * 2, fbar_s, lambda_r, lambda_b, B, H, W, cams + b * 2);
}

// ---------------------------------------------------------------- fixed-order reductions
// alt_flag (may be null): when *alt_flag != 0 the guarded v1 pass has recomputed everything (overflow list overflowed, or the
// row sweep met a degenerate depth): its partial sums (alt_partial, alt_nblk per plane) are the loss.
__global__ __launch_bounds__(kWave) void finalize_pairs_kernel(const float* __restrict__ partial,
                                                               const PairCam* __restrict__ cams, int nblk,
                                                               float lambda_r, float lambda_b,
                                                               float* __restrict__ reproj, float* __restrict__ disp,
                                                               const int* __restrict__ alt_flag,
                                                               const float* __restrict__ alt_partial, int alt_nblk) {
    const int b = blockIdx.x;
    if (alt_flag != nullptr && *alt_flag != 0) { partial = alt_partial; nblk = alt_nblk; }
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
// bench.py brackets the fused pass (row sweep: EVERYTHING the call enqueues; tile kernels: the gradient kernels) with HIP events recorded on the
// stream the kernel is launched on; events are pre-created by cd_profile_begin so recording
// costs ~1 us and never synchronises.  Not thread-safe; meant for one profiling thread.
struct Profiler {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    int* batch = nullptr;
    int pending_batch = 0;
};
static Profiler g_prof;

static void prof_before(hipStream_t s) {
    if (g_prof.on && g_prof.n < g_prof.cap) (void)hipEventRecord(g_prof.start[g_prof.n], s);
}
static void prof_after(hipStream_t s) {
    if (g_prof.on && g_prof.n < g_prof.cap) {
        (void)hipEventRecord(g_prof.stop[g_prof.n], s);
        g_prof.batch[g_prof.n] = g_prof.pending_batch;
        ++g_prof.n;
    }
}

// ---------------------------------------------------------------- workspace
struct Workspace {
    WorkspaceHeader* hdr;   // 256 B at offset 0 (a FIXED place whatever B: state that outlives a call, cd_consistency_loss_workspace_init)
    PairCam* cams;     // [B*2]
    float* mask_sum;   // [B*2]            (when the caller passes none)
    float* partial;    // [B*2][ntiles][2] (v2) or [B*2][nblk][2] (v1 forward-only)
    float* partial_fb; // [B*2][nblk1][2]  scratch partials of the guarded v1 fallback
    void* wins;        // tile windows     (when the caller passes none)
    void* ovf;         // 256 B header + idx[cap] + val[cap]
    int ovf_cap;
    float* slabs;      // v3: [min(B,chunk)*2][ntiles][48*48] floats (only each tile's window is touched)
};

static inline int ovf_capacity(int B, int HW) {
    long long c = (long long)B * 2 * HW / 4;
    if (c < 4096) c = 4096;
    if (c > (1ll << 28)) c = 1ll << 28;
    return (int)c;
}

static inline size_t ws_layout(int B, int H, int W, void* base, Workspace* w) {
    const int HW = H * W;
    const int np = v1_blocks_per_plane(HW, 1) > owner_ntiles(H, W) ? v1_blocks_per_plane(HW, 1) : owner_ntiles(H, W);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t o_hdr = take(sizeof(WorkspaceHeader));
    const size_t o_cams = take(sizeof(PairCam) * (size_t)B * 2);
    const size_t o_msum = take(sizeof(float) * (size_t)B * 2);
    const size_t o_part = take(sizeof(float) * (size_t)B * 2 * np * 2);
    const size_t o_pfb = take(sizeof(float) * (size_t)B * 2 * v1_blocks_per_plane(HW, 1) * 2);
    const size_t o_wins = take(owner_windows_bytes(B, H, W));
    const int cap = ovf_capacity(B, HW);
    const size_t o_ovf = take(256 + (size_t)cap * 8);
    const size_t o_slab = take(sizeof(float) * slab_floats(B, H, W));
    if (w) {
        char* p = (char*)base;
        w->hdr = (WorkspaceHeader*)(p + o_hdr);
        w->cams = (PairCam*)(p + o_cams); w->mask_sum = (float*)(p + o_msum); w->partial = (float*)(p + o_part);
        w->partial_fb = (float*)(p + o_pfb); w->wins = p + o_wins; w->ovf = p + o_ovf; w->ovf_cap = cap;
        w->slabs = (float*)(p + o_slab);
    }
    return off;
}

static int g_force_overflow_cap = -1;  // test hook: shrink the overflow list (cd_debug_set_overflow_capacity)
static int g_loss_variant = 0;         // 0 = by batch and geometry (row sweep v4 when it fills the chip, else v3), 3 / 4 forced

static int run_loss(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                    const float* mask_sum, const void* tile_windows, const float* intr, const float* extr,
                    float lambda_r, float lambda_b, int depth_mode, int B, int H, int W, float* reproj, float* disp,
                    float* total, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    if (!depth || !ff || !fb || !mf || !mb || !intr || !extr || !reproj || !disp || !total || !workspace)
        return CD_ERR_INVALID_ARG;
    if (B <= 0 || H < 2 || W < 2 || depth_mode < 0 || depth_mode > 2) return CD_ERR_INVALID_ARG;
    if ((long long)H * W > (1ll << 26) || H > 32767 || W > 32767 || (long long)B * 2 * H * W >= (1ll << 32))
        return CD_ERR_UNSUPPORTED;
    if (workspace_bytes < ws_layout(B, H, W, nullptr, nullptr)) return CD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    Workspace w;
    ws_layout(B, H, W, workspace, &w);
    int rc;
    if (!mask_sum) {
        if ((rc = launch_mask_sums(mf, mb, B, HW, w.mask_sum, s)) != CD_OK) return rc;
        mask_sum = w.mask_sum;
    }
    const bool r_on = lambda_r > 0.f, d_on = lambda_b > 0.f;
    const bool vec4 = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(depth) | reinterpret_cast<uintptr_t>(ff) |
                                        reinterpret_cast<uintptr_t>(fb) | reinterpret_cast<uintptr_t>(mf) |
                                        reinterpret_cast<uintptr_t>(mb)) % 16 == 0);
    // the row sweep moves the pixels of a thread with one vector access per plane and row: planes must start on 16-byte boundaries
    const bool sweep_aligned = grad && ((reinterpret_cast<uintptr_t>(depth) | reinterpret_cast<uintptr_t>(ff) | reinterpret_cast<uintptr_t>(fb) |
                                          reinterpret_cast<uintptr_t>(mf) | reinterpret_cast<uintptr_t>(mb) | reinterpret_cast<uintptr_t>(grad)) % 16 == 0) &&
                               ((size_t)HW * 4) % 16 == 0;
    const bool use_sweep = grad && d_on && sweep_aligned && sweep_supported(H, W) &&
                           (g_loss_variant == 4 || (g_loss_variant == 0 && sweep_preferred(B, H, W)));
    if (!use_sweep) {      // (the row sweep computes the per-pair constants in its own units kernel: one launch less)
        hipLaunchKernelGGL(prep_kernel, dim3(1), dim3(kBlock), 0, s, intr, extr, mask_sum, lambda_r, lambda_b, B, H, W, w.cams);
        CD_CHECK_LAUNCH();
    }
    int nparts, alt_nparts = 0;
    const int* alt_flag = nullptr;
    if (!grad) {
        g_prof.pending_batch = -B;
        prof_before(s);
        rc = launch_v1(depth, ff, fb, mf, mb, w.cams, depth_mode, r_on, d_on, vec4, B, H, W, w.partial, nullptr, nullptr, s);
        prof_after(s);
        if (rc != CD_OK) return rc;
        nparts = v1_blocks_per_plane(HW, vec4 ? 4 : 1);
    } else {
        if (!tile_windows) {
            if ((rc = launch_tile_windows(ff, fb, mf, mb, B, H, W, w.wins, s)) != CD_OK) return rc;
            tile_windows = w.wins;
        }
        g_prof.pending_batch = B;
        const int cap = (g_force_overflow_cap >= 0 && g_force_overflow_cap < w.ovf_cap) ? g_force_overflow_cap : w.ovf_cap;
        if (use_sweep)      // ONE kernel: gradient, per-pair losses and their mean are complete when it is (loss_sweep.hip)
            return launch_sweep(depth, ff, fb, mf, mb, w.cams, tile_windows, depth_mode, r_on, B, H, W, grad, w.ovf, cap, s, prof_before,
                                prof_after, intr, extr, mask_sum, lambda_r, lambda_b, reproj, disp, total, w.hdr);
        rc = launch_slab(depth, ff, fb, mf, mb, w.cams, tile_windows, depth_mode, r_on, B, H, W, w.partial, grad, w.slabs,
                         w.ovf, cap, s, prof_before, prof_after);
        if (rc != CD_OK) return rc;
        // device-side fallback: idle unless the overflow list overflowed (then it recomputes the gradient)
        const int* flag = owner_fallback_flag(w.ovf);
        const bool fb_vec4 = vec4 && ((size_t)B * 2 * HW) % 4 == 0;
        if ((rc = launch_zero_guarded(grad, (size_t)B * 2 * HW, flag, s)) != CD_OK) return rc;
        if ((rc = launch_v1(depth, ff, fb, mf, mb, w.cams, depth_mode, r_on, d_on, fb_vec4, B, H, W, w.partial_fb, grad, flag, s)) != CD_OK)
            return rc;
        alt_flag = flag;
        alt_nparts = v1_blocks_per_plane(HW, fb_vec4 ? 4 : 1);
        nparts = owner_ntiles(H, W);
    }
    hipLaunchKernelGGL(finalize_pairs_kernel, dim3(B), dim3(kWave), 0, s, w.partial, w.cams, nparts, lambda_r, lambda_b, reproj, disp,
                       alt_flag, w.partial_fb, alt_nparts);
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

int cd_debug_set_loss_chunk(int pairs) {
    cd::set_slab_chunk_pairs(pairs);
    return CD_OK;
}

int cd_debug_set_loss_variant(int v) {
    if (v != 0 && v != 3 && v != 4) return CD_ERR_INVALID_ARG;
    cd::g_loss_variant = v;
    return CD_OK;
}

int cd_debug_set_loss_sweep(int pixels_per_thread) {
    if (pixels_per_thread != 0 && pixels_per_thread != 1 && pixels_per_thread != 2 && pixels_per_thread != 4) return CD_ERR_INVALID_ARG;
    cd::set_sweep_pxt(pixels_per_thread);
    return CD_OK;
}

int cd_debug_set_overflow_capacity(int cap) {
    cd::g_force_overflow_cap = cap;
    return CD_OK;
}

size_t cd_consistency_loss_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return cd::ws_layout(B, H, W, nullptr, nullptr);
}

int cd_consistency_loss_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < sizeof(cd::WorkspaceHeader)) return CD_ERR_INVALID_ARG;
        // (the 8 bytes travel inside the command: no host buffer to keep alive, capturable)
    if (hipMemsetD32Async((hipDeviceptr_t)((char*)workspace + 4), 0, 1, (hipStream_t)stream) != hipSuccess) return CD_ERR_LAUNCH;
    if (hipMemsetD32Async((hipDeviceptr_t)workspace, (int)cd::kWorkspaceMagic, 1, (hipStream_t)stream) != hipSuccess) return CD_ERR_LAUNCH;
    return CD_OK;
}

int cd_mask_sums(const float* mask_fwd, const float* mask_bwd, int B, int H, int W, float* mask_sum, void* stream) {
    if (!mask_fwd || !mask_bwd || !mask_sum || B <= 0 || H <= 0 || W <= 0) return CD_ERR_INVALID_ARG;
    return cd::launch_mask_sums(mask_fwd, mask_bwd, B, H * W, mask_sum, (hipStream_t)stream);
}

size_t cd_tile_windows_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return cd::owner_windows_bytes(B, H, W);
}

int cd_tile_windows(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int B,
                    int H, int W, void* tile_windows, void* stream) {
    if (!flow_fwd || !flow_bwd || !mask_fwd || !mask_bwd || !tile_windows || B <= 0 || H < 2 || W < 2 || H > 32767 || W > 32767)
        return CD_ERR_INVALID_ARG;
    return cd::launch_tile_windows(flow_fwd, flow_bwd, mask_fwd, mask_bwd, B, H, W, tile_windows, (hipStream_t)stream);
}

int cd_consistency_loss_fwd_bwd(const float* depth, const float* flow_fwd, const float* flow_bwd,
                                const float* mask_fwd, const float* mask_bwd, const float* mask_sum,
                                const void* tile_windows, const float* intr, const float* extr, float lambda_r,
                                float lambda_b, int depth_mode, int B, int H, int W, float* reproj, float* disp,
                                float* total, float* grad_in, void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad_in) return CD_ERR_INVALID_ARG;
    return cd::run_loss(depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, mask_sum, tile_windows, intr, extr, lambda_r,
                        lambda_b, depth_mode, B, H, W, reproj, disp, total, grad_in, workspace, workspace_bytes, stream);
}

int cd_consistency_loss_fwd(const float* depth, const float* flow_fwd, const float* flow_bwd,
                            const float* mask_fwd, const float* mask_bwd, const float* mask_sum,
                            const float* intr, const float* extr, float lambda_r, float lambda_b, int depth_mode,
                            int B, int H, int W, float* reproj, float* disp, float* total, void* workspace,
                            size_t workspace_bytes, void* stream) {
    return cd::run_loss(depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, mask_sum, nullptr, intr, extr, lambda_r, lambda_b,
                        depth_mode, B, H, W, reproj, disp, total, nullptr, workspace, workspace_bytes, stream);
}

}  // extern "C"
