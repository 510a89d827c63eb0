// Shared by the consistency-loss kernels (loss_fused.hip: v1 scatter-by-atomics, now the
// forward-only path and the device-side fallback; loss_tiles.hip: tile windows + overflow list; loss_api.hip:
// C-ABI entry points and launch sequence).
#pragma once
#include "cd_common.h"
#include "loss_math.h"   // PairCam, Taps, tap_coords, to_depth, depth_jac (host/device neutral)

namespace cd {

static_assert(kDepthIdentity == CD_DEPTH_IDENTITY && kDepthExp == CD_DEPTH_EXP && kDepthReciprocal == CD_DEPTH_RECIPROCAL,
              "loss_math.h depth modes must match the public header");

// ---- v1 (loss_fused.hip)
int v1_blocks_per_plane(int HW, int vec);
// run_flag: nullptr = always run; else the kernel body runs only if *run_flag != 0 (device-side fallback).
int launch_v1(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, const void* cams,
              int mode, bool reproj, bool disp, bool vec4, int B, int H, int W, float* partial, float* grad,
              const int* run_flag, hipStream_t s);
int launch_zero_guarded(float* buf, size_t n, const int* run_flag, hipStream_t s);

// ---- tile tables and the overflow list (loss_tiles.hip; the names keep round 1's "owner" prefix: 32x32 tiles of a gradient plane)
int owner_tiles_x(int W);
int owner_ntiles(int H, int W);
size_t owner_windows_bytes(int B, int H, int W);
int launch_tile_windows(const float* ff, const float* fb, const float* mf, const float* mb, int B, int H, int W,
                        void* wins, hipStream_t s);
// ovf_mem = 256-byte header + idx[cap] + val[cap].
const int* owner_fallback_flag(void* ovf_mem);
int launch_overflow_apply(void* ovf_mem, int ovf_cap, float* grad, hipStream_t s);

// The first 256 bytes of the loss workspace: state that OUTLIVES a call (cd_consistency_loss_workspace_init writes it once after the
// workspace is allocated).  `finished` counts the workgroups of the running row-sweep launch that are done; the last one averages the
// per-pair losses and puts the counter back to zero, so no per-call reset dispatch is needed.
constexpr unsigned kWorkspaceMagic = 0xC0D1A6D5u;
struct WorkspaceHeader { unsigned magic; unsigned finished; unsigned reserved[62]; };
static_assert(sizeof(WorkspaceHeader) == 256, "header = one 256-byte workspace slot");

// ---- v4 (loss_sweep.hip): one workgroup per pair, row rings in LDS; plans live in the tile-windows blob
size_t pair_record_bytes(int H, int W);   // bytes of one pair's record of the blob: tile windows [+ sweep plan]
int launch_sweep_plan(const float* ff, const float* fb, const float* mf, const float* mb, int B, int H, int W, void* blob,
                      hipStream_t s);
bool sweep_supported(int H, int W);
bool sweep_preferred(int B, int H, int W);
void set_sweep_pxt(int pxt);
// ONE kernel per call: per-pair constants, sweep, the pair's overflow entries, exact mode for pairs the sweep cannot take, per-pair
// losses and their batch mean (reproj / disp / total are complete when it is).  ovf_mem: 256-byte header + idx[cap] + val[cap], the list
// is cut into B per-pair segments.  before / after: the profiling hooks around everything the call enqueues.
int launch_sweep(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, void* cams,
                 const void* blob, int mode, bool reproj, int B, int H, int W, float* grad, void* ovf_mem, int ovf_cap, hipStream_t s,
                 void (*before)(hipStream_t), void (*after)(hipStream_t), const float* intr, const float* extr, const float* mask_sum,
                 float lambda_r, float lambda_b, float* reproj_out, float* disp_out, float* total_out, WorkspaceHeader* hdr);

// ---- v3 (loss_slab.hip): source pass + gather pass; slabs = slab_floats(B,H,W) floats of scratch
size_t slab_floats(int B, int H, int W);
int slab_chunk_pairs(int H, int W);     // pairs per source+gather launch pair (slabs sized to stay cache resident)
void set_slab_chunk_pairs(int pairs);   // test/measurement hook; 0 restores the default
int launch_slab(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                const void* cams, const void* wins, int mode, bool reproj, int B, int H, int W, float* partial,
                float* grad, float* slabs, void* ovf_mem, int ovf_cap, hipStream_t s, void (*before_main)(hipStream_t),
                void (*after_main)(hipStream_t));

}  // namespace cd
