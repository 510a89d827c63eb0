/*
 * consistent_depth_amd -- C ABI of the MI355X-native (gfx950) fine-tuning hot path.
 *
 * This is the drop-in boundary for the depth_fine_tuning.py step of
 * facebookresearch/consistent_depth.  The reference has no FFI of its own (it is pure
 * Python on stock PyTorch ops); each entry point below replaces
 * the chain of ATen ops the cited reference lines launch.  Conventions:
 *
 *   - every pointer is a DEVICE pointer to contiguous fp32 (NCHW / row-major) unless
 *     the parameter says otherwise; nothing is allocated inside the library -- the
 *     caller passes workspaces sized by the matching *_workspace_bytes() call;
 *   - `stream` is the caller's hipStream_t (as void*); all work is enqueued on it and
 *     nothing synchronises the device;
 *   - return value: 0 = ok, negative = cd_status error; no exceptions cross the boundary;
 *   - no torch / C++ types appear in any signature.
 */
#ifndef CONSISTENT_DEPTH_AMD_H
#define CONSISTENT_DEPTH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cd_status {
    CD_OK = 0,
    CD_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, bad enum           */
    CD_ERR_WORKSPACE = -2,   /* workspace_bytes smaller than *_workspace_bytes()      */
    CD_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after a launch        */
    CD_ERR_UNSUPPORTED = -4  /* shape outside what the kernels were built for         */
} cd_status;

/* How the `depth` argument of the loss relates to the network output (the reference
 * applies these as separate ATen ops before the loss; here they are fused into it and
 * the returned gradient is w.r.t. the tensor that was passed in):
 *   CD_DEPTH_IDENTITY    depth = x                      (loss/consistency_loss.py:210 contract)
 *   CD_DEPTH_EXP         depth = exp(x)                 (monodepth/mannequin_challenge_model.py:66)
 *   CD_DEPTH_RECIPROCAL  depth = 1/x                    (monodepth/midas_v2_model.py:67)        */
typedef enum cd_depth_mode {
    CD_DEPTH_IDENTITY = 0,
    CD_DEPTH_EXP = 1,
    CD_DEPTH_RECIPROCAL = 2
} cd_depth_mode;

/* ABI version, bumped on any change of an exported signature, of the meaning of an argument, or of the export list
 * (6: cd_conv2d_fwd_grouped / cd_conv2d_wgrad_grouped / cd_conv2d_wgrad_desc / cd_conv2d_wgrad_table /
 * cd_conv2d_fwd_multi added, cd_bn_relu_bwd's last
 * argument became a flags bitfield, cd_debug_set_loss_variant(2) is refused;
 * 9: cd_consistency_loss_workspace_init added -- a loss workspace must be initialised once before its first use; cd_copy_segments,
 *    cd_counters_add, cd_zero_bytes, cd_debug_set_layers_mode added).
 * The loader (consistent_depth_amd/_native.py) refuses a library whose cd_abi_version() differs from this constant. */
#define CD_ABI_VERSION 9
int cd_abi_version(void);

/* Batch-statistics buffers (the `stats` arguments below) hold CD_BN_STAT_SLOTS partial copies:
 *   double stats[CD_BN_STAT_SLOTS][ctot][2]   -- (sum, sum of squares) per channel of the concat buffer.
 * The convolution epilogue adds each workgroup's contribution into one of the copies (same-address fp64 atomics
 * serialise in the memory system; spreading them keeps the epilogue off the critical path), the BatchNorm entry points
 * sum the copies in slot order.  A caller that synthesises statistics (eval mode) writes slot 0 and zeroes the rest. */
#define CD_BN_STAT_SLOTS 16
/* Human-readable build string ("gfx950 hipcc x.y ..."); static storage. */
const char* cd_build_info(void);

/* ------------------------------------------------------------------------------------
 * Geometric-consistency loss  (reference: loss/consistency_loss.py:98-253 +
 * utils/geometry.py:9-128,201-208; closed form in SURVEY.md appendix A.1)
 * ---------------------------------------------------------------------------------- */

/* Bytes of scratch the loss entry points need for a batch of B pairs of HxW frames. */
size_t cd_consistency_loss_workspace_bytes(int B, int H, int W);
/* ABI 9.  A loss workspace carries a 256-byte header at its start whose contents OUTLIVE a call (the finished-workgroup counter of the
 * row-sweep gradient kernel: the workgroup that finishes last puts it back to zero, so a call needs no reset dispatch of its own).
 * Call this ONCE after allocating (or re-allocating) a workspace, on the stream of the loss calls that follow or synchronised with it,
 * before the first cd_consistency_loss_fwd_bwd / _fwd on it; the same workspace may then serve any (B, H, W) that fits it, one call at a
 * time.  A gradient call on a workspace that was never initialised reports total = NaN (never a stale number).  Re-initialise after a
 * call that was aborted (device reset).  Enqueues two 4-byte fills; capturable.  (Replaces nothing in the reference: torch's
 * allocator zero-fills nothing either -- this is the price of keeping per-call state out of the entry point.) */
int cd_consistency_loss_workspace_init(void* workspace, size_t workspace_bytes, void* stream);

/* Per-pair, per-direction mask sums S[b,k] = sum(mask_k[b])  (weighted_mean_loss,
 * loss/consistency_loss.py:85).  They depend only on the dataset, so a caller may
 * compute them once per pair and pass them to the loss; mask_sum is [B,2] fp32. */
int cd_mask_sums(const float* mask_fwd, const float* mask_bwd, int B, int H, int W,
                 float* mask_sum, void* stream);

/* Per-tile source windows of the gradient kernel (opaque, cd_tile_windows_bytes(B,H,W) bytes for B
 * pairs).  Like the mask sums they depend only on flows and masks, i.e. on the dataset: compute them
 * once per pair (rows of B pairs can be gathered: the layout is [B][2][tiles] x 8 bytes) and pass
 * them to cd_consistency_loss_fwd_bwd; NULL there = recomputed on every call (one extra read of the
 * flows and masks).  They must belong to EXACTLY the flows and masks of the call (like mask_sum): the
 * row-sweep kernel trusts the plan's statement that a row group's valid sources sample resident rows
 * (no clamp, no vote for such groups) -- windows of other flows or masks give undefined gradients. */
size_t cd_tile_windows_bytes(int B, int H, int W);
int cd_tile_windows(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd,
                    const float* mask_bwd, int B, int H, int W, void* tile_windows, void* stream);

/*
 * Fused forward + analytic backward of ConsistencyLoss.__call__ in one pass over the
 * frame pairs (replaces geometry.py pixel_grid/pixels_to_points/reproject_points/
 * project/sample, the two weighted_mean_loss reductions per direction and the whole
 * autograd backward of that chain).
 *
 *   depth      [B,2,H,W]  network output for the two frames of each pair (see depth_mode)
 *   flow_fwd   [B,2,H,W]  metadata["geometry_consistency"]["flows"][0]  (dx,dy) px, frame0->frame1
 *   flow_bwd   [B,2,H,W]  ...["flows"][1], frame1->frame0
 *   mask_fwd   [B,1,H,W]  ...["masks"][0]  fp32 {0,1};  mask_bwd = ["masks"][1]
 *   mask_sum   [B,2] or NULL  precomputed cd_mask_sums(); NULL = computed here (one extra
 *                          read of the masks)
 *   tile_windows  cd_tile_windows() result for these B pairs, or NULL = computed here
 *   intr       [B,2,4]    fx,fy,cx,cy per frame      (metadata["intrinsics"])
 *   extr       [B,2,3,4]  [R|t] camera-to-world      (metadata["extrinsics"])
 *   lambda_r / lambda_b   opt.lambda_reprojection / opt.lambda_view_baseline; a term whose
 *                          lambda <= 0 is skipped and reported as zeros (consistency_loss.py:169,176,198,204)
 * outputs
 *   reproj[B], disp[B]    batch_losses["reprojection"], ["disparity"] (lambda-weighted, :194-205)
 *   total[1]              mean_b(reproj + disp)                         (:208)
 *   grad_in [B,2,H,W]     d total / d depth-argument (chain rule through depth_mode included)
 */
int cd_consistency_loss_fwd_bwd(
    const float* depth, const float* flow_fwd, const float* flow_bwd,
    const float* mask_fwd, const float* mask_bwd, const float* mask_sum,
    const void* tile_windows, const float* intr, const float* extr,
    float lambda_r, float lambda_b, int depth_mode, int B, int H, int W,
    float* reproj, float* disp, float* total, float* grad_in,
    void* workspace, size_t workspace_bytes, void* stream);

/* Forward only (validation sweep, depth_fine_tuning.py:312-406 under no_grad). */
int cd_consistency_loss_fwd(
    const float* depth, const float* flow_fwd, const float* flow_bwd,
    const float* mask_fwd, const float* mask_bwd, const float* mask_sum,
    const float* intr, const float* extr,
    float lambda_r, float lambda_b, int depth_mode, int B, int H, int W,
    float* reproj, float* disp, float* total,
    void* workspace, size_t workspace_bytes, void* stream);

/* Measurement hook (bench.py): while active, every loss call brackets ITS FUSED PASS (the
 * loss_main kernel only, not prep/finalize) with a pair of HIP events recorded on the caller's
 * stream.  cd_profile_end waits for them and returns per-launch milliseconds and the batch
 * size of each launch (negative = forward-only launch); it ends the session. */
int cd_profile_begin(int max_records);
int cd_profile_end(float* ms_out, int* batch_out, int capacity, int* n_out);

/* Test hook: cap the gradient kernel's overflow list at `cap` records (< 0 restores the default), to
 * force the overflow-apply and the device-side fallback paths in tests. */
int cd_debug_set_overflow_capacity(int cap);
/* Test / A-B hook: gradient-kernel formulation.  0 = default dispatch (the row sweep when the batch gives every CU a pair
 * and the image is narrow enough for a >= 24-row ring, else the tile kernels), 4 = row sweep (one workgroup per pair, row
 * rings in LDS; falls back to 3 where its geometry is unsupported), 3 = evaluate once + slab reduce.  Any other value: CD_ERR_INVALID_ARG. */
int cd_debug_set_loss_variant(int variant);
/* Row sweep: pixels per thread (1, 2, 4; 0 = default rule).  It fixes the rows per item, hence the plan stored in the
 * tile-windows blob: set it BEFORE cd_tile_windows_bytes / cd_tile_windows and keep it for the loss calls that use the blob. */
int cd_debug_set_loss_sweep(int pixels_per_thread);
/* pairs per source+gather launch pair of the evaluate-once gradient kernel (0 = default: slabs of one chunk sized
 * to stay resident in the Infinity Cache).  Changes cd_consistency_loss_workspace_bytes(); set it before the query. */
int cd_debug_set_loss_chunk(int pairs);

/* ------------------------------------------------------------------------------------
 * Device-resident frame-pair store: mini-batch gather  (reference: loaders/video_dataset.py:131-207
 * __getitem__ + default collate + utils/torch_helpers.py:10-23 to_device -- file reads and H2D copies per step;
 * here the dataset is uploaded once and a batch is ONE launch copying 12 H W floats per pair)
 * ---------------------------------------------------------------------------------- */
typedef struct cd_pair_store {
    const float* color;          /* [F][3][H][W] RGB in [0,1]                                   */
    const float* flows;          /* [P][2][2][H][W] (pair, direction fwd/bwd, (dx, dy)) pixels   */
    const void* masks;           /* [P][2][H][W] uint8 (mask_u8 = 1; nonzero = valid, as in the reference's PNGs) or fp32 {0,1} */
    const float* intrinsics;     /* [F][4] fx, fy, cx, cy                                        */
    const float* extrinsics;     /* [F][3][4] [R|t] camera-to-world                              */
    const int64_t* pair_frames;  /* [P][2] rows of the two frames of each pair                   */
    const int64_t* frame_ids;    /* [F] original frame numbers for metadata "indices", or NULL   */
    const float* mask_sums;      /* [P][2] cd_mask_sums per pair, or NULL                        */
    const uint8_t* plans;        /* [P][plan_bytes] cd_tile_windows per pair (B = 1 records), or NULL */
    int64_t plan_bytes;
    int32_t F, P, H, W, mask_u8, reserved;
} cd_pair_store;
typedef struct cd_pair_batch {   /* destinations, laid out like the reference's collated batch   */
    float* images;               /* [B][2][3][H][W]                                              */
    float* flow_fwd;             /* [B][2][H][W]   metadata["geometry_consistency"]["flows"][0]  */
    float* flow_bwd;             /* [B][2][H][W]                                  ["flows"][1]   */
    float* mask_fwd;             /* [B][1][H][W]   fp32 {0,1}                     ["masks"][0]   */
    float* mask_bwd;             /* [B][1][H][W]                                  ["masks"][1]   */
    float* intrinsics;           /* [B][2][4]                                                    */
    float* extrinsics;           /* [B][2][3][4]                                                 */
    int64_t* indices;            /* [B][2] or NULL                                               */
    float* mask_sums;            /* [B][2] or NULL                                               */
    uint8_t* plans;              /* [B][plan_bytes] or NULL                                      */
} cd_pair_batch;
/* pair_ids: B int64 rows of the store ON THE DEVICE (an epoch's index lists are uploaded once). */
int cd_gather_pairs(const cd_pair_store* store, const int64_t* pair_ids, int B, const cd_pair_batch* batch, void* stream);

/* utils/geometry.py:201-208 `sample`: bilinear, border padding, align_corners=False on an
 * align_corners=True style normalisation.  data [B,C,H,W], uv [B,2,H,W] px -> out [B,C,H,W]. */
int cd_sample_bilinear_border(const float* data, const float* uv, int B, int C, int H, int W,
                              float* out, void* stream);

/* Flow-consistency masks of B frame pairs -- the reference's mask_valid_correspondences (flow.py:199-228 ->
 * utils/consistency.py:32-67): mask_k = the flow of direction k stays inside the image, agrees with the opposite flow
 * warped by it within flow_thresh pixels, and the colours agree within color_thresh per channel (RMS).  The warp is
 * consistency.py's own sampler (grid in fp64 -> fp32, border padding, ix = u - 0.5).  flows (B,2,H,W), colours (B,C,H,W)
 * of frame 0 / frame 1 of each pair, masks out (B,1,H,W) as 0.0 / 1.0 -- the form the loss kernels consume.
 * Bit-identical to the reference's boolean masks (every rounding step is reproduced). */
int cd_flow_consistency_masks(const float* flow_fwd, const float* flow_bwd, const float* color0,
                              const float* color1, int C, double flow_thresh, double color_thresh, int B, int H,
                              int W, float* mask_fwd, float* mask_bwd, void* stream);

/* Depth-based warp of frames into each other (offline stages around the hot path: scale_calibration.py:84-120 ->
 * geometry.py:179-227 warping_field / warp_image):  uv_out[i] (2,H,W) = where pixel (x,y) of frame i, lifted with
 * depths[i] and moved through the two poses, lands in frame tgt_ids[i]; warped_out[i] (C,H,W) = images[tgt_ids[i]]
 * sampled there (the `sample` of cd_sample_bilinear_border).  N frames: images (N,C,H,W), depths (N,1,H,W),
 * intrinsics (N,4) = fx,fy,cx,cy, extrinsics (N,3,4) = [R|t], tgt_ids (N) int32 on the device.  Either output may be NULL. */
int cd_warp_image(const float* images, const float* depths, const float* intrinsics, const float* extrinsics,
                  const int* tgt_ids, int N, int C, int H, int W, float* uv_out, float* warped_out, void* stream);
/* Camera-space points depth * ray (geometry.py:130-139 depth_to_points) -> points_out (N,3,H,W) and/or their per-frame
 * sums over the pixels -> sums_out[N][3] (fp64, zeroed inside) -- the scene centres of geometry.py:142-176 calibrate_scale. */
int cd_depth_to_points(const float* depths, const float* intrinsics, int N, int H, int W, float* points_out,
                       double* sums_out, void* stream);

/* Per-frame median scale of the initial depth maps against COLMAP's dense depth (scale_calibration.py:253-278: the stage that
 * produces scales.csv and metadata_scaled.npz, the camera file the fine-tuning path reads).  inv_src, inv_cmp (N,H,W): inverse
 * depths of the depth model and of COLMAP (NaN where COLMAP has no value), same resolution.  Per frame: scales_out[i] =
 * np.median((inv_src / inv_cmp)[isfinite(inv_cmp)]) -- EXACT, bit for bit numpy's float32 result incl. the mean of the two middle
 * values for an even count (NaN when no pixel is valid or a selected ratio is NaN); n_valid_out[i] = number of finite COLMAP pixels
 * (the caller applies --dense_pixel_ratio); scaled_out (N,H,W) or NULL = inv_src / scale. */
int cd_frame_median_scales(const float* inv_src, const float* inv_cmp, int N, int H, int W, float* scales_out, int* n_valid_out,
                           float* scaled_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Depth CNN layers (reference: the un-vendored Mannequin-Challenge hourglass called at
 * monodepth/mannequin_challenge_model.py:60; architecture SURVEY.md appendix A.3).
 * All tensors NCHW fp32; a tensor argument is (pointer, total channels of the buffer it lives in,
 * channel offset) so layers read/write channel slices of concat buffers in place.
 * ---------------------------------------------------------------------------------- */

/* Convolution weights are consumed in a packed, zero-padded layout.  transposed = 0 packs the
 * forward filter of w[Cout][Cin][ks][ks]; transposed = 1 packs the input-gradient filter
 * (flipped, in/out swapped) of the same w, so dgrad is cd_conv2d_fwd on the output gradient. */
size_t cd_conv2d_packed_weight_floats(int Cout, int Cin, int ks, int transposed);
int cd_conv2d_pack_weights(const float* w, int Cout, int Cin, int ks, int transposed, float* packed,
                           void* stream);

/* Pack many filters in ONE launch: table_dev = device array of n cd_pack_desc (48 bytes each).
 * Several descriptors may target one `packed` buffer: a fused convolution whose output channels (forward
 * form) or input channels (transposed form) concatenate several weights -- OC/IC are the logical channel
 * counts of the fused conv (they size the layout: cd_conv2d_packed_weight_floats(OC, IC, ks, 0)), oc_off/ic_off
 * this source's position.  Padding elements are never written: zero the buffer once. */
typedef struct cd_pack_desc {
    const float* w;   /* [Cout][Cin][ks][ks] */
    float* packed;
    int Cout, Cin, ks, transposed;
    int OC, IC, oc_off, ic_off;
} cd_pack_desc;
int cd_conv2d_pack_weights_table(const void* table_dev, int n, void* stream);

/* y[:, y_coff : y_coff+Cout] = conv2d(act(x[:, x_coff : x_coff+Cin]), w) + bias, stride 1, zero
 * padding (ks-1)/2, ks in {1,3,5,7,11}, at fp32 accuracy: on the BF16 matrix cores from exactly split operands or on the fp32
 * matrix instruction, by cd_set_conv_arith (below).
 *   act(v) = relu?(v * in_scale[c] + in_shift[c])   when in_scale/in_shift are given (the producer's
 *            BatchNorm-apply [+ReLU] fused into the load), relu only when in_relu and no scale, else v;
 *   stats (optional, [CD_BN_STAT_SLOTS][y_ctot][2] doubles, caller-zeroed): per-channel sum and sum of squares of the
 *            raw output are ADDED -- the batch statistics of the following train-mode BatchNorm;
 *   accumulate != 0: y += result instead of y = result (gradient fan-in of the dgrad convolutions). */
int cd_conv2d_fwd(const float* x, int x_ctot, int x_coff, int Cin, const float* packed_w,
                  const float* bias, const float* in_scale, const float* in_shift, int in_relu,
                  float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate,
                  int N, int H, int W, int ks, void* stream);
/* The same with the launch shape chosen by the caller: tile_rows in {4, 8, 16, 32} (32, ABI 9: a shape of the split-bf16 k x k kernels
 * with 32 output channels per column tile only -- 8 row tiles AND two channel chunks per barrier round; CD_ERR_UNSUPPORTED elsewhere) and
 * co_tiles in {1, 2, 4, 8, 16} 16-wide output-channel slices per workgroup (clamped to cd_conv2d_packed_co_tiles; 8 and
 * 16 exist for 1x1 filters only, with at most 8 / 4 tile rows: one workgroup then computes up to 256 channels);
 * 0 = built-in heuristic.  The result does not depend on the choice (the order of accumulation is fixed), only the
 * speed does: a caller with many launches of few shapes times the candidates once (the hourglass engine does). */
int cd_conv2d_fwd_cfg(const float* x, int x_ctot, int x_coff, int Cin, const float* packed_w,
                      const float* bias, const float* in_scale, const float* in_shift, int in_relu,
                      float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate,
                      int N, int H, int W, int ks, int tile_rows, int co_tiles, void* stream);
/* Grouped convolution (nn.Conv2d(groups=G), ResNeXt's 32 x 8d 3x3 of the MiDaS v2 encoder, reference call site
 * monodepth/midas_v2_model.py:61): group g is a dense convolution of input channels [x_coff + g*cin_g, +cin_g) into output channels
 * [y_coff + g*cout_g, +cout_g) with its own packed filter at packed_w + g*packed_group_stride (floats; each packed by
 * cd_conv2d_pack_weights[_table] for (cout_g, cin_g, ks), stride a multiple of 4 and >= cd_conv2d_packed_weight_floats) -- ONE launch
 * for all groups in the split arithmetic modes (k >= 3, cin_g >= 8), the dense kernels group by group otherwise.  The input gradient
 * is the same call on the transposed packs with the roles of x / y swapped.  bias: [groups*cout_g] or NULL. */
int cd_conv2d_fwd_grouped(const float* x, int x_ctot, int x_coff, int cin_g, const float* packed_w, size_t packed_group_stride,
                          const float* bias, float* y, int y_ctot, int y_coff, int cout_g, int groups, int accumulate, int N, int H, int W,
                          int ks, void* stream);
/* SEVERAL convolutions in ONE dispatch: the n <= 4 members share N, H, W and Cout, have filter sizes in {3, 5, 7, 11} and
 * >= 8 input channels each -- the three k x k branches of an inception (monodepth/mannequin_challenge), forward
 * (inputs: the mid activations with their BatchNorm (scale, shift), outputs: the branch slices of the concat buffer) or input
 * gradient (transposed packs, roles swapped).  Fields as the arguments of cd_conv2d_fwd_cfg; tile_rows / co_tiles: the launch shape
 * hints, shared by the members.  Every workgroup does what it does in the member's own cd_conv2d_fwd_cfg launch with the same hints:
 * identical bits.  Order the members largest filter first.  CD_ERR_UNSUPPORTED (nothing launched): arithmetic mode 0, or a member the
 * split kernels do not take -- launch the members one by one. */
typedef struct cd_conv_desc {
    const float* x; const float* packed_w; const float* bias; const float* in_scale; const float* in_shift; float* y; double* stats;
    int x_ctot, x_coff, Cin, in_relu, y_ctot, y_coff, Cout, accumulate, N, H, W, ks;
} cd_conv_desc;
int cd_conv2d_fwd_multi(const cd_conv_desc* descs, int n, int tile_rows, int co_tiles, void* stream);
/* dw [groups*cout_g][cin_g][ks][ks] (+)= the weight gradient of the grouped convolution; workspace: groups * workspace_group_stride
 * floats, workspace_group_stride >= cd_conv2d_wgrad_workspace_floats(cout_g, cin_g, ks). */
int cd_conv2d_wgrad_grouped(const float* x, int x_ctot, int x_coff, int cin_g, const float* dy, int dy_ctot, int dy_coff, int cout_g,
                            int groups, float* dw, int accumulate, float* workspace, size_t workspace_group_stride, int N, int H, int W,
                            int ks, void* stream);

/* Arithmetic of the convolutions (forward, input gradient and weight gradient), process-wide; start-up value from
 * CD_AMD_CONV_ARITH = "split" (2, default) | "split3" (1) | "fp32" (0):
 *   1: k = 3, 5, 7, 11 with >= 8 input channels: every fp32 operand is split exactly into three bf16 terms and the six
 *     significant cross products run on the BF16 matrix cores with fp32 accumulation -- as close to fp64 as the fp32
 *     instruction (profiles/mfma_split_exp_r02.txt), 1.5-1.7x faster; not bitwise the fmaf chain, +-inf inputs give NaN;
 *   2: 1, plus the 1x1 forward / input gradient on images of >= 4096 row tiles (conv1x1_split.hip: filter slice resident in
 *     LDS, activations split in registers; 1.3-2x faster per launch) and the 1x1 weight gradient with >= 96 output channels
 *     (wgrad1x1_split.hip: both operands straight from global memory into matrix fragments; 1.4-1.5x);
 *   0: the fp32 matrix instruction everywhere (bitwise a k-ordered fmaf chain).
 * The packed filter holds every layout, so the mode may change between launches without re-packing.  The weight gradient of
 * the filters follows the same switch (wgrad_split.hip, wgrad1x1_split.hip); its packed layout (cd_conv2d_wgrad_plan) differs
 * between the modes, so a plan -- and a cd_hourglass handle -- belongs to the mode it was made under: cd_hourglass_forward /
 * _backward return CD_ERR_INVALID_ARG after a mode change (re-create the handle), and a captured HIP graph keeps replaying the
 * kernels of the mode it was captured under.
 * Semantics that differ from the fp32 instruction in the split modes (1, 2): an operand that is +-inf or in the top binade
 * (|x| >= 2^127) yields NaN instead of +-inf (the split terms overflow), and results are not bitwise those of mode 0; finite
 * inputs give fp32-accurate results.  cd_conv2d_wgrad returns CD_ERR_UNSUPPORTED (it does not silently change arithmetic) when
 * a 1x1 split launch would need offsets beyond 32 bits (N * ctot * H * W >= 2^30 elements): use mode 0 for such buffers. */
int cd_set_conv_arith(int mode);
int cd_get_conv_arith(void);
/* The upper bound of co_tiles for (Cout, ks) under the current arithmetic mode (split modes, k >= 3: tile_rows 4 / 8 / 16 selects
 * 4 / 8 row tiles per workgroup, co_tiles 1 / 2 one or two 32-channel column tiles). */
int cd_conv2d_packed_co_tiles(int Cout, int ks);

/* Test hook: force the output-tile height (4, 8, 16; 0 = automatic) of cd_conv2d_fwd so every
 * template instantiation can be parity-tested at small sizes. */
int cd_debug_force_conv_tile_rows(int ty);
/* Force the number of 16-wide output-channel tiles a conv workgroup computes (1, 2, 4, 8, 16; ignored when larger than
 * cd_conv2d_packed_co_tiles; 0 = heuristic), and switch the register-prefetch software pipeline (default on). */
int cd_debug_force_conv_co_tiles(int co_tiles);
int cd_debug_set_conv_pipeline(int on);
/* Measurement hook for cd_conv2d_wgrad: bit 0 skips the store of the partial sums, bit 1 the matrix instructions
 * (the result is then wrong); bit 2 switches the wide 1x1 plan off, bit 3 the few-input-channel (stem) kernel (correct
 * results, for A/B timing and tests);
 * 0 restores normal operation. */
int cd_debug_set_wgrad_mode(int bits);

/* Weight gradient dw[Cout][Cin][ks][ks] (=, or += when accumulate) of the same convolution:
 * sum over n,y,x of dy[n][co][y][x] * act(x)[n][ci][y+ky-P][x+kx-P]  (act as in cd_conv2d_fwd).
 * workspace: cd_conv2d_wgrad_workspace_floats(Cout, Cin, ks) floats, no initialisation needed: every workgroup stores its
 * partial sums into its own slice and the unpack step adds the slices in a fixed order (fp64 accumulator) -- no atomics,
 * the result is bit-reproducible.  (Bit 1 of `accumulate`, "workspace already zeroed" in ABI 3, is accepted and ignored.) */
size_t cd_conv2d_wgrad_workspace_floats(int Cout, int Cin, int ks);
int cd_conv2d_wgrad(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale,
                    const float* in_shift, int in_relu, const float* dy, int dy_ctot, int dy_coff,
                    int Cout, float* dw, int accumulate, float* workspace, int N, int H, int W, int ks,
                    void* stream);
/* Deferred form: with bit 2 of `accumulate` (value 4) cd_conv2d_wgrad leaves the result in `workspace` in the packed
 * layout [split][co group][ci group][tap][cob][cib] that cd_conv2d_wgrad_plan reports for the same arguments (dw may be
 * NULL; `splits` slices of split_stride = (co groups * ci groups * ks*ks * cob * cib) floats), and ONE
 * cd_conv2d_wgrad_unpack_table launch later writes any number of gradients -- one descriptor per destination tensor
 * dw[rows][Cin][ks][ks] taking the output-channel rows [row0, row0+rows) of a packed buffer (a fused convolution's rows
 * belong to several nn.Conv2d weights).  Device-resident table of cd_unpack_desc (56 bytes each). */
typedef struct cd_unpack_desc {
    const float* packed;
    float* dw;
    int Cin, ks, cob, cib, ci_groups, row0, rows, accumulate, splits, split_stride;
} cd_unpack_desc;
int cd_conv2d_wgrad_plan(int Cout, int Cin, int ks, int N, int H, int W, int* cob, int* cib, int* splits);
int cd_conv2d_wgrad_unpack_table(const void* table_dev, int n, void* stream);
/* MANY weight gradients in ONE launch (the hourglass' 82 k x k gradients of a step: the backward only needs them before the
 * optimiser, their operands -- the forward activations and the gradient buffers of every inception -- stay valid until the end of
 * the backward pass, and the deep levels' gradients on their own launch a few dozen short workgroups on an idle chip).
 *   cd_conv2d_wgrad_desc   fills the derived fields of ONE host descriptor whose first 16 fields (x .. ks: the arguments of
 *                          cd_conv2d_wgrad in deferred form, `workspace` receiving the packed partial sums exactly as
 *                          cd_conv2d_wgrad(accumulate = 4) would leave them) the caller has set.  klass = the kernel class
 *                          (0 .. 4) or -1 when this gradient is not one the table kernels compute (1x1, the RGB stem, arithmetic
 *                          mode 0): launch it with cd_conv2d_wgrad.  blocks = its workgroups.
 *   cd_conv2d_wgrad_table  launches the n <= 64 descriptors of ONE class from a DEVICE-resident table; the caller has set
 *                          block_end of entry i to blocks[0] + .. + blocks[i] and passes the total.  Order the table heaviest
 *                          first (workgroups are dispatched in table order).  Results are bit-identical to one cd_conv2d_wgrad
 *                          per descriptor; finish with cd_conv2d_wgrad_unpack_table as usual. */
typedef struct cd_wgrad_desc {
    const float* x; const float* in_scale; const float* in_shift; const float* dy; float* workspace;
    int x_ctot, x_coff, Cin, in_relu, dy_ctot, dy_coff, Cout, N, H, W, ks;
    int klass, splits, cigs, zpg, cogs, tiles_x, tiles_y, blocks, block_end, pad[2];
} cd_wgrad_desc;
int cd_conv2d_wgrad_desc(cd_wgrad_desc* desc);
int cd_conv2d_wgrad_table(const void* table_dev, int n, int klass, int total_blocks, void* stream);

/* ------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (reference: nn.DataParallel, monodepth/midas_v2_model.py:41-43 and the batch scaling of
 * depth_fine_tuning.py:155-159): buf <- sum over ranks / world, in place, ONE ncclAllReduce(sum, fp32) on the caller's RCCL
 * communicator (ncclComm_t passed as void*) + one scale launch, both on `stream`.  librccl is resolved at run time
 * (the copy already loaded into the process, else dlopen): cd_rccl_available() says whether it was found.
 * world = 1 with a communicator still runs the collective (smoke tests); world = 1 without one is a no-op.
 * ---------------------------------------------------------------------------------- */
int cd_rccl_available(void);
int cd_allreduce_mean_f32(float* buf, size_t n, void* nccl_comm, int world, void* stream);

/* ------------------------------------------------------------------------------------
 * The hourglass as ONE object: plan, buffers, forward and explicit backward behind a handle, for hosts without Python
 * (reference call site: monodepth/mannequin_challenge_model.py:52-69 netG.forward + autograd, depth_fine_tuning.py:282).
 * A whole fine-tuning step through this header:
 *     cd_hourglass_zero_grad; cd_hourglass_forward(images -> pred);
 *     cd_consistency_loss_fwd_bwd(depth = pred, CD_DEPTH_EXP -> grad);  cd_hourglass_backward(grad);
 *     [all-reduce cd_hourglass_grads()];  cd_adam_step_flat(cd_hourglass_params(), cd_hourglass_grads(), m, v, ...).
 * Parameters: ONE flat fp32 buffer, the tensors of the PyTorch module's named_parameters() in order (seq.0.weight,
 * seq.0.bias, seq.1.weight, ..., uncertainty_layer.0.*, pred_layer.*), each starting on a 64-float boundary -- the layout
 * consistent_depth_amd.optimizer.FlatAdam uses; cd_hourglass_param_info reports offset and shape of tensor `index`.
 * BatchNorm running statistics: bn_flat = for every BatchNorm2d in module order [running_mean(C), running_var(C)].
 * cd_hourglass_create allocates the engine's device memory (activations, gradients, packed filters, workspaces:
 * ~10 GB for 8 x 384 x 224); all other calls only enqueue work on `stream`.  One input shape per handle.
 * ---------------------------------------------------------------------------------- */
typedef struct cd_hourglass cd_hourglass;
int cd_hourglass_create(int N, int H, int W, cd_hourglass** out);   /* H, W multiples of 16 */
int cd_hourglass_destroy(cd_hourglass* h);
size_t cd_hourglass_param_floats(const cd_hourglass* h);
size_t cd_hourglass_bn_floats(const cd_hourglass* h);
int cd_hourglass_param_count(const cd_hourglass* h);
int cd_hourglass_param_info(const cd_hourglass* h, int index, size_t* offset, int* shape4);
float* cd_hourglass_params(cd_hourglass* h);   /* device pointers owned by the handle */
float* cd_hourglass_grads(cd_hourglass* h);
/* params_flat / bn_flat: host or device memory (hipMemcpyDefault); bn_flat may be NULL (keep / skip the statistics). */
int cd_hourglass_load_state(cd_hourglass* h, const float* params_flat, const float* bn_flat, void* stream);
int cd_hourglass_save_state(cd_hourglass* h, float* params_flat, float* bn_flat, void* stream);
int cd_hourglass_zero_grad(cd_hourglass* h, void* stream);
/* n floats between any two host / device buffers, stream ordered (reads the handle-owned buffers from a binding). */
int cd_copy_f32(const float* src, float* dst, size_t n, void* stream);
/* images [N][3][H][W] RGB in [0,1] -> pred [N][1][H][W] (log depth).  training != 0: batch statistics, running statistics
 * updated (nn.BatchNorm2d semantics, momentum 0.1); 0: running statistics. */
int cd_hourglass_forward(cd_hourglass* h, const float* images, float* pred, int training, void* stream);
/* dpred = d loss / d pred of the LAST training forward; every parameter gradient is ADDED into cd_hourglass_grads():
 * PRECONDITION cd_hourglass_zero_grad (or an equivalent clear of cd_hourglass_grads()) once per step before the forward --
 * a host that skips it gets the sum of this and the previous steps' gradients (deliberate: gradients of other loss terms,
 * e.g. the parameter regulariser, can be placed there first). */
int cd_hourglass_backward(cd_hourglass* h, const float* dpred, void* stream);

/* BatchNorm2d in training mode, forward.  stats[CD_BN_STAT_SLOTS][ctot][2] = per-channel (sum, sum of squares) of the raw
 * tensor over N*H*W (what cd_conv2d_fwd accumulates).  In place: x <- (x - mean) * rsqrt(var + eps)
 * (x_hat, PRE-ReLU: consumers apply relu / the affine part while loading); writes
 * mean_invstd[ctot][2] (mean, 1/std) for the backward and updates running_mean/var[C] (momentum, unbiased
 * variance) like nn.BatchNorm2d when they are given. */
int cd_bn_normalize(float* x, int ctot, int coff, int C, const double* stats, float eps,
                    float* running_mean, float* running_var, float momentum, float* mean_invstd,
                    int N, int H, int W, void* stream);

/* The same BatchNorm WITHOUT a pass over the activation: from stats compute per channel (buffer-indexed arrays of
 * ctot floats) scale = gamma*invstd and shift = beta - gamma*mean*invstd, so consumers evaluate
 * relu(raw*scale + shift) while loading the RAW conv output; saves mean_invstd, updates running stats.
 * count = N*H*W.  gamma/beta/running_* are slice-local ([C]) and optional. */
int cd_bn_finalize(const double* stats, int ctot, int coff, int C, double count, float eps,
                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float momentum, float* mean_invstd, float* scale, float* shift, void* stream);

/* Backward of relu(gamma * x_hat + beta) + train-mode BatchNorm in one call: dA (gradient w.r.t. the
 * activated output) is replaced IN PLACE by the gradient w.r.t. the raw (pre-BN) tensor.  gamma/beta
 * NULL = BatchNorm2d(affine=False).  dgamma/dbeta[C] (optional): the affine gradients are ADDED to them (+=, torch's
 * p.grad convention: the caller zeroes its gradients once per step -- FlatAdam.zero_grad / cd_hourglass_zero_grad) unless
 * CD_BN_BWD_OVERWRITE_AFFINE is set in `flags`, which assigns them.
 * scale/shift (buffer-indexed, from cd_bn_finalize) given: `xhat` holds the RAW conv output and the ReLU mask is
 * evaluated on exactly fma(raw, scale, shift), the expression the consumers applied on load; NULL: `xhat` is the
 * normalised tensor of cd_bn_normalize.  sums: scratch of 2*C doubles, zeroed inside unless CD_BN_BWD_SUMS_PREZEROED.
 * flags: 0 or an OR of the two bits below (bit 0 is the former `sums_prezeroed` argument: 0 / 1 keep their meaning). */
#define CD_BN_BWD_SUMS_PREZEROED 1
#define CD_BN_BWD_OVERWRITE_AFFINE 2
int cd_bn_relu_bwd(float* dA, int d_ctot, int d_coff, const float* xhat, int x_ctot, int x_coff, int C,
                   const float* gamma, const float* beta, const float* mean_invstd, const float* scale,
                   const float* shift, double* sums, int flags, float* dgamma, float* dbeta,
                   int N, int H, int W, void* stream);

/* AvgPool2d(2) of act(x) and its adjoint (dx = gradient w.r.t. the ACTIVATED input, (+)= when accumulate). */
int cd_avgpool2_fwd(const float* x, int x_ctot, int x_coff, const float* in_scale, const float* in_shift,
                    int in_relu, float* y, int y_ctot, int y_coff, int C, int N, int H, int W, void* stream);
int cd_avgpool2_bwd(const float* dy, int dy_ctot, int dy_coff, float* dx, int dx_ctot, int dx_coff, int C,
                    int N, int H, int W, int accumulate, void* stream);

/* out = UpsamplingBilinear2d(2)(act(lo)) [+ act(hi)]  (align_corners=True; hi may be NULL), lo is h x w;
 * cd_upsample2x_bwd is the adjoint of the bilinear part as a gather (no atomics): dlo (+)= U^T dout. */
int cd_upsample2x_add_fwd(const float* lo, int lo_ctot, int lo_coff, const float* lo_scale,
                          const float* lo_shift, int lo_relu, const float* hi, int hi_ctot, int hi_coff,
                          const float* hi_scale, const float* hi_shift, int hi_relu, float* out, int o_ctot,
                          int o_coff, int C, int N, int h, int w, void* stream);
int cd_upsample2x_bwd(const float* dout, int d_ctot, int d_coff, float* dlo, int l_ctot, int l_coff, int C,
                      int N, int h, int w, int accumulate, void* stream);
/* out = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=False) (half-pixel centres, borders clamped) and its adjoint as
 * a gather -- the last up-sampling of the MiDaS output head (reference call site: monodepth/midas_v2_model.py:61-67 -> MidasNet.forward;
 * the un-vendored `midas_net.py::Interpolate`).  (ABI v8) */
int cd_upsample2x_halfpixel_fwd(const float* lo, int lo_ctot, int lo_coff, float* out, int o_ctot, int o_coff, int C,
                                int N, int h, int w, void* stream);
int cd_upsample2x_halfpixel_bwd(const float* dout, int d_ctot, int d_coff, float* dlo, int l_ctot, int l_coff, int C,
                                int N, int h, int w, int accumulate, void* stream);

/* ---- blocks of an autograd-driven backbone (MiDaS v2 / ResNeXt-101, BASELINE configs[4]; reference call site
 * monodepth/midas_v2_model.py:58-67 -> the un-vendored MidasNet; torchvision Bottleneck: conv-bn-relu x2, conv-bn, + identity, relu).  (ABI v8)
 * cd_bn_block_fwd: y = act(BatchNorm2d_train(x) [+ res]), act = ReLU (relu != 0) or identity; batch statistics in fp64 across blocks,
 *   running statistics updated like nn.BatchNorm2d (momentum, unbiased variance); gamma / beta NULL = not affine; res NULL = no residual.
 *   Saves mean_invstd [C][2] for the backward; scale / shift [C] and stats [CD_BN_STAT_SLOTS][C][2] doubles are scratch (zeroed by the call).
 * cd_bn_block_bwd: from dy = d loss / d y:  dv = relu ? (y > 0 ? dy : 0) : dy;  dres = dv (NULL: none);
 *   dx = gamma invstd (dv - mean(dv) - xhat mean(dv xhat));  dgamma = sum dv xhat, dbeta = sum dv (assigned; NULL: none).  sums [C][2]
 *   doubles scratch.  x, y: the forward's input and output. */
int cd_bn_block_fwd(const float* x, const float* gamma, const float* beta, const float* res, int relu, float* running_mean,
                    float* running_var, float momentum, float eps, float* y, float* mean_invstd, float* scale, float* shift,
                    double* stats, int C, int N, int H, int W, void* stream);
int cd_bn_block_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* mean_invstd, int relu,
                    float* dx, float* dres, float* dgamma, float* dbeta, double* sums, int C, int N, int H, int W, void* stream);
/* y = max(a, 0) (op 0) | a + b (op 1) | b > 0 ? a : 0 (op 2: ReLU backward, a = dy, b = the ReLU's input or output) |
 * a * b[0] (op 3, ABI 9: scale by a DEVICE scalar -- the upstream gradient of the loss's autograd node); 16-byte aligned. */
int cd_eltwise(const float* a, const float* b, float* y, size_t n, int op, void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1): y (N, C, (H-1)/2+1, (W-1)/2+1), argmax = position inside the window (0..8; ATen's first-maximum
 * rule); the backward is a gather over the <= 4 windows that contain an input pixel. */
int cd_maxpool3s2_fwd(const float* x, float* y, unsigned char* argmax, int C, int N, int H, int W, void* stream);
int cd_maxpool3s2_bwd(const float* dy, const unsigned char* argmax, float* dx, int C, int N, int H, int W, void* stream);

/* dst[:, d_coff:+C] (+)= src[:, s_coff:+C]  -- gradient fan-in of a tensor with several consumers. */
int cd_add_slice(const float* src, int s_ctot, int s_coff, float* dst, int d_ctot, int d_coff, int C, int N,
                 int H, int W, int accumulate, void* stream);
/* Test / measurement hook for the streaming layers above (avgpool2, upsample2x, add_slice): bit 0 selects the scalar kernels of rounds
 * 1-5 instead of the 16-byte / LDS-band kernels of round 6.  Both give the same bits (tests/test_layers_gpu.py); 0 = normal. */
int cd_debug_set_layers_mode(int bits);
/* ABI 9.  Host-side chores of a step as ONE launch each (they were framework launches inside the captured step):
 * cd_copy_segments: table[i] = {src, dst, n}: n floats copied per entry (the biases of the four branch-entry 1x1 convolutions of every
 * inception into the fused convolution's bias vector: 22 torch.cat per forward before); the table lives in device memory.
 * cd_counters_add: *table[i] += delta for n device pointers to int64 scalars (nn.BatchNorm2d.num_batches_tracked of every layer,
 * advanced by every train-mode forward: /root/reference/depth_fine_tuning.py:241,327-328 keeps train mode during validation too).
 * cd_zero_bytes: p[0 .. bytes) = 0 by a kernel of this library on the caller's stream (gradient / statistics arenas; p 16-byte aligned,
 * bytes a multiple of 4). */
typedef struct cd_copy_seg {
    const float* src;
    float* dst;
    long long n;
} cd_copy_seg;
int cd_copy_segments(const cd_copy_seg* table_dev, int n, void* stream);
int cd_counters_add(long long* const* table_dev, int n, long long delta, void* stream);
int cd_zero_bytes(void* p, size_t bytes, void* stream);
/* out[c] (+)= sum over n,y,x of src[n][coff+c]  -- bias gradient of a conv not followed by BatchNorm. */
int cd_channel_sum(const float* src, int ctot, int coff, int C, int N, int H, int W, float* out,
                   int accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimiser (reference: optimizer/__init__.py:16-17 -> torch.optim.Adam,
 * depth_fine_tuning.py:231-236,283; betas (0.9,0.999), eps 1e-8, no weight decay)
 * ---------------------------------------------------------------------------------- */

/* One Adam step over a flat parameter buffer.  `step` is 1-based.  grad_scale multiplies
 * every gradient first (1/world_size after a sum all-reduce; 1.0 otherwise). */
int cd_adam_step_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                      size_t n, float lr, float beta1, float beta2, float eps, int step,
                      float grad_scale, void* stream);

/* The same step with the reference's NaN guard (depth_fine_tuning.py:278-280: a NaN loss skips
 * backward() and step()) and the step counter kept on the device, so the training loop needs
 * no host synchronisation: if loss[0] is NaN nothing is updated and *step_counter (device int,
 * number of steps taken so far) is not advanced; otherwise the step uses *step_counter + 1 and
 * the counter is advanced.  loss may be NULL (no guard). */
int cd_adam_step_flat_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              size_t n, float lr, float beta1, float beta2, float eps,
                              int* step_counter, const float* loss, float grad_scale, void* stream);

/* sum_i |p_i - p0_i|  (loss/parameter_loss.py:14-18, lambda applied by the caller);
 * out[1]; workspace of cd_l1_distance_workspace_bytes(n). */
size_t cd_l1_distance_workspace_bytes(size_t n);
int cd_l1_distance(const float* p, const float* p0, size_t n, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONSISTENT_DEPTH_AMD_H */
