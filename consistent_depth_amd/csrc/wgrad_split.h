// Interface between conv_wgrad.hip (layout, dispatch, unpack) and wgrad_split.hip (the split-bf16 weight gradient).
#pragma once
#include <hip/hip_runtime.h>

namespace cd {

// image-tile rows per work item of the split-bf16 weight-gradient kernel (16 x 16 channels per block, 32-pixel rows)
__host__ __device__ constexpr int wgrad_split_tile_rows(int ks) { return ks == 11 ? 8 : 6; }
// resident blocks per CU (k = 11: one 8-wave block with a 109 KB tile; else two 4-wave blocks)
__host__ __device__ constexpr int wgrad_split_blocks_per_cu(int ks) { return ks == 11 ? 1 : 2; }

// partial sums into packed[split][co group][ci group][tap][16][16]; `splits` blocks per channel-group pair
int launch_wgrad_split(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift, int in_relu,
                       const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H, int W, int ks, int splits,
                       hipStream_t s);

}  // namespace cd
