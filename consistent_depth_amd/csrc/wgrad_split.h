// Interface between conv_wgrad.hip (layout, dispatch, unpack) and wgrad_split.hip (the split-bf16 weight gradient).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "wgrad_stage_map.h"   // wgrad_split_tile_rows, WsCfg, the staging map (host-testable)

namespace cd {

// resident blocks per CU (k = 11: one 8-wave block with a 109 KB tile; else two 4-wave blocks)
__host__ __device__ constexpr int wgrad_split_blocks_per_cu(int ks) { return ks == 11 ? 1 : 2; }

// 16-channel OUTPUT groups per block: k = 3 with >= 32 output channels takes TWO (a 32 x 16 channel block).  With one, the 9 taps
// are shared by three of the four waves (3 accumulator tiles each, the fourth wave idle) and the input tile with its halo -- 2/3 of
// the staged bytes -- is staged once per 16 output channels; with two, every wave has a sub-tile's 4-5 taps and the input tile
// serves twice the multiply-adds (profiles/wgrad3x3_r03.txt).  The packed result keeps its 16 x 16 tiles: same unpack, same bits.
__host__ __device__ constexpr int wgrad_split_cot(int ks, int Cout) { return (ks == 3 && Cout >= 32) ? 2 : 1; }

// partial sums into packed[split][co group][ci group][tap][16][16]; `splits` blocks per channel-group pair
int launch_wgrad_split(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift, int in_relu,
                       const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H, int W, int ks, int splits,
                       hipStream_t s, int groups = 1, size_t ws_group_stride = 0, int cot = 1);   // groups > 1: Cin / Cout per group, slices g * Cin / g * Cout;
                                                                                                  // cot: 16-channel output groups per block (the layout's choice)

// ---- many weight gradients in one launch (cd_conv2d_wgrad_desc / cd_conv2d_wgrad_table; = cd_wgrad_desc of the public header)
struct WgradDesc {
    const float* x; const float* in_scale; const float* in_shift; const float* dy; float* workspace;
    int x_ctot, x_coff, Cin, in_relu, dy_ctot, dy_coff, Cout, N, H, W, ks;
    int klass, splits, cigs, zpg, cogs, tiles_x, tiles_y, blocks, block_end, pad[2];
};
static_assert(sizeof(WgradDesc) == 128, "cd_wgrad_desc layout");
int wgrad_split_class(int ks, int cot);
void wgrad_split_desc_geometry(WgradDesc* d, int splits, int cot);    // fills klass .. blocks from the layout's choice of (splits, cot)
int launch_wgrad_split_table(const void* table_dev, int n, int klass, int total_blocks, hipStream_t s);

// ---- 1x1 weight gradient (wgrad1x1_split.hip): a wave owns a 64 x 128 (co x ci) patch of dW
constexpr int WGRAD1X1_COB = 64, WGRAD1X1_CIB = 128;
// usable with >= 96 output channels, when H * W is a multiple of 16 (whole 16-pixel steps inside an image), the tensors have < 2^30 elements per image set and
// the image is large enough to give every wave several steps
// CD_AMD_CONV1X1_KC=0: wide 1x1 filters (>= 512 channels) back on the staged fp32 kernels (A/B switch of round 6, read once)
inline bool wide_1x1_enabled() {
    static const bool on = [] { const char* e = getenv("CD_AMD_CONV1X1_KC"); return !(e && e[0] == '0'); }();
    return on;
}
inline bool wgrad1x1_split_ok(int Cout, int Cin, int N, int H, int W, int x_ctot, int dy_ctot) {
    const long long hw = (long long)H * W;
    // few pixels: the staged fp32 kernel, EXCEPT the wide filters of MiDaS' encoder (round 6: 1024 x 1024 on 24 x 24 x 16 pixels is 128
    // patches x 16 pixel splits = 512 workgroups of 36 K-steps; the hourglass has no such filter and keeps its bits)
    const bool enough_pixels = (long long)N * hw >= 20000 || (wide_1x1_enabled() && Cout >= 512 && Cin >= 512 && (long long)N * hw >= 2048);
    return Cout >= 96 && Cin >= 32 &&   // (fewer output channels: a single half-empty 64 x 128 patch -- the staged kernel is faster)
           hw % 16 == 0 && enough_pixels && hw * (x_ctot > dy_ctot ? x_ctot : dy_ctot) < (1LL << 30);
}
void wgrad1x1_split_shape(int Cout, int Cin, int* cogs, int* cigs, int* pg, int* sub, int* groups);
int wgrad1x1_split_blocks(int Cout, int Cin, long long steps);   // grid.x; slices of the packed result = blocks * sub
int launch_wgrad1x1_split(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift, int in_relu,
                          const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H, int W, int blocks_x,
                          hipStream_t s);

}  // namespace cd
