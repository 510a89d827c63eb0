// Interface between conv_mfma.hip (dispatch, packing tables) and conv_split.hip (the split-bf16 convolution).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace cd {

// filter sizes the split-bf16 kernel covers (k = 1, 3 stay on the fp32 instruction: memory / launch bound shapes)
__host__ __device__ constexpr bool split_supported(int ks) { return ks == 3 || ks == 5 || ks == 7 || ks == 11; }

// One source of a packed filter (cd_pack_desc of the header).
struct PackDesc {
    const float* w; float* packed;
    int Cout, Cin, ks, transposed;   // the source tensor w[Cout][Cin][ks][ks] and which form to pack
    int OC, IC, oc_off, ic_off;      // logical channels of the (fused) packed conv and this source's offset in it
};
static_assert(sizeof(PackDesc) == 48, "cd_pack_desc layout");

// 16-wide output-channel tiles per packed group of the fp32 layout for (k, Cout)
__host__ __device__ inline int pick_co_tiles(int ks, int cout) {
    const int need = (cout + 15) / 16;
    int cap = (ks >= 11) ? 1 : 4;  // LDS: 121 taps x CI x COBP floats must leave room for >= 2 blocks per CU
    int t = need < cap ? need : cap;
    if (t == 3) t = 4;
    return t < 1 ? 1 : t;
}
__host__ __device__ constexpr int co_stride_padded(int cob) { return (cob % 32 == 0) ? cob + 16 : cob; }

// floats of the fp32 packed layout [co group][ci chunk][tap][ci in chunk][COBP] of the logical convolution IC -> OC
__host__ __device__ inline size_t fp32_packed_floats(int OC, int IC, int ks) {
    const int cot = pick_co_tiles(ks, OC), cob = cot * 16, cobp = co_stride_padded(cob);
    const int ci_chunk = ks >= 7 ? 4 : (ks == 1 ? 32 : 8);
    const int groups = (OC + cob - 1) / cob, chunks = (IC + ci_chunk - 1) / ci_chunk;
    return (size_t)groups * chunks * ks * ks * ci_chunk * cobp;
}

// 1x1 filters (conv1x1_split.hip; forward / input gradient only): more than 16 output and at least 32 input channels, and the
// block's filter slice (64 output channels x all input channels x 6 bytes) must fit the LDS
// (IC <= 384: the filter slice stays in LDS, conv1x1_split_kernel; up to 2048 since round 6: the chunked kernel conv1x1_split_kc_kernel)
__host__ __device__ constexpr bool split_1x1_supported(int ks, int OC, int IC) { return ks == 1 && OC > 16 && IC >= 32 && IC <= 2048; }
__host__ __device__ constexpr bool split_1x1_resident(int IC) { return IC <= 384; }
size_t split_1x1_packed_floats(int OC, int IC);
int launch_pack_1x1_table(const void* table_dev, int n, hipStream_t s);
int launch_pack_1x1(const float* w, int Cout, int Cin, int transposed, float* packed_split, hipStream_t s);
int launch_conv1x1_split(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                         const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                         int H, int W, hipStream_t s);
bool conv1x1_split_kc_ok(int Cin, int Cout, int N, int H, int W);
int launch_conv1x1_split_kc(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                         const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                         int H, int W, hipStream_t s);

// floats appended to the fp32 packed filter of the logical convolution IC -> OC (0 when unsupported)
size_t split_packed_floats(int OC, int IC, int ks);

// 32-wide column tiles of the split layout for OC output channels (1 when OC <= 16: 16 channels x 2 output rows)
int split_column_tiles(int OC);

int launch_pack_split_table(const void* table_dev, int n, hipStream_t s);
int launch_pack_split(const float* w, int Cout, int Cin, int ks, int transposed, float* packed_split, hipStream_t s);

// Grouped launches: group g works on input channels [x_coff + g * x_stride, + Cin), output channels [y_coff + g * y_stride, + Cout)
// and the packed filter at wsplit + g * w_stride (16-byte units); n = 1 and zero strides = a dense convolution.
struct ConvGroups { int n = 1, x_stride = 0, y_stride = 0; size_t w_stride = 0; };

int launch_conv_split(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                      const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                      int H, int W, int ks, int ty, int cot, hipStream_t s, const ConvGroups& grp = ConvGroups());

// Several convolutions of one launch shape in ONE dispatch (cd_conv2d_fwd_multi): the members of `c` share N, H, W and Cout.
struct SplitConv {
    const float* x; const float* wsplit; const float* bias; const float* in_scale; const float* in_shift; float* y; double* stats;
    int x_ctot, x_coff, Cin, in_relu, y_ctot, y_coff, accumulate, ks;
};
int launch_conv_split_multi(const SplitConv* c, int n, int N, int H, int W, int Cout, int ty, int cot, hipStream_t s);

}  // namespace cd
