// Direct 2-D convolution at fp32 accuracy on the gfx950 BF16 matrix cores (forward and input gradient, k in {5, 7, 11}).
//
// CDNA4 has no TF32/xf32 and its fp32 matrix instruction runs at the vector rate (157 TFLOP/s); the bf16 instruction
// v_mfma_f32_16x16x32_bf16 is ~10-15x faster per multiply-add.  Every fp32 operand is split EXACTLY into three bf16 terms
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (RNE; 3 x 8 = 24 mantissa bits)
// and a product is evaluated as the six cross terms of weight  >= 2^-16:  hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi
// (each bf16 x bf16 product is exact in fp32; the dropped mid*lo, lo*mid, lo*lo terms are < 2^-25 |x*y|, below half an fp32
// ulp of the product), accumulated in fp32 by the matrix core.  Measured against fp64 this is as close as -- on long sums
// closer than -- the native fp32 instruction (profiles/mfma_split_exp_r02.txt: error / sum|a*b| 1.3e-7 vs 1.3e-7 at K = 2048,
// identical with all nine terms), so no precision is traded; what IS different: the result is not bitwise the fmaf chain
// of conv_mfma.hip (cd_set_conv_arith(0) selects that kernel), and an infinite input gives NaN instead of +-inf.
//
// Mapping (implicit GEMM, no im2col buffer): M = 16 consecutive output pixels of one row, N = 16 output channels,
// K = 32 = 8 input channels x 4 consecutive filter taps (taps flattened ky * KS + kx, padded to a multiple of 4 with zero
// weights: 121 -> 124, 49 -> 52, 25 -> 28).
//   A[i = lane&15][8 * (lane>>4) + e] = act(in)[ci0 + e][y + ky][x0 + i + kx],  (ky, kx) = tap 4*step + (lane>>4)
//        one ds_read_b128 per split: the LDS tile is channels-last, 8 bf16 channels = 16 bytes per pixel, three planes
//   B[8 * (lane>>4) + e][j = lane&15] = w[co0 + j][ci0 + e][tap]
//        pre-packed in exactly this fragment order (3 x 1 KB per step), read straight from global/L2 one step ahead
//   D: lane holds channel co0 + (lane&15), pixels x0 + 4*(lane>>4) + {0..3}  (same as the fp32 kernel: same epilogue).
// A block (4 waves) owns a TY x 32 output tile for CO_T*16 output channels; input channels stream through LDS 8 at a time
// (global fp32 -> producer's BN-apply + ReLU -> split -> LDS).  Per step a wave reads 3 KB of LDS per pixel tile and issues
// 6 MFMAs per (pixel tile, channel tile): with one channel tile (Cout = 16) the kernel is LDS-bandwidth bound, with two
// or more MFMA bound.  Fusions (input affine/ReLU, bias, accumulate, BatchNorm statistics) are those of conv_mfma.hip.
#include "cd_common.h"
#include "conv_split.h"

namespace cd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SP_TX = 32;

// two fp32 -> packed bf16 pair (round to nearest even), low half = a
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// (a, b) -> packed pairs of the three split terms
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(a, b);
    const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
    m = cvt_pk_bf16(ra, rb);
    l = cvt_pk_bf16(ra - bf16_lo(m), rb - bf16_hi(m));
}

template <int KS, int TY_> struct SplitCfg {
    static constexpr int TY = TY_;
    static constexpr int TAPS = KS * KS, KSTEPS = (TAPS + 3) / 4;
    static constexpr int ROWS = TY + KS - 1;
    static constexpr int PADL = (((KS - 1) / 2) + 3) & ~3;          // aligned superset rows, as in conv_mfma.hip
    static constexpr int RSP = SP_TX + 2 * PADL, COFF = PADL - (KS - 1) / 2;
    static constexpr int PLANE = ROWS * RSP;                        // pixels (16-byte slots) per split plane
    static constexpr size_t LDS = (size_t)3 * PLANE * 16;
};

// ---------------------------------------------------------------- weight packing (split layout)
// [co tile][ci chunk of 8][step][split][lane][8 bf16]; element e of lane (j, g) = w[tile*16 + j][chunk*8 + e][tap 4*step + g]
__device__ __forceinline__ void pack_split_elements(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout_src, int Cin_src,
                                                    int KS, int transposed, int OC, int IC, int oc_off, int ic_off, size_t first,
                                                    size_t stride) {
    const int oc_n = transposed ? Cin_src : Cout_src, ic_n = transposed ? Cout_src : Cin_src;
    const int taps = KS * KS, ksteps = (taps + 3) / 4, chunks = (IC + 7) / 8, tiles = (OC + 15) / 16;
    const size_t total = (size_t)tiles * chunks * ksteps * 512;
    for (size_t i = first; i < total; i += stride) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int lane = (int)(r & 63); r >>= 6;
        const int step = (int)(r % ksteps); r /= ksteps;
        const int chunk = (int)(r % chunks); r /= chunks;
        const int tile = (int)r;
        const int tap = step * 4 + (lane >> 4);
        const int oc = tile * 16 + (lane & 15) - oc_off, ic = chunk * 8 + e - ic_off;
        if (tap >= taps || (unsigned)oc >= (unsigned)oc_n || (unsigned)ic >= (unsigned)ic_n) continue;   // padding stays zero
        const int ky = tap / KS, kx = tap - ky * KS;
        const float v = transposed ? w[(((size_t)ic * Cin_src + oc) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)]
                                   : w[(((size_t)oc * Cin_src + ic) * KS + ky) * KS + kx];
        unsigned h, m, l;
        split_pair(v, 0.f, h, m, l);
        const size_t base = ((((size_t)tile * chunks + chunk) * ksteps + step) * 3) * 512 + (size_t)lane * 8 + e;
        out[base] = (unsigned short)h; out[base + 512] = (unsigned short)m; out[base + 1024] = (unsigned short)l;
    }
}

__global__ void pack_split_kernel(const float* __restrict__ w, int Cout, int Cin, int KS, int transposed, unsigned short* __restrict__ out) {
    const int OC = transposed ? Cin : Cout, IC = transposed ? Cout : Cin;
    pack_split_elements(w, out, Cout, Cin, KS, transposed, OC, IC, 0, 0, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                        (size_t)gridDim.x * blockDim.x);
}

// All filters of a network: blockIdx.y selects the descriptor (the split layout follows the fp32 layout in `packed`)
__global__ void pack_split_table_kernel(const PackDesc* __restrict__ table) {
    const PackDesc d = table[blockIdx.y];
    if (!split_supported(d.ks)) return;
    float* out = d.packed + fp32_packed_floats(d.OC, d.IC, d.ks);
    pack_split_elements(d.w, reinterpret_cast<unsigned short*>(out), d.Cout, d.Cin, d.ks, d.transposed, d.OC, d.IC, d.oc_off, d.ic_off,
                        (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

int launch_pack_split_table(const void* table_dev, int n, hipStream_t s) {
    hipLaunchKernelGGL(pack_split_table_kernel, dim3(16, n), dim3(256), 0, s, (const PackDesc*)table_dev);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- the convolution
template <int KS, int CO_T, int TYP>
__global__ __launch_bounds__(kBlock) void conv_fwd_split_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const u32x4* __restrict__ wsp, int pack_tiles, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    float* __restrict__ y, int y_ctot, int y_coff, int Cout,
    double* __restrict__ stats, int accumulate, int H, int W, int tiles_x, int tiles_img, int tiles_total, int chunk_tiles,
    int slices) {
    using Cfg = SplitCfg<KS, TYP>;
    constexpr int TY = Cfg::TY, ROWS = Cfg::ROWS, RSP = Cfg::RSP, COFF = Cfg::COFF, PADL = Cfg::PADL, PLANE = Cfg::PLANE;
    constexpr int P = (KS - 1) / 2, TAPS = Cfg::TAPS, KSTEPS = Cfg::KSTEPS;
    constexpr int COB = CO_T * 16;
    constexpr int RPW = TY / 4, MT = RPW * 2;
    constexpr int UNITS = ROWS * (RSP / 4);           // staging units: (row, 4-pixel quad) x 8 channels

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* s_in = reinterpret_cast<u32x4*>(smem_raw);   // [3][ROWS][RSP] 16-byte slots (8 bf16 channels of one pixel)

    // XCD-aware block -> (image tile, channel slice) mapping (see conv_mfma.hip)
    const int xcd = blockIdx.x & 7, kx_ = blockIdx.x >> 3;
    const int tg = kx_ / slices, slice = kx_ - tg * slices;
    const int t_lin = xcd * chunk_tiles + tg;
    if (tg >= chunk_tiles || t_lin >= tiles_total) return;   // block-uniform
    const int n = t_lin / tiles_img, tile = t_lin - n * tiles_img;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * SP_TX, Y0 = ty * TY;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const size_t HW = (size_t)H * W;
    const float* xin = x + ((size_t)n * x_ctot + x_coff) * HW;
    const int n_chunks = (Cin + 7) / 8;
    const bool vec_in = (W & 3) == 0;

    f32x4 acc[MT][CO_T];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < CO_T; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the weight stream of channel tile t: [chunk][step][split][lane], linear in (chunk, step)
    const u32x4* wt[CO_T];
    bool live[CO_T];
#pragma unroll
    for (int t = 0; t < CO_T; ++t) {
        const int gt = slice * CO_T + t;
        live[t] = gt < pack_tiles;   // block-uniform
        wt[t] = wsp + (size_t)(live[t] ? gt : 0) * n_chunks * KSTEPS * 192 + lane;
    }
    const int steps_total = n_chunks * KSTEPS;
    bf16x8 bcur[CO_T][3], bnext[CO_T][3];
    auto load_b = [&](bf16x8 (&dst)[CO_T][3], int lin) {
#pragma unroll
        for (int t = 0; t < CO_T; ++t)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const u32x4 v = wt[t][(size_t)lin * 192 + sp * 64];
                dst[t][sp] = __builtin_bit_cast(bf16x8, v);
            }
    };
    load_b(bcur, 0);

    int arow[MT];   // LDS slot of this lane's pixel for pixel tile m at tap (0, 0)
#pragma unroll
    for (int m = 0; m < MT; ++m) arow[m] = (wid * RPW + (m >> 1)) * RSP + (m & 1) * 16 + li + COFF;

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        __syncthreads();   // the previous chunk's fragments are consumed
        // ---- stage 8 input channels: global fp32 -> (affine, relu) -> three bf16 planes, channels-last
        for (int u = threadIdx.x; u < UNITS; u += kBlock) {
            const int r = u / (RSP / 4), q4 = (u - r * (RSP / 4)) * 4;
            const int gy = Y0 - P + r, gx = X0 - PADL + q4;
            float v[8][4];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = chunk * 8 + c;
                v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
                if (ci < Cin && (unsigned)gy < (unsigned)H) {
                    const float* src = xin + (size_t)ci * HW + (size_t)gy * W + gx;
                    bool ok[4];
                    if (vec_in) {
                        const bool in = (unsigned)gx < (unsigned)W;   // W % 4 == 0: an aligned quad is inside or outside as a whole
                        if (in) { const float4 f = *reinterpret_cast<const float4*>(src); v[c][0] = f.x; v[c][1] = f.y; v[c][2] = f.z; v[c][3] = f.w; }
                        ok[0] = ok[1] = ok[2] = ok[3] = in;
                    } else {
#pragma unroll
                        for (int p = 0; p < 4; ++p) { ok[p] = (unsigned)(gx + p) < (unsigned)W; if (ok[p]) v[c][p] = src[p]; }
                    }
                    if (in_scale) {
                        const float sc = in_scale[ci], sh = in_shift[ci];
#pragma unroll
                        for (int p = 0; p < 4; ++p) if (ok[p]) v[c][p] = __fmaf_rn(v[c][p], sc, sh);   // zero padding stays zero
                    }
                    if (in_relu) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) v[c][p] = fmaxf(v[c][p], 0.f);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                u32x4 hh, mm, ll;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    unsigned h, m, l;
                    split_pair(v[2 * c2][p], v[2 * c2 + 1][p], h, m, l);
                    hh[c2] = h; mm[c2] = m; ll[c2] = l;
                }
                const int slot = r * RSP + q4 + p;
                s_in[slot] = hh; s_in[PLANE + slot] = mm; s_in[2 * PLANE + slot] = ll;
            }
        }
        __syncthreads();

        // ---- MFMA over the tap steps of this chunk
        int ky = 0, kx = g;   // this lane's tap of the current step: 4 * step + g  (g < 4 < KS)
#pragma unroll 1
        for (int s = 0; s < KSTEPS; ++s) {
            const int lin = chunk * KSTEPS + s;
            if (lin + 1 < steps_total) load_b(bnext, lin + 1);
            const int toff = (s * 4 + g < TAPS) ? ky * RSP + kx : 0;   // padded taps carry zero weights: any valid slot
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int slot = arow[m] + toff;
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, s_in[slot]);
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, s_in[PLANE + slot]);
                const bf16x8 a2 = __builtin_bit_cast(bf16x8, s_in[2 * PLANE + slot]);
#pragma unroll
                for (int t = 0; t < CO_T; ++t) {
                    if (!live[t]) continue;
                    f32x4 c = acc[m][t];   // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bcur[t][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bcur[t][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bcur[t][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bcur[t][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bcur[t][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bcur[t][0], c, 0, 0, 0);
                    acc[m][t] = c;
                }
            }
            kx += 4;
            if (kx >= KS) { kx -= KS; ++ky; }
#pragma unroll
            for (int t = 0; t < CO_T; ++t)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bcur[t][sp] = bnext[t][sp];
        }
    }

    // ---- epilogue: bias, store, batch statistics of the raw output (identical to conv_mfma.hip: same D layout)
    const int co_l = li, px4 = g * 4;
    const int co_base = slice * COB;
    float* yout = y + ((size_t)n * y_ctot + y_coff) * HW;
    double s1[CO_T], s2[CO_T];
#pragma unroll
    for (int t = 0; t < CO_T; ++t) {
        const int co = co_base + t * 16 + co_l;
        const float bv = (bias != nullptr && co < Cout) ? bias[co] : 0.f;
        s1[t] = 0.0; s2[t] = 0.0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int gy = Y0 + wid * RPW + (m >> 1), gx = X0 + (m & 1) * 16 + px4;
            f32x4 v = acc[m][t];
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            if (co < Cout && gy < H) {
                float* dst = yout + (size_t)co * HW + (size_t)gy * W + gx;
                if (gx + 3 < W && ((W & 3) == 0)) {
                    if (accumulate) {
                        const float4 o = *reinterpret_cast<const float4*>(dst);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(v.x, v.y, v.z, v.w);
                    if (stats != nullptr) {
                        const double a = v.x, b = v.y, c = v.z, d = v.w;
                        s1[t] += (a + b) + (c + d);
                        s2[t] += (a * a + b * b) + (c * c + d * d);
                    }
                } else {
                    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gx + q < W) {
                            if (accumulate) e[q] += dst[q];
                            dst[q] = e[q]; s1[t] += (double)e[q]; s2[t] += (double)e[q] * (double)e[q];
                        }
                }
            }
        }
    }
    if (stats != nullptr) {  // block-uniform; reduction as in conv_mfma.hip
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_raw);   // [4 waves][COB][2]
#pragma unroll
        for (int t = 0; t < CO_T; ++t) {
            double a = s1[t], b = s2[t];
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (lane < 16) { red[(wid * COB + t * 16 + co_l) * 2] = a; red[(wid * COB + t * 16 + co_l) * 2 + 1] = b; }
        }
        __syncthreads();
        if (threadIdx.x < COB) {
            const int co = co_base + threadIdx.x;
            if (co < Cout) {
                const double a = (red[threadIdx.x * 2] + red[(COB + threadIdx.x) * 2]) + (red[(2 * COB + threadIdx.x) * 2] + red[(3 * COB + threadIdx.x) * 2]);
                const double b = (red[threadIdx.x * 2 + 1] + red[(COB + threadIdx.x) * 2 + 1]) + (red[(2 * COB + threadIdx.x) * 2 + 1] + red[(3 * COB + threadIdx.x) * 2 + 1]);
                const int slot = t_lin & (CD_BN_STAT_SLOTS - 1);
                double* st = stats + ((size_t)slot * y_ctot + y_coff + co) * 2;
                atomicAdd(st, a);
                atomicAdd(st + 1, b);
            }
        }
    }
}

template <int KS, int CO_T, int TYP>
static int launch_split_t(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                          const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate,
                          int N, int H, int W, hipStream_t s) {
    using Cfg = SplitCfg<KS, TYP>;
    const int tiles_x = (W + SP_TX - 1) / SP_TX, tiles_y = (H + Cfg::TY - 1) / Cfg::TY;
    const size_t lds = Cfg::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_split_kernel<KS, CO_T, TYP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (lds > 160 * 1024) return CD_ERR_UNSUPPORTED;
    const int pack_tiles = (Cout + 15) / 16, slices = (pack_tiles + CO_T - 1) / CO_T;
    const int tiles_img = tiles_x * tiles_y, tiles_total = tiles_img * N, chunk_tiles = (tiles_total + 7) / 8;
    hipLaunchKernelGGL((conv_fwd_split_kernel<KS, CO_T, TYP>), dim3((unsigned)chunk_tiles * 8u * (unsigned)slices), dim3(kBlock), lds, s, x, x_ctot,
                       x_coff, Cin, reinterpret_cast<const u32x4*>(wsplit), pack_tiles, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff,
                       Cout, stats, accumulate, H, W, tiles_x, tiles_img, tiles_total, chunk_tiles, slices);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

size_t split_packed_floats(int OC, int IC, int ks) {
    if (!split_supported(ks)) return 0;
    const size_t taps = (size_t)ks * ks, ksteps = (taps + 3) / 4, chunks = ((size_t)IC + 7) / 8, tiles = ((size_t)OC + 15) / 16;
    return tiles * chunks * ksteps * 3 * 64 * 4;   // 16 bytes = 4 floats per lane per split
}

int launch_pack_split(const float* w, int Cout, int Cin, int ks, int transposed, float* packed_split, hipStream_t s) {
    const int OC = transposed ? Cin : Cout, IC = transposed ? Cout : Cin;
    const size_t total = split_packed_floats(OC, IC, ks) * 2 / 3;   // elements (bf16 triples)
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, Cout, Cin, ks, transposed,
                       reinterpret_cast<unsigned short*>(packed_split));
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

int launch_conv_split(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                      const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                      int H, int W, int ks, int ty, int cot, hipStream_t s) {
    if (cot > 2) cot = 2;
    if (Cout <= 16) cot = 1;
#define CD_SP(K, T, Y) return launch_split_t<K, T, Y>(x, x_ctot, x_coff, Cin, wsplit, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, s)
#define CD_SP_T(K, T)                  \
    {                                  \
        if (ty == 16) CD_SP(K, T, 16); \
        if (ty == 8) CD_SP(K, T, 8);   \
        CD_SP(K, T, 4);                \
    }
#define CD_SP_K(K)                    \
    if (ks == K) {                    \
        if (cot == 1) CD_SP_T(K, 1)   \
        CD_SP_T(K, 2)                 \
    }
    CD_SP_K(5) CD_SP_K(7) CD_SP_K(11)
#undef CD_SP_K
#undef CD_SP_T
#undef CD_SP
    return CD_ERR_UNSUPPORTED;
}

}  // namespace cd
