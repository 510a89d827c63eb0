// Direct 2-D convolution on the gfx950 matrix cores in exact fp32
// (v_mfma_f32_16x16x4_f32: bitwise a k-ordered fmaf chain, 157 TFLOP/s peak -- there is no
// TF32/xf32 on CDNA4 and the reference computes in fp32, so no precision is traded).
//
// Replaces the nn.Conv2d forward / input-gradient launches of the (un-vendored) hourglass
// (/root/reference/monodepth/mannequin_challenge_model.py:60 -> netG.forward; architecture
// SURVEY.md appendix A.3): stride 1, "same" zero padding, k in {1,3,5,7,11}, NCHW fp32.
//
// Mapping (implicit GEMM, no im2col buffer):  M = 16 consecutive output pixels of one row,
// N = 16 output channels, K = 4 input channels of ONE filter tap (ky,kx).
//   A[i = lane&15][k = lane>>4] = in[ci0+k][y+ky][x0+i+kx]   (LDS input tile with halo; the 16
//                                  pixels are contiguous, the 4 channels sit 16 banks apart)
//   B[k = lane>>4][j = lane&15] = w[co0+j][ci0+k][ky][kx]     (LDS, pre-packed [tap][ci][co])
//   D: lane holds channel co0+(lane&15), pixels x0+4*(lane>>4)+{0..3}  -> one 16-byte store.
// A block (4 waves) owns a TY x 32 output tile of one image for CO_T*16 output channels; each wave
// keeps (TY/4 rows x 2) M-tiles x CO_T N-tiles of accumulators in registers and walks the taps
// with immediate LDS offsets; input channels are streamed through LDS in chunks of CI_CHUNK.
//
// Fusions: the producer's affine + ReLU is applied while the input tile is staged (per-channel
// scale/shift), and the per-channel sum / sum-of-squares of the raw output (the batch statistics the
// FOLLOWING train-mode BatchNorm needs) are accumulated in the epilogue (fp64, one atomic pair per
// channel per block into one of CD_BN_STAT_SLOTS copies).
// The input-gradient convolution (dgrad) is the same kernel on flipped/transposed packed weights.
// Launch shape (tile rows x channel slices per block) is a run-time choice with bit-identical results;
// staging is 16-byte wide (aligned superset rows) and software-pipelined through registers where it fits.
#include <stdlib.h>
#include <string.h>

#include "cd_common.h"
#include "conv_split.h"

namespace cd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CV_TX = 32;  // output tile width (2 M-tiles)
#ifndef CD_CONV_PIPE_MAX_REGS
#define CD_CONV_PIPE_MAX_REGS 72   // prefetch registers (data + offsets) a block may spend on the software pipeline
#endif

template <int KS, int TY_> struct ConvCfg {
    static constexpr int TY = TY_;                          // output rows per block (4 waves x TY/4 rows): 4, 8 or 16
    static constexpr int CI_CHUNK = (KS >= 7) ? 4 : (KS == 1 ? 32 : 8);  // input channels staged per round (1x1 = pure GEMM: long K chunks)
    static constexpr int RS = CV_TX + KS - 1;              // logical tile row (floats): [X0 - P, X0 + 32 + P)
    static constexpr int ROWS = TY + KS - 1;
    // physical LDS row: the 16-byte aligned superset [X0 - PADL, X0 + 32 + PADL), staged with 16-byte global loads;
    // logical column c sits at c + COFF
    static constexpr int PADL = (((KS - 1) / 2) + 3) & ~3;
    static constexpr int RSP = CV_TX + 2 * PADL, COFF = PADL - (KS - 1) / 2;
    static constexpr int PLANE_RAW = ROWS * RSP;
    // plane stride == 16 (mod 32): the 4 channels of an A fragment hit disjoint bank halves
    static constexpr int PS = PLANE_RAW + ((16 - (PLANE_RAW % 32)) + 32) % 32;
};

// ---------------------------------------------------------------- weight packing
// w [Cout][Cin][KS][KS] -> packed [co_group][ci_chunk][tap][ci_in_chunk][COBP]  (zero padded)
// flip_transpose: build the dgrad filter  w'[ci][co][KS-1-ky][KS-1-kx]  instead (roles of Cin/Cout swap).
__global__ void pack_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int KS, int ci_chunk, int cob,
                                    int cobp, int flip_transpose, float* __restrict__ out, size_t total) {
    // logical conv: out-channels OC, in-channels IC
    const int OC = flip_transpose ? Cin : Cout, IC = flip_transpose ? Cout : Cin;
    const int taps = KS * KS;
    const int n_chunks = (IC + ci_chunk - 1) / ci_chunk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int j = (int)(r % cobp); r /= cobp;
        const int cc = (int)(r % ci_chunk); r /= ci_chunk;
        const int tap = (int)(r % taps); r /= taps;
        const int chunk = (int)(r % n_chunks); r /= n_chunks;
        const int grp = (int)r;
        const int oc = grp * cob + j, ic = chunk * ci_chunk + cc;
        float v = 0.f;
        if (j < cob && oc < OC && ic < IC) {
            const int ky = tap / KS, kx = tap - ky * KS;
            if (!flip_transpose) v = w[(((size_t)oc * Cin + ic) * KS + ky) * KS + kx];
            else v = w[(((size_t)ic * Cin + oc) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)];
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------- the convolution
// CO_T = 16-wide output-channel tiles per block.  The packed filter rows are cobp_pack floats wide (the layout
// is chosen once per filter, pick_co_tiles); a block may take only a CO_T*16-column slice of them
// (which gives the small deep-level images enough workgroups) or, for 1x1, span several groups (input read once).
template <int KS, int CO_T, int TYP>
__global__ __launch_bounds__(kBlock) void conv_fwd_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const float* __restrict__ wpk, int cobp_pack, int pack_cot, int pack_tiles, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    float* __restrict__ y, int y_ctot, int y_coff, int Cout,
    double* __restrict__ stats, int accumulate, int H, int W, int tiles_x, int tiles_img, int tiles_total, int chunk,
    int slices, int pipe) {
    using Cfg = ConvCfg<KS, TYP>;
    constexpr int TY = Cfg::TY, CI = Cfg::CI_CHUNK, RS = Cfg::RS, PS = Cfg::PS, ROWS = Cfg::ROWS;
    constexpr int RSP = Cfg::RSP, COFF = Cfg::COFF, PADL = Cfg::PADL;
    constexpr int P = (KS - 1) / 2, TAPS = KS * KS;
    constexpr int COB = CO_T * 16, COBP = co_stride_padded(COB);
    constexpr int RPW = TY / 4;          // output rows per wave
    constexpr int MT = RPW * 2;          // M-tiles per wave
    constexpr int ROW4 = COB / 4;        // float4 per staged filter row
    constexpr int W_ROWS = TAPS * CI;    // filter rows (tap, ci) per chunk

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                  // [CI][PS]
    float* s_w = smem + CI * PS;         // [TAPS][CI][COBP]

    // XCD-aware block -> (image tile, channel slice) mapping.  Workgroups are dealt round-robin to the 8 XCDs (own L2
    // each): XCD x takes the contiguous run of tiles [x * chunk, (x+1) * chunk), and the slices of one tile are
    // consecutive in its dispatch order -- so the workgroups that read the same input tile (all channel slices) or
    // overlapping halos (neighbouring tiles) share an L2 and run close in time.
    const int xcd = blockIdx.x & 7, kx_ = blockIdx.x >> 3;
    const int tg = kx_ / slices, slice = kx_ - tg * slices;
    const int t_lin = xcd * chunk + tg;
    if (tg >= chunk || t_lin >= tiles_total) return;   // block-uniform
    const int n = t_lin / tiles_img, tile = t_lin - n * tiles_img;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * CV_TX, Y0 = ty * TY;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t HW = (size_t)H * W;
    const float* xin = x + ((size_t)n * x_ctot + x_coff) * HW;
    const int n_chunks = (Cin + CI - 1) / CI;
    // this block's CO_T channel tiles are the global tiles slice * CO_T + t; tile g lives in packed group g / pack_cot
    // at column (g % pack_cot) * 16 -- a block may take part of a packed group or span several
    const int grp_stride = n_chunks * W_ROWS * cobp_pack;

    f32x4 acc[MT][CO_T];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < CO_T; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_lane = (lane >> 4) * PS + (lane & 15);      // A fragment: channel k = lane>>4, pixel i = lane&15
    const int b_lane = (lane >> 4) * COBP + (lane & 15);    // B fragment: channel k, out-channel j

    // Software pipeline: the next chunk's global loads (input tile and filter slice) are in flight in registers
    // while the MFMAs of the current chunk run; the producer's BN-apply + ReLU is applied when the registers are
    // written to LDS (scale/shift staged once in LDS).  Per-thread element offsets are 32-bit and precomputed; the
    // chunk base pointers are wave-uniform.  Used when the registers fit (PIPE_OK) and the launch asks for it.
    // the input tile is staged as float4 of the aligned superset rows (needs W % 4 == 0: an aligned float4 is then inside
    // or outside the image as a whole); otherwise scalar, unpipelined
    constexpr bool VEC_IN = true;
    constexpr int IN_ELEMS = CI * ROWS * (RSP / 4);
    constexpr int PF_IN = (IN_ELEMS + kBlock - 1) / kBlock;
    constexpr int PF_W = (W_ROWS * ROW4 + kBlock - 1) / kBlock;
    constexpr bool PIPE_OK = (PF_IN * 5 + PF_W * 5) <= CD_CONV_PIPE_MAX_REGS;
    const bool vec_in = (W & 3) == 0;
    const bool pipelined = PIPE_OK && pipe && vec_in && (size_t)CI * HW < (1u << 30);
    float* s_aff = s_w + W_ROWS * COBP;  // [2][n_chunks * CI] scale, shift of the input channels (pipelined path)

    // ---- generic (non-pipelined) staging helpers
    auto in_load4 = [&](int chunk, int i) -> float4 {
        const int cc = i / (ROWS * (RSP / 4)), rem = i - cc * (ROWS * (RSP / 4));
        const int r = rem / (RSP / 4), q4 = (rem - r * (RSP / 4)) * 4;
        const int ci = chunk * CI + cc, gy = Y0 - P + r, gx = X0 - PADL + q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            v = *reinterpret_cast<const float4*>(xin + (size_t)ci * HW + (size_t)gy * W + gx);
            if (in_scale) { const float sc = in_scale[ci], sh = in_shift[ci]; v.x = __fmaf_rn(v.x, sc, sh); v.y = __fmaf_rn(v.y, sc, sh); v.z = __fmaf_rn(v.z, sc, sh); v.w = __fmaf_rn(v.w, sc, sh); }
            if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        return v;
    };
    auto in_load1 = [&](int chunk, int i, int pad) -> float {
        const int cc = i / (ROWS * RS), rem = i - cc * (ROWS * RS);
        const int r = rem / RS, c = rem - r * RS;
        const int ci = chunk * CI + cc, gy = Y0 - pad + r, gx = X0 - pad + c;
        float v = 0.f;
        if (ci < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            v = xin[(size_t)ci * HW + (size_t)gy * W + gx];
            if (in_scale) v = __fmaf_rn(v, in_scale[ci], in_shift[ci]);  // same fma as the BN backward's mask
            if (in_relu) v = fmaxf(v, 0.f);
        }
        return v;
    };
    auto in_lds4 = [&](int i) -> int {
        const int cc = i / (ROWS * (RSP / 4)), rem = i - cc * (ROWS * (RSP / 4));
        const int r = rem / (RSP / 4), q4 = (rem - r * (RSP / 4)) * 4;
        return cc * PS + r * RSP + q4;
    };
    auto in_lds1 = [&](int i) -> int {
        const int cc = i / (ROWS * RS), rem = i - cc * (ROWS * RS);
        const int r = rem / RS;
        return cc * PS + r * RSP + (rem - r * RS) + COFF;
    };
    auto w_src = [&](int i) -> int {   // offset of the float4 from the chunk's base in packed group 0; -1 = zero fill
        const int row = i / ROW4, c4 = i - row * ROW4;
        const int g = slice * CO_T + (c4 >> 2);
        if (g >= pack_tiles) return -1;
        const int pg = g / pack_cot;
        return pg * grp_stride + row * cobp_pack + (g - pg * pack_cot) * 16 + (c4 & 3) * 4;
    };
    auto w_lds = [&](int i) -> int { const int row = i / ROW4; return row * COBP + (i - row * ROW4) * 4; };

    // ---- pipelined path state
    int in_off[PIPE_OK ? PF_IN : 1], w_off[PIPE_OK ? PF_W : 1];   // element offsets from the chunk base; -1 = zero padding
    float4 pf_in4[(PIPE_OK && VEC_IN) ? PF_IN : 1];
    float pf_in1[(PIPE_OK && !VEC_IN) ? PF_IN : 1];
    float4 pf_w[PIPE_OK ? PF_W : 1];
    if constexpr (PIPE_OK) {
        if (pipelined) {
#pragma unroll
            for (int q = 0; q < PF_IN; ++q) {
                const int i = threadIdx.x + q * kBlock;
                int off = -1;
                if (i < IN_ELEMS) {
                    const int cc = i / (ROWS * (RSP / 4)), rem = i - cc * (ROWS * (RSP / 4));
                    const int r = rem / (RSP / 4), q4 = (rem - r * (RSP / 4)) * 4;
                    const int gy = Y0 - P + r, gx = X0 - PADL + q4;
                    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) off = cc * (int)HW + gy * W + gx;
                }
                in_off[q] = off;
            }
#pragma unroll
            for (int q = 0; q < PF_W; ++q) {
                const int i = threadIdx.x + q * kBlock;
                w_off[q] = i < W_ROWS * ROW4 ? w_src(i) : -1;
            }
            if (in_scale)
                for (int i = threadIdx.x; i < n_chunks * CI; i += kBlock) {
                    s_aff[i] = i < Cin ? in_scale[i] : 0.f;
                    s_aff[n_chunks * CI + i] = i < Cin ? in_shift[i] : 0.f;
                }
        }
    }
    auto pf_load = [&](int chunk) {
        if constexpr (PIPE_OK) {
            const float* xc = xin + (size_t)chunk * CI * HW;                       // uniform
            const float* wc = wpk + (size_t)chunk * W_ROWS * cobp_pack;          // uniform
            const int ci_left = Cin - chunk * CI;                                 // channels of this chunk that exist
#pragma unroll
            for (int q = 0; q < PF_IN; ++q) {
                const int i = threadIdx.x + q * kBlock;
                const int cc = i / (ROWS * (RSP / 4));
                const bool ok = in_off[q] >= 0 && cc < ci_left;
                if constexpr (VEC_IN) pf_in4[q] = ok ? *reinterpret_cast<const float4*>(xc + in_off[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
                else pf_in1[q] = ok ? xc[in_off[q]] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < PF_W; ++q)
                pf_w[q] = w_off[q] >= 0 ? *reinterpret_cast<const float4*>(wc + w_off[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto pf_store = [&](int chunk) {
        if constexpr (PIPE_OK) {
            const int ci_left = Cin - chunk * CI;
#pragma unroll
            for (int q = 0; q < PF_IN; ++q) {
                const int i = threadIdx.x + q * kBlock;
                if (i < IN_ELEMS) {
                    const int cc = i / (ROWS * (RSP / 4));
                    const bool live = in_off[q] >= 0 && cc < ci_left;   // zero padding stays zero
                    float sc = 1.f, sh = 0.f;
                    if (in_scale) { sc = s_aff[chunk * CI + cc]; sh = s_aff[n_chunks * CI + chunk * CI + cc]; }
                    if constexpr (VEC_IN) {
                        float4 v = pf_in4[q];
                        if (live) {
                            if (in_scale) { v.x = __fmaf_rn(v.x, sc, sh); v.y = __fmaf_rn(v.y, sc, sh); v.z = __fmaf_rn(v.z, sc, sh); v.w = __fmaf_rn(v.w, sc, sh); }
                            if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        }
                        *reinterpret_cast<float4*>(s_in + in_lds4(i)) = v;
                    } else {
                        float v = pf_in1[q];
                        if (live) {
                            if (in_scale) v = __fmaf_rn(v, sc, sh);
                            if (in_relu) v = fmaxf(v, 0.f);
                        }
                        s_in[in_lds1(i)] = v;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < PF_W; ++q) {
                const int i = threadIdx.x + q * kBlock;
                if (i < W_ROWS * ROW4) *reinterpret_cast<float4*>(s_w + w_lds(i)) = pf_w[q];
            }
        }
    };
    if (pipelined) pf_load(0);

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        __syncthreads();  // previous round's fragments are consumed (and, first round, s_aff is written)
        if (pipelined) {
            pf_store(chunk);
            if (chunk + 1 < n_chunks) pf_load(chunk + 1);
        } else {
            // ---- stage the input tile (zero padding, fused BN-apply + ReLU of the producer) and the filter slice
            if (vec_in) {
                for (int i = threadIdx.x; i < IN_ELEMS; i += kBlock) *reinterpret_cast<float4*>(s_in + in_lds4(i)) = in_load4(chunk, i);
            } else {
                for (int i = threadIdx.x; i < CI * ROWS * RS; i += kBlock) s_in[in_lds1(i)] = in_load1(chunk, i, P);
            }
            const float* wc = wpk + (size_t)chunk * W_ROWS * cobp_pack;
            for (int i = threadIdx.x; i < W_ROWS * ROW4; i += kBlock) {
                const int off = w_src(i);
                *reinterpret_cast<float4*>(s_w + w_lds(i)) = off >= 0 ? *reinterpret_cast<const float4*>(wc + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();

        // ---- MFMA over (channel quad, tap)
#pragma unroll
        for (int c4 = 0; c4 < CI / 4; ++c4) {
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float bf[CO_T];
#pragma unroll
                    for (int t = 0; t < CO_T; ++t) bf[t] = s_w[((ky * KS + kx) * CI + c4 * 4) * COBP + t * 16 + b_lane];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int row = wid * RPW + (m >> 1), ct = m & 1;
                        const float af = s_in[c4 * 4 * PS + (row + ky) * RSP + ct * 16 + kx + COFF + a_lane];
#pragma unroll
                        for (int t = 0; t < CO_T; ++t)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[t], acc[m][t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: bias, store, batch statistics of the raw output
    const int co_l = lane & 15, px4 = (lane >> 4) * 4;
    const int co_base = slice * COB;
    float* yout = y + ((size_t)n * y_ctot + y_coff) * HW;
    // statistics partials in double: the sums must not depend on how the launch shape groups the pixels
    // (fp32 partials differ at 1e-7 between tile shapes, which a deep train-mode-BN network amplifies)
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
                    if (accumulate) {  // gradient fan-in: y += conv
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
    if (stats != nullptr) {  // block-uniform
        // Same-address fp64 atomics serialise in the memory system (~50 ns each, measured), so: reduce over the lanes
        // and the 4 waves of the block first (LDS), then ONE atomic pair per channel per block, spread over
        // CD_BN_STAT_SLOTS copies of the statistics that the BatchNorm kernels sum.
        __syncthreads();   // the MFMA operands in LDS are dead: reuse the front of it
        double* red = reinterpret_cast<double*>(smem);   // [4 waves][COB][2]
#pragma unroll
        for (int t = 0; t < CO_T; ++t) {
            double a = s1[t], b = s2[t];   // lanes l, l+16, l+32, l+48 hold the same channel
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

// pack_cot = co tiles per packed group (the filter's layout), CO_T = co tiles per block (any of 1, 2, 4, 8, 16)
template <int KS, int CO_T, int TYP>
static int launch_conv_t(const float* x, int x_ctot, int x_coff, int Cin, const float* wpk, int pack_cot, const float* bias,
                         const float* in_scale, const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff,
                         int Cout, double* stats, int accumulate, int N, int H, int W, int pipe, hipStream_t s) {
    using Cfg = ConvCfg<KS, TYP>;
    constexpr int COB = CO_T * 16, COBP = co_stride_padded(COB);
    const int tiles_x = (W + CV_TX - 1) / CV_TX, tiles_y = (H + Cfg::TY - 1) / Cfg::TY;
    const int pack_cob = pack_cot * 16, groups = (Cout + pack_cob - 1) / pack_cob;
    const int n_chunks = (Cin + Cfg::CI_CHUNK - 1) / Cfg::CI_CHUNK;
    const size_t lds = sizeof(float) * ((size_t)Cfg::CI_CHUNK * Cfg::PS + (size_t)KS * KS * Cfg::CI_CHUNK * COBP + 2 * (size_t)n_chunks * Cfg::CI_CHUNK);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<KS, CO_T, TYP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (lds > 160 * 1024 || lds < sizeof(double) * 8 * COB) return CD_ERR_UNSUPPORTED;   // (the statistics reduction reuses 8*COB doubles)
    const int slices = (Cout + COB - 1) / COB;   // channel slices with at least one live channel
    const int tiles_img = tiles_x * tiles_y, tiles_total = tiles_img * N, chunk = (tiles_total + 7) / 8;
    hipLaunchKernelGGL((conv_fwd_kernel<KS, CO_T, TYP>), dim3((unsigned)chunk * 8u * (unsigned)slices), dim3(kBlock), lds, s, x, x_ctot,
                       x_coff, Cin, wpk, co_stride_padded(pack_cob), pack_cot, groups * pack_cot, bias, in_scale, in_shift, in_relu, y,
                       y_ctot, y_coff, Cout, stats, accumulate, H, W, tiles_x, tiles_img, tiles_total, chunk, slices, pipe);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// (pick_co_tiles: 16-wide output-channel tiles per packed group, conv_split.h)

// (tile rows, co tiles per block) for one launch when the caller does not say (cd_conv2d_fwd_cfg lets a caller that
// has timed the candidates choose -- the hourglass engine does, once per distinct shape).  Rules read off
// tools/conv_sweep.py on the hourglass shapes (profiles/conv_sweep_r01.txt): large images want tall tiles and wide
// channel slices (operand reuse); from 96x56 down there are too few tiles to fill 256 CUs, so a block takes
// a single 16-channel slice and a short tile.
// widest channel slice a workgroup may take: the packed group for k > 1; for 1x1 (a pure GEMM whose input would
// otherwise be re-read once per 64-channel group) up to 8 or 16 tiles, i.e. ALL output channels up to 256
static inline int max_co_tiles(int ks, int Cout) {
    const int pack_cot = pick_co_tiles(ks, Cout);
    if (ks != 1) return pack_cot;
    const int need = (Cout + 15) / 16;
    return need > 8 ? 16 : (need > 4 ? 8 : pack_cot);
}

static inline void pick_conv_tile(int ks, int pack_cot, int Cout, int N, int H, int W, int* ty_out, int* cot_out) {
    const long long px = (long long)N * H * W;
    const long long L1 = 8LL * 192 * 112, L2 = 8LL * 96 * 56, L3 = 8LL * 48 * 28;
    int ty, cot;
    if (ks == 1) {   // measured: 4-row tiles with the full packed group win nearly everywhere; the wide (8/16-tile) slices do not
        cot = px > L3 ? pack_cot : (pack_cot < 2 ? pack_cot : 2);
        // many output channels on few pixels (the MiDaS / ResNeXt bottlenecks at 24x24 and 12x12: 512 .. 2048 channels; the hourglass has
        // at most 256): with two channel tiles per workgroup the input is re-read once per 32 output channels -- the full packed group
        // again (measured on the configs[4] step: 58.0 -> 60.1 pairs/s; 8 tiles 59.5, 16 tiles 55.4).  Same bits for every launch shape.
        if (Cout >= 512) cot = pack_cot;
        ty = 4;
    } else {
        cot = px > L2 ? (pack_cot < 2 ? pack_cot : 2) : 1;
        ty = px > L2 ? ((ks == 7) ? 8 : 16) : (px > L3 ? 8 : 4);
    }
    (void)L1;
    *ty_out = ty;
    *cot_out = cot;
}

// All filters of a network in ONE launch: blockIdx.y selects the descriptor, blockIdx.x grid-strides the elements
// of its destination.  Several descriptors may target ONE packed filter (a fused convolution whose output --
// or, for the transposed/dgrad form, input -- channels are the concatenation of several nn.Conv2d weights):
// each writes only the (oc, ic) range it owns; padding elements are zeroed once when the arena is allocated.
__global__ void pack_weights_table_kernel(const PackDesc* __restrict__ table) {
    const PackDesc d = table[blockIdx.y];
    const int KS = d.ks, OC = d.OC, IC = d.IC;
    const int oc_n = d.transposed ? d.Cin : d.Cout, ic_n = d.transposed ? d.Cout : d.Cin;  // this source's extent
    const int cot = pick_co_tiles(KS, OC), cob = cot * 16, cobp = co_stride_padded(cob);
    const int ci_chunk = KS >= 7 ? 4 : (KS == 1 ? 32 : 8), taps = KS * KS;
    const int n_chunks = (IC + ci_chunk - 1) / ci_chunk, groups = (OC + cob - 1) / cob;
    const size_t total = (size_t)groups * n_chunks * taps * ci_chunk * cobp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int j = (int)(r % cobp); r /= cobp;
        const int cc = (int)(r % ci_chunk); r /= ci_chunk;
        const int tap = (int)(r % taps); r /= taps;
        const int chunk = (int)(r % n_chunks); r /= n_chunks;
        const int grp = (int)r;
        if (j >= cob) continue;
        const int oc = grp * cob + j - d.oc_off, ic = chunk * ci_chunk + cc - d.ic_off;
        if ((unsigned)oc >= (unsigned)oc_n || (unsigned)ic >= (unsigned)ic_n) continue;
        const int ky = tap / KS, kx = tap - ky * KS;
        d.packed[i] = d.transposed ? d.w[(((size_t)ic * d.Cin + oc) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)]
                                   : d.w[(((size_t)oc * d.Cin + ic) * KS + ky) * KS + kx];
    }
}

}  // namespace cd

namespace cd {
static int g_force_conv_ty = 0, g_force_conv_cot = 0, g_conv_pipe = 1;
// arithmetic of the convolutions: 0 = the fp32 matrix instruction, 1 = split-bf16 for k >= 3 (conv_split.hip, wgrad_split.hip),
// 2 = 1 plus the 1x1 forward / input gradient (conv1x1_split.hip).  Start-up value from CD_AMD_CONV_ARITH
// ("fp32" / "split3" / "split"), default split = 2.
static int initial_conv_arith() {
    const char* e = getenv("CD_AMD_CONV_ARITH");
    if (e && (!strcmp(e, "fp32") || !strcmp(e, "0"))) return 0;
    if (e && (!strcmp(e, "split3") || !strcmp(e, "1"))) return 1;
    return 2;
}
static int g_conv_arith = initial_conv_arith();
// CD_AMD_CONV1X1_KC=0: wide 1x1 filters (>= 512 channels) back on the staged fp32 kernel (A/B of conv1x1_split_kc_kernel, round 6)
static const bool g_conv1x1_kc = [] { const char* e = getenv("CD_AMD_CONV1X1_KC"); return !(e && e[0] == '0'); }();
}

extern "C" {

int cd_debug_force_conv_tile_rows(int ty) {
    if (!(ty == 0 || ty == 4 || ty == 8 || ty == 16)) return CD_ERR_INVALID_ARG;
    cd::g_force_conv_ty = ty;
    return CD_OK;
}

int cd_debug_force_conv_co_tiles(int cot) {
    if (!(cot == 0 || cot == 1 || cot == 2 || cot == 4 || cot == 8 || cot == 16)) return CD_ERR_INVALID_ARG;
    cd::g_force_conv_cot = cot;
    return CD_OK;
}

int cd_debug_set_conv_pipeline(int on) {
    cd::g_conv_pipe = on ? 1 : 0;
    return CD_OK;
}

int cd_set_conv_arith(int mode) {
    if (!(mode == 0 || mode == 1 || mode == 2)) return CD_ERR_INVALID_ARG;
    cd::g_conv_arith = mode;
    return CD_OK;
}

int cd_get_conv_arith(void) { return cd::g_conv_arith; }

// fp32 layout, followed (where a split kernel exists: k >= 3, and 1x1 with > 16 output channels) by its bf16 layout: both are
// always packed, the launch picks one
size_t cd_conv2d_packed_weight_floats(int Cout, int Cin, int ks, int transposed) {
    if (Cout <= 0 || Cin <= 0 || !(ks == 1 || ks == 3 || ks == 5 || ks == 7 || ks == 11)) return 0;
    const int OC = transposed ? Cin : Cout, IC = transposed ? Cout : Cin;
    return cd::fp32_packed_floats(OC, IC, ks) + (cd::split_1x1_supported(ks, OC, IC) ? cd::split_1x1_packed_floats(OC, IC) : cd::split_packed_floats(OC, IC, ks));
}

int cd_conv2d_pack_weights(const float* w, int Cout, int Cin, int ks, int transposed, float* packed, void* stream) {
    if (Cout <= 0 || Cin <= 0 || !(ks == 1 || ks == 3 || ks == 5 || ks == 7 || ks == 11)) return CD_ERR_INVALID_ARG;
    const int OC = transposed ? Cin : Cout, IC = transposed ? Cout : Cin;
    const size_t total = cd::fp32_packed_floats(OC, IC, ks);
    if (!w || !packed || total == 0) return CD_ERR_INVALID_ARG;
    const int cot = cd::pick_co_tiles(ks, OC), cob = cot * 16, cobp = cd::co_stride_padded(cob);
    const int ci_chunk = ks >= 7 ? 4 : (ks == 1 ? 32 : 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cd::pack_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, ks,
                       ci_chunk, cob, cobp, transposed, packed, total);
    CD_CHECK_LAUNCH();
    if (cd::split_1x1_supported(ks, OC, IC)) {
        const size_t nsplit = cd::split_1x1_packed_floats(OC, IC);
        if (hipMemsetAsync(packed + total, 0, nsplit * sizeof(float), (hipStream_t)stream) != hipSuccess) return CD_ERR_LAUNCH;
        return cd::launch_pack_1x1(w, Cout, Cin, transposed, packed + total, (hipStream_t)stream);
    }
    if (cd::split_supported(ks)) {   // the split layout's padding must be zero: the one-filter form clears it itself
        const size_t nsplit = cd::split_packed_floats(OC, IC, ks);
        if (hipMemsetAsync(packed + total, 0, nsplit * sizeof(float), (hipStream_t)stream) != hipSuccess) return CD_ERR_LAUNCH;
        return cd::launch_pack_split(w, Cout, Cin, ks, transposed, packed + total, (hipStream_t)stream);
    }
    return CD_OK;
}

int cd_conv2d_pack_weights_table(const void* table_dev, int n, void* stream) {
    if (!table_dev || n <= 0 || n > 65535) return CD_ERR_INVALID_ARG;
    // (64 workgroups per descriptor since round 6 -- 16 left the largest filters to 4096 threads each, a latency chain at the top of every
    // forward: 180.5 -> 182.1 pairs/s, four alternations on one box, profiles/conv_phases_r06.txt)
    hipLaunchKernelGGL(cd::pack_weights_table_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, (const cd::PackDesc*)table_dev);
    CD_CHECK_LAUNCH();
    const int rc = cd::launch_pack_split_table(table_dev, n, (hipStream_t)stream);
    return rc != CD_OK ? rc : cd::launch_pack_1x1_table(table_dev, n, (hipStream_t)stream);
}

int cd_conv2d_fwd_cfg(const float* x, int x_ctot, int x_coff, int Cin, const float* packed_w, const float* bias,
                  const float* in_scale, const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout,
                  double* stats, int accumulate, int N, int H, int W, int ks, int tile_rows, int co_tiles, void* stream) {
    if (!x || !packed_w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return CD_ERR_INVALID_ARG;
    if (x_coff < 0 || x_coff + Cin > x_ctot || y_coff < 0 || y_coff + Cout > y_ctot) return CD_ERR_INVALID_ARG;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int pack_cot = cd::pick_co_tiles(ks, Cout);
    if (!(tile_rows == 0 || tile_rows == 4 || tile_rows == 8 || tile_rows == 16 || tile_rows == 32)) return CD_ERR_INVALID_ARG;
    if (!(co_tiles == 0 || co_tiles == 1 || co_tiles == 2 || co_tiles == 4 || co_tiles == 8 || co_tiles == 16)) return CD_ERR_INVALID_ARG;
    // (small images -- fewer 32-pixel tiles than the chip has waves -- stay on the staged fp32 kernel: measured equal or faster there)
    if (cd::g_conv_arith == 2 && cd::split_1x1_supported(ks, Cout, Cin) && cd::split_1x1_resident(Cin) && (size_t)N * x_ctot * H * W < ((size_t)1 << 30) &&
        (long long)N * H * ((W + 31) / 32) >= 4096)   // one launch shape: the hints are not used
        return cd::launch_conv1x1_split(x, x_ctot, x_coff, Cin, packed_w + cd::fp32_packed_floats(Cout, Cin, ks), bias, in_scale, in_shift, in_relu,
                                        y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, s);
    // wide filters (>= 512 channels on either side: the ResNeXt-101 encoder of MiDaS; the hourglass has none, its small-image 1x1
    // convolutions keep the staged fp32 kernel and their bits): the chunked split-bf16 kernel, any image size
    if (cd::g_conv_arith == 2 && cd::split_1x1_supported(ks, Cout, Cin) && (Cin >= 512 || Cout >= 512) && cd::g_conv1x1_kc &&
        (size_t)N * x_ctot * H * W < ((size_t)1 << 30) && (size_t)N * y_ctot * H * W < ((size_t)1 << 30) && cd::conv1x1_split_kc_ok(Cin, Cout, N, H, W))
        return cd::launch_conv1x1_split_kc(x, x_ctot, x_coff, Cin, packed_w + cd::fp32_packed_floats(Cout, Cin, ks), bias, in_scale, in_shift, in_relu,
                                           y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, s);
    if (cd::g_conv_arith >= 1 && cd::split_supported(ks) && Cin >= 8) {   // (the 3-channel stem would pad K 8/3-fold: fp32 kernel)
        // launch-shape hints: tile_rows <= 4 -> 4 M-tiles per block, else 8; co_tiles >= 2 -> two 32-column tiles per block (then 4
        // M-tiles).  Unhinted: 4 M-tiles, two column tiles when the filter has them and the image is large (conv_split_bench)
        int sty = cd::g_force_conv_ty ? cd::g_force_conv_ty : tile_rows, scot = cd::g_force_conv_cot ? cd::g_force_conv_cot : co_tiles;
        // unhinted (profiles/conv_sweep_r02.txt): up to 96x56 (x 8 images) two channel chunks per barrier round win everywhere
        // (hint 16: latency chains); above, two column tiles when the filter has them (also with two chunks), else 8 M-tiles
        const bool small = (long long)N * H * W <= 8LL * 96 * 56;
        if (scot == 0) scot = small ? 1 : 2;
        if (sty == 0) sty = (small || cd::split_column_tiles(Cout) >= 2) ? 16 : 8;
        return cd::launch_conv_split(x, x_ctot, x_coff, Cin, packed_w + cd::fp32_packed_floats(Cout, Cin, ks), bias, in_scale, in_shift,
                                     in_relu, y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, ks, sty, scot, s);
    }
    if (tile_rows == 32) return CD_ERR_UNSUPPORTED;       // (a launch shape of the split-bf16 k x k kernels only)
    int ty, cot;
    cd::pick_conv_tile(ks, pack_cot, Cout, N, H, W, &ty, &cot);
    const int max_cot = cd::max_co_tiles(ks, Cout);
    if (tile_rows) ty = tile_rows;
    if (co_tiles) cot = co_tiles < max_cot ? co_tiles : max_cot;
    if (cd::g_force_conv_ty) ty = cd::g_force_conv_ty;
    if (cd::g_force_conv_cot && cd::g_force_conv_cot <= max_cot) cot = cd::g_force_conv_cot;
    if (cot == 16 && ty > 4) ty = 4;    // accumulator budget: 16 channel tiles x (TY/4 x 2) pixel tiles x 4 registers
    if (cot == 8 && ty > 8) ty = 8;
    // conv_fwd_kernel<7, 1, 16> is miscompiled by this toolchain (hipcc 7.2 / gfx950): the register allocator
    // rotates the 8 accumulator tiles through AGPRs around the tap loop and the last element of the last tile comes
    // back wrong (tests/test_conv_gpu.py::test_launch_shapes_are_bit_identical catches it; every other
    // instantiation is bit-identical across launch shapes).  8-row tiles are within a few % on the shapes concerned.
    if (ks == 7 && cot == 1 && ty == 16) ty = 8;
    const int pipe = cd::g_conv_pipe;
#define CD_CONV(K, T, Y) return cd::launch_conv_t<K, T, Y>(x, x_ctot, x_coff, Cin, packed_w, pack_cot, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, pipe, s)
#define CD_CONV_T(K, T)                     \
    {                                       \
        if (ty == 16) CD_CONV(K, T, 16);    \
        if (ty == 8) CD_CONV(K, T, 8);      \
        CD_CONV(K, T, 4);                   \
    }
#define CD_CONV_K(K)                        \
    if (ks == K) {                          \
        if (cot == 1) CD_CONV_T(K, 1)       \
        if (cot == 2) CD_CONV_T(K, 2)       \
        if (cot == 4) CD_CONV_T(K, 4)       \
    }
    CD_CONV_K(1) CD_CONV_K(3) CD_CONV_K(5) CD_CONV_K(7)
    if (ks == 11 && cot == 1) CD_CONV_T(11, 1)
    if (ks == 1 && cot == 8) { if (ty == 8) CD_CONV(1, 8, 8); CD_CONV(1, 8, 4); }
    if (ks == 1 && cot == 16) CD_CONV(1, 16, 4);
#undef CD_CONV_K
#undef CD_CONV_T
#undef CD_CONV
    return CD_ERR_UNSUPPORTED;
}

int cd_conv2d_fwd(const float* x, int x_ctot, int x_coff, int Cin, const float* packed_w, const float* bias,
                  const float* in_scale, const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout,
                  double* stats, int accumulate, int N, int H, int W, int ks, void* stream) {
    return cd_conv2d_fwd_cfg(x, x_ctot, x_coff, Cin, packed_w, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats,
                             accumulate, N, H, W, ks, 0, 0, stream);
}

int cd_conv2d_fwd_multi(const cd_conv_desc* d, int n, int tile_rows, int co_tiles, void* stream) {
    if (!d || n < 1 || n > 4) return CD_ERR_INVALID_ARG;
    if (!(tile_rows == 0 || tile_rows == 4 || tile_rows == 8 || tile_rows == 16 || tile_rows == 32) || !(co_tiles == 0 || co_tiles == 1 || co_tiles == 2)) return CD_ERR_INVALID_ARG;
    if (cd::g_conv_arith < 1) return CD_ERR_UNSUPPORTED;      // the fp32-instruction kernels have no multi-convolution dispatch
    cd::SplitConv c[4];
    for (int i = 0; i < n; ++i) {
        const cd_conv_desc& e = d[i];
        if (!e.x || !e.packed_w || !e.y || e.N <= 0 || e.H <= 0 || e.W <= 0 || e.Cin <= 0 || e.Cout <= 0) return CD_ERR_INVALID_ARG;
        if (e.x_coff < 0 || e.x_coff + e.Cin > e.x_ctot || e.y_coff < 0 || e.y_coff + e.Cout > e.y_ctot) return CD_ERR_INVALID_ARG;
        if ((e.in_scale == nullptr) != (e.in_shift == nullptr)) return CD_ERR_INVALID_ARG;
        if (e.N != d[0].N || e.H != d[0].H || e.W != d[0].W || e.Cout != d[0].Cout) return CD_ERR_INVALID_ARG;   // one launch shape
        if (!cd::split_supported(e.ks) || e.Cin < 8) return CD_ERR_UNSUPPORTED;
        c[i].x = e.x; c[i].wsplit = e.packed_w + cd::fp32_packed_floats(e.Cout, e.Cin, e.ks); c[i].bias = e.bias; c[i].in_scale = e.in_scale;
        c[i].in_shift = e.in_shift; c[i].y = e.y; c[i].stats = e.stats; c[i].x_ctot = e.x_ctot; c[i].x_coff = e.x_coff; c[i].Cin = e.Cin;
        c[i].in_relu = e.in_relu; c[i].y_ctot = e.y_ctot; c[i].y_coff = e.y_coff; c[i].accumulate = e.accumulate; c[i].ks = e.ks;
    }
    int sty = cd::g_force_conv_ty ? cd::g_force_conv_ty : tile_rows, scot = cd::g_force_conv_cot ? cd::g_force_conv_cot : co_tiles;
    const bool small = (long long)d[0].N * d[0].H * d[0].W <= 8LL * 96 * 56;      // (the unhinted rule of cd_conv2d_fwd_cfg)
    if (scot == 0) scot = small ? 1 : 2;
    if (sty == 0) sty = (small || cd::split_column_tiles(d[0].Cout) >= 2) ? 16 : 8;
    return cd::launch_conv_split_multi(c, n, d[0].N, d[0].H, d[0].W, d[0].Cout, sty, scot, (hipStream_t)stream);
}

int cd_conv2d_fwd_grouped(const float* x, int x_ctot, int x_coff, int cin_g, const float* packed_w, size_t packed_group_stride,
                          const float* bias, float* y, int y_ctot, int y_coff, int cout_g, int groups, int accumulate, int N, int H, int W,
                          int ks, void* stream) {
    if (!x || !packed_w || !y || groups <= 0 || cin_g <= 0 || cout_g <= 0 || N <= 0 || H <= 0 || W <= 0) return CD_ERR_INVALID_ARG;
    if (x_coff < 0 || x_coff + groups * cin_g > x_ctot || y_coff < 0 || y_coff + groups * cout_g > y_ctot) return CD_ERR_INVALID_ARG;
    if (groups > 1 && (packed_group_stride < cd_conv2d_packed_weight_floats(cout_g, cin_g, ks, 0) || packed_group_stride % 4)) return CD_ERR_INVALID_ARG;
    if (cd::g_conv_arith >= 1 && cd::split_supported(ks) && cin_g >= 8 && groups <= 65535) {   // ONE launch: blockIdx.y = the group
        const bool small = (long long)N * H * W <= 8LL * 96 * 56;
        cd::ConvGroups grp;
        grp.n = groups; grp.x_stride = cin_g; grp.y_stride = cout_g; grp.w_stride = packed_group_stride / 4;
        return cd::launch_conv_split(x, x_ctot, x_coff, cin_g, packed_w + cd::fp32_packed_floats(cout_g, cin_g, ks), bias, nullptr, nullptr, 0, y,
                                     y_ctot, y_coff, cout_g, nullptr, accumulate, N, H, W, ks, (small || cd::split_column_tiles(cout_g) >= 2) ? 16 : 8,
                                     small ? 1 : 2, (hipStream_t)stream, grp);
    }
    for (int g = 0; g < groups; ++g) {   // other arithmetic modes / filter sizes: the dense kernels, group by group
        const int rc = cd_conv2d_fwd(x, x_ctot, x_coff + g * cin_g, cin_g, packed_w + (size_t)g * packed_group_stride,
                                     bias ? bias + g * cout_g : nullptr, nullptr, nullptr, 0, y, y_ctot, y_coff + g * cout_g, cout_g, nullptr,
                                     accumulate, N, H, W, ks, stream);
        if (rc != CD_OK) return rc;
    }
    return CD_OK;
}

int cd_conv2d_packed_co_tiles(int Cout, int ks) {
    if (cd::g_conv_arith >= 1 && cd::split_supported(ks)) return cd::split_column_tiles(Cout) >= 2 ? 2 : 1;
    return cd::max_co_tiles(ks, Cout);
}

}  // extern "C"
