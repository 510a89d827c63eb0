// Weight gradient of the stride-1 "same" convolution on the gfx950 matrix cores, exact fp32.
//
//   dW[co][ci][ky][kx] = sum_{n,y,x} dY[n][co][y][x] * act(X)[n][ci][y+ky-P][x+kx-P]
//
// (the nn.Conv2d weight-gradient launches of autograd for the hourglass the reference trains,
// /root/reference/depth_fine_tuning.py:282 loss.backward()).  GEMM view per filter tap:
// M = 16 output channels, N = 16 input channels, K = pixels; v_mfma_f32_16x16x4_f32 consumes 4
// consecutive pixels of a row per instruction:
//   A[i = lane&15][k = lane>>4] = dY[co0+i][y][x0+k]            (LDS, channel planes 2 banks apart)
//   B[k = lane>>4][j = lane&15] = act(X)[ci0+j][y+ky][x0+k+kx]  (LDS input tile with halo)
//   D: lane holds ci0+(lane&15), co0 + 4*(lane>>4) + {0..3}.
// A block stages a TY x 32 pixel tile of dY (CO_T*16 channels) and of the input (CI_T*16 channels,
// + halo) in LDS; its 4 waves partition the accumulator set (taps for k >= 5, co x ci tiles for k <= 3 -- never
// the pixels) and keep one accumulator tile per owned (tap, co tile, ci tile) in registers while the block walks all the
// image tiles assigned to it (grid-stride over (image, tile)); every block ("split" s of its channel group) stores its
// partial sums ONCE, with plain stores, into ITS OWN slice of the workspace [split][co grp][ci grp][tap][co][ci]; the unpack
// kernels add the slices in split order (fp64 accumulator) while transposing into the [Cout][Cin][k][k] gradient.
// No atomics, no zero-initialised workspace, and the sum has ONE fixed order: the weight gradient is bit-reproducible
// (round 1 flushed with fp32 atomics -- memory-side on MI355X, ~4x the cost of a store, and order-dependent).
#include "cd_common.h"
#include "conv_split.h"
#include "wgrad_split.h"

namespace cd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_TX = 32;

// The 4 waves of a block partition the accumulator set (tap, co tile, ci tile) -- never the pixels, so no two
// waves hold partial sums of the same dW element (no redundant flush):
//   k >= 5 : taps round-robin over the 4 waves;   k <= 3 : 2 x 2 over (co tiles, ci tiles) when both have >= 2 tiles,
//   else taps (k = 3) / whichever tile dimension has 4 (k = 1).
template <int KS, int CO_T, int CI_T> struct WgSplit {
    static constexpr bool by_tiles = (KS <= 3) && (CO_T >= 2) && (CI_T >= 2);
    static constexpr bool by_co4 = (KS == 1) && !by_tiles && (CO_T >= 4);
    static constexpr bool by_ci4 = (KS == 1) && !by_tiles && !by_co4 && (CI_T >= 4);
    static constexpr int NW_A = by_tiles ? 2 : (by_co4 ? 4 : 1);     // waves over co tiles
    static constexpr int NW_C = by_tiles ? 2 : (by_ci4 ? 4 : 1);     // waves over ci tiles
    static constexpr int NW_T = 4 / (NW_A * NW_C);                   // waves over taps
    static constexpr int TAPS = KS * KS;
    static constexpr int TPW = (TAPS + NW_T - 1) / NW_T;             // taps per wave
    static constexpr int APW = CO_T / NW_A, CPW = CI_T / NW_C;       // tiles per wave
};

template <int KS, int CO_T = 1, int CI_T = 1> struct WgCfg {
    // tile rows.  1x1: a smaller tile (2 blocks per CU); the wide 1x1 shapes (all of dY's and X's channels in one block,
    // so both are read exactly once) keep (CO_T + CI_T) * 16 channel planes in LDS and take 2 rows
    static constexpr int TY = (KS == 1) ? ((CO_T + CI_T > 16) ? 2 : 4) : 8;
    static constexpr int TAPS = KS * KS;
    static constexpr int RS = WG_TX + KS - 1, ROWS = TY + KS - 1;
    // physical row of the input tile in LDS: the 16-byte aligned superset [X0 - PADL, X0 + 32 + PADL) of the logical
    // [X0 - P, X0 + 32 + P), so that rows are staged with 16-byte global loads; logical column c sits at c + COFF
    static constexpr int PADL = (((KS - 1) / 2) + 3) & ~3;
    static constexpr int RSP = WG_TX + 2 * PADL, COFF = PADL - (KS - 1) / 2;
    static constexpr int PS_IN_RAW = ROWS * RSP;
    static constexpr int PS_IN = PS_IN_RAW + ((2 - (PS_IN_RAW % 32)) + 32) % 32;   // == 2 (mod 32)
    static constexpr int PS_DY = TY * WG_TX + 2;                                     // == 2 (mod 32)
};

template <int KS, int CO_T, int CI_T>
__global__ __launch_bounds__(kBlock) void conv_wgrad_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    const float* __restrict__ dy, int dy_ctot, int dy_coff, int Cout,
    float* __restrict__ dw_packed, int N, int H, int W, int tiles_x, int tiles_y, int dbg) {
    using Cfg = WgCfg<KS, CO_T, CI_T>;
    using Sp = WgSplit<KS, CO_T, CI_T>;
    constexpr int P = (KS - 1) / 2, TAPS = Cfg::TAPS, TPW = Sp::TPW, NW_T = Sp::NW_T, NW_A = Sp::NW_A, NW_C = Sp::NW_C;
    constexpr int APW = Sp::APW, CPW = Sp::CPW;
    constexpr int RS = Cfg::RS, ROWS = Cfg::ROWS, PSI = Cfg::PS_IN, PSD = Cfg::PS_DY, WG_TY = Cfg::TY;
    constexpr int RSP = Cfg::RSP, COFF = Cfg::COFF, PADL = Cfg::PADL;
    constexpr int COB = CO_T * 16, CIB = CI_T * 16;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_dy = smem;               // [COB][PSD]
    float* s_in = smem + COB * PSD;   // [CIB][PSI]

    const int cig = blockIdx.y, cog = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wt = wid % NW_T, wa = (wid / NW_T) % NW_A, wc = wid / (NW_T * NW_A);   // this wave's tap / co-tile / ci-tile slot
    const size_t HW = (size_t)H * W;
    const int items = N * tiles_x * tiles_y;

    f32x4 acc[TPW][APW][CPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int a = 0; a < APW; ++a)
#pragma unroll
            for (int c = 0; c < CPW; ++c) acc[t][a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_lane = (lane & 15) * PSD + (lane >> 4);   // dY fragment: channel i = lane&15, pixel k = lane>>4
    const int b_lane = (lane & 15) * PSI + (lane >> 4);   // input fragment: channel j = lane&15, pixel k

    // 1x1 (pure GEMM over pixels): software pipeline -- the next tile's global loads are in flight in registers
    // while the MFMAs of the current tile run (the staged bytes per MFMA are highest here)
    constexpr int PF_DY = (KS == 1) ? (COB * WG_TY * (WG_TX / 4) + kBlock - 1) / kBlock : 1;
    constexpr int PF_IN = (KS == 1) ? (CIB * WG_TY * (WG_TX / 4) + kBlock - 1) / kBlock : 1;
    float4 pf_dy[PF_DY], pf_in[PF_IN];
    const bool pipelined = (KS == 1) && ((W & 3) == 0);
    auto pf_load = [&](int item) {
        const int n = item / (tiles_x * tiles_y), tile = item - n * (tiles_x * tiles_y);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int X0 = tx * WG_TX, Y0 = ty * WG_TY;
        const float* dyn = dy + ((size_t)n * dy_ctot + dy_coff) * HW;
        const float* xn = x + ((size_t)n * x_ctot + x_coff) * HW;
#pragma unroll
        for (int q = 0; q < PF_DY; ++q) {
            const int i = threadIdx.x + q * kBlock;
            const int c = i / (WG_TY * (WG_TX / 4)), rem = i - c * (WG_TY * (WG_TX / 4));
            const int r = rem / (WG_TX / 4), col = (rem - r * (WG_TX / 4)) * 4;
            const int co = cog * COB + c, gy = Y0 + r, gx = X0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < COB * WG_TY * (WG_TX / 4) && co < Cout && gy < H && gx < W)
                v = *reinterpret_cast<const float4*>(dyn + (size_t)co * HW + (size_t)gy * W + gx);
            pf_dy[q] = v;
        }
#pragma unroll
        for (int q = 0; q < PF_IN; ++q) {
            const int i = threadIdx.x + q * kBlock;
            const int c = i / (WG_TY * (WG_TX / 4)), rem = i - c * (WG_TY * (WG_TX / 4));
            const int r = rem / (WG_TX / 4), col = (rem - r * (WG_TX / 4)) * 4;
            const int ci = cig * CIB + c, gy = Y0 + r, gx = X0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < CIB * WG_TY * (WG_TX / 4) && ci < Cin && gy < H && gx < W) {
                v = *reinterpret_cast<const float4*>(xn + (size_t)ci * HW + (size_t)gy * W + gx);
                if (in_scale) { const float sc = in_scale[ci], sh = in_shift[ci]; v.x = __fmaf_rn(v.x, sc, sh); v.y = __fmaf_rn(v.y, sc, sh); v.z = __fmaf_rn(v.z, sc, sh); v.w = __fmaf_rn(v.w, sc, sh); }
                if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            pf_in[q] = v;
        }
    };
    auto pf_store = [&]() {
#pragma unroll
        for (int q = 0; q < PF_DY; ++q) {
            const int i = threadIdx.x + q * kBlock;
            if (i < COB * WG_TY * (WG_TX / 4)) {
                const int c = i / (WG_TY * (WG_TX / 4)), rem = i - c * (WG_TY * (WG_TX / 4));
                const int r = rem / (WG_TX / 4), col = (rem - r * (WG_TX / 4)) * 4;
                float* d = s_dy + c * PSD + r * WG_TX + col;
                *reinterpret_cast<float2*>(d) = make_float2(pf_dy[q].x, pf_dy[q].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(pf_dy[q].z, pf_dy[q].w);
            }
        }
#pragma unroll
        for (int q = 0; q < PF_IN; ++q) {
            const int i = threadIdx.x + q * kBlock;
            if (i < CIB * WG_TY * (WG_TX / 4)) {
                const int c = i / (WG_TY * (WG_TX / 4)), rem = i - c * (WG_TY * (WG_TX / 4));
                const int r = rem / (WG_TX / 4), col = (rem - r * (WG_TX / 4)) * 4;
                float* d = s_in + c * PSI + r * RS + col;
                *reinterpret_cast<float2*>(d) = make_float2(pf_in[q].x, pf_in[q].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(pf_in[q].z, pf_in[q].w);
            }
        }
    };
    if (pipelined && (int)blockIdx.x < items) pf_load(blockIdx.x);

    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        if (pipelined) {
            __syncthreads();
            pf_store();
            if (item + (int)gridDim.x < items) pf_load(item + gridDim.x);
            __syncthreads();
        } else {
        const int n = item / (tiles_x * tiles_y), tile = item - n * (tiles_x * tiles_y);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int X0 = tx * WG_TX, Y0 = ty * WG_TY;
        __syncthreads();
        // ---- stage dY tile (zero outside the image / beyond Cout)
        const float* dyn = dy + ((size_t)n * dy_ctot + dy_coff) * HW;
        if ((W & 3) == 0) {  // 16-byte global loads (rows of the tile are 32 contiguous pixels)
            for (int i = threadIdx.x; i < COB * WG_TY * (WG_TX / 4); i += kBlock) {
                const int c = i / (WG_TY * (WG_TX / 4)), rem = i - c * (WG_TY * (WG_TX / 4));
                const int r = rem / (WG_TX / 4), col = (rem - r * (WG_TX / 4)) * 4;
                const int co = cog * COB + c, gy = Y0 + r, gx = X0 + col;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co < Cout && gy < H && gx < W) v = *reinterpret_cast<const float4*>(dyn + (size_t)co * HW + (size_t)gy * W + gx);
                float* d = s_dy + c * PSD + r * WG_TX + col;   // plane stride is 2 (mod 4): 8-byte aligned only
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
            }
        } else
        for (int i = threadIdx.x; i < COB * WG_TY * WG_TX; i += kBlock) {
            const int c = i / (WG_TY * WG_TX), rem = i - c * (WG_TY * WG_TX);
            const int r = rem / WG_TX, col = rem - r * WG_TX;
            const int co = cog * COB + c, gy = Y0 + r, gx = X0 + col;
            float v = 0.f;
            if (co < Cout && gy < H && gx < W) v = dyn[(size_t)co * HW + (size_t)gy * W + gx];
            s_dy[c * PSD + r * WG_TX + col] = v;
        }
        // ---- stage activated input tile with halo (zero padding)
        const float* xn = x + ((size_t)n * x_ctot + x_coff) * HW;
        if (KS == 1 && (W & 3) == 0) {
            for (int i = threadIdx.x; i < CIB * ROWS * (RS / 4); i += kBlock) {
                const int c = i / (ROWS * (RS / 4)), rem = i - c * (ROWS * (RS / 4));
                const int r = rem / (RS / 4), col = (rem - r * (RS / 4)) * 4;
                const int ci = cig * CIB + c, gy = Y0 + r, gx = X0 + col;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < Cin && gy < H && gx < W) {
                    v = *reinterpret_cast<const float4*>(xn + (size_t)ci * HW + (size_t)gy * W + gx);
                    if (in_scale) { const float sc = in_scale[ci], sh = in_shift[ci]; v.x = __fmaf_rn(v.x, sc, sh); v.y = __fmaf_rn(v.y, sc, sh); v.z = __fmaf_rn(v.z, sc, sh); v.w = __fmaf_rn(v.w, sc, sh); }
                    if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                float* d = s_in + c * PSI + r * RS + col;
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
            }
        } else if ((W & 3) == 0) {
            // halo tile through 16-byte loads of the aligned superset (W % 4 == 0: an aligned float4 is inside or outside as a whole)
            for (int i = threadIdx.x; i < CIB * ROWS * (RSP / 4); i += kBlock) {
                const int c = i / (ROWS * (RSP / 4)), rem = i - c * (ROWS * (RSP / 4));
                const int r = rem / (RSP / 4), q4 = (rem - r * (RSP / 4)) * 4;
                const int ci = cig * CIB + c, gy = Y0 - P + r, gx = X0 - PADL + q4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                    v = *reinterpret_cast<const float4*>(xn + (size_t)ci * HW + (size_t)gy * W + gx);
                    if (in_scale) { const float sc = in_scale[ci], sh = in_shift[ci]; v.x = __fmaf_rn(v.x, sc, sh); v.y = __fmaf_rn(v.y, sc, sh); v.z = __fmaf_rn(v.z, sc, sh); v.w = __fmaf_rn(v.w, sc, sh); }
                    if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                float* d = s_in + c * PSI + r * RSP + q4;   // plane stride is 2 (mod 4): 8-byte aligned only
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
            }
        } else
        for (int i = threadIdx.x; i < CIB * ROWS * RS; i += kBlock) {
            const int c = i / (ROWS * RS), rem = i - c * (ROWS * RS);
            const int r = rem / RS, col = rem - r * RS;
            const int ci = cig * CIB + c, gy = Y0 - P + r, gx = X0 - P + col;
            float v = 0.f;
            if (ci < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                v = xn[(size_t)ci * HW + (size_t)gy * W + gx];
                if (in_scale) v = __fmaf_rn(v, in_scale[ci], in_shift[ci]);  // same fma as the BN backward's mask
                if (in_relu) v = fmaxf(v, 0.f);
            }
            s_in[c * PSI + r * RSP + col + COFF] = v;
        }
        __syncthreads();
        }
        // ---- MFMA: all rows x all 4-pixel groups x this wave's (taps, co tiles, ci tiles)
        if (!(dbg & 2))
#pragma unroll 1
        for (int r = 0; r < WG_TY; ++r) {
#pragma unroll 2
            for (int c4 = 0; c4 < WG_TX / 4; ++c4) {
                float af[APW];
#pragma unroll
                for (int a = 0; a < APW; ++a) af[a] = s_dy[(a * NW_A + wa) * 16 * PSD + r * WG_TX + c4 * 4 + a_lane];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int tap = t * NW_T + wt;   // compile-time stride, wave-uniform offset
                    if (tap < TAPS) {
                        const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
                        for (int c = 0; c < CPW; ++c) {
                            const float bf = s_in[(c * NW_C + wc) * 16 * PSI + (r + ky) * RSP + c4 * 4 + kx + COFF + b_lane];
#pragma unroll
                            for (int a = 0; a < APW; ++a)
                                acc[t][a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf, acc[t][a][c], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- flush: this block's slice, packed [split][cog][cig][tap][COB][CIB]
    if (dbg & 1) return;   // measurement hook (cd_debug_set_wgrad_mode): skip the flush
    const size_t slice = (size_t)gridDim.z * gridDim.y * TAPS * COB * CIB;
    const size_t base = (size_t)blockIdx.x * slice + ((size_t)cog * gridDim.y + cig) * TAPS * COB * CIB;
    const int ci_l = lane & 15, co4 = (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = t * NW_T + wt;
        if (tap < TAPS) {
#pragma unroll
            for (int a = 0; a < APW; ++a)
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    float* dst = dw_packed + base + ((size_t)tap * COB + (a * NW_A + wa) * 16 + co4) * CIB + (c * NW_C + wc) * 16 + ci_l;
                    const f32x4 v = acc[t][a][c];
                    dst[0] = v.x; dst[CIB] = v.y; dst[2 * CIB] = v.z; dst[3 * CIB] = v.w;
                }
        }
    }
}

// the slices of one packed element, added in ONE fixed order (4 interleaved fp64 chains over the splits, combined at the
// end: the loads of a chain do not wait for each other)
__device__ __forceinline__ float sum_splits(const float* __restrict__ p, int splits, size_t stride) {
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int s = 0;
    // (round 6: 16 loads in flight per thread instead of 4 -- the one unpack launch at the end of the backward is a latency chain per
    // element, 0.5 ms on the step's critical path; the additions keep their order: same bits)
    for (; s + 15 < splits; s += 16) {
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = p[(size_t)(s + i) * stride];
#pragma unroll
        for (int i = 0; i < 16; i += 4) { v0 += (double)t[i]; v1 += (double)t[i + 1]; v2 += (double)t[i + 2]; v3 += (double)t[i + 3]; }
    }
    for (; s + 3 < splits; s += 4) {
        const float a = p[(size_t)s * stride], b = p[(size_t)(s + 1) * stride], c = p[(size_t)(s + 2) * stride], d = p[(size_t)(s + 3) * stride];
        v0 += (double)a; v1 += (double)b; v2 += (double)c; v3 += (double)d;
    }
    for (; s < splits; ++s) v0 += (double)p[(size_t)s * stride];
    return (float)((v0 + v1) + (v2 + v3));
}

// One gradient tensor dw[rows][Cin][KS][KS] from the output-channel rows [row0, row0 + rows) of a packed buffer.  Threads
// walk the PACKED order (input channel fastest: 64-byte runs of every slice are read by neighbouring lanes); the transposed
// write into dw happens once per element, the reads `splits` times.
__device__ __forceinline__ void unpack_rows(const float* __restrict__ packed, float* __restrict__ dw, int Cin, int ks, int cob,
                                            int cib, int ci_groups, int row0, int rows, int accumulate, int splits,
                                            size_t split_stride, int first, int step) {
    const int taps = ks * ks, total = rows * Cin * taps;
    for (int j = first; j < total; j += step) {
        const int ci = j % Cin, r = (j / Cin) % rows, tap = j / (Cin * rows);
        const int co = row0 + r, cog = co / cob, cig = ci / cib;
        const float v = sum_splits(packed + (((size_t)cog * ci_groups + cig) * taps + tap) * cob * cib + (size_t)(co - cog * cob) * cib + (ci - cig * cib),
                                   splits, split_stride);
        float* d = dw + ((size_t)r * Cin + ci) * taps + tap;
        *d = accumulate ? *d + v : v;
    }
}

// packed [split][cog][cig][tap][COB][CIB] -> dW[Cout][Cin][KS][KS]  (accumulate = 0: overwrite, 1: add)
__global__ void unpack_wgrad_kernel(const float* __restrict__ packed, int Cout, int Cin, int KS, int COB, int CIB,
                                    int ci_groups, int splits, size_t split_stride, float* __restrict__ dw, int accumulate,
                                    size_t ws_group_stride) {
    // blockIdx.y = the group of a grouped convolution (its own workspace and its own [Cout][Cin][KS][KS] block of dw)
    packed += (size_t)blockIdx.y * ws_group_stride;
    dw += (size_t)blockIdx.y * Cout * Cin * KS * KS;
    unpack_rows(packed, dw, Cin, KS, COB, CIB, ci_groups, 0, Cout, accumulate, splits, split_stride,
                blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------- few input channels (the RGB stem)
// With Cin <= 4 the generic kernel would pad the input channels to a 16-wide N tile (5x wasted MFMAs at Cin = 3).
// Here the N dimension packs (ci, kx) pairs instead -- Cin * KS <= 32 columns = 2 N tiles -- so ONE filter row ky
// of all channels is 2 MFMAs per (co tile, 4 pixels) instead of KS:
//   B[k = pixel][j = (ci, kx)] = act(X)[ci][y+ky][x0+k+kx]      (per-lane LDS offset ci * PSI + kx)
//   D tile (co tile a, row ky, half nt): lane holds column q = nt*16 + (lane&15) -> (ci, kx) = (q / KS, q % KS).
// A block owns 32 output channels; wave w keeps co tile (w & 1), column half (w >> 1), all KS rows: KS tiles.
// Partial sums go to the same packed [cog][0][tap][32][16] buffer as conv_wgrad_kernel<KS, 2, 1>.
template <int KS>
__global__ __launch_bounds__(kBlock) void conv_wgrad_fewcin_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    const float* __restrict__ dy, int dy_ctot, int dy_coff, int Cout,
    float* __restrict__ dw_packed, int N, int H, int W, int tiles_x, int tiles_y) {
    constexpr int P = (KS - 1) / 2, TAPS = KS * KS, TY = 8, CMAX = 4;
    constexpr int RS = WG_TX + KS - 1, ROWS = TY + KS - 1;
    constexpr int PSI = ROWS * RS + 8;                 // planes 8 floats apart (mod 32 it decorrelates the ci groups)
    constexpr int PSD = TY * WG_TX + 2;
    constexpr int COB = 32, CIB = 16;
    static_assert(CMAX * KS <= 32, "two N tiles of (ci, kx) pairs");
    __shared__ float s_dy[COB * PSD];
    __shared__ float s_in[CMAX * PSI];

    const int cog = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wa = wid & 1, nt = wid >> 1;
    const size_t HW = (size_t)H * W;
    const int items = N * tiles_x * tiles_y;
    const int q = nt * 16 + (lane & 15);               // this lane's column: (ci, kx) pair
    const bool q_live = q < Cin * KS;
    const int q_ci = q_live ? q / KS : 0, q_kx = q_live ? q - q_ci * KS : 0;
    const int a_lane = (wa * 16 + (lane & 15)) * PSD + (lane >> 4);
    const int b_lane = q_ci * PSI + q_kx + (lane >> 4);

    f32x4 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int n = item / (tiles_x * tiles_y), tile = item - n * (tiles_x * tiles_y);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int X0 = tx * WG_TX, Y0 = ty * TY;
        __syncthreads();
        const float* dyn = dy + ((size_t)n * dy_ctot + dy_coff) * HW;
        if ((W & 3) == 0) {
            for (int i = threadIdx.x; i < COB * TY * (WG_TX / 4); i += kBlock) {
                const int c = i / (TY * (WG_TX / 4)), rem = i - c * (TY * (WG_TX / 4));
                const int r = rem / (WG_TX / 4), col = (rem - r * (WG_TX / 4)) * 4;
                const int co = cog * COB + c, gy = Y0 + r, gx = X0 + col;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co < Cout && gy < H && gx < W) v = *reinterpret_cast<const float4*>(dyn + (size_t)co * HW + (size_t)gy * W + gx);
                float* d = s_dy + c * PSD + r * WG_TX + col;
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
            }
        } else {
            for (int i = threadIdx.x; i < COB * TY * WG_TX; i += kBlock) {
                const int c = i / (TY * WG_TX), rem = i - c * (TY * WG_TX);
                const int r = rem / WG_TX, col = rem - r * WG_TX;
                const int co = cog * COB + c, gy = Y0 + r, gx = X0 + col;
                float v = 0.f;
                if (co < Cout && gy < H && gx < W) v = dyn[(size_t)co * HW + (size_t)gy * W + gx];
                s_dy[c * PSD + r * WG_TX + col] = v;
            }
        }
        const float* xn = x + ((size_t)n * x_ctot + x_coff) * HW;
        for (int i = threadIdx.x; i < CMAX * ROWS * RS; i += kBlock) {
            const int c = i / (ROWS * RS), rem = i - c * (ROWS * RS);
            const int r = rem / RS, col = rem - r * RS;
            const int gy = Y0 - P + r, gx = X0 - P + col;
            float v = 0.f;
            if (c < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                v = xn[(size_t)c * HW + (size_t)gy * W + gx];
                if (in_scale) v = __fmaf_rn(v, in_scale[c], in_shift[c]);
                if (in_relu) v = fmaxf(v, 0.f);
            }
            s_in[c * PSI + rem] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < TY; ++r) {
#pragma unroll 2
            for (int c4 = 0; c4 < WG_TX / 4; ++c4) {
                const float af = s_dy[r * WG_TX + c4 * 4 + a_lane];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const float bf = s_in[(r + ky) * RS + c4 * 4 + b_lane];
                    acc[ky] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[ky], 0, 0, 0);
                }
            }
        }
    }
    if (!q_live) return;   // (after the last barrier) padding columns carry garbage by construction; the unpack never reads them
    const size_t base = (size_t)blockIdx.x * ((size_t)gridDim.z * TAPS * COB * CIB) + (size_t)cog * TAPS * COB * CIB;
    const int co4 = wa * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        float* dst = dw_packed + base + ((size_t)(ky * KS + q_kx) * COB + co4) * CIB + q_ci;
        const f32x4 v = acc[ky];
        dst[0] = v.x; dst[CIB] = v.y; dst[2 * CIB] = v.z; dst[3 * CIB] = v.w;
    }
}

// Blocks ("splits") per channel group: enough to fill the chip -- ~2 per CU where the LDS tiles allow 2 resident blocks, 1 for
// the wide 1x1 shapes.  ONE definition: the launchers, the workspace size and cd_conv2d_wgrad_plan must agree.
static inline int wgrad_splits(int groups, int per_cu, int items) {
    int splits = (256 * per_cu + groups - 1) / groups;
    if (splits > items) splits = items;
    return splits < 1 ? 1 : splits;
}
static inline int wgrad_items(int N, int H, int W, int ty) { return N * ((W + WG_TX - 1) / WG_TX) * ((H + ty - 1) / ty); }

template <int KS>
static int launch_wgrad_fewcin(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift,
                               int in_relu, const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H,
                               int W, hipStream_t s) {
    const int tiles_x = (W + WG_TX - 1) / WG_TX, tiles_y = (H + 7) / 8;
    const int cogs = (Cout + 31) / 32, items = N * tiles_x * tiles_y;
    const int splits = wgrad_splits(cogs, 3, items);
    hipLaunchKernelGGL((conv_wgrad_fewcin_kernel<KS>), dim3(splits, 1, cogs), dim3(kBlock), 0, s, x, x_ctot, x_coff, Cin, in_scale,
                       in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, tiles_x, tiles_y);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}


static int g_wgrad_wide = 1;  // cd_debug_set_wgrad_mode bit 2 switches the wide 1x1 plan off (A/B measurements, tests)
// measurement hooks: bit 0 skip the atomic flush, bit 1 skip the MFMAs (results are then wrong); bit 4 (CD_AMD_WGRAD_COT1=1, read once:
// workspace sizes depend on it) 16 x 16-channel blocks for every split-bf16 gradient (A/B of the 32 x 16 block of the 3x3 gradient)
static int g_wgrad_dbg = [] { const char* e = getenv("CD_AMD_WGRAD_COT1"); return (e && e[0] == '1') ? 16 : 0; }();

// The same for MANY gradients in one launch (blockIdx.y = descriptor): a network's backward leaves every partial-sum
// buffer packed (cd_conv2d_wgrad accumulate bit 2) and unpacks them all at the end; one descriptor may take only the
// output-channel rows [row0, row0 + rows) of a packed gradient (a fused convolution whose rows belong to several
// nn.Conv2d weights).
struct UnpackDesc {
    const float* packed; float* dw;
    int Cin, ks, cob, cib, ci_groups, row0, rows, accumulate, splits, split_stride;   // split_stride in floats
};
static_assert(sizeof(UnpackDesc) == 56, "cd_unpack_desc layout");

__global__ void unpack_wgrad_table_kernel(const UnpackDesc* __restrict__ table) {
    const UnpackDesc d = table[blockIdx.y];
    unpack_rows(d.packed, d.dw, d.Cin, d.ks, d.cob, d.cib, d.ci_groups, d.row0, d.rows, d.accumulate, d.splits, (size_t)d.split_stride,
                blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

struct WgPlan { int co_t, ci_t; };

static inline WgPlan wgrad_plan(int ks, int cout, int cin) {
    WgPlan p;
    const int co_need = (cout + 15) / 16, ci_need = (cin + 15) / 16;
    if (ks == 11) { p.co_t = 1; p.ci_t = 1; }
    else if (ks == 7) { p.co_t = co_need >= 2 ? 2 : 1; p.ci_t = 1; }
    else if (ks == 5) { p.co_t = co_need >= 2 ? 2 : 1; p.ci_t = 1; }
    else if (ks == 3) { p.co_t = co_need >= 2 ? 2 : 1; p.ci_t = ci_need >= 2 ? 2 : 1; }
    else { p.co_t = co_need >= 4 ? 4 : (co_need >= 2 ? 2 : 1); p.ci_t = ci_need >= 4 ? 4 : (ci_need >= 2 ? 2 : 1); }
    return p;
}

// 1x1 with many channels and many pixels: ONE block holds all (or half) of dY's and X's channels of its pixels, so the
// two tensors are read once instead of once per channel group of the other (measured: the narrow plan spends 56 % of its
// time staging, 2.5 GB of re-reads at 128 -> 208 channels).  Shapes are the instantiated ones; dead tiles are zeros.
static inline bool wgrad_wide_plan(int ks, int cout, int cin, int N, int H, int W, WgPlan* p) {
    if (ks != 1) return false;
    const int co_need = (cout + 15) / 16, ci_need = (cin + 15) / 16;
    if (co_need <= 4 && ci_need <= 4) return false;
    if ((long long)N * ((W + WG_TX - 1) / WG_TX) * ((H + 1) / 2) < 512) return false;   // too few pixel tiles for 256 blocks
    if (co_need > 16 || ci_need > 16) return false;
    // measured (profiles/wgrad_sweep_r01.txt): it pays when the narrow plan (64 x 64 channels per block) would read the
    // tensors >= 6 times in total and the wide shape covers all channels in ONE block
    const int narrow_groups = ((co_need + 3) / 4) * ((ci_need + 3) / 4);
    if (narrow_groups < 6) return false;
    const int co_t = co_need <= 8 ? 8 : (co_need <= 10 ? 10 : (co_need <= 14 ? 14 : 16));
    const int ci_t = ci_need <= 8 ? 8 : 16;
    if (co_t * ci_t > 160) return false;   // accumulator budget: <= 40 tiles per wave
    p->co_t = co_t; p->ci_t = ci_t;
    return true;
}

template <int KS, int CO_T, int CI_T>
static int launch_wgrad_t(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift,
                          int in_relu, const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H,
                          int W, hipStream_t s) {
    using Cfg = WgCfg<KS, CO_T, CI_T>;
    constexpr int COB = CO_T * 16, CIB = CI_T * 16;
    const int tiles_x = (W + WG_TX - 1) / WG_TX, tiles_y = (H + Cfg::TY - 1) / Cfg::TY;
    const int cogs = (Cout + COB - 1) / COB, cigs = (Cin + CIB - 1) / CIB;
    const int items = N * tiles_x * tiles_y;
    // enough blocks to fill the chip (~2 per CU: the LDS tiles allow 2 resident blocks; 1 for the wide 1x1 shapes), few
    // enough that the atomic flush of the partial sums (one per block, all splits hit the same addresses) stays small
    constexpr int per_cu = (CO_T + CI_T > 16) ? 1 : 2;
    const int splits = wgrad_splits(cogs * cigs, per_cu, items);
    const size_t lds = sizeof(float) * ((size_t)COB * Cfg::PS_DY + (size_t)CIB * Cfg::PS_IN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<KS, CO_T, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (lds > 160 * 1024) return CD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, CO_T, CI_T>), dim3(splits, cigs, cogs), dim3(kBlock), lds, s, x, x_ctot, x_coff,
                       Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, tiles_x, tiles_y, g_wgrad_dbg);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// Packed layout and launch shape of one weight gradient -- ONE definition for the launchers, the workspace size,
// cd_conv2d_wgrad_plan and the unpack descriptors.
struct WgLayout { int cob, cib, cogs, cigs, splits, max_splits, fewcin, wide, split_arith, split1x1, blocks_x, cot; size_t slice; };

// the split-bf16 kernel (wgrad_split.hip) takes the k = 3, 5, 7, 11 gradients when that arithmetic is selected (cd_set_conv_arith),
// except the RGB stem (3 input channels would pad a 16-wide tile 5-fold: the few-input-channel fp32 kernel stays)
static inline bool wgrad_uses_split(int ks, int Cin) { return cd_get_conv_arith() >= 1 && split_supported(ks) && Cin >= 8; }

// 1x1 under arithmetic mode 2: wgrad1x1_split.hip (64 x 128 patches per wave, packed [split][cog][cig][64][128])
static WgLayout wgrad_layout_split1x1(int Cout, int Cin, long long steps) {
    WgLayout L;
    L.wide = 0; L.fewcin = 0; L.split_arith = 0; L.split1x1 = 1;
    L.cob = WGRAD1X1_COB; L.cib = WGRAD1X1_CIB;
    int pg, sub, groups;
    wgrad1x1_split_shape(Cout, Cin, &L.cogs, &L.cigs, &pg, &sub, &groups);
    L.slice = (size_t)L.cogs * L.cigs * L.cob * L.cib;
    L.max_splits = wgrad1x1_split_blocks(Cout, Cin, 1LL << 40) * sub;
    L.blocks_x = wgrad1x1_split_blocks(Cout, Cin, steps);
    L.splits = L.blocks_x * sub;
    return L;
}

static WgLayout wgrad_layout_split(int Cout, int Cin, int ks, int N, int H, int W) {
    WgLayout L;
    L.wide = 0; L.fewcin = 0; L.split_arith = 1; L.split1x1 = 0; L.blocks_x = 0;
    L.cob = 16; L.cib = 16;
    L.cogs = (Cout + 15) / 16; L.cigs = (Cin + 15) / 16;
    L.slice = (size_t)L.cogs * L.cigs * ks * ks * 256;
    const int per_cu = wgrad_split_blocks_per_cu(ks);
    // 16-channel output groups per block (the packed tiles stay 16 x 16): two for the 3x3 gradient when that still fills the chip
    // -- measured (profiles/wgrad3x3_r03.txt): 64 -> 32 @192x112 122 -> 99 us, 64 -> 64 @96x56 59 -> 47 us, but 32 -> 32 @96x56
    // 30 -> 38 us with only 320 blocks left
    const int zg1 = L.cogs * L.cigs, zg2 = ((L.cogs + 1) / 2) * L.cigs;
    const bool may2 = !(g_wgrad_dbg & 16) && wgrad_split_cot(ks, Cout) == 2;
    L.cot = 1;
    if (may2 && N > 0) {
        const int items = wgrad_items(N, H, W, wgrad_split_tile_rows(ks));
        if (zg2 * wgrad_splits(zg2, per_cu, items) >= 384) L.cot = 2;
    }
    const int zgroups = L.cot == 2 ? zg2 : zg1;
    L.max_splits = wgrad_splits(may2 ? zg2 : zg1, per_cu, 1 << 30);      // (workspace: room for either choice)
    L.splits = N > 0 ? wgrad_splits(zgroups, per_cu, wgrad_items(N, H, W, wgrad_split_tile_rows(ks))) : L.max_splits;
    return L;
}

// x_ctot / dy_ctot: channel counts of the buffers the operands are slices of (0 = unknown: the 1x1 split kernel's 32-bit offsets are
// then checked against Cin / Cout only, and again at launch)
static WgLayout wgrad_layout(int Cout, int Cin, int ks, int N, int H, int W, int arith = -1, int x_ctot = 0, int dy_ctot = 0) {
    const int mode = arith < 0 ? cd_get_conv_arith() : arith;
    if (mode >= 1 && split_supported(ks) && Cin >= 8) return wgrad_layout_split(Cout, Cin, ks, N, H, W);
    if (mode == 2 && ks == 1 && N > 0 && wgrad1x1_split_ok(Cout, Cin, N, H, W, x_ctot > Cin ? x_ctot : Cin, dy_ctot > Cout ? dy_ctot : Cout))
        return wgrad_layout_split1x1(Cout, Cin, (long long)N * H * W / 16);
    WgLayout L;
    L.split_arith = 0; L.split1x1 = 0; L.blocks_x = 0;
    WgPlan p = wgrad_plan(ks, Cout, Cin);
    L.wide = (N > 0 && g_wgrad_wide && wgrad_wide_plan(ks, Cout, Cin, N, H, W, &p)) ? 1 : 0;
    L.fewcin = (ks == 7 && Cin <= 4 && p.co_t == 2 && !(g_wgrad_dbg & 8)) ? 1 : 0;
    L.cob = p.co_t * 16; L.cib = p.ci_t * 16;
    L.cogs = (Cout + L.cob - 1) / L.cob; L.cigs = L.fewcin ? 1 : (Cin + L.cib - 1) / L.cib;
    L.slice = (size_t)L.cogs * L.cigs * ks * ks * L.cob * L.cib;
    const int per_cu = L.fewcin ? 3 : ((p.co_t + p.ci_t > 16) ? 1 : 2);
    const int ty = (ks == 1) ? ((p.co_t + p.ci_t > 16) ? 2 : 4) : 8;
    L.max_splits = wgrad_splits(L.cogs * L.cigs, per_cu, 1 << 30);
    L.splits = N > 0 ? wgrad_splits(L.cogs * L.cigs, per_cu, wgrad_items(N, H, W, ty)) : L.max_splits;
    return L;
}

}  // namespace cd

extern "C" {

int cd_debug_set_wgrad_mode(int bits) {
    cd::g_wgrad_dbg = (bits & 11) | (cd::g_wgrad_dbg & 16);   // bit 3: generic kernel for the few-input-channel (stem) case; bit 4 is process-wide (below)
    cd::g_wgrad_wide = (bits & 4) ? 0 : 1;
    return CD_OK;
}

size_t cd_conv2d_wgrad_workspace_floats(int Cout, int Cin, int ks) {
    if (Cout <= 0 || Cin <= 0 || !(ks == 1 || ks == 3 || ks == 5 || ks == 7 || ks == 11)) return 0;
    // every block owns a slice: the largest number of blocks per channel group the launcher can choose, for either 1x1 layout
    const cd::WgLayout a = cd::wgrad_layout(Cout, Cin, ks, 0, 0, 0);
    size_t n = a.slice * (size_t)a.max_splits;
    if (ks == 1 && Cin >= 32 && Cout >= 96) {   // the 1x1 split layout (arithmetic mode 2, chosen per launch by the image size)
        const cd::WgLayout b = cd::wgrad_layout_split1x1(Cout, Cin, 1LL << 40);
        if (b.slice * (size_t)b.max_splits > n) n = b.slice * (size_t)b.max_splits;
    }
    if (cd::split_supported(ks) && Cin >= 8) {   // either arithmetic may be selected later: room for both layouts
        const cd::WgLayout b = cd::wgrad_layout(Cout, Cin, ks, 0, 0, 0, cd_get_conv_arith() >= 1 ? 0 : 1);
        if (b.slice * (size_t)b.max_splits > n) n = b.slice * (size_t)b.max_splits;
    }
    cd::WgPlan w;
    if (cd::wgrad_wide_plan(ks, Cout, Cin, 1 << 20, 2, 32, &w)) {   // the wide 1x1 layout (chosen per launch by the image size)
        const int wob = w.co_t * 16, wib = w.ci_t * 16;
        const size_t groups = (size_t)((Cout + wob - 1) / wob) * ((Cin + wib - 1) / wib);
        const size_t m = groups * wob * wib * (size_t)cd::wgrad_splits((int)groups, (w.co_t + w.ci_t > 16) ? 1 : 2, 1 << 30);
        if (m > n) n = m;
    }
    return n;
}

int cd_conv2d_wgrad_plan(int Cout, int Cin, int ks, int N, int H, int W, int* cob, int* cib, int* splits) {
    if (!cob || !cib || !splits || cd_conv2d_wgrad_workspace_floats(Cout, Cin, ks) == 0 || N <= 0 || H <= 0 || W <= 0) return CD_ERR_INVALID_ARG;
    const cd::WgLayout L = cd::wgrad_layout(Cout, Cin, ks, N, H, W);
    *cob = L.cob;
    *cib = L.cib;
    *splits = L.splits;
    return CD_OK;
}

int cd_conv2d_wgrad_unpack_table(const void* table_dev, int n, void* stream) {
    if (!table_dev || n <= 0 || n > 65535) return CD_ERR_INVALID_ARG;
    // 48 workgroups per gradient tensor: the reads of the per-workgroup slices (2 GB per hourglass step) need the whole chip
    hipLaunchKernelGGL(cd::unpack_wgrad_table_kernel, dim3(48, n), dim3(256), 0, (hipStream_t)stream, (const cd::UnpackDesc*)table_dev);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

int cd_conv2d_wgrad(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift,
                    int in_relu, const float* dy, int dy_ctot, int dy_coff, int Cout, float* dw, int accumulate,
                    float* workspace, int N, int H, int W, int ks, void* stream) {
    if (!x || !dy || (!dw && !(accumulate & 4)) || !workspace || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return CD_ERR_INVALID_ARG;
    if (x_coff < 0 || x_coff + Cin > x_ctot || dy_coff < 0 || dy_coff + Cout > dy_ctot) return CD_ERR_INVALID_ARG;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return CD_ERR_INVALID_ARG;
    const size_t wsf = cd_conv2d_wgrad_workspace_floats(Cout, Cin, ks);
    if (wsf == 0) return CD_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    // (the workspace needs no initialisation: every block writes its whole slice; bit 1 of `accumulate` is accepted and ignored)
    cd::WgPlan p = cd::wgrad_plan(ks, Cout, Cin);
    const bool wide = cd::g_wgrad_wide && cd::wgrad_wide_plan(ks, Cout, Cin, N, H, W, &p);
    cd::WgLayout L = cd::wgrad_layout(Cout, Cin, ks, N, H, W);
    int rc = CD_ERR_UNSUPPORTED;
    if (L.split1x1 && !cd::wgrad1x1_split_ok(Cout, Cin, N, H, W, x_ctot, dy_ctot)) return CD_ERR_UNSUPPORTED;   // (buffers too large for 32-bit offsets)
    if (L.split1x1) {
        rc = cd::launch_wgrad1x1_split(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, workspace, N, H, W,
                                       L.blocks_x, s);
        if (rc != CD_OK || (accumulate & 4)) return rc;
        const int total = Cout * Cin;
        hipLaunchKernelGGL(cd::unpack_wgrad_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0, s, workspace,
                           Cout, Cin, 1, L.cob, L.cib, L.cigs, L.splits, L.slice, dw, accumulate & 1, (size_t)0);
        CD_CHECK_LAUNCH();
        return CD_OK;
    }
    if (L.split_arith) {
        rc = cd::launch_wgrad_split(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, workspace, N, H, W, ks,
                                    L.splits, s, 1, 0, L.cot);
        if (rc != CD_OK || (accumulate & 4)) return rc;
        const int total = Cout * Cin * ks * ks;
        hipLaunchKernelGGL(cd::unpack_wgrad_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0, s, workspace,
                           Cout, Cin, ks, L.cob, L.cib, L.cigs, L.splits, L.slice, dw, accumulate & 1, (size_t)0);
        CD_CHECK_LAUNCH();
        return CD_OK;
    }
#define CD_WG(K, A, C) rc = cd::launch_wgrad_t<K, A, C>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, workspace, N, H, W, s)
    if (ks == 11) CD_WG(11, 1, 1);
    else if (ks == 7 && Cin <= 4 && p.co_t == 2 && !(cd::g_wgrad_dbg & 8))   // the RGB stem
        rc = cd::launch_wgrad_fewcin<7>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, workspace, N, H, W, s);
    else if (ks == 7) { if (p.co_t == 2) CD_WG(7, 2, 1); else CD_WG(7, 1, 1); }
    else if (ks == 5) { if (p.co_t == 2) CD_WG(5, 2, 1); else CD_WG(5, 1, 1); }
    else if (ks == 3) {
        if (p.co_t == 2 && p.ci_t == 2) CD_WG(3, 2, 2);
        else if (p.co_t == 2) CD_WG(3, 2, 1);
        else if (p.ci_t == 2) CD_WG(3, 1, 2);
        else CD_WG(3, 1, 1);
    } else if (wide) {
        if (p.co_t == 8 && p.ci_t == 16) CD_WG(1, 8, 16);
        else if (p.co_t == 10 && p.ci_t == 8) CD_WG(1, 10, 8);
        else if (p.co_t == 10 && p.ci_t == 16) CD_WG(1, 10, 16);
        else if (p.co_t == 14 && p.ci_t == 8) CD_WG(1, 14, 8);
        else if (p.co_t == 16 && p.ci_t == 8) CD_WG(1, 16, 8);
    } else {
        if (p.co_t == 4 && p.ci_t == 4) CD_WG(1, 4, 4);
        else if (p.co_t == 4 && p.ci_t == 2) CD_WG(1, 4, 2);
        else if (p.co_t == 4) CD_WG(1, 4, 1);
        else if (p.co_t == 2 && p.ci_t == 4) CD_WG(1, 2, 4);
        else if (p.co_t == 2 && p.ci_t == 2) CD_WG(1, 2, 2);
        else if (p.co_t == 2) CD_WG(1, 2, 1);
        else if (p.ci_t == 4) CD_WG(1, 1, 4);
        else if (p.ci_t == 2) CD_WG(1, 1, 2);
        else CD_WG(1, 1, 1);
    }
#undef CD_WG
    if (rc != CD_OK) return rc;
    if (accumulate & 4) return CD_OK;   // stays packed: cd_conv2d_wgrad_unpack_table finishes it
    const int cob = p.co_t * 16, cib = p.ci_t * 16;
    const int total = Cout * Cin * ks * ks;
    hipLaunchKernelGGL(cd::unpack_wgrad_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0, s,
                       workspace, Cout, Cin, ks, cob, cib, L.cigs, L.splits, L.slice, dw, accumulate & 1, (size_t)0);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_conv2d_wgrad_desc(cd_wgrad_desc* desc) {
    static_assert(sizeof(cd_wgrad_desc) == sizeof(cd::WgradDesc), "cd_wgrad_desc / WgradDesc");
    if (!desc) return CD_ERR_INVALID_ARG;
    cd::WgradDesc* d = reinterpret_cast<cd::WgradDesc*>(desc);
    d->klass = -1; d->blocks = 0; d->block_end = 0; d->pad[0] = d->pad[1] = 0;
    if (!d->x || !d->dy || !d->workspace || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return CD_ERR_INVALID_ARG;
    if (d->x_coff < 0 || d->x_coff + d->Cin > d->x_ctot || d->dy_coff < 0 || d->dy_coff + d->Cout > d->dy_ctot) return CD_ERR_INVALID_ARG;
    if ((d->in_scale == nullptr) != (d->in_shift == nullptr)) return CD_ERR_INVALID_ARG;
    if (cd_conv2d_wgrad_workspace_floats(d->Cout, d->Cin, d->ks) == 0) return CD_ERR_UNSUPPORTED;
    // the layout (blocks per channel group, 16 x 16 or 32 x 16 channel blocks) is the one cd_conv2d_wgrad chooses for the same
    // arguments: same workspace contents, same cd_conv2d_wgrad_plan, same unpack descriptors
    const cd::WgLayout L = cd::wgrad_layout(d->Cout, d->Cin, d->ks, d->N, d->H, d->W);
    if (!L.split_arith) return CD_OK;      // klass stays -1: not one of the table kernels' gradients -- launch it on its own
    cd::wgrad_split_desc_geometry(d, L.splits, L.cot);
    return CD_OK;
}

int cd_conv2d_wgrad_table(const void* table_dev, int n, int klass, int total_blocks, void* stream) {
    if (!table_dev || n <= 0 || n > 64 || klass < 0 || klass > 4 || total_blocks <= 0) return CD_ERR_INVALID_ARG;
    return cd::launch_wgrad_split_table(table_dev, n, klass, total_blocks, (hipStream_t)stream);
}

int cd_conv2d_wgrad_grouped(const float* x, int x_ctot, int x_coff, int cin_g, const float* dy, int dy_ctot, int dy_coff, int cout_g,
                            int groups, float* dw, int accumulate, float* workspace, size_t workspace_group_stride, int N, int H, int W,
                            int ks, void* stream) {
    if (!x || !dy || !dw || !workspace || groups <= 0 || cin_g <= 0 || cout_g <= 0 || N <= 0 || H <= 0 || W <= 0) return CD_ERR_INVALID_ARG;
    if (x_coff < 0 || x_coff + groups * cin_g > x_ctot || dy_coff < 0 || dy_coff + groups * cout_g > dy_ctot) return CD_ERR_INVALID_ARG;
    const size_t wsf = cd_conv2d_wgrad_workspace_floats(cout_g, cin_g, ks);
    if (wsf == 0) return CD_ERR_UNSUPPORTED;
    if (groups > 1 && workspace_group_stride < wsf) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const cd::WgLayout L = cd::wgrad_layout(cout_g, cin_g, ks, N, H, W);
    if (L.split_arith && groups <= 65535 / L.cogs) {   // ONE launch for all groups (+ one unpack)
        const int rc = cd::launch_wgrad_split(x, x_ctot, x_coff, cin_g, nullptr, nullptr, 0, dy, dy_ctot, dy_coff, cout_g, workspace, N, H, W, ks,
                                              L.splits, s, groups, workspace_group_stride, L.cot);
        if (rc != CD_OK) return rc;
        const int total = cout_g * cin_g * ks * ks;
        const int bx = (total + 255) / 256 > 64 ? 64 : (total + 255) / 256;
        hipLaunchKernelGGL(cd::unpack_wgrad_kernel, dim3(bx, groups), dim3(256), 0, s, workspace, cout_g, cin_g, ks, L.cob, L.cib, L.cigs, L.splits,
                           L.slice, dw, accumulate & 1, workspace_group_stride);
        CD_CHECK_LAUNCH();
        return CD_OK;
    }
    for (int g = 0; g < groups; ++g) {   // other arithmetic modes / filter sizes: the dense kernels, group by group
        const int rc = cd_conv2d_wgrad(x, x_ctot, x_coff + g * cin_g, cin_g, nullptr, nullptr, 0, dy, dy_ctot, dy_coff + g * cout_g, cout_g,
                                       dw + (size_t)g * cout_g * cin_g * ks * ks, accumulate & 1, workspace + (size_t)g * workspace_group_stride, N, H,
                                       W, ks, stream);
        if (rc != CD_OK) return rc;
    }
    return CD_OK;
}

}  // extern "C"
