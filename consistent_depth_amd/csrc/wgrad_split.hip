// Weight gradient of the stride-1 "same" convolution at fp32 accuracy on the gfx950 BF16 matrix cores (k in {3, 5, 7, 11}).
//
//   dW[co][ci][ky][kx] = sum_{n,y,x} dY[n][co][y][x] * act(X)[n][ci][y+ky-P][x+kx-P]
//
// Same arithmetic as conv_split.hip: both operands are split exactly into three bf16 terms and the six significant cross
// products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (as close to fp64 as the fp32 instruction, ~2x its speed).
// GEMM view per filter tap: M = 16 output channels, N = 16 input channels, K = the 32 pixels of one tile row:
//   A[i = lane&15][8*(lane>>4) + e] = dY[co0+i][y][x0 + 8*(lane>>4) + e]            one ds_read_b128 per split
//   B[8*(lane>>4) + e][j = lane&15] = act(X)[ci0+j][y+ky][x0 + 8*(lane>>4) + e + kx]
// Both tiles sit in LDS in their natural order (pixels contiguous, 2 bytes each), so B for tap kx starts at an arbitrary
// 2-byte offset.  A lane therefore loads the 24-pixel WINDOW [8g, 8g+24) of the row once (three aligned ds_read_b128 per
// split) and every tap of that filter row is a register range of it -- even shifts directly, odd shifts funnel-shifted by
// one pixel (4 v_alignbit per tap and split): one window serves all KS taps of the row.
// D: lane holds ci0 + (lane&15), co0 + 4*(lane>>4) + {0..3}  (the layout conv_wgrad.hip flushes and its unpack kernels read).
// A block owns 16 x 16 channels and walks image tiles (grid-stride); its waves (8 at k = 11, else 4) divide the taps (contiguous
// ranges of the flattened tap index, compile-time per wave: 7 x 16 + 9 at k = 11), each keeping one accumulator tile per tap.  Every
// block stores its partial sums once into its own slice; the unpack kernels add the slices in a fixed order: no atomics,
// bit-reproducible.
#include "cd_common.h"
#include "wgrad_split.h"

namespace cd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ws_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a, b) -> packed pairs of the three split terms (as conv_split.hip)
__device__ __forceinline__ void ws_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = ws_cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = ws_cvt_pk_bf16(ra, rb);
    l = ws_cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

constexpr int ws_pad(int words) { return words + ((4 - words % 8) + 8) % 8; }   // == 4 (mod 8): 16 channel planes tile the 64 banks

template <int KS, int COT = 1> struct WsCfg {
    static constexpr int TY = wgrad_split_tile_rows(KS);
    static constexpr int NW = KS == 11 ? 8 : 4;               // waves per block (k = 11: 8 x 16 taps = 64 accumulator registers each)
    static constexpr int WPS = NW / COT;                      // waves per 16 x 16 sub-tile (COT output-channel groups per block)
    static constexpr int P = (KS - 1) / 2, TAPS = KS * KS, TPW = (TAPS + WPS - 1) / WPS;
    static constexpr int ROWS = TY + KS - 1;
    static constexpr int XW = 48;                              // pixels per LDS row of X: [X0 - 8, X0 + 40)
    static constexpr int PSX = ws_pad(ROWS * XW / 2);          // 32-bit words per channel plane
    static constexpr int PSD = ws_pad(TY * 32 / 2);
    static constexpr int SPX = 16 * PSX, SPD = 16 * COT * PSD;   // words per split plane set
    static constexpr size_t LDS = (size_t)3 * (SPX + SPD) * 4;
};

// One wave's share of a staged tile: output-channel sub-tile WV / WPS, taps [(WV % WPS) * TPW, + TPW) of the flattened index.
template <int KS, int WV, int COT>
__device__ __forceinline__ void ws_wave(const unsigned* __restrict__ s_x, const unsigned* __restrict__ s_dy, f32x4 (&acc)[WsCfg<KS, COT>::TPW],
                                        int lane) {
    using Cfg = WsCfg<KS, COT>;
    constexpr int TY = Cfg::TY, P = Cfg::P, TAPS = Cfg::TAPS, TPW = Cfg::TPW, PSX = Cfg::PSX, PSD = Cfg::PSD, SPX = Cfg::SPX, SPD = Cfg::SPD;
    constexpr int SUB = WV / Cfg::WPS, T0 = (WV % Cfg::WPS) * TPW, T1 = (T0 + TPW < TAPS) ? T0 + TPW : TAPS;
    if constexpr (T0 < T1) {
        constexpr int KY0 = T0 / KS, KY1 = (T1 - 1) / KS;
        const int li = lane & 15, g = lane >> 4;
        const unsigned* a_ptr = s_dy + (SUB * 16 + li) * PSD + 4 * g;
        const unsigned* w_ptr = s_x + li * PSX + 4 * g;
#pragma unroll 1
        for (int y = 0; y < TY; ++y) {
            bf16x8 a[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_ptr + sp * SPD + y * 16));
#pragma unroll
            for (int ky = KY0; ky <= KY1; ++ky) {
                const int kxa = (T0 - ky * KS > 0) ? T0 - ky * KS : 0, kxb = (T1 - ky * KS < KS) ? T1 - ky * KS : KS;   // folded: ky is unrolled
                // one B split at a time (its window is 23 registers): lo with dY hi; mid with dY mid, hi; hi with dY lo, mid, hi
#pragma unroll
                for (int sp = 2; sp >= 0; --sp) {
                    __builtin_amdgcn_sched_barrier(0);   // keep the windows of later splits / rows out of this group's registers
                    unsigned w[12];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(w_ptr + sp * SPX + (y + ky) * 24 + 4 * q);
                        w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
                    }
#pragma unroll
                    for (int pa = 2 - sp; pa >= 0; --pa)   // dY splits paired with this B split, smallest product first
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {   // round-robin over the taps of the row: a dependent MFMA never follows its producer
                            if (kx < kxa || kx >= kxb) continue;
                            const int t = ky * KS + kx - T0, s = 8 - P + kx, r0 = s >> 1;
                            u32x4 bv;   // pixels [s, s + 8) of the window: a register range, or (odd s) funnel-shifted by one pixel
                            if (s & 1) bv = u32x4{__builtin_amdgcn_alignbit(w[r0 + 1], w[r0], 16), __builtin_amdgcn_alignbit(w[r0 + 2], w[r0 + 1], 16),
                                                  __builtin_amdgcn_alignbit(w[r0 + 3], w[r0 + 2], 16), __builtin_amdgcn_alignbit(w[r0 + 4], w[r0 + 3], 16)};
                            else bv = u32x4{w[r0], w[r0 + 1], w[r0 + 2], w[r0 + 3]};
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa], __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
                        }
                }
            }
        }
    }
}

template <int KS, int COT>
__global__ __launch_bounds__((WsCfg<KS, COT>::NW * 64), 2) void conv_wgrad_split_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    const float* __restrict__ dy, int dy_ctot, int dy_coff, int Cout,
    float* __restrict__ dw_packed, int N, int H, int W, int tiles_x, int tiles_y, int cogs, int zpg, int g_xc, int g_dyc, size_t g_ws) {
    using Cfg = WsCfg<KS, COT>;
    constexpr int TY = Cfg::TY, P = Cfg::P, TAPS = Cfg::TAPS, TPW = Cfg::TPW, ROWS = Cfg::ROWS, PSX = Cfg::PSX, PSD = Cfg::PSD;
    constexpr int SPX = Cfg::SPX, SPD = Cfg::SPD, NT = Cfg::NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned ws_smem[];
    unsigned* s_x = ws_smem;               // [3][16 ci][PSX]
    unsigned* s_dy = ws_smem + 3 * SPX;    // [3][16 * COT co][PSD]

    // blockIdx.z = group * zpg + cog: a grouped convolution is `groups` independent gradients on channel slices, each with its
    // own packed workspace (g_ws floats apart); dense launches have one group.  cog counts blocks of 16 * COT output channels
    // (zpg per group), `cogs` the 16-channel tiles of the packed layout.
    const int grp = blockIdx.z / zpg;
    const int cig = blockIdx.y, cog = blockIdx.z - grp * zpg;
    x_coff += grp * g_xc;
    dy_coff += grp * g_dyc;
    dw_packed += (size_t)grp * g_ws;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t HW = (size_t)H * W;
    const int items = N * tiles_x * tiles_y;
    const bool vec = (W & 3) == 0;

    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int n = item / (tiles_x * tiles_y), tile = item - n * (tiles_x * tiles_y);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int X0 = tx * 32, Y0 = ty * TY;
        __syncthreads();   // the previous tile is consumed
        // ---- staging: fp32 -> (affine, relu) -> three bf16 planes, pixels contiguous.  Elements go in batches of 4 per thread whose
        // loads are UNCONDITIONAL (clamped address, padding zeroed with an AND afterwards): all loads of a batch are in flight before
        // the first wait (a load under a divergent branch is followed by s_waitcnt vmcnt(0): one full latency per element).
        // One element = 4 consecutive pixels of one channel row.  `stage(nch, rows, quads, ...)` covers a [nch][rows][quads] tile.
        auto stage = [&](const float* __restrict__ src_n, int ch0, int ch_n, int nch, int rows, int quads, int y0, int x0, unsigned* __restrict__ dst,
                         int plane_words, int row_words, int split_words, bool affine) {
            const int total = nch * rows * quads;
            constexpr int BATCH = 4;
            for (int i0 = threadIdx.x; i0 < total; i0 += NT * BATCH) {
                float v[BATCH][4], asc[BATCH], ash[BATCH];
                unsigned keep[BATCH][4];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int i = i0 + b * NT;
                    const int c = i / (rows * quads), rem = i - c * (rows * quads), r = rem / quads, q = rem - r * quads;
                    const int ch = ch0 + c, gy = y0 + r, gx = x0 + 4 * q;
                    const bool base_ok = i < total && ch < ch_n && (unsigned)gy < (unsigned)H;
                    const float* row = src_n + (size_t)(ch < ch_n ? ch : ch_n - 1) * HW + (size_t)((unsigned)gy < (unsigned)H ? gy : 0) * W;
                    // the producer's BatchNorm scale / shift travel WITH the data loads (not as a second round trip afterwards)
                    asc[b] = (affine && in_scale) ? in_scale[ch < ch_n ? ch : ch_n - 1] : 1.f;
                    ash[b] = (affine && in_scale) ? in_shift[ch < ch_n ? ch : ch_n - 1] : 0.f;
                    if (vec) {   // W % 4 == 0: an aligned quad is inside or outside the image as a whole
                        const bool in = base_ok && (unsigned)gx < (unsigned)W;
                        const float4 f = *reinterpret_cast<const float4*>(row + ((unsigned)gx < (unsigned)W ? gx : 0));
                        v[b][0] = f.x; v[b][1] = f.y; v[b][2] = f.z; v[b][3] = f.w;
                        keep[b][0] = keep[b][1] = keep[b][2] = keep[b][3] = in ? 0xffffffffu : 0u;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool in = base_ok && (unsigned)(gx + e) < (unsigned)W;
                            v[b][e] = row[(unsigned)(gx + e) < (unsigned)W ? gx + e : 0];
                            keep[b][e] = in ? 0xffffffffu : 0u;
                        }
                    }
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int i = i0 + b * NT;
                    if (i >= total) break;
                    const int c = i / (rows * quads), rem = i - c * (rows * quads), r = rem / quads, q = rem - r * quads;
                    if (affine) {
                        const int ch = ch0 + c < ch_n ? ch0 + c : ch_n - 1;
                        if (in_scale) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[b][e] = __fmaf_rn(v[b][e], asc[b], ash[b]);   // same fma as the BN backward's mask
                        }
                        if (in_relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[b][e] = fmaxf(v[b][e], 0.f);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[b][e] = __uint_as_float(__float_as_uint(v[b][e]) & keep[b][e]);   // zero padding stays an exact zero
                    unsigned h0, m0, l0, h1, m1, l1;
                    ws_split_pair(v[b][0], v[b][1], h0, m0, l0);
                    ws_split_pair(v[b][2], v[b][3], h1, m1, l1);
                    unsigned* d = dst + c * plane_words + r * row_words + 2 * q;
                    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(d + split_words) = u32x2{m0, m1};
                    *reinterpret_cast<u32x2*>(d + 2 * split_words) = u32x2{l0, l1};
                }
            }
        };
        // dY tile (zero outside the image / beyond Cout), then the activated input tile with halo: rows [Y0 - P, Y0 + TY + P),
        // pixels [X0 - 8, X0 + 40)
        stage(dy + ((size_t)n * dy_ctot + dy_coff) * HW, cog * 16 * COT, Cout, 16 * COT, TY, 8, Y0, X0, s_dy, PSD, 16, SPD, false);
        stage(x + ((size_t)n * x_ctot + x_coff) * HW, cig * 16, Cin, 16, ROWS, 12, Y0 - P, X0 - 8, s_x, PSX, 24, SPX, true);
        __syncthreads();
        if (wid == 0) ws_wave<KS, 0, COT>(s_x, s_dy, acc, lane);
        else if (wid == 1) ws_wave<KS, 1, COT>(s_x, s_dy, acc, lane);
        else if (wid == 2) ws_wave<KS, 2, COT>(s_x, s_dy, acc, lane);
        else if (wid == 3) ws_wave<KS, 3, COT>(s_x, s_dy, acc, lane);
        else if constexpr (Cfg::NW == 8) {
            if (wid == 4) ws_wave<KS, 4, COT>(s_x, s_dy, acc, lane);
            else if (wid == 5) ws_wave<KS, 5, COT>(s_x, s_dy, acc, lane);
            else if (wid == 6) ws_wave<KS, 6, COT>(s_x, s_dy, acc, lane);
            else ws_wave<KS, 7, COT>(s_x, s_dy, acc, lane);
        }
    }

    // ---- flush: this block's slice, packed [split][cog16][cig][tap][16 co][16 ci]; the wave's sub-tile is 16-channel group
    // cog * COT + sub of the packed layout (absent when Cout is not a multiple of 16 * COT)
    const int sub = wid / Cfg::WPS, cog16 = cog * COT + sub;
    if (cog16 >= cogs) return;
    const size_t slice = (size_t)cogs * gridDim.y * TAPS * 256;
    const size_t base = (size_t)blockIdx.x * slice + ((size_t)cog16 * gridDim.y + cig) * TAPS * 256;
    const int ci_l = lane & 15, co4 = (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = (wid % Cfg::WPS) * TPW + t;
        if (tap < TAPS) {
            float* dst = dw_packed + base + ((size_t)tap * 16 + co4) * 16 + ci_l;
            const f32x4 v = acc[t];
            dst[0] = v.x; dst[16] = v.y; dst[32] = v.z; dst[48] = v.w;
        }
    }
}

template <int KS, int COT>
static int launch_ws(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift, int in_relu,
                     const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H, int W, int splits, hipStream_t s,
                     int groups, size_t ws_group_stride) {
    using Cfg = WsCfg<KS, COT>;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + Cfg::TY - 1) / Cfg::TY;
    const int cogs = (Cout + 15) / 16, cigs = (Cin + 15) / 16, zpg = (cogs + COT - 1) / COT;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_split_kernel<KS, COT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_wgrad_split_kernel<KS, COT>), dim3(splits, cigs, zpg * groups), dim3(Cfg::NW * 64), Cfg::LDS, s, x, x_ctot, x_coff, Cin,
                       in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, tiles_x, tiles_y, cogs, zpg, groups > 1 ? Cin : 0,
                       groups > 1 ? Cout : 0, ws_group_stride);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

int launch_wgrad_split(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift, int in_relu,
                       const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H, int W, int ks, int splits,
                       hipStream_t s, int groups, size_t ws_group_stride, int cot) {
    if (ks == 11) return launch_ws<11, 1>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, splits, s, groups, ws_group_stride);
    if (ks == 7) return launch_ws<7, 1>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, splits, s, groups, ws_group_stride);
    if (ks == 3 && cot == 2) return launch_ws<3, 2>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, splits, s, groups, ws_group_stride);
    if (ks == 3) return launch_ws<3, 1>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, splits, s, groups, ws_group_stride);
    if (ks == 5) return launch_ws<5, 1>(x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu, dy, dy_ctot, dy_coff, Cout, packed, N, H, W, splits, s, groups, ws_group_stride);
    return CD_ERR_UNSUPPORTED;
}

}  // namespace cd
