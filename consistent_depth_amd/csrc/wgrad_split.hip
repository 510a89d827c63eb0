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

#ifndef CD_WGRAD_AHEAD11   // 1: fetch the next tile ahead at k = 11 too (27 spilled registers: measured slower than all-loads-at-once)
#define CD_WGRAD_AHEAD11 0
#endif
#ifndef CD_WS_DBG        // measurement builds (tools/exp/build_variants.sh), bits: 1 = no MFMA phase, 2 = no staging (fetch / commit), 4 = no operand alignment work (wrong results)
#define CD_WS_DBG 0
#endif

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

// One wave's share of a staged tile: output-channel sub-tile WV / WPS, taps [(WV % WPS) * TPW, + TPW) of the flattened index
// (13 at k = 7, 16 at k = 11: they span two or three filter rows).  Per output row y and B split the windows of ALL the wave's filter
// rows are loaded (three ds_read_b128 = 12 registers each) and the MFMAs walk over all the wave's taps before the next dY split
// revisits an accumulator: v_mfma_f32_16x16x32_bf16 needs many independent accumulators in flight (profiles/mfma_rate_exp_r02.txt:
// 55 % of its rate with 4-8, the full rate with 16); the round-robin over the 5-11 taps of ONE filter row (rounds 2/3) ran the MFMA
// phase at ~60 % even with the operand work and the LDS reads removed (profiles/wgrad_phases_r03.txt).  Every accumulator still
// receives its six products per row in the same order -- lo x hi; mid x mid, mid x hi; hi x lo, hi x mid, hi x hi -- same bits.
template <int KS, int WV, int COT>
__device__ __forceinline__ void ws_wave(const unsigned* __restrict__ s_x, const unsigned* __restrict__ s_dy, f32x4 (&acc)[WsCfg<KS, COT>::TPW],
                                        int lane) {
    using Cfg = WsCfg<KS, COT>;
    constexpr int TY = Cfg::TY, P = Cfg::P, TAPS = Cfg::TAPS, TPW = Cfg::TPW, PSX = Cfg::PSX, PSD = Cfg::PSD, SPX = Cfg::SPX, SPD = Cfg::SPD;
    constexpr int SUB = WV / Cfg::WPS, T0 = (WV % Cfg::WPS) * TPW, T1 = (T0 + TPW < TAPS) ? T0 + TPW : TAPS;
    if constexpr (T0 < T1) {
        constexpr int KY0 = T0 / KS, KY1 = (T1 - 1) / KS, NKY = KY1 - KY0 + 1;
        const int li = lane & 15, g = lane >> 4;
        const unsigned* a_ptr = s_dy + (SUB * 16 + li) * PSD + 4 * g;
        const unsigned* w_ptr = s_x + li * PSX + 4 * g;
#pragma unroll 1
        for (int y = 0; y < TY; ++y) {
            bf16x8 a[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_ptr + sp * SPD + y * 16));
#pragma unroll
            for (int sp = 2; sp >= 0; --sp) {
                __builtin_amdgcn_sched_barrier(0);   // keep the windows of later splits / rows out of this group's registers
                unsigned w[NKY][12];
#pragma unroll
                for (int r = 0; r < NKY; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(w_ptr + sp * SPX + (y + KY0 + r) * 24 + 4 * q);
                        w[r][4 * q] = v[0]; w[r][4 * q + 1] = v[1]; w[r][4 * q + 2] = v[2]; w[r][4 * q + 3] = v[3];
                    }
#pragma unroll
                for (int pa = 2 - sp; pa >= 0; --pa)   // dY splits paired with this B split, smallest product first
#pragma unroll
                    for (int t = 0; t < T1 - T0; ++t) {   // all taps of the wave: a dependent MFMA is T1 - T0 instructions behind its producer
                        const int ky = (T0 + t) / KS, kx = T0 + t - ky * KS, s = 8 - P + kx, r0 = s >> 1;
                        const unsigned (&wc)[12] = w[ky - KY0];
                        u32x4 bv;   // pixels [s, s + 8) of the window: a register range, or (odd s) funnel-shifted by one pixel
                        if (CD_WS_DBG & 4) bv = u32x4{wc[r0 & ~1], wc[(r0 & ~1) + 1], wc[(r0 & ~1) + 2], wc[(r0 & ~1) + 3]};   // (what-if: no operand alignment work)
                        else if (s & 1) bv = u32x4{__builtin_amdgcn_alignbit(wc[r0 + 1], wc[r0], 16), __builtin_amdgcn_alignbit(wc[r0 + 2], wc[r0 + 1], 16),
                                                   __builtin_amdgcn_alignbit(wc[r0 + 3], wc[r0 + 2], 16), __builtin_amdgcn_alignbit(wc[r0 + 4], wc[r0 + 3], 16)};
                        else bv = u32x4{wc[r0], wc[r0 + 1], wc[r0 + 2], wc[r0 + 3]};
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa], __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
                    }
            }
        }
    }
}

// One weight gradient as the kernels see it (a stand-alone launch passes its grid: splits = gridDim.x, cigs = gridDim.y).
struct WsArgs {
    const float* x; const float* in_scale; const float* in_shift; const float* dy; float* dw_packed;
    int x_ctot, x_coff, Cin, in_relu, dy_ctot, dy_coff, Cout, N, H, W, tiles_x, tiles_y, cogs, zpg, g_xc, g_dyc;
    size_t g_ws;
    int splits, cigs;
};

// The work of block (bx, by, bz) of the stand-alone grid (splits, cigs, zpg * groups) of one weight gradient.
template <int KS, int COT>
__device__ __forceinline__ void ws_block(const WsArgs& a, const int bx, const int by, const int bz) {
    const float* __restrict__ x = a.x; const float* __restrict__ in_scale = a.in_scale; const float* __restrict__ in_shift = a.in_shift;
    const float* __restrict__ dy = a.dy; float* __restrict__ dw_packed = a.dw_packed;
    int x_coff = a.x_coff, dy_coff = a.dy_coff;
    const int x_ctot = a.x_ctot, Cin = a.Cin, in_relu = a.in_relu, dy_ctot = a.dy_ctot, Cout = a.Cout, N = a.N, H = a.H, W = a.W;
    const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, cogs = a.cogs, zpg = a.zpg, g_xc = a.g_xc, g_dyc = a.g_dyc;
    const size_t g_ws = a.g_ws;
    const int n_splits = a.splits, n_cigs = a.cigs;
    using Cfg = WsCfg<KS, COT>;
    constexpr int TY = Cfg::TY, P = Cfg::P, TAPS = Cfg::TAPS, TPW = Cfg::TPW, ROWS = Cfg::ROWS, PSX = Cfg::PSX, PSD = Cfg::PSD;
    constexpr int SPX = Cfg::SPX, SPD = Cfg::SPD, NT = Cfg::NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned ws_smem[];
    unsigned* s_x = ws_smem;               // [3][16 ci][PSX]
    unsigned* s_dy = ws_smem + 3 * SPX;    // [3][16 * COT co][PSD]

    // blockIdx.z = group * zpg + cog: a grouped convolution is `groups` independent gradients on channel slices, each with its
    // own packed workspace (g_ws floats apart); dense launches have one group.  cog counts blocks of 16 * COT output channels
    // (zpg per group), `cogs` the 16-channel tiles of the packed layout.
    const int grp = bz / zpg;
    const int cig = by, cog = bz - grp * zpg;
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

    // ---- staging: fp32 -> (affine, relu) -> three bf16 planes, pixels contiguous.  One element ("quad") = 4 consecutive pixels of
    // one channel row; a tile is QDY quads of dY (zero outside the image / beyond Cout) followed by QX quads of the activated input
    // with its halo, rows [Y0 - P, Y0 + TY + P), pixels [X0 - 8, X0 + 40); thread t owns the quads t + j * NT, j < NQ (QDY is a
    // multiple of NT: the dY / X decision is per j, compile-time).  Staging is split in two halves around the MFMA phase:
    //   fetch(item)  -- the raw global loads of a tile into NQ x 4 registers, UNCONDITIONAL (clamped address; padding is zeroed with
    //                   an AND in the second half) so that all of them are in flight at once;
    //   commit()     -- transform + split + LDS writes of the fetched tile.
    // The fetch of tile i + 1 is issued BEFORE the MFMAs of tile i and committed after them: the memory latency of a tile (three
    // serialised batches of loads in round 2/3's kernel, as long as its MFMA phase at k <= 7) is hidden behind the matrix cores;
    // between two MFMA phases a block only pays the commit's vector ALU work.
    // (the map thread / slot -> tile element -> LDS word is ws_stage_quad of wgrad_stage_map.h: tests/test_wgrad_map_cpu.py runs it on the host)
    constexpr int NQ = Cfg::NQ;
    float* s_aff = reinterpret_cast<float*>(ws_smem + 3 * (SPX + SPD));   // [16][2]: scale, shift of this block's input channels
    if (threadIdx.x < 16) {
        const int ch = cig * 16 + (int)threadIdx.x < Cin ? cig * 16 + (int)threadIdx.x : Cin - 1;
        s_aff[2 * threadIdx.x] = in_scale ? in_scale[ch] : 1.f;
        s_aff[2 * threadIdx.x + 1] = in_scale ? in_shift[ch] : 0.f;
    }
    float4 pv[NQ];
    unsigned long long pkeep = 0;
    auto fetch = [&](int item) {
        const int n = item / (tiles_x * tiles_y), tile = item - n * (tiles_x * tiles_y);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int X0 = tx * 32, Y0 = ty * TY;
        const float* dy_n = dy + ((size_t)n * dy_ctot + dy_coff) * HW;
        const float* x_n = x + ((size_t)n * x_ctot + x_coff) * HW;
        pkeep = 0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const WsQuad sq = ws_stage_quad<KS, COT>((int)threadIdx.x, j);
            const bool is_dy = sq.is_dy, live = sq.live;
            const int c = sq.c, r = sq.r, q = sq.q;
            const int ch = (is_dy ? cog * 16 * COT : cig * 16) + c, ch_n = is_dy ? Cout : Cin;
            const int gy = (is_dy ? Y0 : Y0 - P) + r, gx = (is_dy ? X0 : X0 - 8) + 4 * q;
            const bool base_ok = live && ch < ch_n && (unsigned)gy < (unsigned)H;
            const float* row = (is_dy ? dy_n : x_n) + (size_t)(ch < ch_n ? ch : ch_n - 1) * HW + (size_t)((unsigned)gy < (unsigned)H ? gy : 0) * W;
            if (vec) {   // W % 4 == 0: an aligned quad is inside or outside the image as a whole
                pv[j] = *reinterpret_cast<const float4*>(row + ((unsigned)gx < (unsigned)W ? gx : 0));
                if (base_ok && (unsigned)gx < (unsigned)W) pkeep |= 15ull << (4 * j);
            } else {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k] = row[(unsigned)(gx + k) < (unsigned)W ? gx + k : 0];
                    if (base_ok && (unsigned)(gx + k) < (unsigned)W) pkeep |= 1ull << (4 * j + k);
                }
                pv[j] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const WsQuad sq = ws_stage_quad<KS, COT>((int)threadIdx.x, j);
            if (!sq.live) break;
            const bool is_dy = sq.is_dy;
            const int c = sq.c;
            float v[4] = {pv[j].x, pv[j].y, pv[j].z, pv[j].w};
            if (!is_dy) {
                if (in_scale) {
                    const float2 a = *reinterpret_cast<const float2*>(s_aff + 2 * c);
                    // One v_fma_f32 per pixel, spelled out: left to the compiler the four fmas become two v_pk_fma_f32 that broadcast
                    // scale / shift out of the loaded register PAIR with op_sel (dst overlapping the pair) -- and on gfx950 that form
                    // returned wrong sums NON-deterministically (the shift term only; 1e-3 relative on dW of the 32 x 16-channel 3x3
                    // blocks, found by tests/test_hourglass_engine_gpu.py's per-block test; tools/exp/wgrad_dbg.py bisects it:
                    // plain v_fma_f32, or the same packed fma on registers that are not the loaded pair, are exact).
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v[e]) : "v"(v[e]), "v"(a.x), "v"(a.y));   // = fmaf: the BN backward's mask sees the same bits
                }
                if (in_relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
            }
            const unsigned kb = (unsigned)(pkeep >> (4 * j));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(__float_as_uint(v[e]) & (0u - ((kb >> e) & 1u)));   // zero padding stays an exact zero
            unsigned h0, m0, l0, h1, m1, l1;
            ws_split_pair(v[0], v[1], h0, m0, l0);
            ws_split_pair(v[2], v[3], h1, m1, l1);
            unsigned* d = ws_smem + sq.lds_word;
            const int split_words = sq.split_words;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + split_words) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(d + 2 * split_words) = u32x2{l0, l1};
        }
    };

    constexpr bool AHEAD = KS != 11 || CD_WGRAD_AHEAD11;   // (k = 11: 64 accumulator registers + the 23-register windows; see the header)
    if (AHEAD && !(CD_WS_DBG & 2) && bx < items) fetch(bx);
    for (int item = bx; item < items; item += n_splits) {
        if (!AHEAD && !(CD_WS_DBG & 2)) fetch(item);   // all loads of the tile in flight at once, landing while the slower waves finish the previous tile
        __syncthreads();   // the previous tile is consumed (first trip: s_aff is written)
        if (!(CD_WS_DBG & 2)) commit();
        __syncthreads();
        if (AHEAD && !(CD_WS_DBG & 2) && item + n_splits < items) fetch(item + n_splits);   // in flight during the MFMAs below
        if (CD_WS_DBG & 1) continue;
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
    const size_t slice = (size_t)cogs * n_cigs * TAPS * 256;
    const size_t base = (size_t)bx * slice + ((size_t)cog16 * n_cigs + cig) * TAPS * 256;
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
__global__ __launch_bounds__((WsCfg<KS, COT>::NW * 64), 2) void conv_wgrad_split_kernel(const WsArgs a) {
    ws_block<KS, COT>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// MANY weight gradients of one (KS, COT) class in ONE launch: a flat grid, block b belongs to the descriptor d with
// table[d - 1].block_end <= b < table[d].block_end and is block (b - start) of that gradient's stand-alone grid, x fastest.  Every
// block does exactly what it does in its own launch (same items, same order, same slice of the same workspace): the packed partial
// sums -- hence the weight gradients -- are bit-identical to per-convolution launches.  What the table buys: the small images of the
// deep hourglass levels give a stand-alone launch a few dozen short blocks and an idle chip; here they ride along with the large
// ones (the caller orders the table heaviest first), and there is one launch tail per class instead of one per convolution.
template <int KS, int COT>
__global__ __launch_bounds__((WsCfg<KS, COT>::NW * 64), 2) void conv_wgrad_split_table_kernel(const WgradDesc* __restrict__ table, int n) {
    const int b = (int)blockIdx.x, lane = (int)(threadIdx.x & 63);
    // the descriptor: every lane looks at one entry (n <= 64 per launch), the first one whose range ends beyond b is ours
    const int end = table[lane < n ? lane : n - 1].block_end;
    const unsigned long long m = __ballot(lane < n && b < end);
    const int d = __builtin_amdgcn_readfirstlane(m ? __builtin_ctzll(m) : n - 1);
    const WgradDesc& e = table[d];
    const int local = b - (d > 0 ? table[d - 1].block_end : 0);
    WsArgs a;
    a.x = e.x; a.in_scale = e.in_scale; a.in_shift = e.in_shift; a.dy = e.dy; a.dw_packed = e.workspace;
    a.x_ctot = e.x_ctot; a.x_coff = e.x_coff; a.Cin = e.Cin; a.in_relu = e.in_relu; a.dy_ctot = e.dy_ctot; a.dy_coff = e.dy_coff;
    a.Cout = e.Cout; a.N = e.N; a.H = e.H; a.W = e.W; a.tiles_x = e.tiles_x; a.tiles_y = e.tiles_y; a.cogs = e.cogs; a.zpg = e.zpg;
    a.g_xc = 0; a.g_dyc = 0; a.g_ws = 0; a.splits = e.splits; a.cigs = e.cigs;
    const int bx = local % e.splits, rest = local / e.splits;
    ws_block<KS, COT>(a, bx, rest % e.cigs, rest / e.cigs);
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
    WsArgs a;
    a.x = x; a.in_scale = in_scale; a.in_shift = in_shift; a.dy = dy; a.dw_packed = packed;
    a.x_ctot = x_ctot; a.x_coff = x_coff; a.Cin = Cin; a.in_relu = in_relu; a.dy_ctot = dy_ctot; a.dy_coff = dy_coff; a.Cout = Cout;
    a.N = N; a.H = H; a.W = W; a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.cogs = cogs; a.zpg = zpg;
    a.g_xc = groups > 1 ? Cin : 0; a.g_dyc = groups > 1 ? Cout : 0; a.g_ws = ws_group_stride; a.splits = splits; a.cigs = cigs;
    hipLaunchKernelGGL((conv_wgrad_split_kernel<KS, COT>), dim3(splits, cigs, zpg * groups), dim3(Cfg::NW * 64), Cfg::LDS, s, a);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

template <int KS, int COT>
static int launch_ws_table(const WgradDesc* table_dev, int n, int total_blocks, hipStream_t s) {
    using Cfg = WsCfg<KS, COT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_split_table_kernel<KS, COT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_wgrad_split_table_kernel<KS, COT>), dim3(total_blocks), dim3(Cfg::NW * 64), Cfg::LDS, s, table_dev, n);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// class index of (KS, COT) for the table launches: 0 (3,1)  1 (3,2)  2 (5,1)  3 (7,1)  4 (11,1)
int wgrad_split_class(int ks, int cot) {
    if (ks == 3) return cot == 2 ? 1 : 0;
    return ks == 5 ? 2 : (ks == 7 ? 3 : (ks == 11 ? 4 : -1));
}

void wgrad_split_desc_geometry(WgradDesc* d, int splits, int cot) {
    const int ty = wgrad_split_tile_rows(d->ks);
    d->tiles_x = (d->W + 31) / 32; d->tiles_y = (d->H + ty - 1) / ty;
    d->cogs = (d->Cout + 15) / 16; d->cigs = (d->Cin + 15) / 16; d->zpg = (d->cogs + cot - 1) / cot;
    d->splits = splits;
    d->klass = wgrad_split_class(d->ks, cot);
    d->blocks = d->splits * d->cigs * d->zpg;
}

int launch_wgrad_split_table(const void* table_dev, int n, int klass, int total_blocks, hipStream_t s) {
    const WgradDesc* t = (const WgradDesc*)table_dev;
    switch (klass) {
        case 0: return launch_ws_table<3, 1>(t, n, total_blocks, s);
        case 1: return launch_ws_table<3, 2>(t, n, total_blocks, s);
        case 2: return launch_ws_table<5, 1>(t, n, total_blocks, s);
        case 3: return launch_ws_table<7, 1>(t, n, total_blocks, s);
        case 4: return launch_ws_table<11, 1>(t, n, total_blocks, s);
    }
    return CD_ERR_UNSUPPORTED;
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
