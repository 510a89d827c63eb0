// Direct 2-D convolution at fp32 accuracy on the gfx950 BF16 matrix cores (forward and input gradient, k in {3, 5, 7, 11}).
//
// CDNA4 has no TF32/xf32 and its fp32 matrix instruction runs at the vector rate (157 TFLOP/s); the bf16 instruction
// v_mfma_f32_32x32x16_bf16 is ~14x faster per multiply-add (2.1 PFLOP/s measured, profiles/mfma_rate_exp_r02.txt).  Every
// fp32 operand is split EXACTLY into three bf16 terms
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (RNE; 3 x 8 = 24 mantissa bits)
// and a product is evaluated as the six cross terms of weight  >= 2^-16:  hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi
// (each bf16 x bf16 product is exact in fp32; the dropped mid*lo, lo*mid, lo*lo terms are < 2^-25 |x*y|, below half an fp32
// ulp of the product), accumulated in fp32 by the matrix core.  Measured against fp64 this is as close as -- on long sums
// closer than -- the native fp32 instruction (profiles/mfma_split_exp_r02.txt: error / sum|a*b| 1.3e-7 vs 1.3e-7 at K = 2048,
// identical with all nine terms), so no precision is traded; what IS different: the result is not bitwise the fmaf chain
// of conv_mfma.hip (cd_set_conv_arith(0) selects that kernel), and an infinite input -- or one in the top binade, |x| >= 2^127 * (2 - 2^-8),
// whose hi term rounds to infinity -- gives NaN instead of +-inf (tests/test_split_arith_cpu.py restates the arithmetic in numpy).
//
// Mapping (implicit GEMM, no im2col buffer): M = the 32 output pixels of one tile row, N = 32 output columns,
// K = 16 = 8 input channels x 2 consecutive filter taps (taps flattened ky * KS + kx).
//   A[i = lane&31][8 * (lane>>5) + e] = act(in)[ci0 + e][y + ky][x0 + i + kx],  (ky, kx) = tap 2*step + (lane>>5)
//        one ds_read_b128 per split: the LDS tile is channels-last, 8 bf16 channels = 16 bytes per pixel, three planes
//   B[8 * (lane>>5) + e][n = lane&31] = w[column n][ci0 + e][tap]
//        pre-packed in exactly this fragment order (3 x 1 KB per step), read straight from global/L2 one step ahead
//   D: lane holds column lane&31, pixels 8*q + 4*(lane>>5) + {0..3}, q = 0..3  -> four 16-byte stores.
// Columns: 32 output channels -- or, when the convolution has at most 16 of them (the 64->16 branches of the finest level,
// the largest share of the network's multiply-adds), 16 channels x 2 OUTPUT ROWS: column (co, dy) of the M-tile of row y is
// output row y + dy, with the filter shifted down by dy (taps over KS + 1 rows, zero weights where the shift leaves the
// filter), so an M-tile exists only for every other row.  That keeps all 32 columns busy at (KS+1)/KS of the multiply-adds
// and halves the LDS bytes per multiply-add (a pixel fragment feeds 32 columns either way).
// A block (4 waves) owns a TY x 32 output tile for NT*32 columns; input channels stream through LDS 8 at a time
// (global fp32 -> producer's BN-apply + ReLU -> split -> LDS).  Per step a wave reads 3 KB of LDS per M-tile and issues
// 6 MFMAs (32 cycles each) per (M-tile, column tile): LDS <= 50% busy.  The six products of one accumulator are issued
// round-robin over the wave's accumulators (a dependent MFMA right behind its producer stalls the pipe).
// Fusions (input affine/ReLU, bias, accumulate, BatchNorm statistics) are those of conv_mfma.hip.
#include "cd_common.h"
#include "conv_split.h"

#ifndef CD_SP_DBG        // measurement builds (tools/exp/build_variants.sh): 1 = no MFMA phase, 2 = no staging (profiles/conv_phases_r03.txt), 4 = staging without its global loads
#define CD_SP_DBG 0
#endif

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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// DY = output rows per M-tile (2 when the convolution has <= 16 output channels)
__host__ __device__ constexpr int split_dy(int OC) { return OC <= 16 ? 2 : 1; }
__host__ __device__ constexpr int split_taps(int ks, int dy) { return (ks + dy - 1) * ks; }      // flattened (ky', kx), ky' over KS + DY - 1 rows
__host__ __device__ constexpr int split_steps(int ks, int dy) { return (split_taps(ks, dy) + 1) / 2; }
__host__ __device__ constexpr int split_ntiles(int OC) { return split_dy(OC) == 2 ? 1 : (OC + 31) / 32; }

template <int KS, int TY_, int DY> struct SplitCfg {
    static constexpr int TY = TY_;
    static constexpr int TAPS = split_taps(KS, DY), KSTEPS = split_steps(KS, DY);
    static constexpr int ROWS = TY + KS - 1;
    static constexpr int PADL = (((KS - 1) / 2) + 3) & ~3;          // aligned superset rows, as in conv_mfma.hip
    static constexpr int RSP = SP_TX + 2 * PADL, COFF = PADL - (KS - 1) / 2;
    static constexpr int PLANE = ROWS * RSP;                        // pixels (16-byte slots) per split plane
    static constexpr size_t LDS = (size_t)3 * PLANE * 16;
};

// ---------------------------------------------------------------- weight packing (split layout)
// [column tile][ci chunk of 8][step][split][lane][8 bf16]; element e of lane (n = lane&31, g = lane>>5) is
// w[column n][chunk*8 + e][tap 2*step + g]; column n = output channel tile*32 + n, or (OC <= 16) channel n&15 of output row
// dy = n>>4, whose filter row is ky' - dy.
__device__ __forceinline__ void pack_split_elements(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout_src, int Cin_src,
                                                    int KS, int transposed, int OC, int IC, int oc_off, int ic_off, size_t first,
                                                    size_t stride) {
    const int oc_n = transposed ? Cin_src : Cout_src, ic_n = transposed ? Cout_src : Cin_src;
    const int dy_n = split_dy(OC), taps = split_taps(KS, dy_n), ksteps = split_steps(KS, dy_n), chunks = (IC + 7) / 8, tiles = split_ntiles(OC);
    const size_t total = (size_t)tiles * chunks * ksteps * 512;
    for (size_t i = first; i < total; i += stride) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int lane = (int)(r & 63); r >>= 6;
        const int step = (int)(r % ksteps); r /= ksteps;
        const int chunk = (int)(r % chunks); r /= chunks;
        const int tile = (int)r;
        const int tap = step * 2 + (lane >> 5), n = lane & 31;
        const int kyp = tap / KS, kx = tap - kyp * KS;
        const int ky = dy_n == 2 ? kyp - (n >> 4) : kyp;
        const int oc = (dy_n == 2 ? (n & 15) : tile * 32 + n) - oc_off, ic = chunk * 8 + e - ic_off;
        if (tap >= taps || (unsigned)ky >= (unsigned)KS || (unsigned)oc >= (unsigned)oc_n || (unsigned)ic >= (unsigned)ic_n) continue;   // stays zero
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
    hipLaunchKernelGGL(pack_split_table_kernel, dim3(64, n), dim3(256), 0, s, (const PackDesc*)table_dev);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- the convolution
// NT = column tiles (32 wide) per block, TYP = output rows per block, DY = output rows per M-tile.
// Work split inside a block: ALL four waves hold accumulators for ALL MB = TYP / DY M-tiles of the block and divide the
// reduction dimension -- wave w takes the steps (chunk * KSTEPS + step) % 4 == w -- so a weight fragment is fetched by one
// wave only (with the M-tiles divided instead, every wave fetched every fragment: 64 B/clk/CU of L1 traffic, the measured
// bottleneck).  At the end the four partial sums of an M-tile are added in a fixed order through LDS by the wave that
// stores it.  The order of accumulation depends on nothing but (Cin, KS, DY): every launch shape gives the same bits.
constexpr size_t SPLIT_REDUCE_LDS = 4 * 3 * 4096;   // one round of the cross-wave reduction: 4 owners x 3 foreign partials x 4 KB
static inline size_t split_aff_bytes(int Cin) { return (size_t)((Cin + 7) / 8) * 8 * 2 * sizeof(float); }   // scale / shift of every (padded) input channel

// One convolution as the kernels see it (the arguments of a stand-alone launch).
struct SplitArgs {
    const float* x; const u32x4* wsp; const float* bias; const float* in_scale; const float* in_shift; float* y; double* stats;
    int x_ctot, x_coff, Cin, pack_tiles, in_relu, y_ctot, y_coff, Cout, accumulate, H, W, tiles_x, tiles_img, tiles_total, chunk_tiles, slices;
    ConvGroups grp;
};

// The work of block (bx, by) of the stand-alone grid (chunk_tiles * 8 * slices, groups).
template <int KS, int NT, int TYP, int DY, int CGS>
__device__ __forceinline__ void conv_fwd_split_block(const SplitArgs& a, const int bx, const int by) {
    const float* __restrict__ x = a.x; const u32x4* __restrict__ wsp = a.wsp; const float* __restrict__ bias = a.bias;
    const float* __restrict__ in_scale = a.in_scale; const float* __restrict__ in_shift = a.in_shift; float* __restrict__ y = a.y;
    double* __restrict__ stats = a.stats;
    int x_coff = a.x_coff, y_coff = a.y_coff;
    const int x_ctot = a.x_ctot, Cin = a.Cin, pack_tiles = a.pack_tiles, in_relu = a.in_relu, y_ctot = a.y_ctot, Cout = a.Cout;
    const int accumulate = a.accumulate, H = a.H, W = a.W, tiles_x = a.tiles_x, tiles_img = a.tiles_img, tiles_total = a.tiles_total;
    const int chunk_tiles = a.chunk_tiles, slices = a.slices;
    const ConvGroups grp = a.grp;
    using Cfg = SplitCfg<KS, TYP, DY>;
    // grouped convolution (ResNeXt's 32 x 8d 3x3): blockIdx.y = the group, a dense convolution on its channel slices with its
    // own packed filter; dense launches have one group and zero strides
    x_coff += by * grp.x_stride;
    y_coff += by * grp.y_stride;
    wsp += (size_t)by * grp.w_stride;
    if (bias != nullptr) bias += by * grp.y_stride;
    constexpr int TY = Cfg::TY, ROWS = Cfg::ROWS, RSP = Cfg::RSP, COFF = Cfg::COFF, PADL = Cfg::PADL, PLANE = Cfg::PLANE;
    constexpr int P = (KS - 1) / 2, TAPS = Cfg::TAPS, KSTEPS = Cfg::KSTEPS;
    constexpr int MB = TY / DY;                       // M-tiles (tile rows, DY output rows each) per block
    constexpr int MG = 4;                             // M-tiles whose fragments are in registers at a time
    constexpr int CPT = DY == 2 ? 16 : 32;            // output channels per column tile
    constexpr int COB = NT * CPT;
    constexpr int UNITS = ROWS * (RSP / 4);           // staging units: (row, 4-pixel quad) x 8 channels
    static_assert(MB % 4 == 0 && MB * NT <= 8, "M-tiles are owned round-robin by the 4 waves; 128 accumulator registers");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* s_in = reinterpret_cast<u32x4*>(smem_raw);   // [3][ROWS][RSP] 16-byte slots (8 bf16 channels of one pixel)

    // XCD-aware block -> (image tile, channel slice) mapping (see conv_mfma.hip)
    const int xcd = bx & 7, kx_ = bx >> 3;
    const int tg = kx_ / slices, slice = kx_ - tg * slices;
    const int t_lin = xcd * chunk_tiles + tg;
    if (tg >= chunk_tiles || t_lin >= tiles_total) return;   // block-uniform
    const int n = t_lin / tiles_img, tile = t_lin - n * tiles_img;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * SP_TX, Y0 = ty * TY;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 31, g = lane >> 5;
    const size_t HW = (size_t)H * W;
    const float* xin = x + ((size_t)n * x_ctot + x_coff) * HW;
    const int n_chunks = (Cin + 7) / 8;
    const bool vec_in = (W & 3) == 0;

    f32x16 acc[MB][NT];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][t][q] = 0.f;

    // the weight stream of column tile t: [chunk][step][split][lane], linear in (chunk, step).  A column tile beyond the
    // filter (odd tile count, NT = 2) re-computes tile 0 and is dropped by the epilogue's channel bound.
    const u32x4* wt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int gt = slice * NT + t;
        wt[t] = wsp + (size_t)(gt < pack_tiles ? gt : 0) * n_chunks * KSTEPS * 192 + lane;
    }
    const int steps_total = n_chunks * KSTEPS;
    // weight fragments: two register buffers used alternately (explicitly -- a "next = load; ...; cur = next" rotation was folded
    // away by the compiler, which then waited for every fragment right after issuing its load)
    bf16x8 bA[NT][3], bB[NT][3];
    auto load_b = [&](bf16x8 (&dst)[NT][3], int lin) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const u32x4 v = wt[t][(size_t)(lin < steps_total ? lin : steps_total - 1) * 192 + sp * 64];   // (always a load: no branch)
                dst[t][sp] = __builtin_bit_cast(bf16x8, v);
            }
    };
    int lin = wid;   // this wave's next step, linear over (chunk, step)
    load_b(bA, lin);   // invariant at every step-loop entry: bA holds the fragments of step `lin`

    const int abase = li + COFF;   // LDS slot of this lane's pixel in tile row 0 at tap (0, 0)

    // The producer's BatchNorm scale / shift of ALL input channels sit in LDS behind the images (round 6; padded channels: 0): every
    // staging unit used to fetch its chunk's 16 values from global memory with its data (round 3: fetched in the second half of the
    // staging they were a second memory round trip per unit).
    const int nc8 = n_chunks * 8;
    float* s_aff = reinterpret_cast<float*>(smem_raw + (size_t)CGS * Cfg::LDS);     // [2][nc8]
    if (in_scale != nullptr)
        for (int i = threadIdx.x; i < nc8; i += kBlock) {
            s_aff[i] = i < Cin ? in_scale[i] : 0.f;
            s_aff[nc8 + i] = i < Cin ? in_shift[i] : 0.f;
        }       // (visible after the first barrier of the round loop)

    // staging of one unit = (tile row r, 4-pixel quad) x the 8 channels of a chunk, in two halves: the raw loads (32 registers), then
    // transform + split + LDS writes.  Everything else a unit needs (its position, the padding masks, the affine) is recomputed / read
    // from LDS in the second half: a unit's raw loads are 32 registers and nothing else.
    auto stage_load = [&](int chunk, int u, float (&v)[8][4]) {
        const int r = u / (RSP / 4), q4 = (u - r * (RSP / 4)) * 4;
        const int gy = Y0 - P + r, gx = X0 - PADL + q4;
        const bool row_in = (unsigned)gy < (unsigned)H;
        const int gyc = row_in ? gy : 0;
        if (CD_SP_DBG & 4) {       // what-if: the staging's arithmetic and LDS writes without its memory round trip (wrong results)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c][0] = v[c][1] = v[c][2] = v[c][3] = __int_as_float(0x3f800000 + u + c + gy);
            return;
        }
        if (vec_in) {       // W % 4 == 0: an aligned quad is inside or outside the image as a whole
            const bool in = row_in && (unsigned)gx < (unsigned)W;
            const float* src = xin + (size_t)gyc * W + (in ? gx : 0);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = chunk * 8 + c;
                const float4 f = *reinterpret_cast<const float4*>(src + (size_t)(ci < Cin ? ci : Cin - 1) * HW);
                v[c][0] = f.x; v[c][1] = f.y; v[c][2] = f.z; v[c][3] = f.w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool in = row_in && (unsigned)(gx + p) < (unsigned)W;
                const float* src = xin + (size_t)gyc * W + (in ? gx + p : 0);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int ci = chunk * 8 + c;
                    v[c][p] = src[(size_t)(ci < Cin ? ci : Cin - 1) * HW];
                }
            }
        }
    };
    auto stage_finish = [&](int chunk, int u, float (&v)[8][4], int gbase) {
        const int r = u / (RSP / 4), q4 = (u - r * (RSP / 4)) * 4;
        const int gy = Y0 - P + r, gx = X0 - PADL + q4;
        const bool row_in = (unsigned)gy < (unsigned)H;
        unsigned keep[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
            keep[p] = ((CD_SP_DBG & 4) || (row_in && (unsigned)(gx + (vec_in ? 0 : p)) < (unsigned)W)) ? 0xffffffffu : 0u;
        float sc[8], sh[8];
        if (in_scale) {
            const f32x4* a4 = reinterpret_cast<const f32x4*>(s_aff + chunk * 8);
            const f32x4* b4 = reinterpret_cast<const f32x4*>(s_aff + nc8 + chunk * 8);
            const f32x4 s0 = a4[0], s1 = a4[1], h0 = b4[0], h1 = b4[1];
#pragma unroll
            for (int c = 0; c < 4; ++c) { sc[c] = s0[c]; sc[4 + c] = s1[c]; sh[c] = h0[c]; sh[4 + c] = h1[c]; }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = chunk * 8 + c;
            const unsigned kc = ci < Cin ? 0xffffffffu : 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float t = v[c][p];
                if (in_scale) t = __fmaf_rn(t, sc[c], sh[c]);
                if (in_relu) t = fmaxf(t, 0.f);
                v[c][p] = __uint_as_float(__float_as_uint(t) & keep[p] & kc);   // zero padding (pixels and channels) stays an exact zero
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
            const int slot = gbase + r * RSP + q4 + p;
            s_in[slot] = hh; s_in[PLANE + slot] = mm; s_in[2 * PLANE + slot] = ll;
        }
    };
    // (Round 6 also built a PREFETCH of every thread's first unit across the MFMA phase -- 32 registers, requested after the
    // barrier, consumed one round later -- and took it out again: hipcc's wait in front of a weight fragment is one immediate for
    // every path that reaches it, the smallest one, so the first fragment wait of every phase waited for the whole prefetch (no
    // gain: 179.7-181.0 pairs/s against 180.2-180.3 without it); with the first steps of a phase peeled out of the loop and the
    // scalar / vector staging paths as two copies of the round loop the 128-accumulator class spilled 728 bytes per lane.  Of the
    // 1.7 ms per step the staging costs this family, 0.86 ms is that memory round trip: profiles/conv_phases_r06.txt.)
    const int n_rounds = (n_chunks + CGS - 1) / CGS;
    // CGS channel chunks (8 channels each) are staged per barrier round, each into its own LDS image: the deep levels of the
    // hourglass have few tiles per launch and are a latency chain of rounds -- two chunks per round halve it
    for (int round = 0; round < n_rounds; ++round) {
        __syncthreads();   // the previous round's fragments are consumed
        // ---- stage: global fp32 -> (affine, relu) -> three bf16 planes, channels-last.  The 8 loads of a unit are UNCONDITIONAL
        // (clamped address; padding zeroed with an AND afterwards) so that they are all in flight before the first wait: a load
        // under a divergent branch is followed by s_waitcnt vmcnt(0), which serialised the 8 channels.
        // (Fetching a thread's unit for round r + 1 BEFORE the MFMA phase of round r was tried in round 3 and is slower: the weight
        // fragments below are global loads too, vmcnt retires in order, so the first fragment wait of the phase also waits for the
        // whole prefetch -- nothing overlaps; profiles/conv_phases_r03.txt.)
        for (int uu = threadIdx.x; uu < ((CD_SP_DBG & 2) ? 0 : CGS * UNITS); uu += kBlock) {
            const int g2 = uu / UNITS, u = uu - g2 * UNITS;
            float v[8][4];
            stage_load(round * CGS + g2, u, v);     // (a chunk beyond the last one: channels >= Cin, zeroed)
            stage_finish(round * CGS + g2, u, v, g2 * 3 * PLANE);
        }
        __syncthreads();

        // ---- MFMA over this wave's tap steps of the chunk
        const int lin_end = ((round + 1) * CGS < n_chunks ? (round + 1) * CGS : n_chunks) * KSTEPS;
        auto step = [&](const bf16x8 (&bcur)[NT][3], bf16x8 (&bfill)[NT][3], int l) {
            load_b(bfill, l + 4);   // this wave's next step: in flight during the MFMAs below
            const int chunk = l / KSTEPS;   // (wave-uniform)
            const int tap = (l - chunk * KSTEPS) * 2 + g;
            const int ky = tap / KS, kx = tap - ky * KS;
            // a padded tap carries zero weights: any valid slot
            const int slot0 = abase + (chunk - round * CGS) * 3 * PLANE + ((tap < TAPS) ? ky * RSP + kx : 0);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, smallest first
#pragma unroll
            for (int mg = 0; mg < MB; mg += MG) {
                bf16x8 a[MG][3];
#pragma unroll
                for (int m = 0; m < MG; ++m) {
                    const int slot = slot0 + (mg + m) * DY * RSP;
                    a[m][0] = __builtin_bit_cast(bf16x8, s_in[slot]);
                    a[m][1] = __builtin_bit_cast(bf16x8, s_in[PLANE + slot]);
                    a[m][2] = __builtin_bit_cast(bf16x8, s_in[2 * PLANE + slot]);
                }
#pragma unroll
                for (int p = 0; p < 6; ++p)   // round-robin over the accumulators: a dependent MFMA never follows its producer
#pragma unroll
                    for (int m = 0; m < MG; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[mg + m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[p]], bcur[t][PB[p]], acc[mg + m][t], 0, 0, 0);
            }
        };
        if (CD_SP_DBG & 1) { lin += ((lin_end - lin + 3) / 4) * 4; continue; }
#pragma unroll 1
        for (; lin + 4 < lin_end; lin += 8) {   // two steps per trip: the buffers swap roles without a register copy
            step(bA, bB, lin);
            step(bB, bA, lin + 4);
        }
        if (lin < lin_end) {   // odd step count of this wave in this chunk: one copy (and wait) per chunk
            step(bA, bB, lin);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bA[t][sp] = bB[t][sp];
            lin += 4;
        }
    }

    // ---- cross-wave reduction: M-tile 4j + o belongs to wave o; per round (j, t) the three other waves hand their partial
    // sums over through LDS and the owner adds (P0 + P1) + (P2 + P3)
    f32x4* s_red = reinterpret_cast<f32x4*>(smem_raw);   // [owner][foreign slot 0..2][4 register quads][64 lanes]
#pragma unroll
    for (int j = 0; j < MB / 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __syncthreads();   // the MFMA operands / the previous round are consumed
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o == wid) continue;   // wave-uniform
                const int fs = wid < o ? wid : wid - 1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    s_red[((o * 3 + fs) * 4 + q) * 64 + lane] = f32x4{acc[4 * j + o][t][4 * q], acc[4 * j + o][t][4 * q + 1], acc[4 * j + o][t][4 * q + 2], acc[4 * j + o][t][4 * q + 3]};
            }
            __syncthreads();
            // the owner's own partial sits at position `wid` of the fixed order
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 pw[4];
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    // (selected per wave below: all four acc tiles are compile-time indexed)
                    pw[w2] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (o != wid) continue;   // wave-uniform: o is this wave
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        if (w2 == o) pw[w2] = f32x4{acc[4 * j + o][t][4 * q], acc[4 * j + o][t][4 * q + 1], acc[4 * j + o][t][4 * q + 2], acc[4 * j + o][t][4 * q + 3]};
                        else pw[w2] = s_red[((o * 3 + (w2 < o ? w2 : w2 - 1)) * 4 + q) * 64 + lane];
                    }
                    const f32x4 sum = (pw[0] + pw[1]) + (pw[2] + pw[3]);
                    acc[4 * j + o][t][4 * q] = sum[0]; acc[4 * j + o][t][4 * q + 1] = sum[1]; acc[4 * j + o][t][4 * q + 2] = sum[2]; acc[4 * j + o][t][4 * q + 3] = sum[3];
                }
            }
        }

    // ---- epilogue: bias, store, batch statistics of the raw output.  D: column = lane&31, pixels 8q + 4g + {0..3};
    // this wave stores the M-tiles 4j + wid
    const int co_base = slice * COB;
    const int dy_l = DY == 2 ? (li >> 4) : 0, co_l = DY == 2 ? (li & 15) : li;
    float* yout = y + ((size_t)n * y_ctot + y_coff) * HW;
    double s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = co_base + t * CPT + co_l;
        const float bv = (bias != nullptr && co < Cout) ? bias[co] : 0.f;
        s1[t] = 0.0; s2[t] = 0.0;
#pragma unroll
        for (int j = 0; j < MB / 4; ++j) {
            const int gy = Y0 + (4 * j + wid) * DY + dy_l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gx = X0 + 8 * q + 4 * g;
                float e[4];
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o == wid) {   // wave-uniform select of the owned accumulator (compile-time register indices)
                        e[0] = acc[4 * j + o][t][4 * q] + bv; e[1] = acc[4 * j + o][t][4 * q + 1] + bv;
                        e[2] = acc[4 * j + o][t][4 * q + 2] + bv; e[3] = acc[4 * j + o][t][4 * q + 3] + bv;
                    }
                if (co < Cout && gy < H) {
                    float* dst = yout + (size_t)co * HW + (size_t)gy * W + gx;
                    if (gx + 3 < W && ((W & 3) == 0)) {
                        if (accumulate) {
                            const float4 o4 = *reinterpret_cast<const float4*>(dst);
                            e[0] += o4.x; e[1] += o4.y; e[2] += o4.z; e[3] += o4.w;
                        }
                        *reinterpret_cast<float4*>(dst) = make_float4(e[0], e[1], e[2], e[3]);
                        if (stats != nullptr) {
                            const double a = e[0], b = e[1], c = e[2], d = e[3];
                            s1[t] += (a + b) + (c + d);
                            s2[t] += (a * a + b * b) + (c * c + d * d);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (gx + k < W) {
                                if (accumulate) e[k] += dst[k];
                                dst[k] = e[k]; s1[t] += (double)e[k]; s2[t] += (double)e[k] * (double)e[k];
                            }
                    }
                }
            }
        }
    }
    if (stats != nullptr) {  // block-uniform; lanes of one channel: +32 (pixel half), and for DY = 2 also +16 (the other row)
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_raw);   // [4 waves][COB][2]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double a = s1[t], b = s2[t];
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (DY == 2) { a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64); }
            if (lane < CPT) { red[(wid * COB + t * CPT + lane) * 2] = a; red[(wid * COB + t * CPT + lane) * 2 + 1] = b; }
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

// (__launch_bounds__(.., 3) for the classes with 4 accumulator tiles per wave -- 168 registers, 16 bytes of scratch in the two-chunk
// classes, three resident workgroups -- measured in round 6: 176.6-177.7 against 176.9-177.1 pairs/s, no gain)
template <int KS, int NT, int TYP, int DY, int CGS>
__global__ __launch_bounds__(kBlock, 2) void conv_fwd_split_kernel(const SplitArgs a) {
    conv_fwd_split_block<KS, NT, TYP, DY, CGS>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// SEVERAL convolutions of one launch shape in ONE dispatch: blockIdx.y = the convolution (the three k x k branches of an inception:
// same image, same output channels, different filter sizes and input slices), blockIdx.x its stand-alone grid.  Kernels from
// different streams or graph branches do not share the chip on this stack (profiles/branch_overlap_r04.txt); workgroups of one
// dispatch do -- and at 96x56 and below a single branch launches fewer workgroups than the chip has CUs, each a latency chain of
// channel-chunk rounds.  Every workgroup does exactly what it does in its own launch: same bits.  The caller orders the branches
// largest filter first (workgroups are dispatched y-major: the long ones must not start last).
constexpr int kSplitMultiMax = 4;
struct SplitMulti { SplitArgs b[kSplitMultiMax]; int ks[kSplitMultiMax]; };

template <int NT, int TYP, int DY, int CGS>
__global__ __launch_bounds__(kBlock, 2) void conv_fwd_split_multi_kernel(const SplitMulti m) {
    const int br = (int)blockIdx.y;
    const SplitArgs& a = m.b[br];
    switch (m.ks[br]) {     // block-uniform
        case 3: conv_fwd_split_block<3, NT, TYP, DY, CGS>(a, (int)blockIdx.x, 0); break;
        case 5: conv_fwd_split_block<5, NT, TYP, DY, CGS>(a, (int)blockIdx.x, 0); break;
        case 7: conv_fwd_split_block<7, NT, TYP, DY, CGS>(a, (int)blockIdx.x, 0); break;
        default: conv_fwd_split_block<11, NT, TYP, DY, CGS>(a, (int)blockIdx.x, 0); break;
    }
}

template <int KS, int NT, int TYP, int DY, int CGS = 1>
static int launch_split_t(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                          const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate,
                          int N, int H, int W, const ConvGroups& grp, hipStream_t s) {
    using Cfg = SplitCfg<KS, TYP, DY>;
    const int tiles_x = (W + SP_TX - 1) / SP_TX, tiles_y = (H + Cfg::TY - 1) / Cfg::TY;
    const size_t img = CGS * Cfg::LDS + split_aff_bytes(Cin);      // the LDS images + the producer's scale / shift table behind them
    const size_t lds = img > SPLIT_REDUCE_LDS ? img : SPLIT_REDUCE_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_split_kernel<KS, NT, TYP, DY, CGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (lds > 160 * 1024) return CD_ERR_UNSUPPORTED;
    const int pack_tiles = split_ntiles(Cout), slices = (pack_tiles + NT - 1) / NT;
    const int tiles_img = tiles_x * tiles_y, tiles_total = tiles_img * N, chunk_tiles = (tiles_total + 7) / 8;
    SplitArgs a;
    a.x = x; a.wsp = reinterpret_cast<const u32x4*>(wsplit); a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.y = y; a.stats = stats;
    a.x_ctot = x_ctot; a.x_coff = x_coff; a.Cin = Cin; a.pack_tiles = pack_tiles; a.in_relu = in_relu; a.y_ctot = y_ctot; a.y_coff = y_coff;
    a.Cout = Cout; a.accumulate = accumulate; a.H = H; a.W = W; a.tiles_x = tiles_x; a.tiles_img = tiles_img; a.tiles_total = tiles_total;
    a.chunk_tiles = chunk_tiles; a.slices = slices; a.grp = grp;
    hipLaunchKernelGGL((conv_fwd_split_kernel<KS, NT, TYP, DY, CGS>), dim3((unsigned)chunk_tiles * 8u * (unsigned)slices, (unsigned)grp.n), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

template <int NT, int TYP, int DY, int CGS>
static int launch_split_multi_t(const SplitConv* c, int n, int N, int H, int W, int Cout, hipStream_t s) {
    const int tiles_x = (W + SP_TX - 1) / SP_TX, tiles_y = (H + TYP - 1) / TYP;
    const int pack_tiles = split_ntiles(Cout), slices = (pack_tiles + NT - 1) / NT;
    const int tiles_img = tiles_x * tiles_y, tiles_total = tiles_img * N, chunk_tiles = (tiles_total + 7) / 8;
    SplitMulti m;
    size_t lds = SPLIT_REDUCE_LDS;
    for (int i = 0; i < n; ++i) {
        SplitArgs& a = m.b[i];
        a.x = c[i].x; a.wsp = reinterpret_cast<const u32x4*>(c[i].wsplit); a.bias = c[i].bias; a.in_scale = c[i].in_scale; a.in_shift = c[i].in_shift;
        a.y = c[i].y; a.stats = c[i].stats;
        a.x_ctot = c[i].x_ctot; a.x_coff = c[i].x_coff; a.Cin = c[i].Cin; a.pack_tiles = pack_tiles; a.in_relu = c[i].in_relu; a.y_ctot = c[i].y_ctot;
        a.y_coff = c[i].y_coff; a.Cout = Cout; a.accumulate = c[i].accumulate; a.H = H; a.W = W; a.tiles_x = tiles_x; a.tiles_img = tiles_img;
        a.tiles_total = tiles_total; a.chunk_tiles = chunk_tiles; a.slices = slices; a.grp = ConvGroups();
        m.ks[i] = c[i].ks;
        const int ks = c[i].ks;
        const size_t need = (size_t)CGS * 3 * (TYP + ks - 1) * (SP_TX + 2 * ((((ks - 1) / 2) + 3) & ~3)) * 16 + split_aff_bytes(c[i].Cin);   // = CGS * SplitCfg<ks, TYP, DY>::LDS + the affine table
        if (need > lds) lds = need;
    }
    for (int i = n; i < kSplitMultiMax; ++i) { m.b[i] = m.b[0]; m.ks[i] = m.ks[0]; }
    if (lds > 160 * 1024) return CD_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_split_multi_kernel<NT, TYP, DY, CGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_fwd_split_multi_kernel<NT, TYP, DY, CGS>), dim3((unsigned)chunk_tiles * 8u * (unsigned)slices, (unsigned)n), dim3(kBlock), lds, s, m);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// n <= 4 convolutions with the SAME N, H, W, Cout and filter sizes in {3, 5, 7, 11}; (ty, cot) as launch_conv_split.
int launch_conv_split_multi(const SplitConv* c, int n, int N, int H, int W, int Cout, int ty, int cot, hipStream_t s) {
    if (n < 1 || n > kSplitMultiMax) return CD_ERR_UNSUPPORTED;
    for (int i = 0; i < n; ++i)
        if (!split_supported(c[i].ks)) return CD_ERR_UNSUPPORTED;
    const int nt = (cot >= 2 && split_ntiles(Cout) >= 2) ? 2 : 1;
    if (ty >= 32) {     // hint 32 (round 6): 8 M-tiles AND two channel chunks per barrier round -- 32 output channels per column tile only
        if (split_dy(Cout) == 2 || nt == 2) return CD_ERR_UNSUPPORTED;
        return launch_split_multi_t<1, 8, 1, 2>(c, n, N, H, W, Cout, s);
    }
    const bool two = ty >= 16;
    const int mb = (nt == 2 || ty <= 4 || two) ? 4 : 8;
    if (split_dy(Cout) == 2) {     // <= 16 output channels: 16 channels x 2 output rows per column tile (the shapes of launch_conv_split)
        if (mb == 8) return launch_split_multi_t<1, 16, 2, 1>(c, n, N, H, W, Cout, s);
        return two ? launch_split_multi_t<1, 8, 2, 2>(c, n, N, H, W, Cout, s) : launch_split_multi_t<1, 8, 2, 1>(c, n, N, H, W, Cout, s);
    }
    if (nt == 2) return two ? launch_split_multi_t<2, 4, 1, 2>(c, n, N, H, W, Cout, s) : launch_split_multi_t<2, 4, 1, 1>(c, n, N, H, W, Cout, s);
    if (mb == 8) return launch_split_multi_t<1, 8, 1, 1>(c, n, N, H, W, Cout, s);
    return two ? launch_split_multi_t<1, 4, 1, 2>(c, n, N, H, W, Cout, s) : launch_split_multi_t<1, 4, 1, 1>(c, n, N, H, W, Cout, s);
}

int split_column_tiles(int OC) { return split_ntiles(OC); }

size_t split_packed_floats(int OC, int IC, int ks) {
    if (!split_supported(ks)) return 0;
    const size_t chunks = ((size_t)IC + 7) / 8;
    return (size_t)split_ntiles(OC) * chunks * split_steps(ks, split_dy(OC)) * 3 * 64 * 4;   // 16 bytes = 4 floats per lane per split
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

// (ty, cot) are the launch-shape hints of cd_conv2d_fwd_cfg.  A block has 4 (ty <= 4) or 8 M-tiles (rows for DY = 1, row pairs
// for DY = 2) and 1 or 2 column tiles (cot >= 2 and the filter has >= 2 of them -> 2, then 4 M-tiles); ty = 16 selects 4 M-tiles
// with two 8-channel chunks staged per barrier round.
int launch_conv_split(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                      const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                      int H, int W, int ks, int ty, int cot, hipStream_t s, const ConvGroups& grp) {
    const int dy = split_dy(Cout);
    const int nt = (cot >= 2 && split_ntiles(Cout) >= 2) ? 2 : 1;
    const bool wide2 = ty >= 32;                      // hint 32: 8 M-tiles, two channel chunks per barrier round (half the rounds of the 8-tile class)
    if (wide2 && (dy == 2 || nt == 2)) return CD_ERR_UNSUPPORTED;
    const bool two = ty >= 16;                        // hint 16: 4 M-tiles, two channel chunks per barrier round (latency-bound small images)
    const int mb = (nt == 2 || ty <= 4 || two) ? 4 : 8;
#define CD_SP(K, T, Y, D, G) return launch_split_t<K, T, Y, D, G>(x, x_ctot, x_coff, Cin, wsplit, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, grp, s)
#define CD_SP_K(K)                                                   \
    if (ks == K) {                                                   \
        if (dy == 2) { if (mb == 8) CD_SP(K, 1, 16, 2, 1); if (two) CD_SP(K, 1, 8, 2, 2); CD_SP(K, 1, 8, 2, 1); } \
        if (nt == 2) { if (two) CD_SP(K, 2, 4, 1, 2); CD_SP(K, 2, 4, 1, 1); }  \
        if (wide2) CD_SP(K, 1, 8, 1, 2);                             \
        if (mb == 8) CD_SP(K, 1, 8, 1, 1);                           \
        if (two) CD_SP(K, 1, 4, 1, 2);                               \
        CD_SP(K, 1, 4, 1, 1);                                        \
    }
    CD_SP_K(3) CD_SP_K(5) CD_SP_K(7) CD_SP_K(11)
#undef CD_SP_K
#undef CD_SP
    return CD_ERR_UNSUPPORTED;
}

}  // namespace cd
