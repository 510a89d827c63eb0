// 1x1 convolution (a GEMM over pixels) at fp32 accuracy on the gfx950 BF16 matrix cores, forward and input gradient.
//
// Same arithmetic as conv_split.hip (three exact bf16 terms per fp32 operand, six products, fp32 accumulate), different
// data flow: with one tap there is nothing to reuse in an LDS input tile -- the staged-tile kernels spend their time in
// global -> registers -> LDS -> fragments and two barriers per 64 MFMAs.  Here
//   * the block's slice of the filter (NT x 32 output channels x all input channels, three bf16 planes, <= 96 KB) is copied
//     into LDS ONCE and stays for the block's lifetime (a block walks many pixel tiles);
//   * activations never touch LDS: a lane loads the 8 channels of ITS pixel straight from global memory (32 consecutive
//     pixels per half wave and channel: full 128-byte lines), applies the producer's BN-affine / ReLU and splits them in
//     registers -- that IS the A fragment of v_mfma_f32_32x32x16_bf16 (M = 32 pixels, K = 16 channels: lane (i, g) holds
//     channels 8g..8g+7 of pixel i);
//   * waves are independent (no barrier in the main loop); the next K-step's loads are in flight during the MFMAs.
// D: lane holds output channel lane&31, pixels 8q + 4*(lane>>5) + {0..3} -> 16-byte stores; bias, gradient accumulation and
// the BatchNorm statistics of the raw output (fp64, reduced per block, one atomic pair per channel) as in conv_mfma.hip.
// The order of accumulation is K-step by K-step for every launch: results do not depend on the launch shape.
#include "cd_common.h"
#include "conv_split.h"

namespace cd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned p1_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void p1_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = p1_cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = p1_cvt_pk_bf16(ra, rb);
    l = p1_cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

// ---------------------------------------------------------------- weight packing
// [column tile of 32][K-step of 16 channels][split][lane][8 bf16]: element e of lane (n = lane&31, g = lane>>5) = w[tile*32 + n][step*16 + 8g + e]
__device__ __forceinline__ void pack_1x1_elements(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout_src, int Cin_src,
                                                  int transposed, int OC, int IC, int oc_off, int ic_off, size_t first, size_t stride) {
    const int oc_n = transposed ? Cin_src : Cout_src, ic_n = transposed ? Cout_src : Cin_src;
    const int ksteps = (IC + 15) / 16, tiles = (OC + 31) / 32;
    const size_t total = (size_t)tiles * ksteps * 512;
    for (size_t i = first; i < total; i += stride) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int lane = (int)(r & 63); r >>= 6;
        const int step = (int)(r % ksteps); r /= ksteps;
        const int tile = (int)r;
        const int oc = tile * 32 + (lane & 31) - oc_off, ic = step * 16 + (lane >> 5) * 8 + e - ic_off;
        if ((unsigned)oc >= (unsigned)oc_n || (unsigned)ic >= (unsigned)ic_n) continue;   // padding stays zero
        const float v = transposed ? w[(size_t)ic * Cin_src + oc] : w[(size_t)oc * Cin_src + ic];
        unsigned h, m, l;
        p1_split_pair(v, 0.f, h, m, l);
        const size_t base = (((size_t)tile * ksteps + step) * 3) * 512 + (size_t)lane * 8 + e;
        out[base] = (unsigned short)h; out[base + 512] = (unsigned short)m; out[base + 1024] = (unsigned short)l;
    }
}

__global__ void pack_1x1_kernel(const float* __restrict__ w, int Cout, int Cin, int transposed, unsigned short* __restrict__ out) {
    const int OC = transposed ? Cin : Cout, IC = transposed ? Cout : Cin;
    pack_1x1_elements(w, out, Cout, Cin, transposed, OC, IC, 0, 0, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

__global__ void pack_1x1_table_kernel(const PackDesc* __restrict__ table) {
    const PackDesc d = table[blockIdx.y];
    if (!split_1x1_supported(d.ks, d.OC, d.IC)) return;
    float* out = d.packed + fp32_packed_floats(d.OC, d.IC, d.ks);
    pack_1x1_elements(d.w, reinterpret_cast<unsigned short*>(out), d.Cout, d.Cin, d.transposed, d.OC, d.IC, d.oc_off, d.ic_off,
                      (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

size_t split_1x1_packed_floats(int OC, int IC) { return (size_t)((OC + 31) / 32) * ((IC + 15) / 16) * 3 * 64 * 4; }

int launch_pack_1x1_table(const void* table_dev, int n, hipStream_t s) {
    // (64 workgroups per descriptor since the end of round 6: with the 512...2048-channel filters of MiDaS in this layout 8 were a latency
    // chain of 1.5 ms per step at the top of every forward)
    hipLaunchKernelGGL(pack_1x1_table_kernel, dim3(64, n), dim3(256), 0, s, (const PackDesc*)table_dev);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

int launch_pack_1x1(const float* w, int Cout, int Cin, int transposed, float* packed_split, hipStream_t s) {
    const int OC = transposed ? Cin : Cout, IC = transposed ? Cout : Cin;
    const size_t total = split_1x1_packed_floats(OC, IC) * 2 / 3;
    size_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_1x1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, Cout, Cin, transposed, reinterpret_cast<unsigned short*>(packed_split));
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- the convolution

// NT = column tiles (32 output channels each) per block, NW = waves per block, DEPTH = register buffers of raw activations
// (DEPTH - 1 K-steps of loads in flight), WPE = waves per SIMD the register budget is set for.
template <int NT, int NW, int DEPTH, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void conv1x1_split_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const u32x4* __restrict__ wsp, int col_tiles, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    float* __restrict__ y, int y_ctot, int y_coff, int Cout,
    double* __restrict__ stats, int accumulate, int H, int W, int tiles_x, int tiles_total, int slices, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char p1_smem[];
    const int ksteps_real = (Cin + 15) / 16, ksteps = (ksteps_real + DEPTH - 1) / DEPTH * DEPTH;   // padded K-steps: zero weights, nothing loaded
    u32x4* s_w = reinterpret_cast<u32x4*>(p1_smem);                                   // [kstep][split][NT][64 lanes]
    float* s_aff = reinterpret_cast<float*>(p1_smem + (size_t)ksteps * 3 * NT * 1024);   // [2][ksteps * 16]: scale, shift (padded channels: 0)
    double* s_red = reinterpret_cast<double*>(s_aff + 2 * ksteps * 16);                // [NW][NT * 32][2]

    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (own L2 each); the channel slices of one pixel range are
    // consecutive in the dispatch order of ONE XCD, so the input they all read is fetched into a single L2
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int slice = jx % slices, blk = (jx / slices) * 8 + xcd;
    if (blk >= nblk) return;   // block-uniform (grid padded to a multiple of 8 pixel blocks)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 31, g = lane >> 5;
    const size_t HW = (size_t)H * W;

    // ---- the filter slice and the input transform of every channel, once per block
    for (int i = threadIdx.x; i < ksteps * 3 * NT * 64; i += NW * 64) {
        const int ln = i & 63, t = (i >> 6) % NT, sp = (i / (64 * NT)) % 3, ks = i / (64 * NT * 3);
        const int gt = slice * NT + t;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (gt < col_tiles && ks < ksteps_real) v = wsp[(((size_t)gt * ksteps_real + ks) * 3 + sp) * 64 + ln];
        s_w[i] = v;
    }
    for (int i = threadIdx.x; i < ksteps * 16; i += NW * 64) {
        s_aff[i] = (in_scale && i < Cin) ? in_scale[i] : (i < Cin ? 1.f : 0.f);
        s_aff[ksteps * 16 + i] = (in_shift && i < Cin) ? in_shift[i] : 0.f;
    }
    __syncthreads();

    double s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }
    const int co_base = slice * NT * 32;
    const bool aligned4 = (W & 3) == 0;

    // this lane's pixel of a tile: 32-bit offset of channel x_coff (the launcher guarantees < 2^30 elements).  Every load below is
    // UNCONDITIONAL (clamped address, value zeroed by a select): a load under a divergent branch makes the compiler wait for ALL
    // outstanding loads (s_waitcnt vmcnt(0)) at every use, which serialised the prefetch ring (measured: matrix cores 28 % busy).
    auto tile_px = [&](int tile, unsigned& px, bool& ok) {
        const int n = tile / (H * tiles_x), rem = tile - n * (H * tiles_x);
        const int gy = rem / tiles_x, gx0 = (rem - gy * tiles_x) * 32;
        ok = tile < tiles_total && gx0 + li < W;
        px = ok ? (unsigned)(((size_t)n * x_ctot + x_coff) * HW + (size_t)gy * W + gx0 + li) : (unsigned)((size_t)x_coff * HW);
    };
    const unsigned hw32 = (unsigned)HW;
    auto load_raw = [&](float (&dst)[8], unsigned px, bool ok, int ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = ks * 16 + g * 8 + e;
            const unsigned cc = (unsigned)(ci < Cin ? ci : Cin - 1);
            const unsigned keep = (ok && ci < Cin) ? 0xffffffffu : 0u;   // (an AND, not a select: the compiler sinks a load into a select's branch)
            dst[e] = __uint_as_float(__float_as_uint(x[px + cc * hw32]) & keep);
        }
    };
    // Raw activations travel through a ring of DEPTH register buffers that runs ACROSS tiles: while K-step ks of a tile is
    // computed, the loads of the next DEPTH - 1 steps -- of this tile or the first ones of the wave's next tile -- are in flight
    // (a CU sustains bandwidth = bytes in flight / latency; refilling the ring per tile left the matrix cores 28 % busy).
    const int stride = nblk * NW;
    unsigned px, px_n;
    bool ok, ok_n;
    tile_px(blk * NW + wid, px, ok);
    tile_px(blk * NW + wid + stride, px_n, ok_n);
    float r[DEPTH][8];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) load_raw(r[j], px, ok, j);

    for (int tile = blk * NW + wid; tile < tiles_total; tile += stride) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

        auto process = [&](const float (&raw)[8], int ks) {
            // the producer's affine + ReLU on this lane's 8 channels (a padded channel or pixel is an exact zero)
            float v[8];
            if (in_scale) {
                const float4 sc0 = *reinterpret_cast<const float4*>(s_aff + ks * 16 + g * 8), sc1 = *reinterpret_cast<const float4*>(s_aff + ks * 16 + g * 8 + 4);
                const float4 sh0 = *reinterpret_cast<const float4*>(s_aff + ksteps * 16 + ks * 16 + g * 8), sh1 = *reinterpret_cast<const float4*>(s_aff + ksteps * 16 + ks * 16 + g * 8 + 4);
                const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w}, sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ok ? __fmaf_rn(raw[e], sc[e], sh[e]) : 0.f;   // same fma as the BN backward's mask
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = raw[e];
            }
            if (in_relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x4 hh, mm, ll;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                unsigned h, mi, l;
                p1_split_pair(v[2 * c2], v[2 * c2 + 1], h, mi, l);
                hh[c2] = h; mm[c2] = mi; ll[c2] = l;
            }
            const bf16x8 a[3] = {__builtin_bit_cast(bf16x8, hh), __builtin_bit_cast(bf16x8, mm), __builtin_bit_cast(bf16x8, ll)};
            bf16x8 b[NT][3];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) b[t][sp] = __builtin_bit_cast(bf16x8, s_w[((ks * 3 + sp) * NT + t) * 64 + lane]);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, smallest first, round-robin over the accumulators
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]], b[t][PB[p]], acc[t], 0, 0, 0);
        };
#pragma unroll 1
        for (int ks = 0; ks < ksteps; ks += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {   // compile-time buffer indices: ksteps is a multiple of DEPTH
                const int sn = ks + j + DEPTH - 1;   // refill the buffer consumed one step ago
                if (sn < ksteps) load_raw(r[(j + DEPTH - 1) % DEPTH], px, ok, sn);
                else load_raw(r[(j + DEPTH - 1) % DEPTH], px_n, ok_n, sn - ksteps);
                process(r[j], ks + j);
                __builtin_amdgcn_sched_barrier(0);   // (keep the fragments of the next step out of this step's registers)
            }
        }
        const int n = tile / (H * tiles_x), rem_ = tile - n * (H * tiles_x);
        const int gy = rem_ / tiles_x, gx0 = (rem_ - gy * tiles_x) * 32;
        px = px_n; ok = ok_n;
        tile_px(tile + 2 * stride, px_n, ok_n);

        // ---- epilogue of the tile: bias, store, statistics partials.  (A transposed store through a per-wave LDS tile -- full
        // 128-byte lines per instruction instead of 64 x 16 bytes -- was measured 10-25 % slower: the stores are not the limit.)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = co_base + t * 32 + li;
            const float bv = (bias != nullptr && co < Cout) ? bias[co] : 0.f;
            if (co >= Cout) continue;
            float* yrow = y + ((size_t)n * y_ctot + y_coff + co) * HW + (size_t)gy * W;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gx = gx0 + 8 * q + 4 * g;
                float e4[4] = {acc[t][4 * q] + bv, acc[t][4 * q + 1] + bv, acc[t][4 * q + 2] + bv, acc[t][4 * q + 3] + bv};
                float* dst = yrow + gx;
                if (gx + 3 < W && aligned4) {
                    if (accumulate) {
                        const float4 o4 = *reinterpret_cast<const float4*>(dst);
                        e4[0] += o4.x; e4[1] += o4.y; e4[2] += o4.z; e4[3] += o4.w;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(e4[0], e4[1], e4[2], e4[3]);
                    if (stats != nullptr) {
                        const double a0 = e4[0], a1 = e4[1], a2 = e4[2], a3 = e4[3];
                        s1[t] += (a0 + a1) + (a2 + a3);
                        s2[t] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (gx + k < W) {
                            if (accumulate) e4[k] += dst[k];
                            dst[k] = e4[k];
                            s1[t] += (double)e4[k]; s2[t] += (double)e4[k] * (double)e4[k];
                        }
                }
            }
        }
    }

    if (stats != nullptr) {   // block-uniform: per channel, lanes lane and lane+32 -> waves -> one atomic pair per block
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double a = s1[t], b = s2[t];
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (lane < 32) { s_red[((wid * NT + t) * 32 + lane) * 2] = a; s_red[((wid * NT + t) * 32 + lane) * 2 + 1] = b; }
        }
        __syncthreads();
        if (threadIdx.x < NT * 32) {
            const int co = co_base + threadIdx.x;
            if (co < Cout) {
                double a = 0.0, b = 0.0;
                for (int w2 = 0; w2 < NW; ++w2) { a += s_red[((w2 * NT) * 32 + threadIdx.x) * 2]; b += s_red[((w2 * NT) * 32 + threadIdx.x) * 2 + 1]; }
                const int slot = blk & (CD_BN_STAT_SLOTS - 1);
                double* st = stats + ((size_t)slot * y_ctot + y_coff + co) * 2;
                atomicAdd(st, a);
                atomicAdd(st + 1, b);
            }
        }
    }
}

template <int NT, int NW, int DEPTH, int WPE>
static int launch_1x1_t(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                        const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                        int H, int W, size_t lds, int blocks_per_cu, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv1x1_split_kernel<NT, NW, DEPTH, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int col_tiles = (Cout + 31) / 32, slices = (col_tiles + NT - 1) / NT;
    const int tiles_x = (W + 31) / 32, tiles_total = N * H * tiles_x;
    int nblk = (256 * blocks_per_cu + slices - 1) / slices;          // blocks per slice: fill the chip, every block loads the filter once
    const int max_blk = (tiles_total + NW - 1) / NW;
    if (nblk > max_blk) nblk = max_blk;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL((conv1x1_split_kernel<NT, NW, DEPTH, WPE>), dim3((unsigned)((nblk + 7) / 8) * 8u * (unsigned)slices), dim3(NW * 64), lds, s, x, x_ctot, x_coff, Cin,
                       reinterpret_cast<const u32x4*>(wsplit), col_tiles, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats,
                       accumulate, H, W, tiles_x, tiles_total, slices, nblk);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- the same GEMM for filters too large to stay in LDS (round 6)
// Dense 1x1 convolutions with 512...2048 channels on small images (the ResNeXt-101 encoder of MiDaS v2, BASELINE configs[4]: 46 of its
// 1x1 convolutions are 1024 x 1024 on 24 x 24 pixels) ran on the staged fp32-MFMA kernel at 44 TFLOP/s: a 64 x 1024 filter slice is
// 393 KB of bf16 planes.  Here the slice travels through LDS in CHUNKS of 4 K-steps (64 input channels, 48 KB for 128 output
// channels), double-buffered: a chunk is fetched into registers before the 96 MFMAs of the current one and written to the other
// buffer after them, one barrier per chunk.  Everything else is the kernel above: activations straight from global memory into the A
// fragment through a 4-deep register ring, six products per K-step, the same epilogue.  Pixel tiles are 32 consecutive pixels of the
// FLATTENED (image, y, x) index (H*W % 4 == 0: a quad of a lane's outputs never straddles two images) -- 12 x 12 and 24 x 24 planes
// waste no lanes, which row tiles would (37 % / 75 % occupancy).  One tile per wave, NW waves and NT x 32 output channels per
// workgroup (launched with NW = 8, NT = 4).
// The order of accumulation is K-step by K-step, as above: the result does not depend on NW.
template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void conv1x1_split_kc_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const u32x4* __restrict__ wsp, int col_tiles, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    float* __restrict__ y, int y_ctot, int y_coff, int Cout,
    double* __restrict__ stats, int accumulate, int HW, int P_total, int tiles_total, int slices, int ngroups) {
    constexpr int DEPTH = 4, KC = 4, CHUNK = KC * 3 * NT * 64;   // u32x4 per chunk
    constexpr int FQ = (CHUNK + NW * 64 - 1) / (NW * 64);
    extern __shared__ __attribute__((aligned(16))) unsigned char p1_smem[];
    const int ksteps_real = (Cin + 15) / 16, ksteps = (ksteps_real + KC - 1) / KC * KC;
    u32x4* s_w = reinterpret_cast<u32x4*>(p1_smem);                                        // [2][KC][split][NT][64 lanes]
    float* s_aff = reinterpret_cast<float*>(p1_smem + (size_t)2 * CHUNK * 16);               // [2][ksteps * 16]
    double* s_red = reinterpret_cast<double*>(p1_smem);                                     // [NW][NT * 32][2], over the filter buffers after the loop

    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int slice = jx % slices, grp = (jx / slices) * 8 + xcd;
    if (grp >= ngroups) return;   // block-uniform
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 31, g = lane >> 5;
    const unsigned hw32 = (unsigned)HW;

    for (int i = threadIdx.x; i < ksteps * 16; i += NW * 64) {
        s_aff[i] = (in_scale && i < Cin) ? in_scale[i] : (i < Cin ? 1.f : 0.f);
        s_aff[ksteps * 16 + i] = (in_shift && i < Cin) ? in_shift[i] : 0.f;
    }
    // filter chunk `c` (K-steps 4c .. 4c + 3) of this block's NT column tiles: global -> registers, registers -> LDS buffer
    // (fetch only ISSUES the loads -- unconditional, on a clamped address -- and commit masks the dead elements: a use of the loaded
    // value inside fetch, even the masking AND, makes the compiler wait for it there with vmcnt(0), which also drains the activation
    // ring at the top of every chunk: measured as 42 % matrix-core utilisation in the first version of this kernel)
    u32x4 wreg[FQ];
    auto chunk_src = [&](int c, int q, bool& live) -> size_t {
        const int i = q * (NW * 64) + threadIdx.x;
        const int ln = i & 63, t = (i >> 6) % NT, sp = (i / (64 * NT)) % 3, ksl = i / (64 * NT * 3);
        const int gt = slice * NT + t, ks = c * KC + ksl;
        live = i < CHUNK && gt < col_tiles && ks < ksteps_real;
        return live ? (((size_t)gt * ksteps_real + ks) * 3 + sp) * 64 + ln : 0;
    };
    auto fetch = [&](int c) {
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            bool live;
            wreg[q] = wsp[chunk_src(c, q, live)];
        }
    };
    auto commit = [&](int c, int buf) {
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int i = q * (NW * 64) + threadIdx.x;
            bool live;
            (void)chunk_src(c, q, live);
            const unsigned keep = live ? 0xffffffffu : 0u;
            u32x4 v = wreg[q];
            v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
            if (i < CHUNK) s_w[buf * CHUNK + i] = v;
        }
    };
    fetch(0);
    commit(0, 0);

    // this lane's pixel
    const int tile = grp * NW + wid;
    const int P = tile * 32 + li;
    const bool ok = tile < tiles_total && P < P_total;
    const int pn = ok ? P / HW : 0, pp = ok ? P - pn * HW : 0;
    const unsigned px = (unsigned)(((size_t)pn * x_ctot + x_coff) * HW + (size_t)pp);
    auto load_raw = [&](float (&dst)[8], int ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = ks * 16 + g * 8 + e;
            const unsigned cc = (unsigned)(ci < Cin ? ci : Cin - 1);
            const unsigned keep = (ok && ci < Cin) ? 0xffffffffu : 0u;
            dst[e] = __uint_as_float(__float_as_uint(x[px + cc * hw32]) & keep);
        }
    };
    float r[DEPTH][8];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) load_raw(r[j], j);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    __syncthreads();

    int buf = 0;
#pragma unroll 1
    for (int ks = 0; ks < ksteps; ks += KC) {
        const bool more = ks + KC < ksteps;
        if (more) fetch(ks / KC + 1);
        const u32x4* sw = s_w + buf * CHUNK;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const int sn = ks + j + DEPTH - 1;
            load_raw(r[(j + DEPTH - 1) % DEPTH], sn < ksteps ? sn : ksteps - 1);   // (past the end: a harmless reload, never used)
            {
                const float (&raw)[8] = r[j];
                const int kk = ks + j;
                float v[8];
                if (in_scale) {
                    const float4 sc0 = *reinterpret_cast<const float4*>(s_aff + kk * 16 + g * 8), sc1 = *reinterpret_cast<const float4*>(s_aff + kk * 16 + g * 8 + 4);
                    const float4 sh0 = *reinterpret_cast<const float4*>(s_aff + ksteps * 16 + kk * 16 + g * 8), sh1 = *reinterpret_cast<const float4*>(s_aff + ksteps * 16 + kk * 16 + g * 8 + 4);
                    const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w}, sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ok ? __fmaf_rn(raw[e], sc[e], sh[e]) : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = raw[e];
                }
                if (in_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                u32x4 hh, mm, ll;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    unsigned h, mi, l;
                    p1_split_pair(v[2 * c2], v[2 * c2 + 1], h, mi, l);
                    hh[c2] = h; mm[c2] = mi; ll[c2] = l;
                }
                const bf16x8 a[3] = {__builtin_bit_cast(bf16x8, hh), __builtin_bit_cast(bf16x8, mm), __builtin_bit_cast(bf16x8, ll)};
                bf16x8 b[NT][3];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) b[t][sp] = __builtin_bit_cast(bf16x8, sw[((j * 3 + sp) * NT + t) * 64 + lane]);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, smallest first (as above)
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]], b[t][PB[p]], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) commit(ks / KC + 1, buf ^ 1);
        __syncthreads();   // the next chunk is in LDS, and nobody reads `buf` any more
        buf ^= 1;
    }

    // ---- epilogue: bias, store, statistics partials (the layout of the kernel above; the pixel quad of (q, g) is 4 consecutive
    // flattened pixels of one image)
    double s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.0; s2[t] = 0.0; }
    const int co_base = slice * NT * 32;
    if (tile < tiles_total) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = co_base + t * 32 + li;
            if (co >= Cout) continue;
            const float bv = bias != nullptr ? bias[co] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int P0 = tile * 32 + 8 * q + 4 * g;
                if (P0 >= P_total) continue;
                const int n = P0 / HW, p0 = P0 - n * HW;
                float* dst = y + ((size_t)n * y_ctot + y_coff + co) * HW + p0;
                float e4[4] = {acc[t][4 * q] + bv, acc[t][4 * q + 1] + bv, acc[t][4 * q + 2] + bv, acc[t][4 * q + 3] + bv};
                if (accumulate) {
                    const float4 o4 = *reinterpret_cast<const float4*>(dst);
                    e4[0] += o4.x; e4[1] += o4.y; e4[2] += o4.z; e4[3] += o4.w;
                }
                *reinterpret_cast<float4*>(dst) = make_float4(e4[0], e4[1], e4[2], e4[3]);
                if (stats != nullptr) {
                    const double a0 = e4[0], a1 = e4[1], a2 = e4[2], a3 = e4[3];
                    s1[t] += (a0 + a1) + (a2 + a3);
                    s2[t] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
        }
    }
    if (stats != nullptr) {   // block-uniform
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double a = s1[t], b = s2[t];
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (lane < 32) { s_red[((wid * NT + t) * 32 + lane) * 2] = a; s_red[((wid * NT + t) * 32 + lane) * 2 + 1] = b; }
        }
        __syncthreads();
        if (threadIdx.x < NT * 32) {
            const int co = co_base + threadIdx.x;
            if (co < Cout) {
                double a = 0.0, b = 0.0;
                for (int w2 = 0; w2 < NW; ++w2) { a += s_red[((w2 * NT) * 32 + threadIdx.x) * 2]; b += s_red[((w2 * NT) * 32 + threadIdx.x) * 2 + 1]; }
                const int slot = grp & (CD_BN_STAT_SLOTS - 1);
                double* st = stats + ((size_t)slot * y_ctot + y_coff + co) * 2;
                atomicAdd(st, a);
                atomicAdd(st + 1, b);
            }
        }
    }
}

template <int NT, int NW>
static int launch_1x1_kc_t(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                           const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                           int H, int W, hipStream_t s) {
    const int ksteps = ((Cin + 15) / 16 + 3) / 4 * 4;
    const size_t lds = (size_t)2 * 4 * 3 * NT * 1024 + (size_t)2 * ksteps * 16 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv1x1_split_kc_kernel<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int col_tiles = (Cout + 31) / 32, slices = (col_tiles + NT - 1) / NT;
    const int P_total = N * H * W, tiles_total = (P_total + 31) / 32, ngroups = (tiles_total + NW - 1) / NW;
    hipLaunchKernelGGL((conv1x1_split_kc_kernel<NT, NW>), dim3((unsigned)((ngroups + 7) / 8) * 8u * (unsigned)slices), dim3(NW * 64), lds, s, x, x_ctot,
                       x_coff, Cin, reinterpret_cast<const u32x4*>(wsplit), col_tiles, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout,
                       stats, accumulate, H * W, P_total, tiles_total, slices, ngroups);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// wide filters / small planes: see conv1x1_split_kc_kernel.  H*W % 4 == 0 and N*H*W < 2^26 (the caller checks the element count)
bool conv1x1_split_kc_ok(int Cin, int Cout, int N, int H, int W) {
    return Cin >= 64 && Cout >= 64 && ((H * W) & 3) == 0 && (long long)N * H * W < (1LL << 26) && (size_t)((Cin + 63) / 64 * 64) * 2 * 4 <= 48 * 1024;
}
int launch_conv1x1_split_kc(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                            const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate,
                            int N, int H, int W, hipStream_t s) {
    if (!conv1x1_split_kc_ok(Cin, Cout, N, H, W)) return CD_ERR_UNSUPPORTED;
    // One shape: 8 waves x 128 output channels (216 registers: two waves per SIMD).  A 12-wave workgroup would balance the 24 x 24 planes
    // of MiDaS' layer3 better (288 pixel tiles x 8 slices = 288 workgroups of 8 waves take two rounds on 256 CUs, 192 of 12 waves one
    // round of 1.5x the length) but needs 168 registers per wave and spills 156 bytes: measured 210 us against 183 for 1024 x 1024, not shipped.
    // (NT = 5 -- 160 output channels per workgroup, 250 registers, no spill -- makes the 1024 x 1024 convolutions on 288 pixel tiles ONE round of
    // 252 workgroups: measured 215 us against 194 for the two rounds of NT = 4, the longer workgroup loses more than the round saves.)
    return launch_1x1_kc_t<4, 8>(x, x_ctot, x_coff, Cin, wsplit, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats, accumulate, N, H, W, s);
}

int launch_conv1x1_split(const float* x, int x_ctot, int x_coff, int Cin, const float* wsplit, const float* bias, const float* in_scale,
                         const float* in_shift, int in_relu, float* y, int y_ctot, int y_coff, int Cout, double* stats, int accumulate, int N,
                         int H, int W, hipStream_t s) {
    const int ksteps = (Cin + 15) / 16, col_tiles = (Cout + 31) / 32;
    // LDS of a block: filter slice + affine table (both over the K-steps padded to a multiple of the ring depth) + statistics scratch
    auto lds_of = [&](int nt, int nw, int depth) {
        const size_t kp = (size_t)(ksteps + depth - 1) / depth * depth;
        return kp * 3 * nt * 1024 + 2 * kp * 16 * 4 + (size_t)nw * nt * 32 * 16;
    };
#define CD_1X1(NT_, NW_, D_, WPE_, BPC_)                                                                                                   \
    return launch_1x1_t<NT_, NW_, D_, WPE_>(x, x_ctot, x_coff, Cin, wsplit, bias, in_scale, in_shift, in_relu, y, y_ctot, y_coff, Cout, stats, \
                                            accumulate, N, H, W, lds_of(NT_, NW_, D_), BPC_, s)
    // 128 output channels per block (the input is read and split once per 128 instead of once per 64) when the filter slice fits
    // and the convolution has them: one 8-wave block per CU, 256 registers
    if (col_tiles > 2 && lds_of(4, 8, 4) <= 150 * 1024) CD_1X1(4, 8, 4, 2, 1);
    // 64 output channels per block: three 4-wave blocks per CU for small filter slices, else one 8-wave block
    if (lds_of(2, 4, 4) <= 52 * 1024) CD_1X1(2, 4, 4, 3, 3);
    if (lds_of(2, 8, 3) <= 160 * 1024) CD_1X1(2, 8, 3, 3, 1);
    return CD_ERR_UNSUPPORTED;
#undef CD_1X1
}

}  // namespace cd
