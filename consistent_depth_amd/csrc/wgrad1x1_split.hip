// Weight gradient of the 1x1 convolution at fp32 accuracy on the gfx950 BF16 matrix cores:
//
//   dW[co][ci] = sum over pixels p of dY[co][p] * act(X)[ci][p]          (a GEMM whose reduction dimension is the pixels)
//
// Same arithmetic as the other split kernels (three exact bf16 terms per fp32 operand, six products, fp32 accumulate).
// Both operands are pixel-contiguous in memory, which is exactly the K-contiguous layout v_mfma_f32_32x32x16_bf16 wants
// (M = 32 output channels, N = 32 input channels, K = 16 pixels: lane (c = lane&31, g = lane>>5) holds pixels 8g..8g+7 of
// channel c): a lane loads its 8 pixels with two 16-byte loads straight from global memory and splits them in registers.
// No LDS, no barriers.  A wave owns a patch of 64 x 128 (co x ci) = 8 accumulator tiles and streams a contiguous range of
// 16-pixel steps; the (up to) 4 waves of a block own different patches over the SAME pixels (shared L1/L2 lines).  Every
// wave stores its partial sums once into its own slice of the workspace, packed [split][co grp][ci grp][64][128] -- the
// layout the fixed-order unpack kernels of conv_wgrad.hip read: no atomics, bit-reproducible.
// All loads are unconditional (clamped channel, masked with an AND): see conv1x1_split.hip for why.
#include "cd_common.h"
#include "wgrad_split.h"

namespace cd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned w1_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 8 fp32 -> three bf16x8 fragments (hi, mid, lo)
__device__ __forceinline__ void w1_split8(const float (&v)[8], bf16x8 (&f)[3]) {
    u32x4 hh, mm, ll;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a = v[2 * c], b = v[2 * c + 1];
        const unsigned h = w1_cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = w1_cvt_pk_bf16(ra, rb);
        const unsigned l = w1_cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
        hh[c] = h; mm[c] = m; ll[c] = l;
    }
    f[0] = __builtin_bit_cast(bf16x8, hh); f[1] = __builtin_bit_cast(bf16x8, mm); f[2] = __builtin_bit_cast(bf16x8, ll);
}

constexpr int W1_CO_T = WGRAD1X1_COB / 32, W1_CI_T = WGRAD1X1_CIB / 32;   // accumulator tiles of a patch: 2 x 4

// grid: x = pixel split, y = patch group.  A block's 4 waves: patch = group * PG + wid % PG, sub-split = wid / PG.
__global__ __launch_bounds__(kBlock, 2) void wgrad1x1_split_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int Cin,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
    const float* __restrict__ dy, int dy_ctot, int dy_coff, int Cout,
    float* __restrict__ dw_packed, int N, int HW16 /* 16-pixel steps per image */, int cigs, int patches, int pg, int sub) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 31, g = lane >> 5;
    const int patch = blockIdx.y * pg + wid % pg, ssub = wid / pg;
    if (patch >= patches || ssub >= sub) return;   // wave-uniform: an idle wave (3 patches per group)
    const int cog = patch / cigs, cig = patch - cog * cigs;
    const int split = blockIdx.x * sub + ssub, splits = gridDim.x * sub;
    const size_t HW = (size_t)HW16 * 16;
    // this wave's contiguous range of 16-pixel steps over (image, pixel)
    const long long steps = (long long)N * HW16;
    const long long s0 = steps * split / splits, s1 = steps * (split + 1) / splits;

    // per-lane channels of every tile: clamped for the address, masked for the value
    unsigned a_off[W1_CO_T], a_keep[W1_CO_T], b_off[W1_CI_T], b_keep[W1_CI_T];
    float b_sc[W1_CI_T], b_sh[W1_CI_T];
#pragma unroll
    for (int t = 0; t < W1_CO_T; ++t) {
        const int co = cog * WGRAD1X1_COB + t * 32 + c;
        a_keep[t] = co < Cout ? 0xffffffffu : 0u;
        a_off[t] = (unsigned)((size_t)(dy_coff + (co < Cout ? co : Cout - 1)) * HW) + 8u * g;
    }
#pragma unroll
    for (int t = 0; t < W1_CI_T; ++t) {
        const int ci = cig * WGRAD1X1_CIB + t * 32 + c;
        const int cc = ci < Cin ? ci : Cin - 1;
        b_keep[t] = ci < Cin ? 0xffffffffu : 0u;
        b_off[t] = (unsigned)((size_t)(x_coff + cc) * HW) + 8u * g;
        b_sc[t] = in_scale ? in_scale[cc] : 1.f;
        b_sh[t] = in_scale ? in_shift[cc] : 0.f;
    }

    f32x16 acc[W1_CO_T][W1_CI_T];
#pragma unroll
    for (int a = 0; a < W1_CO_T; ++a)
#pragma unroll
        for (int b = 0; b < W1_CI_T; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    // raw operands of one step: [tile][8 pixels]
    float ra[2][W1_CO_T][8], rb[2][W1_CI_T][8];
    auto load_step = [&](int buf, long long s) {
        const long long sc = s < s1 ? s : s1 - 1;   // (past the end: a harmless repeat, never used)
        const int n = (int)(sc / HW16);
        const unsigned p0 = (unsigned)(sc - (long long)n * HW16) * 16u;
        const float* dyn = dy + (size_t)n * dy_ctot * HW + p0;
        const float* xn = x + (size_t)n * x_ctot * HW + p0;
#pragma unroll
        for (int t = 0; t < W1_CO_T; ++t) {
            const float4 u = *reinterpret_cast<const float4*>(dyn + a_off[t]), w = *reinterpret_cast<const float4*>(dyn + a_off[t] + 4);
            ra[buf][t][0] = u.x; ra[buf][t][1] = u.y; ra[buf][t][2] = u.z; ra[buf][t][3] = u.w;
            ra[buf][t][4] = w.x; ra[buf][t][5] = w.y; ra[buf][t][6] = w.z; ra[buf][t][7] = w.w;
        }
#pragma unroll
        for (int t = 0; t < W1_CI_T; ++t) {
            const float4 u = *reinterpret_cast<const float4*>(xn + b_off[t]), w = *reinterpret_cast<const float4*>(xn + b_off[t] + 4);
            rb[buf][t][0] = u.x; rb[buf][t][1] = u.y; rb[buf][t][2] = u.z; rb[buf][t][3] = u.w;
            rb[buf][t][4] = w.x; rb[buf][t][5] = w.y; rb[buf][t][6] = w.z; rb[buf][t][7] = w.w;
        }
    };
    auto compute = [&](int buf) {
        bf16x8 fa[W1_CO_T][3];
#pragma unroll
        for (int t = 0; t < W1_CO_T; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(__float_as_uint(ra[buf][t][e]) & a_keep[t]);
            w1_split8(v, fa[t]);
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, smallest first
#pragma unroll
        for (int t = 0; t < W1_CI_T; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float u = rb[buf][t][e];
                if (in_scale) u = __fmaf_rn(u, b_sc[t], b_sh[t]);   // same fma as the BN backward's mask
                if (in_relu) u = fmaxf(u, 0.f);
                v[e] = __uint_as_float(__float_as_uint(u) & b_keep[t]);
            }
            bf16x8 fb[3];
            w1_split8(v, fb);
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int a = 0; a < W1_CO_T; ++a)
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[p]], fb[PB[p]], acc[a][t], 0, 0, 0);
        }
    };
    if (s0 < s1) {
        load_step(0, s0);
#pragma unroll 1
        for (long long s = s0; s < s1; s += 2) {   // two steps per trip: the buffers alternate without a register copy
            load_step(1, s + 1);
            compute(0);
            if (s + 1 < s1) {
                load_step(0, s + 2);
                compute(1);
            }
        }
    }

    // ---- flush this wave's slice: packed [split][cog][cig][64 co][128 ci].  D: column = lane&31 (ci), rows 8q + 4g + j (co)
    const size_t slice = (size_t)patches * WGRAD1X1_COB * WGRAD1X1_CIB;
    float* dst0 = dw_packed + (size_t)split * slice + (size_t)patch * WGRAD1X1_COB * WGRAD1X1_CIB;
#pragma unroll
    for (int a = 0; a < W1_CO_T; ++a)
#pragma unroll
        for (int b = 0; b < W1_CI_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co_l = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                dst0[(size_t)co_l * WGRAD1X1_CIB + b * 32 + c] = acc[a][b][r];
            }
}

void wgrad1x1_split_shape(int Cout, int Cin, int* cogs, int* cigs, int* pg, int* sub, int* groups) {
    *cogs = (Cout + WGRAD1X1_COB - 1) / WGRAD1X1_COB;
    *cigs = (Cin + WGRAD1X1_CIB - 1) / WGRAD1X1_CIB;
    const int patches = *cogs * *cigs;
    *pg = patches < 4 ? patches : 4;
    *sub = 4 / *pg;
    *groups = (patches + *pg - 1) / *pg;
}

int wgrad1x1_split_blocks(int Cout, int Cin, long long steps) {
    int cogs, cigs, pg, sub, groups;
    wgrad1x1_split_shape(Cout, Cin, &cogs, &cigs, &pg, &sub, &groups);
    long long s = (512 + groups - 1) / groups;        // ~2 blocks per CU over all patch groups
    if (s * sub > steps / 4) s = steps / 4 / sub;     // at least 4 steps per wave
    return s < 1 ? 1 : (int)s;
}

int launch_wgrad1x1_split(const float* x, int x_ctot, int x_coff, int Cin, const float* in_scale, const float* in_shift, int in_relu,
                          const float* dy, int dy_ctot, int dy_coff, int Cout, float* packed, int N, int H, int W, int blocks_x,
                          hipStream_t s) {
    int cogs, cigs, pg, sub, groups;
    wgrad1x1_split_shape(Cout, Cin, &cogs, &cigs, &pg, &sub, &groups);
    hipLaunchKernelGGL(wgrad1x1_split_kernel, dim3(blocks_x, groups), dim3(kBlock), 0, s, x, x_ctot, x_coff, Cin, in_scale, in_shift, in_relu,
                       dy, dy_ctot, dy_coff, Cout, packed, N, H * W / 16, cigs, cogs * cigs, pg, sub);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // namespace cd
