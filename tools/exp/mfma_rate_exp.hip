// Issue rate of the bf16 matrix instructions under different occupancies / accumulator counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC> __global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned seed) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + ((threadIdx.x * 7 + e * 13 + seed) & 0x7f)); b[e] = (short)(0x3f00 + ((threadIdx.x * 5 + e * 3 + seed) & 0xff)); }
    float s = 0;
    if (KIND == 0) {
        f32x4 acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    } else {
        f32x16 acc[NACC];
        for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NACC> void run(float* dO, int blocks, const char* name) {
    const int iters = 20000 / NACC * (KIND ? 1 : 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        rate_kernel<KIND, NACC><<<blocks, 256>>>(dO, iters, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * iters * NACC * (KIND ? 32 * 32 * 16 * 2.0 : 16 * 16 * 32 * 2.0);
    const double per_simd = (double)blocks * 4 * iters * NACC / 1024.0;   // MFMAs per SIMD (256 CUs x 4)
    printf("%s acc=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.0f TFLOP/s  %.1f ns/MFMA/SIMD\n", name, NACC, blocks, blocks / 256.0, ms,
           flops / ms * 1e-9, ms * 1e6 / per_simd);
}

int main() {
    float* dO;
    hipMalloc(&dO, 8192 * 256 * 4);
    for (int blocks : {256, 512, 1024}) {
        run<0, 4>(dO, blocks, "16x16x32"); run<0, 8>(dO, blocks, "16x16x32"); run<0, 16>(dO, blocks, "16x16x32");
        run<1, 2>(dO, blocks, "32x32x16"); run<1, 4>(dO, blocks, "32x32x16"); run<1, 8>(dO, blocks, "32x32x16");
    }
    return 0;
}
