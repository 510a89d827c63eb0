// Experiment: fp32 GEMM accuracy through bf16 matrix instructions with 3-way operand splits (x = hi + mid + lo, each a
// bf16; 9 / 6 / 3 cross products accumulated in fp32 by v_mfma_f32_16x16x32_bf16) against the native fp32 instruction
// (v_mfma_f32_16x16x4_f32) and an fp64 host reference; plus the issue rate of both instructions.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_split_exp tools/exp/mfma_split_exp.hip && tools/exp/mfma_split_exp
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_rne(x);
    const float r1 = x - bf16_f32(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_f32(m);
    l = bf16_rne(r2);
}

// A[16][K] row-major, B[K][16] row-major; out[variant][16][16]; one wave.
// variants: 0 native fp32, 1 x9, 2 x6, 3 x3 (hi*hi, hi*mid, mid*hi), 4 x1 (plain bf16)
__global__ void numerics_kernel(const float* A, const float* B, int K, float* out) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    f32x4 acc0 = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + g], B[(k + g) * 16 + i], acc0, 0, 0, 0);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int k = 0; k < K; k += 32) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            unsigned short h, m, l;
            split3(A[i * K + k + g * 8 + e], h, m, l);
            a[0][e] = (short)h; a[1][e] = (short)m; a[2][e] = (short)l;
            split3(B[(k + g * 8 + e) * 16 + i], h, m, l);
            b[0][e] = (short)h; b[1][e] = (short)m; b[2][e] = (short)l;
        }
        // smallest terms first
        const int order9[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
        for (int t = 0; t < 9; ++t) {
            const int p = order9[t][0], q = order9[t][1];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], b[q], acc[0], 0, 0, 0);
            if (p + q <= 2) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], b[q], acc[1], 0, 0, 0);
            if (p + q <= 1) acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], b[q], acc[2], 0, 0, 0);
            if (p + q == 0) acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], b[q], acc[3], 0, 0, 0);
        }
    }
    // D: col = lane & 15, row = g * 4 + r
    for (int r = 0; r < 4; ++r) {
        out[0 * 256 + (g * 4 + r) * 16 + i] = acc0[r];
        for (int v = 0; v < 4; ++v) out[(v + 1) * 256 + (g * 4 + r) * 16 + i] = acc[v][r];
    }
}

template <int KIND> __global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(threadIdx.x * 3 + e); }
    const float fa = threadIdx.x * 0.5f, fb = threadIdx.x * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    const char* names[5] = {"native f32 mfma", "bf16 x9", "bf16 x6", "bf16 x3", "bf16 x1"};
    for (int dist = 0; dist < 3; ++dist)
        for (int K : {32, 256, 2048, 7744}) {
            std::vector<float> A(16 * K), B(K * 16);
            for (auto& v : A) v = (float)(dist == 0 ? nrand() : dist == 1 ? fabs(nrand()) : nrand() * exp(4 * nrand()));
            for (auto& v : B) v = (float)(dist == 0 ? nrand() : dist == 1 ? fabs(nrand()) : nrand() * exp(4 * nrand()));
            float *dA, *dB, *dO;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 5 * 256 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            numerics_kernel<<<1, 64>>>(dA, dB, K, dO);
            std::vector<float> O(5 * 256);
            hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
            double worst[5] = {0, 0, 0, 0, 0}, rms[5] = {0, 0, 0, 0, 0};
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[r * K + k] * B[k * 16 + c]; ref += p; mag += fabs(p); }
                    for (int v = 0; v < 5; ++v) {
                        const double e = fabs(O[v * 256 + r * 16 + c] - ref) / mag;
                        if (e > worst[v]) worst[v] = e;
                        rms[v] += e * e / 256;
                    }
                }
            printf("dist %d K %5d  |err|/sum|ab|  ", dist, K);
            for (int v = 0; v < 5; ++v) printf(" %s max %.2e rms %.2e |", names[v], worst[v], sqrt(rms[v]));
            printf("\n");
            hipFree(dA); hipFree(dB); hipFree(dO);
        }
    float* dO;
    hipMalloc(&dO, 4096 * 256 * 4);
    for (int kind = 0; kind < 2; ++kind) {
        const int iters = kind ? 20000 : 4000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) rate_kernel<0><<<2048, 256>>>(dO, iters); else rate_kernel<1><<<2048, 256>>>(dO, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2048.0 * 4 * iters * 8 * (kind ? 16 * 16 * 32 * 2 : 16 * 16 * 4 * 2);
        printf("rate %s: %.3f ms  %.1f TFLOP/s\n", kind ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32", ms, flops / ms * 1e-9);
    }
    return 0;
}
