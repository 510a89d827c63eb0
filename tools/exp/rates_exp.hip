// Experiment: issue rates on gfx950 that size the instruction budget of the fused loss kernel:
//   plain / packed fp32 VALU, transcendentals, the f32->f64 fixed-point conversion, LDS integer atomics (32 / 64 bit),
//   random-ish 4-tap LDS gathers.  256-thread blocks, 8 blocks per CU (full occupancy), results per CU-cycle at the
//   measured wall time (clock assumed 2.4 GHz; compare rows with each other, not with the absolute).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/rates_exp tools/exp/rates_exp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NACC = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ unsigned long long s64[2048];
    __shared__ unsigned s32[4096];
    for (int i = threadIdx.x; i < 2048; i += 256) { s64[i] = 0ull; s32[i] = 0u; s32[i + 2048] = 0u; }
    __syncthreads();
    const int t = threadIdx.x;
    float a[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) a[i] = seed + (float)(t + i) * 1e-3f;
    double dacc = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
            if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
            if (MODE == 6) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 7) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        }
        if (MODE == 8) {   // packed fp32: 4 v_pk_fma_f32 on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < NACC; i += 2) {
                f2 v = {a[i], a[i + 1]};
                f2 sd = {seed, seed};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(sd));
                a[i] = v.x; a[i + 1] = v.y;
            }
        }
        if (MODE == 9) {   // the 2^-40 fixed-point conversion of round 1: cvt_f64_f32 + fma_f64 + 64-bit subtract
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const double d = __fma_rn((double)a[i], 1099511627776.0, 6755399441055744.0);
                dacc += __longlong_as_double(__double_as_longlong(d) - __double_as_longlong(6755399441055744.0));
                a[i] += 1e-3f;
            }
        }
        if (MODE == 10) {  // fp32-only split of c*2^40 into (hi, lo) 32-bit integers
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const float hi = floorf(a[i] * 256.f);             // c * 2^8 -> integer part (value * 2^32 units)
                const float lo = (a[i] * 256.f - hi) * 4294967296.f;
                const long long v = ((long long)(int)hi << 32) + (long long)(unsigned)lo;
                dacc += __longlong_as_double(v);
                a[i] += 1e-3f;
            }
        }
        if (MODE == 11) {  // ds_add_u64, 8 per iteration, conflict-light addresses
#pragma unroll
            for (int i = 0; i < NACC; ++i) atomicAdd(&s64[(t * 1 + it * 64 + i * 257) & 2047], 1ull);
        }
        if (MODE == 12) {  // ds_add_u32
#pragma unroll
            for (int i = 0; i < NACC; ++i) atomicAdd(&s32[(t * 1 + it * 64 + i * 257) & 4095], 1u);
        }
        if (MODE == 13) {  // ds_read_b32 gathers (the depth taps)
#pragma unroll
            for (int i = 0; i < NACC; ++i) a[i] += __uint_as_float(s32[(t + it * 64 + i * 49) & 4095]);
        }
        if (MODE == 14) {  // ds_add_u64 where neighbouring lanes hit neighbouring addresses AND the 4 taps overlap (scatter-like)
            const int base = (t + it * 64) & 1023;
            atomicAdd(&s64[base], 1ull); atomicAdd(&s64[base + 1], 1ull);
            atomicAdd(&s64[base + 48], 1ull); atomicAdd(&s64[base + 49], 1ull);
            atomicAdd(&s64[base + 512], 1ull); atomicAdd(&s64[base + 513], 1ull);
            atomicAdd(&s64[base + 560], 1ull); atomicAdd(&s64[base + 561], 1ull);
        }
        if (MODE == 16) {  // ds_add_f64 (LDS double atomic add)
#pragma unroll
            for (int i = 0; i < NACC; ++i) atomicAdd(reinterpret_cast<double*>(&s64[(t + it * 64 + i * 257) & 2047]), 1.0);
        }
        if (MODE == 17) {  // f32 -> f64 convert + nothing else
#pragma unroll
            for (int i = 0; i < NACC; ++i) { dacc += (double)a[i]; a[i] += 1e-3f; }
        }
        if (MODE == 15) {  // ds_add_f32 for reference
#pragma unroll
            for (int i = 0; i < NACC; ++i) atomicAdd(reinterpret_cast<float*>(&s32[(t + it * 64 + i * 257) & 4095]), 1.f);
        }
    }
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += a[i];
    out[blockIdx.x * 256 + t] = r + (float)dacc + (float)s64[t] + (float)s32[t];
}

template <int MODE> int run(const char* name, float* out, int per_iter) {
    const int iters = 2048, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(out, 16, 1.0001f);
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_ops = (double)blocks * 4 * iters * per_iter;
    printf("%-34s %.3f ms  %8.1f G lane-ops/s  %.2f CU-cycles per wave-op (2.4 GHz)\n", name, ms, wave_ops * 64 / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / wave_ops);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_fma_f32", out, NACC); run<4>("v_mul_f32", out, NACC); run<8>("v_pk_fma_f32 (per pk instr)", out, NACC / 2);
    run<5>("v_cndmask_b32", out, NACC); run<6>("v_floor_f32", out, NACC); run<7>("v_cvt_i32_f32", out, NACC);
    run<1>("v_exp_f32", out, NACC); run<2>("v_rcp_f32", out, NACC); run<3>("v_rsq_f32", out, NACC);
    run<9>("to_fixed f64 path (per value)", out, NACC); run<10>("to_fixed f32 split (per value)", out, NACC);
    run<11>("ds_add_u64", out, NACC); run<12>("ds_add_u32", out, NACC); run<13>("ds_read_b32 gather", out, NACC);
    run<14>("ds_add_u64 tap pattern", out, 8); run<15>("ds_add_f32", out, NACC); run<16>("ds_add_f64", out, NACC);
    run<17>("cvt_f64_f32 + add_f64 + add_f32", out, NACC);
    return 0;
}
