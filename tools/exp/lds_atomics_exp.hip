// Experiment: LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs plain ds_write_b32 / ds_read_b32,
// conflict-free consecutive addresses, 256-thread blocks, 4 blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    __syncthreads();
    float acc = 0.f;
    const int t = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int a = (t + it * 64) & 4095;        // consecutive addresses within a wave
        if (MODE == 0) atomicAdd(&s[a], 1.0f);                                  // ds_add_f32
        if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&s[a]), 1u);       // ds_add_u32
        if (MODE == 2) s[a] = (float)it;                                        // ds_write_b32
        if (MODE == 3) acc += s[a];                                             // ds_read_b32
        if (MODE == 4) { unsigned r = atomicAdd(reinterpret_cast<unsigned*>(&s[a]), 1u); acc += (float)r; }  // ds_add_rtn_u32
        if (MODE == 5) atomicAdd(&s[(t * 17 + it * 64) & 4095], 1.0f);          // f32, strided (17): still conflict-free banks
    }
    __syncthreads();
    out[blockIdx.x * 256 + t] = s[t] + acc;
}

template <int MODE> int run(const char* name, float* out) {
    const int iters = 4096, blocks = 1024;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(out, 16);
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * iters;
    // 256 CUs; cycles per wave-instruction per CU at ~2.3 GHz
    printf("%-22s %.3f ms  %.1f G lane-ops/s  ~%.1f CU-cycles per wave-instruction\n", name, ms, wave_instr * 64 / ms / 1e6,
           ms * 1e-3 * 2.3e9 * 256 / wave_instr);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    run<2>("ds_write_b32", out); run<3>("ds_read_b32", out); run<1>("ds_add_u32", out); run<4>("ds_add_rtn_u32", out);
    run<0>("ds_add_f32", out); run<5>("ds_add_f32 strided", out);
    return 0;
}
