// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "consistent_depth_amd.h"

namespace cd {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

#define CD_CHECK_LAUNCH()                                   \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return CD_ERR_LAUNCH; \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}

// Sum over a 256-thread block; result valid in thread 0.  `lds` needs kBlock/kWave floats.
__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kBlock / kWave; ++i) r += lds[i];
    }
    __syncthreads();
    return r;
}

// Hardware fp32 atomic add (global_atomic_add_f32, no return, no CAS loop).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace cd
