// Experiment: where do fp32 atomics execute on MI355X and how fast are they?
//   A: agent-scope atomic add (memory-side on a multi-XCD part), coalesced 4-tap pattern
//   W: workgroup-scope atomic add (should execute in the XCD's L2), XCD-affine work assignment
//      via HW_REG_XCC_ID + per-XCD work counters; result checked for lost updates
//   S: plain streaming store baseline
// build: hipcc --offload-arch=gfx950 -O3 -o atomics_exp tools/exp/atomics_exp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int SCOPE>
__device__ __forceinline__ void add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SCOPE);
}

// every element i of a tile receives 4 adds (taps i, i+1, i+W, i+W+1 from 4 different sources)
template <int SCOPE>
__global__ __launch_bounds__(256) void tap_kernel(float* buf, int n, int W) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n - W - 1; i += gridDim.x * 256) {
        add<SCOPE>(buf + i, 1.f);
        add<SCOPE>(buf + i + 1, 1.f);
        add<SCOPE>(buf + i + W, 1.f);
        add<SCOPE>(buf + i + W + 1, 1.f);
    }
}

// XCD-affine: the buffer is cut into `nchunk` chunks; chunk c may only be touched from XCD (c % 8).
// Persistent blocks pull chunk indices from the counter of their own XCD.
template <int SCOPE>
__global__ __launch_bounds__(256) void tap_xcd_kernel(float* buf, int chunk_elems, int nchunk, int W, int* counters,
                                                      int* xcd_hist) {
    __shared__ int s_c;
    const unsigned xcd = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcd_hist[xcd], 1);
    const int per_xcd = nchunk / 8;
    const int tiles_per_chunk = chunk_elems / 1024;  // 1024 elems per tile (256 thr x 4)
    for (;;) {
        if (threadIdx.x == 0) s_c = atomicAdd(&counters[xcd], 1);
        __syncthreads();
        const int t = s_c;
        __syncthreads();
        if (t >= per_xcd * tiles_per_chunk) break;
        const int c = (t / tiles_per_chunk) * 8 + xcd;  // chunk owned by this XCD
        float* base = buf + (size_t)c * chunk_elems;
        const int i0 = (t % tiles_per_chunk) * 1024;
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * 256 + threadIdx.x;
            if (i < chunk_elems - W - 1) {
                add<SCOPE>(base + i, 1.f);
                add<SCOPE>(base + i + 1, 1.f);
                add<SCOPE>(base + i + W, 1.f);
                add<SCOPE>(base + i + W + 1, 1.f);
            }
        }
    }
}

__global__ __launch_bounds__(256) void store_kernel(float4* buf, size_t n4) {
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) buf[i] = make_float4(1, 2, 3, 4);
}

static double check(const std::vector<float>& h, int n, int W) {
    // interior elements must be exactly 4
    size_t bad = 0;
    for (int i = W + 1; i < n - W - 1; ++i) if (h[i] != 4.f) ++bad;
    return (double)bad;
}

int main() {
    const int W = 224, HW = 384 * 224;
    const int planes = 512;                     // 512 planes x 344 KB = 176 MB
    const int n = planes * HW;
    float* buf; CK(hipMalloc(&buf, sizeof(float) * (size_t)n));
    int* counters; CK(hipMalloc(&counters, 64 * sizeof(int)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    std::vector<float> h(HW);

    // S: store baseline
    CK(hipMemset(buf, 0, sizeof(float) * (size_t)n));
    for (int r = 0; r < 2; ++r) {
        CK(hipEventRecord(e0));
        store_kernel<<<2048, 256>>>((float4*)buf, (size_t)n / 4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("S  plain store        : %.3f ms  %.1f GB/s\n", ms, 4.0 * n / ms / 1e6);

    // A: agent scope
    CK(hipMemset(buf, 0, sizeof(float) * (size_t)n));
    CK(hipEventRecord(e0));
    tap_kernel<__HIP_MEMORY_SCOPE_AGENT><<<2048, 256>>>(buf, n, W);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), buf + 5 * HW, sizeof(float) * HW, hipMemcpyDeviceToHost));
    printf("A  agent-scope atomics: %.3f ms  %.2f G atomics/s  bad=%g\n", ms, 4.0 * n / ms / 1e6, check(h, HW, W));

    // W (non-affine): workgroup scope from arbitrary XCDs -> expect lost updates if L2-local
    CK(hipMemset(buf, 0, sizeof(float) * (size_t)n));
    CK(hipEventRecord(e0));
    tap_kernel<__HIP_MEMORY_SCOPE_WORKGROUP><<<2048, 256>>>(buf, n, W);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), buf + 5 * HW, sizeof(float) * HW, hipMemcpyDeviceToHost));
    printf("W0 wg-scope, any XCD  : %.3f ms  %.2f G atomics/s  bad=%g (lost updates expected if L2-local)\n", ms,
           4.0 * n / ms / 1e6, check(h, HW, W));

    // W (XCD-affine): chunk = one plane
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(buf, 0, sizeof(float) * (size_t)n));
        CK(hipMemset(counters, 0, 64 * sizeof(int)));
        CK(hipEventRecord(e0));
        tap_xcd_kernel<__HIP_MEMORY_SCOPE_WORKGROUP><<<2048, 256>>>(buf, HW, planes, W, counters, counters + 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        double bad = 0;
        for (int pl : {0, 1, 7, 8, 100, 511}) {
            CK(hipMemcpy(h.data(), buf + (size_t)pl * HW, sizeof(float) * HW, hipMemcpyDeviceToHost));
            size_t b = 0; for (int i = W + 1; i < HW - W - 1; ++i) if (h[i] != 4.f) ++b; bad += b;
        }
        int hist[8]; CK(hipMemcpy(hist, counters + 16, sizeof(hist), hipMemcpyDeviceToHost));
        printf("W1 wg-scope, XCD-affine: %.3f ms  %.2f G atomics/s  bad=%g  blocks/xcd=%d %d %d %d %d %d %d %d\n", ms,
               4.0 * n / ms / 1e6, bad, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    }
    // A (XCD-affine) for comparison
    CK(hipMemset(buf, 0, sizeof(float) * (size_t)n));
    CK(hipMemset(counters, 0, 64 * sizeof(int)));
    CK(hipEventRecord(e0));
    tap_xcd_kernel<__HIP_MEMORY_SCOPE_AGENT><<<2048, 256>>>(buf, HW, planes, W, counters, counters + 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("A1 agent-scope, XCD-affine: %.3f ms  %.2f G atomics/s\n", ms, 4.0 * n / ms / 1e6);
    return 0;
}
