// Flat-buffer optimiser kernels for gfx950 (HBM-bound, 16 B/lane streams).
//
// cd_adam_step_flat replaces torch.optim.Adam.step as the reference drives it
// (/root/reference/optimizer/__init__.py:16-17, depth_fine_tuning.py:231-236,283):
// ~316 per-tensor launches there, ONE launch over the flat parameter buffer here.
// cd_l1_distance replaces loss/parameter_loss.py:14-18 (sum |p - p0|).
#include <math.h>

#include "cd_common.h"

namespace cd {

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float gs, float b1, float b2,
                                         float omb1, float omb2, float step_size, float inv_bc2s, float eps) {
    g *= gs;
    m = m + omb1 * (g - m);           // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + omb2 * g * g;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = __builtin_amdgcn_sqrtf(v) * inv_bc2s + eps;
    p = p - step_size * (m / denom);  // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(kBlock) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, size_t n,
                                                           float gs, float b1, float b2, float step_size,
                                                           float inv_bc2s, float eps) {
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        adam_one(pp.y, gg.y, mm.y, vv.y, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        adam_one(pp.z, gg.z, mm.z, vv.z, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        adam_one(pp.w, gg.w, mm.w, vv.w, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n % 4) by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
    }
}

// Same step, but the "skip this step if the loss is NaN" guard of depth_fine_tuning.py:278-280
// and the optimiser step counter live on the device, so the training loop never has to
// synchronise with the host: if *loss is NaN nothing is written and *step_counter is not
// advanced (the reference `continue`s before backward()/step()).
__global__ __launch_bounds__(kBlock) void adam_flat_guarded_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                   float* __restrict__ m, float* __restrict__ v,
                                                                   size_t n, float gs, float lr, float b1, float b2,
                                                                   float eps, const int* __restrict__ step_counter,
                                                                   const float* __restrict__ loss) {
    if (loss != nullptr && isnan(loss[0])) return;  // block-uniform
    const int step = step_counter[0] + 1;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2s = (float)(1.0 / sqrt(bc2));
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        adam_one(pp.y, gg.y, mm.y, vv.y, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        adam_one(pp.z, gg.z, mm.z, vv.z, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        adam_one(pp.w, gg.w, mm.w, vv.w, gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], gs, b1, b2, omb1, omb2, step_size, inv_bc2s, eps);
    }
}

// runs after the guarded step on the same stream: advance the counter unless the step was skipped
__global__ void adam_advance_kernel(int* step_counter, const float* __restrict__ loss) {
    if (loss == nullptr || !isnan(loss[0])) step_counter[0] += 1;
}

__global__ __launch_bounds__(kBlock) void l1_partial_kernel(const float* __restrict__ p, const float* __restrict__ p0,
                                                            size_t n, float* __restrict__ partial) {
    __shared__ float lds[kBlock / kWave];
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc += fabsf(p[i] - p0[i]);
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kBlock) void l1_final_kernel(const float* __restrict__ partial, int nblk,
                                                          float* __restrict__ out) {
    __shared__ double lds[kBlock];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) acc += (double)partial[i];
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)lds[0];
}

static inline int l1_blocks(size_t n) {
    size_t b = (n + kBlock * 8 - 1) / (kBlock * 8);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace cd

extern "C" {

int cd_adam_step_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                      float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return CD_ERR_INVALID_ARG;
    if (n == 0) return CD_OK;
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
         reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 != 0)
        return CD_ERR_INVALID_ARG;
    // bias corrections in double on the host, like torch's python-scalar path
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2s = (float)(1.0 / sqrt(bc2));
    size_t blocks = (n / 4 + cd::kBlock - 1) / cd::kBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride the rest
    hipLaunchKernelGGL(cd::adam_flat_kernel, dim3((unsigned)blocks), dim3(cd::kBlock), 0, (hipStream_t)stream, params,
                       grads, exp_avg, exp_avg_sq, n, grad_scale, beta1, beta2, step_size, inv_bc2s, eps);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_adam_step_flat_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                              float beta1, float beta2, float eps, int* step_counter, const float* loss,
                              float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step_counter) return CD_ERR_INVALID_ARG;
    if (n == 0) return CD_OK;
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
         reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 != 0)
        return CD_ERR_INVALID_ARG;
    size_t blocks = (n / 4 + cd::kBlock - 1) / cd::kBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(cd::adam_flat_guarded_kernel, dim3((unsigned)blocks), dim3(cd::kBlock), 0, (hipStream_t)stream,
                       params, grads, exp_avg, exp_avg_sq, n, grad_scale, lr, beta1, beta2, eps, step_counter, loss);
    CD_CHECK_LAUNCH();
    hipLaunchKernelGGL(cd::adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter, loss);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

size_t cd_l1_distance_workspace_bytes(size_t n) { return sizeof(float) * (size_t)cd::l1_blocks(n); }

int cd_l1_distance(const float* p, const float* p0, size_t n, float* out, void* workspace, size_t workspace_bytes,
                   void* stream) {
    if (!p || !p0 || !out || !workspace) return CD_ERR_INVALID_ARG;
    if (workspace_bytes < cd_l1_distance_workspace_bytes(n)) return CD_ERR_WORKSPACE;
    const int nb = cd::l1_blocks(n);
    hipLaunchKernelGGL(cd::l1_partial_kernel, dim3(nb), dim3(cd::kBlock), 0, (hipStream_t)stream, p, p0, n, (float*)workspace);
    CD_CHECK_LAUNCH();
    hipLaunchKernelGGL(cd::l1_final_kernel, dim3(1), dim3(cd::kBlock), 0, (hipStream_t)stream, (const float*)workspace, nb, out);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

}  // extern "C"
