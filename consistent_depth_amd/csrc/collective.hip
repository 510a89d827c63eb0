// Gradient all-reduce of the data-parallel step on RCCL, straight from the C ABI (no torch.distributed in between):
//     buf <- (sum over ranks of buf) / world        one in-place ncclAllReduce(sum, fp32) over xGMI + one scale launch.
//
// Replaces (reference, /root/reference): nn.DataParallel's replicate / scatter / gather / reduce_add of every step
// (monodepth/midas_v2_model.py:41-43, depth_fine_tuning.py:155-159).  The buffer is the flat gradient of the engine
// (cd_hourglass_grads, or FlatAdam's flat_grad [+ loss slot]): 21.4 MB for the hourglass, ONE collective per step.
// librccl is resolved at run time (dlopen): inside a PyTorch-ROCm process the already loaded librccl is used, so the
// communicator may come from anywhere in the process; a host without RCCL can still load this library.
#include <dlfcn.h>

#include "cd_common.h"

namespace cd {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // rccl.h: ncclFloat32, ncclSum

static nccl_allreduce_fn resolve_allreduce() {
    static nccl_allreduce_fn fn = nullptr;
    static bool tried = false;
    if (tried) return fn;
    tried = true;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");        // librccl already in the process (torch loads its own)
    if (!sym) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h && (sym = dlsym(h, "ncclAllReduce"))) break;
        }
    }
    fn = reinterpret_cast<nccl_allreduce_fn>(sym);
    return fn;
}

__global__ __launch_bounds__(kBlock) void scale_kernel(float* __restrict__ buf, size_t n, float f) {
    const size_t n4 = n / 4;
    float4* b4 = reinterpret_cast<float4*>(buf);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        float4 v = b4[i];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        b4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) buf[n4 * 4 + threadIdx.x] *= f;
}

}  // namespace cd

extern "C" {

int cd_rccl_available(void) { return cd::resolve_allreduce() != nullptr; }

int cd_allreduce_mean_f32(float* buf, size_t n, void* nccl_comm, int world, void* stream) {
    if (!buf || n == 0 || world < 1 || (world > 1 && !nccl_comm) || reinterpret_cast<uintptr_t>(buf) % 16 != 0) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (world > 1 || nccl_comm) {
        cd::nccl_allreduce_fn ar = cd::resolve_allreduce();
        if (!ar) return CD_ERR_UNSUPPORTED;
        if (ar(buf, buf, n, cd::kNcclFloat32, cd::kNcclSum, nccl_comm, s) != 0) return CD_ERR_LAUNCH;
    }
    if (world > 1) {
        unsigned blocks = (unsigned)((n / 4 + cd::kBlock - 1) / cd::kBlock);
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(cd::scale_kernel, dim3(blocks), dim3(cd::kBlock), 0, s, buf, n, 1.f / (float)world);
        CD_CHECK_LAUNCH();
    }
    return CD_OK;
}

}  // extern "C"
