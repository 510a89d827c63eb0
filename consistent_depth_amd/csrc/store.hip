// Device-resident frame-pair store: ONE launch gathers a mini-batch of pairs into the step's input buffers.
//
// Replaces (reference, /root/reference): loaders/video_dataset.py:131-207 (__getitem__: 2 colour .raw, 2 flow .raw, 2 mask
// .png per pair, intrinsics / extrinsics rows) + default_collate + utils/torch_helpers.py:10-23 (to_device).  The whole dataset
// lives in HBM (loaders/pair_store.py); a batch is a copy of 12 H W floats per pair (images 2 x 3HW, flows 2 x 2HW, masks
// 2 x HW) plus a few hundred bytes of cameras and dataset constants.  Masks are kept the way the reference's PNGs hold them,
// one byte per pixel (video_dataset.py:71-77: `> 0` -> float), and are widened to the loss kernels' fp32 {0,1} on the way.
#include <string.h>

#include "cd_common.h"

namespace cd {

struct StoreDesc {          // = cd_pair_store of include/consistent_depth_amd.h
    const float* color;         // [F][3][H][W]
    const float* flows;         // [P][2][2][H][W]   (pair, direction, (dx, dy))
    const void* masks;          // [P][2][H][W]      uint8 (mask_u8 = 1) or fp32
    const float* intrinsics;    // [F][4]
    const float* extrinsics;    // [F][3][4]
    const int64_t* pair_frames; // [P][2] row indices into the frame arrays
    const int64_t* frame_ids;   // [F] original frame numbers (metadata "indices"), may be null
    const float* mask_sums;     // [P][2]            dataset constant (may be null)
    const uint8_t* plans;       // [P][plan_bytes]   dataset constant: tile windows + sweep plan (may be null)
    int64_t plan_bytes;
    int32_t F, P, H, W, mask_u8, reserved;
};
struct BatchDesc {          // = cd_pair_batch
    float* images;      // [B][2][3][H][W]
    float* flow_fwd;    // [B][2][H][W]
    float* flow_bwd;    // [B][2][H][W]
    float* mask_fwd;    // [B][1][H][W]
    float* mask_bwd;    // [B][1][H][W]
    float* intrinsics;  // [B][2][4]
    float* extrinsics;  // [B][2][3][4]
    int64_t* indices;   // [B][2]            (may be null)
    float* mask_sums;   // [B][2]            (may be null)
    uint8_t* plans;     // [B][plan_bytes]   (may be null)
};

// grid (chunks, 8 segments, B): segment 0,1 = images of frame 0,1; 2,3 = flows fwd,bwd; 4,5 = masks fwd,bwd; 6 = small stuff; 7 = plan
__global__ __launch_bounds__(kBlock) void gather_pairs_kernel(const StoreDesc s, const int64_t* __restrict__ ids, const BatchDesc o) {
    const int b = blockIdx.z, seg = blockIdx.y;
    const int64_t p = ids[b];
    const size_t HW = (size_t)s.H * s.W;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, nth = (size_t)gridDim.x * kBlock;
    if (seg < 4) {
        const float* src;
        float* dst;
        size_t n;
        if (seg < 2) {
            const int64_t fr = s.pair_frames[p * 2 + seg];
            src = s.color + (size_t)fr * 3 * HW; dst = o.images + ((size_t)b * 2 + seg) * 3 * HW; n = 3 * HW;
        } else {
            src = s.flows + ((size_t)p * 2 + (seg - 2)) * 2 * HW; dst = (seg == 2 ? o.flow_fwd : o.flow_bwd) + (size_t)b * 2 * HW; n = 2 * HW;
        }
        if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (size_t i = tid; i < n / 4; i += nth) d4[i] = s4[i];
        } else {
            for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
        }
    } else if (seg < 6) {
        float* dst = (seg == 4 ? o.mask_fwd : o.mask_bwd) + (size_t)b * HW;
        if (s.mask_u8) {
            const uint8_t* src = static_cast<const uint8_t*>(s.masks) + ((size_t)p * 2 + (seg - 4)) * HW;
            if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
                float4* d4 = reinterpret_cast<float4*>(dst);
                for (size_t i = tid; i < HW / 4; i += nth) {
                    const uint32_t v = s4[i];
                    d4[i] = make_float4((v & 0xffu) ? 1.f : 0.f, (v & 0xff00u) ? 1.f : 0.f, (v & 0xff0000u) ? 1.f : 0.f, (v & 0xff000000u) ? 1.f : 0.f);
                }
            } else {
                for (size_t i = tid; i < HW; i += nth) dst[i] = src[i] ? 1.f : 0.f;
            }
        } else {
            const float* src = static_cast<const float*>(s.masks) + ((size_t)p * 2 + (seg - 4)) * HW;
            for (size_t i = tid; i < HW; i += nth) dst[i] = src[i];
        }
    } else if (seg == 6) {
        if (blockIdx.x != 0) return;
        for (int i = threadIdx.x; i < 2 * 4; i += kBlock) o.intrinsics[b * 8 + i] = s.intrinsics[s.pair_frames[p * 2 + i / 4] * 4 + i % 4];
        for (int i = threadIdx.x; i < 2 * 12; i += kBlock) o.extrinsics[b * 24 + i] = s.extrinsics[s.pair_frames[p * 2 + i / 12] * 12 + i % 12];
        if (threadIdx.x < 2) {
            const int64_t fr = s.pair_frames[p * 2 + threadIdx.x];
            if (o.indices) o.indices[b * 2 + threadIdx.x] = s.frame_ids ? s.frame_ids[fr] : fr;
            if (o.mask_sums && s.mask_sums) o.mask_sums[b * 2 + threadIdx.x] = s.mask_sums[p * 2 + threadIdx.x];
        }
    } else {
        if (!o.plans || !s.plans) return;
        const uint8_t* src = s.plans + (size_t)p * s.plan_bytes;
        uint8_t* dst = o.plans + (size_t)b * s.plan_bytes;
        if ((s.plan_bytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
            const uint4* s4 = reinterpret_cast<const uint4*>(src);
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            for (size_t i = tid; i < (size_t)s.plan_bytes / 16; i += nth) d4[i] = s4[i];
        } else {
            for (size_t i = tid; i < (size_t)s.plan_bytes; i += nth) dst[i] = src[i];
        }
    }
}

}  // namespace cd

extern "C" {

static_assert(sizeof(cd::StoreDesc) == sizeof(cd_pair_store) && sizeof(cd::BatchDesc) == sizeof(cd_pair_batch), "descriptor layout");

int cd_gather_pairs(const cd_pair_store* store, const int64_t* pair_ids, int B, const cd_pair_batch* batch, void* stream) {
    if (!store || !pair_ids || !batch || B <= 0) return CD_ERR_INVALID_ARG;
    if (!store->color || !store->flows || !store->masks || !store->intrinsics || !store->extrinsics || !store->pair_frames ||
        store->H <= 0 || store->W <= 0)
        return CD_ERR_INVALID_ARG;
    if (!batch->images || !batch->flow_fwd || !batch->flow_bwd || !batch->mask_fwd || !batch->mask_bwd || !batch->intrinsics ||
        !batch->extrinsics)
        return CD_ERR_INVALID_ARG;
    const cd::StoreDesc s = *reinterpret_cast<const cd::StoreDesc*>(store);   // same layout (static_assert above)
    const cd::BatchDesc o = *reinterpret_cast<const cd::BatchDesc*>(batch);
    const size_t HW = (size_t)s.H * s.W;
    unsigned chunks = (unsigned)((3 * HW / 4 + cd::kBlock * 4 - 1) / (cd::kBlock * 4));
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(cd::gather_pairs_kernel, dim3(chunks, 8, B), dim3(cd::kBlock), 0, (hipStream_t)stream, s, pair_ids, o);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // extern "C"
