// BatchNorm2d (+ residual) (+ ReLU) as ONE block, max-pool 3x3 / 2 and the plain element-wise pieces, for networks whose backward pass is
// driven by autograd: the MiDaS v2 backbone of BASELINE configs[4] (reference: monodepth/midas_v2_model.py:58-67 -> the un-vendored
// MidasNet; torchvision's ResNeXt Bottleneck: conv-bn-relu, conv-bn-relu, conv-bn, + identity, relu).  The hourglass engine does not use
// these: its BatchNorm statistics come out of the convolution epilogues and are applied on load (layers.hip).
//
// All HBM-bound, one pass each:
//   forward   stats   : per-channel (sum, sum of squares)  -- fp32 per block, fp64 across blocks (CD_BN_STAT_SLOTS partial copies)
//             finalize: cd_bn_finalize (layers.hip): mean, invstd, scale = gamma invstd, shift = beta - gamma mean invstd, running statistics
//             apply   : y = act(fma(x, scale, shift) [+ res])
//   backward  reduce  : dv = relu ? (y > 0 ? dy : 0) : dy;  T1 = sum dv, T2 = sum dv * xhat;  [dres = dv]
//             apply   : dx = gamma invstd (dv - T1 / cnt - xhat T2 / cnt);  dgamma = T2, dbeta = T1
// (ATen / MIOpen: batch_norm, add, relu, threshold_backward, batch_norm_backward = 5 kernels and 13 plane passes for a bottleneck exit;
// here 4 kernels, 9 plane passes.)
#include "cd_common.h"

namespace cd {

__host__ inline dim3 bnb_grid(int HW, int C, int N, int per_thread) {
    int bx = (HW + kBlock * per_thread - 1) / (kBlock * per_thread);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    return dim3(bx, C, N);
}

__global__ __launch_bounds__(kBlock) void bnb_stats_kernel(const float* __restrict__ x, int C, int HW, double* __restrict__ stats) {
    __shared__ float lds[kBlock / kWave];
    const int c = blockIdx.y, n = blockIdx.z;
    const float* p = x + ((size_t)n * C + c) * HW;
    float s = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            const float4 v = p4[i];
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) { const float v = p[i]; s += v; q += v * v; }
    }
    s = block_sum(s, lds);
    q = block_sum(q, lds);
    if (threadIdx.x == 0) {
        const int slot = (blockIdx.x + n) % CD_BN_STAT_SLOTS;
        atomicAdd(&stats[((size_t)slot * C + c) * 2], (double)s);
        atomicAdd(&stats[((size_t)slot * C + c) * 2 + 1], (double)q);
    }
}

__global__ __launch_bounds__(kBlock) void bnb_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ res, int relu,
                                                           float* __restrict__ y, int C, int HW) {
    const int c = blockIdx.y, n = blockIdx.z;
    const size_t base = ((size_t)n * C + c) * HW;
    const float sc = scale[c], sh = shift[c];
    auto f = [&](float v, float r) -> float {
        const float o = __fmaf_rn(v, sc, sh) + r;
        return relu ? fmaxf(o, 0.f) : o;
    };
    if ((HW & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
        float4* y4 = reinterpret_cast<float4*>(y + base);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            const float4 v = x4[i];
            const float4 r = r4 ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            y4[i] = make_float4(f(v.x, r.x), f(v.y, r.y), f(v.z, r.z), f(v.w, r.w));
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) y[base + i] = f(x[base + i], res ? res[base + i] : 0.f);
    }
}

__global__ __launch_bounds__(kBlock) void bnb_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ y, const float* __restrict__ mean_invstd,
                                                                int relu, float* __restrict__ dres, double* __restrict__ sums, int C, int HW) {
    __shared__ float lds[kBlock / kWave];
    const int c = blockIdx.y, n = blockIdx.z;
    const size_t base = ((size_t)n * C + c) * HW;
    const float mean = mean_invstd[2 * c], invstd = mean_invstd[2 * c + 1];
    float t1 = 0.f, t2 = 0.f;
    auto term = [&](float d, float xv, float yv) -> float {
        const float dv = (!relu || yv > 0.f) ? d : 0.f;
        t1 += dv;
        t2 += dv * ((xv - mean) * invstd);
        return dv;
    };
    if ((HW & 3) == 0) {
        const float4* d4 = reinterpret_cast<const float4*>(dy + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* y4 = relu ? reinterpret_cast<const float4*>(y + base) : nullptr;
        float4* r4 = dres ? reinterpret_cast<float4*>(dres + base) : nullptr;
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            const float4 d = d4[i], xv = x4[i];
            const float4 yv = y4 ? y4[i] : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 o = make_float4(term(d.x, xv.x, yv.x), term(d.y, xv.y, yv.y), term(d.z, xv.z, yv.z), term(d.w, xv.w, yv.w));
            if (r4) r4[i] = o;
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) {
            const float o = term(dy[base + i], x[base + i], relu ? y[base + i] : 1.f);
            if (dres) dres[base + i] = o;
        }
    }
    t1 = block_sum(t1, lds);
    t2 = block_sum(t2, lds);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], (double)t1);
        atomicAdd(&sums[2 * c + 1], (double)t2);
    }
}

__global__ __launch_bounds__(kBlock) void bnb_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ y, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean_invstd, int relu,
                                                               const double* __restrict__ sums, double count, float* __restrict__ dx,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int HW) {
    const int c = blockIdx.y, n = blockIdx.z;
    const size_t base = ((size_t)n * C + c) * HW;
    const float mean = mean_invstd[2 * c], invstd = mean_invstd[2 * c + 1];
    const float m1 = (float)(sums[2 * c] / count), m2 = (float)(sums[2 * c + 1] / count);
    const float k = (gamma ? gamma[c] : 1.f) * invstd;
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = (float)sums[2 * c + 1];
        if (dbeta) dbeta[c] = (float)sums[2 * c];
    }
    auto f = [&](float d, float xv, float yv) -> float {
        const float dv = (!relu || yv > 0.f) ? d : 0.f;
        return k * (dv - m1 - ((xv - mean) * invstd) * m2);
    };
    if ((HW & 3) == 0) {
        const float4* d4 = reinterpret_cast<const float4*>(dy + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* y4 = relu ? reinterpret_cast<const float4*>(y + base) : nullptr;
        float4* o4 = reinterpret_cast<float4*>(dx + base);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            const float4 d = d4[i], xv = x4[i];
            const float4 yv = y4 ? y4[i] : make_float4(1.f, 1.f, 1.f, 1.f);
            o4[i] = make_float4(f(d.x, xv.x, yv.x), f(d.y, xv.y, yv.y), f(d.z, xv.z, yv.z), f(d.w, xv.w, yv.w));
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock)
            dx[base + i] = f(dy[base + i], x[base + i], relu ? y[base + i] : 1.f);
    }
}

// ---------------------------------------------------------------- element-wise pieces of the decoder (ResidualConvUnit: conv(relu(x)), out + x)
// op 0: y = max(a, 0)      op 1: y = a + b      op 2: y = b > 0 ? a : 0   (ReLU backward: a = dy, b = the ReLU's input or output)
__global__ __launch_bounds__(kBlock) void eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                         size_t n, int op) {
    if (op == 3) {      // y = a * b[0]: one device scalar for the whole array (block-uniform branch)
        const float sc = b[0];
        const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
            const float4 av = a4[i];
            y4[i] = make_float4(av.x * sc, av.y * sc, av.z * sc, av.w * sc);
        }
        for (size_t i = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = a[i] * sc;
        return;
    }
    auto f = [&](float av, float bv) -> float { return op == 0 ? fmaxf(av, 0.f) : (op == 1 ? av + bv : (bv > 0.f ? av : 0.f)); };
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const float4 av = a4[i], bv = b ? b4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        y4[i] = make_float4(f(av.x, bv.x), f(av.y, bv.y), f(av.z, bv.z), f(av.w, bv.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = f(a[i], b ? b[i] : 0.f);
}

// ---------------------------------------------------------------- MaxPool2d(3, stride 2, padding 1): the stem of the ResNeXt encoder
// forward: the first maximum of the window in row-major order (ATen's rule: `val > max || isnan(val)`), its position inside the window (0..8) kept for
// the backward; backward: a gather over the <= 4 outputs whose window contains the input pixel (no atomics).
__global__ __launch_bounds__(kBlock) void maxpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                unsigned char* __restrict__ arg, int H, int W, int Ho, int Wo) {
    const size_t plane = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
    const float* p = x + plane * H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < Ho * Wo; i += gridDim.x * kBlock) {
        const int oy = i / Wo, ox = i - oy * Wo;
        float best = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = 2 * oy - 1 + dy, xx = 2 * ox - 1 + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float v = p[yy * W + xx];
                if (bi < 0) bi = dy * 3 + dx;                     // (ATen: the index starts at the window's first pixel)
                if (v > best || v != v) { best = v; bi = dy * 3 + dx; }
            }
        y[plane * Ho * Wo + i] = best;
        arg[plane * Ho * Wo + i] = (unsigned char)bi;
    }
}

__global__ __launch_bounds__(kBlock) void maxpool3s2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                float* __restrict__ dx, int H, int W, int Ho, int Wo) {
    const size_t plane = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
    const float* d = dy + plane * Ho * Wo;
    const unsigned char* a = arg + plane * Ho * Wo;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < H * W; i += gridDim.x * kBlock) {
        const int yy = i / W, xx = i - yy * W;
        float acc = 0.f;
        // outputs oy with 2 oy - 1 <= yy <= 2 oy + 1:  oy in {(yy) / 2, (yy + 1) / 2}
#pragma unroll
        for (int ky = 0; ky < 2; ++ky)
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const int oy = (yy + ky) / 2, ox = (xx + kx) / 2;
                if ((ky == 1 && (yy & 1) == 0) || (kx == 1 && (xx & 1) == 0)) continue;      // (even coordinates belong to one window per axis)
                if (oy >= Ho || ox >= Wo) continue;
                const int dyw = yy - (2 * oy - 1), dxw = xx - (2 * ox - 1);
                if ((int)a[oy * Wo + ox] == dyw * 3 + dxw) acc += d[oy * Wo + ox];
            }
        dx[plane * H * W + i] = acc;
    }
}

}  // namespace cd

#define CD_ARGCHK(cond) do { if (!(cond)) return CD_ERR_INVALID_ARG; } while (0)

extern "C" int cd_bn_block_fwd(const float* x, const float* gamma, const float* beta, const float* res, int relu, float* running_mean,
                               float* running_var, float momentum, float eps, float* y, float* mean_invstd, float* scale, float* shift,
                               double* stats, int C, int N, int H, int W, void* stream) {
    CD_ARGCHK(x && y && mean_invstd && scale && shift && stats && C > 0 && N > 0 && H > 0 && W > 0 && C <= 65535 && N <= 65535);
    CD_ARGCHK((gamma == nullptr) == (beta == nullptr) && (running_mean == nullptr) == (running_var == nullptr));
    // the kernels move 16 bytes per lane when H*W % 4 == 0: planes then start on 16-byte boundaries only if the tensors do (a contiguous
    // VIEW with an odd storage offset does not; the Python face re-homes such tensors, a C caller gets an error instead of a fault)
    CD_ARGCHK(((H * W) & 3) != 0 || ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0));
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * (size_t)C * CD_BN_STAT_SLOTS, s) != hipSuccess) return CD_ERR_LAUNCH;
    hipLaunchKernelGGL(cd::bnb_stats_kernel, cd::bnb_grid(HW, C, N, 16), dim3(cd::kBlock), 0, s, x, C, HW, stats);
    CD_CHECK_LAUNCH();
    const int rc = cd_bn_finalize(stats, C, 0, C, (double)N * HW, eps, gamma, beta, running_mean, running_var, momentum, mean_invstd, scale, shift, stream);
    if (rc != CD_OK) return rc;
    hipLaunchKernelGGL(cd::bnb_apply_kernel, cd::bnb_grid(HW, C, N, 8), dim3(cd::kBlock), 0, s, x, scale, shift, res, relu, y, C, HW);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

extern "C" int cd_bn_block_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* mean_invstd, int relu,
                               float* dx, float* dres, float* dgamma, float* dbeta, double* sums, int C, int N, int H, int W, void* stream) {
    CD_ARGCHK(dy && x && mean_invstd && dx && sums && (y || !relu) && C > 0 && N > 0 && H > 0 && W > 0 && C <= 65535 && N <= 65535);
    CD_ARGCHK((dgamma == nullptr) == (dbeta == nullptr));
    CD_ARGCHK(((H * W) & 3) != 0 || ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0));     // (as cd_bn_block_fwd)
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)C, s) != hipSuccess) return CD_ERR_LAUNCH;
    hipLaunchKernelGGL(cd::bnb_bwd_reduce_kernel, cd::bnb_grid(HW, C, N, 16), dim3(cd::kBlock), 0, s, dy, x, y, mean_invstd, relu, dres, sums, C, HW);
    CD_CHECK_LAUNCH();
    hipLaunchKernelGGL(cd::bnb_bwd_apply_kernel, cd::bnb_grid(HW, C, N, 8), dim3(cd::kBlock), 0, s, dy, x, y, gamma, mean_invstd, relu, sums,
                       (double)N * HW, dx, dgamma, dbeta, C, HW);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

extern "C" int cd_eltwise(const float* a, const float* b, float* y, size_t n, int op, void* stream) {
    CD_ARGCHK(a && y && n > 0 && op >= 0 && op <= 3 && (b || op == 0));
    CD_ARGCHK(((uintptr_t)a & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!b || op == 3 || ((uintptr_t)b & 15) == 0));
    size_t blocks = (n / 4 + cd::kBlock * 4 - 1) / (cd::kBlock * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cd::eltwise_kernel, dim3((unsigned)blocks), dim3(cd::kBlock), 0, (hipStream_t)stream, a, b, y, n, op);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

extern "C" int cd_maxpool3s2_fwd(const float* x, float* y, unsigned char* argmax, int C, int N, int H, int W, void* stream) {
    CD_ARGCHK(x && y && argmax && C > 0 && N > 0 && H > 0 && W > 0 && C <= 65535 && N <= 65535);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(cd::maxpool3s2_fwd_kernel, cd::bnb_grid(Ho * Wo, C, N, 2), dim3(cd::kBlock), 0, (hipStream_t)stream, x, y, argmax, H, W, Ho, Wo);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

extern "C" int cd_maxpool3s2_bwd(const float* dy, const unsigned char* argmax, float* dx, int C, int N, int H, int W, void* stream) {
    CD_ARGCHK(dy && dx && argmax && C > 0 && N > 0 && H > 0 && W > 0 && C <= 65535 && N <= 65535);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(cd::maxpool3s2_bwd_kernel, cd::bnb_grid(H * W, C, N, 2), dim3(cd::kBlock), 0, (hipStream_t)stream, dy, argmax, dx, H, W, Ho, Wo);
    CD_CHECK_LAUNCH();
    return CD_OK;
}
