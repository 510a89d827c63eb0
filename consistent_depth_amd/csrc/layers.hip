// Memory-bound layers of the hourglass around the MFMA convolutions (gfx950, NCHW fp32, 16 B per
// lane where the row length allows).  They replace the BatchNorm2d(train) / ReLU / AvgPool2d(2) /
// UpsamplingBilinear2d(2) / add launches (forward and autograd backward) of the network the
// reference fine-tunes.
//
// Activation convention of the engine: a conv+BN+ReLU unit keeps x_hat = (y - mean) / sqrt(var+eps)
// (normalised, PRE-ReLU) in memory; every consumer applies act(v) = relu(v*scale[c] + shift[c]) while
// loading (scale/shift only for the affine stem BN).  x_hat is exactly what the BN/ReLU backward
// needs, so no separate post-ReLU tensor or mask is ever stored.
#include "cd_common.h"

namespace cd {

__device__ __forceinline__ float act(float v, const float* sc, const float* sh, int c, int relu) {
    if (sc) v = __fmaf_rn(v, sc[c], sh[c]);  // explicit fma: the BN backward re-evaluates exactly this expression
    return relu ? fmaxf(v, 0.f) : v;
}

// ---------------------------------------------------------------- BatchNorm (train) forward
// stats[slot][c] = partial (sum, sumsq) of the raw conv output over N*H*W (accumulated by the conv epilogue into
// CD_BN_STAT_SLOTS copies to keep same-address atomics apart; summed here in slot order).
// In place: x <- (x - mean) * rsqrt(var + eps); saves (mean, invstd); updates the running stats like
// nn.BatchNorm2d(momentum): running_var uses the unbiased variance.
__global__ __launch_bounds__(kBlock) void bn_normalize_kernel(float* __restrict__ x, int ctot, int coff,
                                                              const double* __restrict__ stats, double count, float eps,
                                                              float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float momentum,
                                                              float* __restrict__ mean_invstd, int HW) {
    const int c = blockIdx.y, n = blockIdx.z;
    double sum = 0.0, sumsq = 0.0;
    for (int s = 0; s < CD_BN_STAT_SLOTS; ++s) {   // fixed order
        sum += stats[((size_t)s * ctot + coff + c) * 2];
        sumsq += stats[((size_t)s * ctot + coff + c) * 2 + 1];
    }
    const double mean_d = sum / count;
    double var_d = sumsq / count - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
        mean_invstd[2 * (coff + c)] = mean;
        mean_invstd[2 * (coff + c) + 1] = invstd;
        if (running_mean) {
            const double unb = count > 1.0 ? var_d * count / (count - 1.0) : var_d;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
    float* p = x + ((size_t)n * ctot + coff + c) * HW;
    if ((HW & 3) == 0) {
        float4* p4 = reinterpret_cast<float4*>(p);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            float4 v = p4[i];
            v.x = (v.x - mean) * invstd; v.y = (v.y - mean) * invstd; v.z = (v.z - mean) * invstd; v.w = (v.w - mean) * invstd;
            p4[i] = v;
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) p[i] = (p[i] - mean) * invstd;
    }
}

// BatchNorm (train) WITHOUT a pass over the activation: from the conv epilogue's (sum, sumsq) compute per channel
//   scale = gamma * invstd,  shift = beta - gamma * mean * invstd     (gamma = 1, beta = 0 when not affine)
// so that consumers evaluate relu(raw * scale + shift) while loading the RAW conv output; also saves
// (mean, invstd) for the backward and updates the running statistics.  One thread per channel.
__global__ void bn_finalize_kernel(const double* __restrict__ stats, int ctot, int coff, int C, double count, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                   float* __restrict__ mean_invstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sum = 0.0, sumsq = 0.0;
    for (int s = 0; s < CD_BN_STAT_SLOTS; ++s) {
        sum += stats[((size_t)s * ctot + coff + c) * 2];
        sumsq += stats[((size_t)s * ctot + coff + c) * 2 + 1];
    }
    const double mean_d = sum / count;
    double var_d = sumsq / count - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    mean_invstd[2 * (coff + c)] = mean;
    mean_invstd[2 * (coff + c) + 1] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[coff + c] = g * invstd;
    shift[coff + c] = b - g * mean * invstd;
    if (running_mean) {
        const double unb = count > 1.0 ? var_d * count / (count - 1.0) : var_d;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// ---------------------------------------------------------------- BN(train)+ReLU backward
// Given dA = d loss / d a with a = relu(gamma*x_hat + beta):
//   pass 1: T1[c] = sum dA*mask, T2[c] = sum dA*mask*x_hat          (mask = gamma*x_hat + beta > 0)
//   pass 2: dY = gamma*invstd * (dA*mask - T1/cnt - x_hat*T2/cnt)   in place over dA   (d loss / d raw conv output)
//           and  d gamma = T2, d beta = T1 for the affine stem BN.
__global__ __launch_bounds__(kBlock) void bn_relu_bwd_reduce_kernel(const float* __restrict__ dA, int d_ctot, int d_coff,
                                                                    const float* __restrict__ xhat, int x_ctot,
                                                                    int x_coff, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    const float* __restrict__ mean_invstd,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    double* __restrict__ sums, int HW) {
    __shared__ float lds[kBlock / kWave];
    const int c = blockIdx.y, n = blockIdx.z;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    // raw mode (scale != NULL): `xhat` holds the RAW conv output.  The ReLU mask is evaluated on EXACTLY the
    // expression the consumers applied on load, fma(raw, scale, shift) = gamma*x_hat + beta.
    const bool raw = scale != nullptr;
    const float sc = raw ? scale[x_coff + c] : 0.f, sh = raw ? shift[x_coff + c] : 0.f;
    const float xm = raw ? mean_invstd[2 * (x_coff + c)] : 0.f, xs = raw ? mean_invstd[2 * (x_coff + c) + 1] : 1.f;
    const float* d = dA + ((size_t)n * d_ctot + d_coff + c) * HW;
    const float* xh = xhat + ((size_t)n * x_ctot + x_coff + c) * HW;
    float t1 = 0.f, t2 = 0.f;
    auto term = [&](float rv, float dval) {
        const float pre = raw ? __fmaf_rn(rv, sc, sh) : g * rv + b;
        const float xv = raw ? (gamma ? (rv - xm) * xs : pre) : rv;
        const float dv = pre > 0.f ? dval : 0.f;
        t1 += dv;
        t2 += dv * xv;
    };
    if ((HW & 3) == 0) {   // 16-byte loads (plane starts are 16-byte aligned when HW % 4 == 0)
        const float4* d4 = reinterpret_cast<const float4*>(d);
        const float4* x4 = reinterpret_cast<const float4*>(xh);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            const float4 rv = x4[i], dv = d4[i];
            term(rv.x, dv.x); term(rv.y, dv.y); term(rv.z, dv.z); term(rv.w, dv.w);
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) term(xh[i], d[i]);
    }
    t1 = block_sum(t1, lds);
    t2 = block_sum(t2, lds);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], (double)t1);
        atomicAdd(&sums[2 * c + 1], (double)t2);
    }
}

__global__ __launch_bounds__(kBlock) void bn_relu_bwd_apply_kernel(float* __restrict__ dA, int d_ctot, int d_coff,
                                                                   const float* __restrict__ xhat, int x_ctot, int x_coff,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const double* __restrict__ sums, double count,
                                                                   const float* __restrict__ mean_invstd,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   int overwrite_affine, int HW) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const bool raw = scale != nullptr;
    const float sc = raw ? scale[x_coff + c] : 0.f, sh = raw ? shift[x_coff + c] : 0.f;
    const float xm = raw ? mean_invstd[2 * (x_coff + c)] : 0.f, xs = raw ? mean_invstd[2 * (x_coff + c) + 1] : 1.f;
    const float m1 = (float)(sums[2 * c] / count), m2 = (float)(sums[2 * c + 1] / count);
    const float k = g * mean_invstd[2 * (x_coff + c) + 1];
    if (dgamma && blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
        // accumulated like every parameter gradient of the engine, unless the caller asked for plain assignment
        dgamma[c] = (overwrite_affine ? 0.f : dgamma[c]) + (float)sums[2 * c + 1];
        dbeta[c] = (overwrite_affine ? 0.f : dbeta[c]) + (float)sums[2 * c];
    }
    float* d = dA + ((size_t)n * d_ctot + d_coff + c) * HW;
    const float* xh = xhat + ((size_t)n * x_ctot + x_coff + c) * HW;
    auto apply = [&](float rv, float dval) -> float {
        const float pre = raw ? __fmaf_rn(rv, sc, sh) : g * rv + b;
        const float xv = raw ? (gamma ? (rv - xm) * xs : pre) : rv;
        const float dv = pre > 0.f ? dval : 0.f;
        return k * (dv - m1 - xv * m2);
    };
    if ((HW & 3) == 0) {
        float4* d4 = reinterpret_cast<float4*>(d);
        const float4* x4 = reinterpret_cast<const float4*>(xh);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            const float4 rv = x4[i];
            float4 dv = d4[i];
            dv.x = apply(rv.x, dv.x); dv.y = apply(rv.y, dv.y); dv.z = apply(rv.z, dv.z); dv.w = apply(rv.w, dv.w);
            d4[i] = dv;
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) d[i] = apply(xh[i], d[i]);
    }
}

// ---------------------------------------------------------------- AvgPool2d(2)
__global__ __launch_bounds__(kBlock) void avgpool2_fwd_kernel(const float* __restrict__ x, int x_ctot, int x_coff,
                                                              const float* __restrict__ sc, const float* __restrict__ sh,
                                                              int relu, float* __restrict__ y, int y_ctot, int y_coff,
                                                              int H, int W) {
    const int c = blockIdx.y, n = blockIdx.z, Ho = H / 2, Wo = W / 2;
    const float* xi = x + ((size_t)n * x_ctot + x_coff + c) * H * W;
    float* yo = y + ((size_t)n * y_ctot + y_coff + c) * Ho * Wo;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < Ho * Wo; i += gridDim.x * kBlock) {
        const int oy = i / Wo, ox = i - oy * Wo;
        const float2 a = *reinterpret_cast<const float2*>(xi + (size_t)(2 * oy) * W + 2 * ox);
        const float2 b = *reinterpret_cast<const float2*>(xi + (size_t)(2 * oy + 1) * W + 2 * ox);
        yo[i] = 0.25f * (act(a.x, sc, sh, c, relu) + act(a.y, sc, sh, c, relu) + act(b.x, sc, sh, c, relu) + act(b.y, sc, sh, c, relu));
    }
}

// dIn (gradient w.r.t. the ACTIVATED input) (+)= 0.25 * dOut
__global__ __launch_bounds__(kBlock) void avgpool2_bwd_kernel(const float* __restrict__ dy, int dy_ctot, int dy_coff,
                                                              float* __restrict__ dx, int dx_ctot, int dx_coff, int H,
                                                              int W, int accumulate) {
    const int c = blockIdx.y, n = blockIdx.z, Ho = H / 2, Wo = W / 2;
    const float* d = dy + ((size_t)n * dy_ctot + dy_coff + c) * Ho * Wo;
    float* o = dx + ((size_t)n * dx_ctot + dx_coff + c) * H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < H * W; i += gridDim.x * kBlock) {
        const int yy = i / W, xx = i - yy * W;
        const float v = 0.25f * d[(yy >> 1) * Wo + (xx >> 1)];
        o[i] = accumulate ? o[i] + v : v;
    }
}

// 16-byte variants (round 6; W % 4 == 0 and 16-byte aligned bases -- the launchers check): the same per-element expressions as the
// scalar kernels above, so the results are bit for bit theirs (tests/test_layers_gpu.py compares the two through
// cd_debug_set_layers_mode).  Forward: two outputs per lane from two float4 rows; backward: a 2 x 4 block of dx per lane from one
// float2 of dy.  In the step the scalar forms ran at 1.5-3 TB/s (4 bytes per lane and load, grids capped at 64 workgroups per
// plane with 1.3 iterations per thread); these stream at the rate of the BatchNorm passes.
__global__ __launch_bounds__(kBlock) void avgpool2_fwd_vec_kernel(const float* __restrict__ x, int x_ctot, int x_coff,
                                                                  const float* __restrict__ sc, const float* __restrict__ sh,
                                                                  int relu, float* __restrict__ y, int y_ctot, int y_coff,
                                                                  int H, int W) {
    const int c = blockIdx.y, n = blockIdx.z, Ho = H / 2, Wo = W / 2, Wp = Wo / 2;
    const float* xi = x + ((size_t)n * x_ctot + x_coff + c) * H * W;
    float* yo = y + ((size_t)n * y_ctot + y_coff + c) * Ho * Wo;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < Ho * Wp; i += gridDim.x * kBlock) {
        const int oy = i / Wp, xp = i - oy * Wp;
        const float4 a = *reinterpret_cast<const float4*>(xi + (size_t)(2 * oy) * W + 4 * xp);
        const float4 b = *reinterpret_cast<const float4*>(xi + (size_t)(2 * oy + 1) * W + 4 * xp);
        float2 o;
        o.x = 0.25f * (act(a.x, sc, sh, c, relu) + act(a.y, sc, sh, c, relu) + act(b.x, sc, sh, c, relu) + act(b.y, sc, sh, c, relu));
        o.y = 0.25f * (act(a.z, sc, sh, c, relu) + act(a.w, sc, sh, c, relu) + act(b.z, sc, sh, c, relu) + act(b.w, sc, sh, c, relu));
        *reinterpret_cast<float2*>(yo + (size_t)oy * Wo + 2 * xp) = o;
    }
}

__global__ __launch_bounds__(kBlock) void avgpool2_bwd_vec_kernel(const float* __restrict__ dy, int dy_ctot, int dy_coff,
                                                                  float* __restrict__ dx, int dx_ctot, int dx_coff, int H,
                                                                  int W, int accumulate) {
    const int c = blockIdx.y, n = blockIdx.z, Ho = H / 2, Wo = W / 2, Wq = W / 4;
    const float* d = dy + ((size_t)n * dy_ctot + dy_coff + c) * Ho * Wo;
    float* o = dx + ((size_t)n * dx_ctot + dx_coff + c) * H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < Ho * Wq; i += gridDim.x * kBlock) {
        const int oy = i / Wq, xq = i - oy * Wq;
        const float2 g = *reinterpret_cast<const float2*>(d + (size_t)oy * Wo + 2 * xq);
        const float v0 = 0.25f * g.x, v1 = 0.25f * g.y;
        float4* r0 = reinterpret_cast<float4*>(o + (size_t)(2 * oy) * W + 4 * xq);
        float4* r1 = reinterpret_cast<float4*>(o + (size_t)(2 * oy + 1) * W + 4 * xq);
        float4 a = make_float4(v0, v0, v1, v1), b = a;
        if (accumulate) {
            const float4 p = *r0, q = *r1;
            a.x = p.x + v0; a.y = p.y + v0; a.z = p.z + v1; a.w = p.w + v1;
            b.x = q.x + v0; b.y = q.y + v0; b.z = q.z + v1; b.w = q.w + v1;
        }
        *r0 = a;
        *r1 = b;
    }
}

// ---------------------------------------------------------------- bilinear x2 (align_corners=True) + add
// out[y][x] = bilinear(act_a(lo))[y][x] + act_b(hi)[y][x];   lo is (h, w), hi/out are (2h, 2w)
__global__ __launch_bounds__(kBlock) void upsample2x_add_fwd_kernel(
    const float* __restrict__ lo, int lo_ctot, int lo_coff, const float* __restrict__ lsc, const float* __restrict__ lsh,
    int lrelu, const float* __restrict__ hi, int hi_ctot, int hi_coff, const float* __restrict__ hsc,
    const float* __restrict__ hsh, int hrelu, float* __restrict__ out, int o_ctot, int o_coff, int h, int w) {
    const int c = blockIdx.y, n = blockIdx.z, H = 2 * h, W = 2 * w;
    const float ry = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float* l = lo + ((size_t)n * lo_ctot + lo_coff + c) * h * w;
    const float* hh = hi ? hi + ((size_t)n * hi_ctot + hi_coff + c) * H * W : nullptr;
    float* o = out + ((size_t)n * o_ctot + o_coff + c) * H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < H * W; i += gridDim.x * kBlock) {
        const int y = i / W, x = i - y * W;
        const float sy = ry * (float)y, sx = rx * (float)x;
        const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1), y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ty = sy - (float)y0, tx = sx - (float)x0;
        const float v00 = act(l[y0 * w + x0], lsc, lsh, c, lrelu), v01 = act(l[y0 * w + x1], lsc, lsh, c, lrelu);
        const float v10 = act(l[y1 * w + x0], lsc, lsh, c, lrelu), v11 = act(l[y1 * w + x1], lsc, lsh, c, lrelu);
        float v = (1.f - ty) * ((1.f - tx) * v00 + tx * v01) + ty * ((1.f - tx) * v10 + tx * v11);
        if (hh) v += act(hh[i], hsc, hsh, c, hrelu);
        o[i] = v;
    }
}

// The kernel above WITH THE ROUNDINGS hipcc gave it (read off its ISA: the compiler contracts the two inner sums into
// fma((1 - tx), v00, tx * v01) and leaves the outer one as two products and an add -- it pairs them into v_pk_mul_f32; ty and tx are
// plain subtractions), written out with contraction off so that the column kernel below, where the compiler would choose differently,
// gives the scalar kernel's bits.
__device__ __forceinline__ void upsample2x_coord(float r, int i, int n_lo, int& i0, int& i1, float& t) {
#pragma clang fp contract(off)
    const float s = r * (float)i;
    i0 = min((int)s, n_lo - 1);
    i1 = min(i0 + 1, n_lo - 1);
    t = s - (float)i0;
}
__device__ __forceinline__ float upsample2x_blend(float v00, float v01, float v10, float v11, float tx, float ty) {
#pragma clang fp contract(off)
    const float top = __builtin_fmaf(1.f - tx, v00, tx * v01), bot = __builtin_fmaf(1.f - tx, v10, tx * v11);
    const float p = (1.f - ty) * top, q = ty * bot;
    return p + q;
}

// Round 6: a lane OWNS four consecutive output columns and walks down the rows of its band (w even, 16-byte aligned hi / out: the
// launcher checks).  The column taps and weights of its four pixels are computed once, a row costs its (lane-uniform) y0 / y1 / ty,
// 16 loads of the low-res plane at precomputed offsets, one 16-byte load of the skip tensor and one 16-byte store; 256 / (W/4) rows
// are in flight per workgroup.  The scalar kernel spends ~60 VALU instructions and an integer division per output pixel with every tap
// behind its own branch on the affine: 1.8 TB/s in the step where the BatchNorm passes stream at 5.6.
__global__ __launch_bounds__(kBlock) void upsample2x_add_fwd_col_kernel(
    const float* __restrict__ lo, int lo_ctot, int lo_coff, const float* __restrict__ lsc, const float* __restrict__ lsh,
    int lrelu, const float* __restrict__ hi, int hi_ctot, int hi_coff, const float* __restrict__ hsc,
    const float* __restrict__ hsh, int hrelu, float* __restrict__ out, int o_ctot, int o_coff, int h, int w, int band_rows) {
    const int c = blockIdx.y, n = blockIdx.z, H = 2 * h, W = 2 * w, Wq = W / 4;
    const int rpp = kBlock / Wq;                                  // rows per pass of the workgroup (the launcher guarantees Wq <= kBlock)
    const int rsub = threadIdx.x / Wq, xq = threadIdx.x - rsub * Wq;
    if (rsub >= rpp) return;
    const float ry = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float* l = lo + ((size_t)n * lo_ctot + lo_coff + c) * h * w;
    const float* hh = hi ? hi + ((size_t)n * hi_ctot + hi_coff + c) * H * W : nullptr;
    float* o = out + ((size_t)n * o_ctot + o_coff + c) * H * W;
    int x0[4], x1[4];
    float tx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) upsample2x_coord(rx, 4 * xq + k, w, x0[k], x1[k], tx[k]);
    const bool laff = lsc != nullptr, haff = hsc != nullptr;
    const float lscv = laff ? lsc[c] : 0.f, lshv = laff ? lsh[c] : 0.f, hscv = haff ? hsc[c] : 0.f, hshv = haff ? hsh[c] : 0.f;
    const int y_end = min(H, ((int)blockIdx.x + 1) * band_rows);
    for (int y = blockIdx.x * band_rows + rsub; y < y_end; y += rpp) {
        int y0, y1;
        float ty;
        upsample2x_coord(ry, y, h, y0, y1, ty);
        const float* r0 = l + y0 * w;
        const float* r1 = l + y1 * w;
        float t[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) { t[4 * k] = r0[x0[k]]; t[4 * k + 1] = r0[x1[k]]; t[4 * k + 2] = r1[x0[k]]; t[4 * k + 3] = r1[x1[k]]; }
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hh) s4 = *reinterpret_cast<const float4*>(hh + (size_t)y * W + 4 * xq);
        if (laff) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t[k]) : "v"(t[k]), "v"(lscv), "v"(lshv));   // = act()'s fmaf, never v_pk_fma_f32 (wgrad_split.hip:204)
        }
        if (lrelu) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = fmaxf(t[k], 0.f);
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = upsample2x_blend(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3], tx[k], ty);
        if (hh) {
            float sv[4] = {s4.x, s4.y, s4.z, s4.w};
            if (haff) {
#pragma unroll
                for (int k = 0; k < 4; ++k) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(sv[k]) : "v"(sv[k]), "v"(hscv), "v"(hshv));
            }
            if (hrelu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k] = fmaxf(sv[k], 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += sv[k];
        }
        *reinterpret_cast<float4*>(o + (size_t)y * W + 4 * xq) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// adjoint of the bilinear part, as a GATHER over the high-res gradient (no atomics):
// dlo[yy][xx] (+)= sum_{y,x} wy(y,yy) * wx(x,xx) * dout[y][x]
__device__ __forceinline__ float upsample2x_bwd_pixel(const float* __restrict__ d, int h, int w, int H, int W, float ry, float rx,
                                                      int yy, int xx) {
    // high-res rows whose source coordinate lies in (yy-1, yy+1): y in ((yy-1)/ry, (yy+1)/ry)
    // (one extra candidate on each side: the weights below are exact, candidates outside get weight 0)
    const int ya = ry > 0.f ? max(0, (int)floorf((float)(yy - 1) / ry) - 1) : 0;
    const int yb = ry > 0.f ? min(H - 1, (int)ceilf((float)(yy + 1) / ry) + 1) : H - 1;
    const int xa = rx > 0.f ? max(0, (int)floorf((float)(xx - 1) / rx) - 1) : 0;
    const int xb = rx > 0.f ? min(W - 1, (int)ceilf((float)(xx + 1) / rx) + 1) : W - 1;
    // column weights once per output pixel (identical arithmetic to the forward); the candidate window is at most
    // 2/rx + 4 <= kMaxCand wide for every w >= 2 (rx >= 1/3), rows use the same bound
    constexpr int kMaxCand = 12;
    float wxs[kMaxCand];
    const int nx = min(xb - xa + 1, kMaxCand);
#pragma unroll
    for (int q = 0; q < kMaxCand; ++q) {
        const int x = xa + q;
        const float sx = rx * (float)x;
        const int x0 = min((int)sx, w - 1), x1 = min(x0 + 1, w - 1);
        const float tx = sx - (float)x0;
        wxs[q] = q < nx ? (x0 == xx ? 1.f - tx : 0.f) + (x1 == xx ? tx : 0.f) : 0.f;
    }
    float acc = 0.f;
    if (xb - xa + 1 <= kMaxCand) {
        for (int y = ya; y <= yb; ++y) {
            const float sy = ry * (float)y;
            const int y0 = min((int)sy, h - 1), y1 = min(y0 + 1, h - 1);
            const float ty = sy - (float)y0;
            const float wy = (y0 == yy ? 1.f - ty : 0.f) + (y1 == yy ? ty : 0.f);
            if (wy == 0.f) continue;
            const float* row = d + y * W + xa;
#pragma unroll
            for (int q = 0; q < kMaxCand; ++q)
                if (q < nx) acc += wy * wxs[q] * row[q];
        }
    } else {   // degenerate sizes (w == 1: every column is a candidate): the plain double loop
        for (int y = ya; y <= yb; ++y) {
            const float sy = ry * (float)y;
            const int y0 = min((int)sy, h - 1), y1 = min(y0 + 1, h - 1);
            const float ty = sy - (float)y0;
            const float wy = (y0 == yy ? 1.f - ty : 0.f) + (y1 == yy ? ty : 0.f);
            if (wy == 0.f) continue;
            for (int x = xa; x <= xb; ++x) {
                const float sx = rx * (float)x;
                const int x0 = min((int)sx, w - 1), x1 = min(x0 + 1, w - 1);
                const float tx = sx - (float)x0;
                const float wx = (x0 == xx ? 1.f - tx : 0.f) + (x1 == xx ? tx : 0.f);
                acc += wy * wx * d[y * W + x];
            }
        }
    }
    return acc;
}

__global__ __launch_bounds__(kBlock) void upsample2x_bwd_kernel(const float* __restrict__ dout, int d_ctot, int d_coff,
                                                                float* __restrict__ dlo, int l_ctot, int l_coff, int h,
                                                                int w, int accumulate) {
    const int c = blockIdx.y, n = blockIdx.z, H = 2 * h, W = 2 * w;
    const float ry = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float* d = dout + ((size_t)n * d_ctot + d_coff + c) * H * W;
    float* o = dlo + ((size_t)n * l_ctot + l_coff + c) * h * w;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < h * w; i += gridDim.x * kBlock) {
        const int yy = i / w, xx = i - yy * w;
        const float acc = upsample2x_bwd_pixel(d, h, w, H, W, ry, rx, yy, xx);
        o[i] = accumulate ? o[i] + acc : acc;
    }
}

// The same sums from a BAND of low-res rows per workgroup (round 6).  The gather above spends its time on 12 candidate columns x ~5
// rows of scalar global loads per output (of which 4-5 x 4-5 carry weight) and on recomputing the weights per output: 276 us for the
// 384x224 level where the bytes take 45.  Here the weights of a column depend on the column only and those of a row on the row only:
// they are computed once per workgroup (with the expressions of upsample2x_bwd_pixel), compacted to the first non-zero candidate
// onward (<= kUpWin taps for h, w >= 4: the open interval ((t-1)/r, (t+1)/r) is 4 + 2/(n-1) < 5 long) and kept in LDS; the
// high-res rows the band touches are staged into LDS with 16-byte loads (each row fetched once per band, + 3 halo rows); an output
// is then <= 5 x 5 fused multiply-adds out of LDS IN THE ORDER OF THE GATHER (rows ascending, columns ascending; the taps left out
// have weight 0 and fma(0, x, acc) = acc for finite x), so the result is bit for bit the gather's
// (tests/test_layers_gpu.py::test_streaming_layer_kernels_give_the_bits_of_the_scalar_ones).  Anything unusual -- a non-zero weight
// beyond the window, more rows than the LDS holds -- sends the whole workgroup through upsample2x_bwd_pixel.
constexpr int kUpCand = 12, kUpWin = 5;
struct UpTap { int first, count; float w[kUpWin]; int pad; };   // 32 bytes
__device__ __forceinline__ bool upsample2x_taps(int t, int n_lo, int n_hi, float r, UpTap& tp) {
#pragma clang fp contract(off)   // (the gather's roundings, read off its ISA: s = r * x for the index, t = fma(r, x, -i0) for the weight)
    const int a = r > 0.f ? max(0, (int)floorf((float)(t - 1) / r) - 1) : 0;
    const int b = r > 0.f ? min(n_hi - 1, (int)ceilf((float)(t + 1) / r) + 1) : n_hi - 1;
    float ws[kUpCand];
    const int nc = min(b - a + 1, kUpCand);
    int q_first = kUpCand, q_last = -1;
#pragma unroll
    for (int q = 0; q < kUpCand; ++q) {
        const int x = a + q;
        const float sx = r * (float)x;
        const int x0 = min((int)sx, n_lo - 1), x1 = min(x0 + 1, n_lo - 1);
        const float tx = __builtin_fmaf(r, (float)x, -(float)x0);
        ws[q] = q < nc ? (x0 == t ? 1.f - tx : 0.f) + (x1 == t ? tx : 0.f) : 0.f;
        if (ws[q] != 0.f) { q_first = min(q_first, q); q_last = q; }
    }
    tp.first = a + q_first;
    tp.count = q_last - q_first + 1;
    tp.pad = 0;
#pragma unroll
    for (int j = 0; j < kUpWin; ++j) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < kUpCand; ++q)
            if (q - j >= 0 && q - j <= kUpCand - 1 && q_first == q - j) v = ws[q];
        tp.w[j] = v;
    }
    return b - a + 1 <= kUpCand && q_last >= 0 && tp.count <= kUpWin;
}

template <int RB>
__global__ __launch_bounds__(kBlock) void upsample2x_bwd_band_kernel(const float* __restrict__ dout, int d_ctot, int d_coff,
                                                                     float* __restrict__ dlo, int l_ctot, int l_coff, int h,
                                                                     int w, int accumulate, int rows_cap, int vec) {
    extern __shared__ __attribute__((aligned(16))) float up_smem[];
    __shared__ UpTap s_ytab[RB];
    __shared__ int s_bad;
    UpTap* s_xtab = reinterpret_cast<UpTap*>(up_smem);                 // [w]
    float* s_rows = up_smem + (size_t)w * (sizeof(UpTap) / 4);         // [rows_cap][W] + kUpWin + 3 zeros
    const int c = blockIdx.y, n = blockIdx.z, H = 2 * h, W = 2 * w;
    const int r0 = blockIdx.x * RB, nr = min(RB, h - r0);
    const float ry = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float* d = dout + ((size_t)n * d_ctot + d_coff + c) * H * W;
    float* o = dlo + ((size_t)n * l_ctot + l_coff + c) * h * w;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < w + nr; t += kBlock) {
        UpTap tp;
        bool ok;
        if (t < w) { ok = upsample2x_taps(t, w, W, rx, tp); s_xtab[t] = tp; }
        else { ok = upsample2x_taps(r0 + t - w, h, H, ry, tp); s_ytab[t - w] = tp; }
        if (!ok) s_bad = 1;
    }
    __syncthreads();
    int y_lo = H, y_hi = -1;
    for (int r = 0; r < nr; ++r) { y_lo = min(y_lo, s_ytab[r].first); y_hi = max(y_hi, s_ytab[r].first + s_ytab[r].count - 1); }
    const int nrows = y_hi - y_lo + 1;
    if (s_bad == 0 && nrows >= 1 && nrows <= rows_cap && y_lo >= 0 && y_hi < H) {   // workgroup-uniform
        const float* src = d + (size_t)y_lo * W;
        const int total = nrows * W;
        if (vec) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* t4 = reinterpret_cast<float4*>(s_rows);
            for (int i = threadIdx.x; i < total / 4; i += kBlock) t4[i] = s4[i];
        } else {
            for (int i = threadIdx.x; i < total; i += kBlock) s_rows[i] = src[i];
        }
        if (threadIdx.x < kUpWin + 3) s_rows[total + threadIdx.x] = 0.f;
        __syncthreads();
        for (int i = threadIdx.x; i < nr * w; i += kBlock) {
            const int rr = i / w, xx = i - rr * w;
            const UpTap tx = s_xtab[xx];
            const UpTap* ty = &s_ytab[rr];
            const float* rowp = s_rows + (ty->first - y_lo) * W + tx.first;
            const int ny = ty->count;
            float acc = 0.f;
            for (int r = 0; r < ny; ++r, rowp += W) {
                const float wy = ty->w[r];
#pragma unroll
                for (int j = 0; j < kUpWin; ++j) acc = __fmaf_rn(wy * tx.w[j], rowp[j], acc);
            }
            float* op = o + (size_t)(r0 + rr) * w + xx;
            *op = accumulate ? *op + acc : acc;
        }
    } else {
        for (int i = threadIdx.x; i < nr * w; i += kBlock) {
            const int rr = i / w, xx = i - rr * w;
            const float acc = upsample2x_bwd_pixel(d, h, w, H, W, ry, rx, r0 + rr, xx);
            float* op = o + (size_t)(r0 + rr) * w + xx;
            *op = accumulate ? *op + acc : acc;
        }
    }
}

// ---------------------------------------------------------------- bilinear x2, half-pixel centres (align_corners=False)
// F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) -- the last up-sampling of MiDaS' output head
// (midas_net.py::_Interpolate; upstream intel-isl/MiDaS `Interpolate(scale_factor=2, mode="bilinear")`).  Source index of output o:
// max(0.5 * (o + 0.5) - 0.5, 0): even o = 2i reads (i - 1, i) with weights (0.25, 0.75), odd o = 2i + 1 reads (i, i + 1) with
// (0.75, 0.25), the border taps clamped (weight 1 on the edge pixel).  All weights are exact in fp32.
__device__ __forceinline__ void halfpixel_taps(int o, int n_in, int* i0, int* i1, float* t) {
    const float s = fmaxf(0.5f * ((float)o + 0.5f) - 0.5f, 0.f);
    *i0 = min((int)s, n_in - 1);
    *i1 = min(*i0 + 1, n_in - 1);
    *t = s - (float)*i0;
}
__global__ __launch_bounds__(kBlock) void upsample2x_halfpixel_fwd_kernel(const float* __restrict__ lo, int lo_ctot, int lo_coff,
                                                                          float* __restrict__ out, int o_ctot, int o_coff, int h, int w) {
    const int c = blockIdx.y, n = blockIdx.z, H = 2 * h, W = 2 * w;
    const float* l = lo + ((size_t)n * lo_ctot + lo_coff + c) * h * w;
    float* o = out + ((size_t)n * o_ctot + o_coff + c) * H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < H * W; i += gridDim.x * kBlock) {
        const int y = i / W, x = i - y * W;
        int y0, y1, x0, x1;
        float ty, tx;
        halfpixel_taps(y, h, &y0, &y1, &ty);
        halfpixel_taps(x, w, &x0, &x1, &tx);
        o[i] = (1.f - ty) * ((1.f - tx) * l[y0 * w + x0] + tx * l[y0 * w + x1]) + ty * ((1.f - tx) * l[y1 * w + x0] + tx * l[y1 * w + x1]);
    }
}
// the adjoint as a GATHER (no atomics): input pixel i receives from the outputs 2i - 1 .. 2i + 2 of its axis -- weight of output o on
// input i = (i0(o) == i ? 1 - t : 0) + (i1(o) == i ? t : 0), the forward's own arithmetic (borders included)
__global__ __launch_bounds__(kBlock) void upsample2x_halfpixel_bwd_kernel(const float* __restrict__ dout, int d_ctot, int d_coff,
                                                                          float* __restrict__ dlo, int l_ctot, int l_coff, int h, int w,
                                                                          int accumulate) {
    const int c = blockIdx.y, n = blockIdx.z, H = 2 * h, W = 2 * w;
    const float* d = dout + ((size_t)n * d_ctot + d_coff + c) * H * W;
    float* o = dlo + ((size_t)n * l_ctot + l_coff + c) * h * w;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < h * w; i += gridDim.x * kBlock) {
        const int yy = i / w, xx = i - yy * w;
        float wx[4], wy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int a, b;
            float t;
            const int x = 2 * xx - 1 + q, y = 2 * yy - 1 + q;
            wx[q] = 0.f; wy[q] = 0.f;
            if (x >= 0 && x < W) { halfpixel_taps(x, w, &a, &b, &t); wx[q] = (a == xx ? 1.f - t : 0.f) + (b == xx ? t : 0.f); }
            if (y >= 0 && y < H) { halfpixel_taps(y, h, &a, &b, &t); wy[q] = (a == yy ? 1.f - t : 0.f) + (b == yy ? t : 0.f); }
        }
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = min(max(2 * yy - 1 + r, 0), H - 1);      // (clamped rows / columns carry weight 0)
            float racc = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) racc += wx[q] * d[y * W + min(max(2 * xx - 1 + q, 0), W - 1)];
            acc += wy[r] * racc;
        }
        o[i] = accumulate ? o[i] + acc : acc;
    }
}

// dst[:, coff:coff+C] (+)= src[:, scoff:scoff+C]   (gradient fan-in of an activation with several consumers)
__global__ __launch_bounds__(kBlock) void add_slice_kernel(const float* __restrict__ src, int s_ctot, int s_coff,
                                                           float* __restrict__ dst, int d_ctot, int d_coff, int HW,
                                                           int accumulate) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float* s = src + ((size_t)n * s_ctot + s_coff + c) * HW;
    float* d = dst + ((size_t)n * d_ctot + d_coff + c) * HW;
    if ((HW & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(s);
        float4* d4 = reinterpret_cast<float4*>(d);
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW / 4; i += gridDim.x * kBlock) {
            float4 v = s4[i];
            if (accumulate) { const float4 o = d4[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            d4[i] = v;
        }
    } else {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) d[i] = accumulate ? d[i] + s[i] : s[i];
    }
}

// out[c] (+)= sum over n, y, x of src[n][coff+c]   (bias gradient of a conv that is NOT followed by BatchNorm: in the
// hourglass only the 1-channel head).  ONE workgroup per channel, fixed summation order (per-thread strided sums in 4
// independent fp32 chains, then a fp64 tree): bit-reproducible, unlike the fp32-atomic combination of round 1.
constexpr int kSumBlock = 1024;
__global__ __launch_bounds__(kSumBlock) void channel_sum_kernel(const float* __restrict__ src, int ctot, int coff, int N,
                                                                int HW, float* __restrict__ out, int accumulate) {
    __shared__ double lds[kSumBlock];
    constexpr int kBlock = kSumBlock;
    const int c = blockIdx.y;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* pl = src + ((size_t)n * ctot + coff + c) * HW;
        int p = threadIdx.x;
        for (; p + 3 * kBlock < HW; p += 4 * kBlock) { a0 += pl[p]; a1 += pl[p + kBlock]; a2 += pl[p + 2 * kBlock]; a3 += pl[p + 3 * kBlock]; }
        for (; p < HW; p += kBlock) a0 += pl[p];
    }
    lds[threadIdx.x] = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)lds[0] : (float)lds[0];
}

static inline dim3 plane_grid(int HW, int C, int N, int per_thread) {
    int bx = (HW + kBlock * per_thread - 1) / (kBlock * per_thread);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    return dim3(bx, C, N);
}
// For the pure streaming kernels: `items` thread-iterations per plane in workgroups of EQUAL trip count (a cap of 64 on 84 workgroups'
// worth of items leaves a third of the threads a second iteration and the rest waiting: add_slice ran at 2.8 TB/s where the BatchNorm
// passes, whose 42 workgroups do exactly two iterations each, reach 5.6).  Results do not depend on the grid (element-wise kernels only;
// the reductions keep plane_grid, their summation order is the grid's).
static inline dim3 plane_grid_even(long long items, int C, int N) {
    long long bx = (items + kBlock - 1) / kBlock;
    if (bx < 1) bx = 1;
    if (bx > 64) { const long long it = (bx + 63) / 64; bx = (bx + it - 1) / it; }
    return dim3((unsigned)bx, C, N);
}
// cd_debug_set_layers_mode / CD_AMD_LAYERS_MODE at load: bit 0 = the scalar kernels of rounds 1-5 (bit-identity tests, A/B timing)
static int g_layers_mode = [] { const char* e = getenv("CD_AMD_LAYERS_MODE"); return e ? atoi(e) : 0; }();
static inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// ---------------------------------------------------------------- small host-side chores as ONE launch each (round 6: they were ATen
// launches inside the step -- 22 torch.cat of the fused entry convolutions' biases, a foreach-add over the 155 BatchNorm batch counters)
// table[i]: n floats from src to dst (block i)
__global__ __launch_bounds__(kBlock) void copy_segments_kernel(const cd_copy_seg* __restrict__ table) {
    const cd_copy_seg seg = table[blockIdx.x];
    for (long long i = threadIdx.x; i < seg.n; i += kBlock) seg.dst[i] = seg.src[i];
}
// p[0 .. n) = 0 (32-bit words; p 16-byte aligned): 16 bytes per lane, tail by words
__global__ __launch_bounds__(kBlock) void zero_bytes_kernel(unsigned* __restrict__ p, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * kBlock;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = 0u;
}
// *table[i] += delta (nn.BatchNorm2d.num_batches_tracked of every layer, int64 scalars)
__global__ __launch_bounds__(kBlock) void counters_add_kernel(long long* const* __restrict__ table, int n, long long delta) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) *table[i] += delta;
}

}  // namespace cd

extern "C" {

#define CD_ARGCHK(cond) do { if (!(cond)) return CD_ERR_INVALID_ARG; } while (0)

int cd_bn_normalize(float* x, int ctot, int coff, int C, const double* stats, float eps, float* running_mean,
                    float* running_var, float momentum, float* mean_invstd, int N, int H, int W, void* stream) {
    CD_ARGCHK(x && stats && mean_invstd && C > 0 && coff >= 0 && coff + C <= ctot && N > 0 && H > 0 && W > 0);
    CD_ARGCHK((running_mean == nullptr) == (running_var == nullptr));
    hipLaunchKernelGGL(cd::bn_normalize_kernel, cd::plane_grid(H * W, C, N, 8), dim3(cd::kBlock), 0, (hipStream_t)stream, x,
                       ctot, coff, stats, (double)N * H * W, eps, running_mean, running_var, momentum, mean_invstd, H * W);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_bn_finalize(const double* stats, int ctot, int coff, int C, double count, float eps, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float* mean_invstd,
                   float* scale, float* shift, void* stream) {
    CD_ARGCHK(stats && mean_invstd && scale && shift && C > 0 && coff >= 0 && coff + C <= ctot && count > 0);
    CD_ARGCHK((running_mean == nullptr) == (running_var == nullptr) && (gamma == nullptr) == (beta == nullptr));
    hipLaunchKernelGGL(cd::bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, ctot, coff, C, count, eps,
                       gamma, beta, running_mean, running_var, momentum, mean_invstd, scale, shift);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_bn_relu_bwd(float* dA, int d_ctot, int d_coff, const float* xhat, int x_ctot, int x_coff, int C, const float* gamma,
                   const float* beta, const float* mean_invstd, const float* scale, const float* shift, double* sums,
                   int flags, float* dgamma, float* dbeta, int N, int H, int W, void* stream) {
    const int sums_prezeroed = flags & CD_BN_BWD_SUMS_PREZEROED, overwrite_affine = (flags & CD_BN_BWD_OVERWRITE_AFFINE) ? 1 : 0;
    CD_ARGCHK((flags & ~(CD_BN_BWD_SUMS_PREZEROED | CD_BN_BWD_OVERWRITE_AFFINE)) == 0);
    CD_ARGCHK((scale == nullptr) == (shift == nullptr));
    CD_ARGCHK(dA && xhat && mean_invstd && sums && C > 0 && d_coff >= 0 && d_coff + C <= d_ctot && x_coff >= 0 && x_coff + C <= x_ctot);
    CD_ARGCHK((gamma == nullptr) == (beta == nullptr) && (dgamma == nullptr) == (dbeta == nullptr));
    hipStream_t s = (hipStream_t)stream;
    if (!sums_prezeroed && hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s) != hipSuccess) return CD_ERR_LAUNCH;
    const dim3 grid = cd::plane_grid(H * W, C, N, 8);
    hipLaunchKernelGGL(cd::bn_relu_bwd_reduce_kernel, grid, dim3(cd::kBlock), 0, s, dA, d_ctot, d_coff, xhat, x_ctot, x_coff,
                       gamma, beta, mean_invstd, scale, shift, sums, H * W);
    CD_CHECK_LAUNCH();
    hipLaunchKernelGGL(cd::bn_relu_bwd_apply_kernel, grid, dim3(cd::kBlock), 0, s, dA, d_ctot, d_coff, xhat, x_ctot, x_coff,
                       gamma, beta, sums, (double)N * H * W, mean_invstd, scale, shift, dgamma, dbeta, overwrite_affine, H * W);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_avgpool2_fwd(const float* x, int x_ctot, int x_coff, const float* in_scale, const float* in_shift, int in_relu,
                    float* y, int y_ctot, int y_coff, int C, int N, int H, int W, void* stream) {
    CD_ARGCHK(x && y && C > 0 && (H % 2) == 0 && (W % 2) == 0 && x_coff + C <= x_ctot && y_coff + C <= y_ctot);
    CD_ARGCHK((in_scale == nullptr) == (in_shift == nullptr));
    if (!(cd::g_layers_mode & 1) && (W % 4) == 0 && cd::aligned16(x, y))
        hipLaunchKernelGGL(cd::avgpool2_fwd_vec_kernel, cd::plane_grid_even((long long)(H / 2) * (W / 4), C, N), dim3(cd::kBlock), 0,
                           (hipStream_t)stream, x, x_ctot, x_coff, in_scale, in_shift, in_relu, y, y_ctot, y_coff, H, W);
    else
        hipLaunchKernelGGL(cd::avgpool2_fwd_kernel, cd::plane_grid(H * W / 4, C, N, 4), dim3(cd::kBlock), 0, (hipStream_t)stream,
                           x, x_ctot, x_coff, in_scale, in_shift, in_relu, y, y_ctot, y_coff, H, W);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_avgpool2_bwd(const float* dy, int dy_ctot, int dy_coff, float* dx, int dx_ctot, int dx_coff, int C, int N, int H,
                    int W, int accumulate, void* stream) {
    CD_ARGCHK(dy && dx && C > 0 && (H % 2) == 0 && (W % 2) == 0 && dy_coff + C <= dy_ctot && dx_coff + C <= dx_ctot);
    if (!(cd::g_layers_mode & 1) && (W % 4) == 0 && cd::aligned16(dy, dx))
        hipLaunchKernelGGL(cd::avgpool2_bwd_vec_kernel, cd::plane_grid_even((long long)(H / 2) * (W / 4), C, N), dim3(cd::kBlock), 0,
                           (hipStream_t)stream, dy, dy_ctot, dy_coff, dx, dx_ctot, dx_coff, H, W, accumulate);
    else
        hipLaunchKernelGGL(cd::avgpool2_bwd_kernel, cd::plane_grid(H * W, C, N, 4), dim3(cd::kBlock), 0, (hipStream_t)stream, dy,
                           dy_ctot, dy_coff, dx, dx_ctot, dx_coff, H, W, accumulate);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_upsample2x_add_fwd(const float* lo, int lo_ctot, int lo_coff, const float* lo_scale, const float* lo_shift,
                          int lo_relu, const float* hi, int hi_ctot, int hi_coff, const float* hi_scale,
                          const float* hi_shift, int hi_relu, float* out, int o_ctot, int o_coff, int C, int N, int h, int w,
                          void* stream) {
    CD_ARGCHK(lo && out && C > 0 && h > 0 && w > 0 && lo_coff + C <= lo_ctot && o_coff + C <= o_ctot);
    CD_ARGCHK(hi == nullptr || hi_coff + C <= hi_ctot);
    CD_ARGCHK((lo_scale == nullptr) == (lo_shift == nullptr) && (hi_scale == nullptr) == (hi_shift == nullptr));
    if (!(cd::g_layers_mode & 1) && (w % 2) == 0 && w / 2 <= cd::kBlock && cd::aligned16(hi, out)) {
        // bands of 8 passes of the workgroup's rows (384x224: 4 rows per pass, 12 bands per plane), at least one pass
        const int rpp = cd::kBlock / (w / 2), band_rows = rpp * 8;
        hipLaunchKernelGGL(cd::upsample2x_add_fwd_col_kernel, dim3((unsigned)((2 * h + band_rows - 1) / band_rows), C, N), dim3(cd::kBlock), 0,
                           (hipStream_t)stream, lo, lo_ctot, lo_coff, lo_scale, lo_shift, lo_relu, hi, hi_ctot, hi_coff, hi_scale,
                           hi_shift, hi_relu, out, o_ctot, o_coff, h, w, band_rows);
    } else
        hipLaunchKernelGGL(cd::upsample2x_add_fwd_kernel, cd::plane_grid(4 * h * w, C, N, 4), dim3(cd::kBlock), 0,
                           (hipStream_t)stream, lo, lo_ctot, lo_coff, lo_scale, lo_shift, lo_relu, hi, hi_ctot, hi_coff, hi_scale,
                           hi_shift, hi_relu, out, o_ctot, o_coff, h, w);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_upsample2x_halfpixel_fwd(const float* lo, int lo_ctot, int lo_coff, float* out, int o_ctot, int o_coff, int C, int N, int h, int w,
                                void* stream) {
    CD_ARGCHK(lo && out && C > 0 && N > 0 && h > 0 && w > 0 && lo_coff >= 0 && o_coff >= 0 && lo_coff + C <= lo_ctot && o_coff + C <= o_ctot);
    hipLaunchKernelGGL(cd::upsample2x_halfpixel_fwd_kernel, cd::plane_grid(4 * h * w, C, N, 4), dim3(cd::kBlock), 0, (hipStream_t)stream,
                       lo, lo_ctot, lo_coff, out, o_ctot, o_coff, h, w);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_upsample2x_halfpixel_bwd(const float* dout, int d_ctot, int d_coff, float* dlo, int l_ctot, int l_coff, int C, int N, int h, int w,
                                int accumulate, void* stream) {
    CD_ARGCHK(dout && dlo && C > 0 && N > 0 && h > 0 && w > 0 && d_coff >= 0 && l_coff >= 0 && d_coff + C <= d_ctot && l_coff + C <= l_ctot);
    hipLaunchKernelGGL(cd::upsample2x_halfpixel_bwd_kernel, cd::plane_grid(h * w, C, N, 1), dim3(cd::kBlock), 0, (hipStream_t)stream,
                       dout, d_ctot, d_coff, dlo, l_ctot, l_coff, h, w, accumulate);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_upsample2x_bwd(const float* dout, int d_ctot, int d_coff, float* dlo, int l_ctot, int l_coff, int C, int N, int h,
                      int w, int accumulate, void* stream) {
    CD_ARGCHK(dout && dlo && C > 0 && h > 0 && w > 0 && d_coff + C <= d_ctot && l_coff + C <= l_ctot);
    // the band kernel (LDS-staged rows, weights per row / column once per workgroup) when the band fits the LDS; else the gather
    auto lds_of = [&](int rb) { return ((size_t)w * sizeof(cd::UpTap) / 4 + (size_t)(2 * rb + 6) * 2 * w + cd::kUpWin + 3) * sizeof(float); };
    const int RB = (h >= 64 && lds_of(16) <= 40 * 1024) ? 16 : 8, rows_cap = 2 * RB + 6;   // (16 rows: 1.2x the rows' bytes instead of 1.4x)
    const size_t lds = lds_of(RB);
    if (!(cd::g_layers_mode & 1) && h >= 4 && w >= 4 && lds <= 63 * 1024) {
        const int vec = ((2 * w) % 4) == 0 && cd::aligned16(dout);
        const dim3 grid((unsigned)((h + RB - 1) / RB), C, N);
        if (RB == 16)
            hipLaunchKernelGGL(cd::upsample2x_bwd_band_kernel<16>, grid, dim3(cd::kBlock), lds, (hipStream_t)stream, dout, d_ctot, d_coff, dlo,
                               l_ctot, l_coff, h, w, accumulate, rows_cap, vec);
        else
            hipLaunchKernelGGL(cd::upsample2x_bwd_band_kernel<8>, grid, dim3(cd::kBlock), lds, (hipStream_t)stream, dout, d_ctot, d_coff, dlo,
                               l_ctot, l_coff, h, w, accumulate, rows_cap, vec);
    } else {
        hipLaunchKernelGGL(cd::upsample2x_bwd_kernel, cd::plane_grid(h * w, C, N, 1), dim3(cd::kBlock), 0, (hipStream_t)stream,
                           dout, d_ctot, d_coff, dlo, l_ctot, l_coff, h, w, accumulate);
    }
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_add_slice(const float* src, int s_ctot, int s_coff, float* dst, int d_ctot, int d_coff, int C, int N, int H, int W,
                 int accumulate, void* stream) {
    CD_ARGCHK(src && dst && C > 0 && s_coff + C <= s_ctot && d_coff + C <= d_ctot);
    const long long hw = (long long)H * W;
    const dim3 grid = (cd::g_layers_mode & 1) ? cd::plane_grid(H * W, C, N, 4) : cd::plane_grid_even((hw & 3) == 0 ? hw / 4 : hw, C, N);
    hipLaunchKernelGGL(cd::add_slice_kernel, grid, dim3(cd::kBlock), 0, (hipStream_t)stream, src,
                       s_ctot, s_coff, dst, d_ctot, d_coff, H * W, accumulate);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_copy_segments(const cd_copy_seg* table_dev, int n, void* stream) {
    CD_ARGCHK(table_dev && n > 0);
    hipLaunchKernelGGL(cd::copy_segments_kernel, dim3((unsigned)n), dim3(cd::kBlock), 0, (hipStream_t)stream, table_dev);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_counters_add(long long* const* table_dev, int n, long long delta, void* stream) {
    CD_ARGCHK(table_dev && n > 0);
    hipLaunchKernelGGL(cd::counters_add_kernel, dim3((unsigned)((n + cd::kBlock - 1) / cd::kBlock)), dim3(cd::kBlock), 0, (hipStream_t)stream,
                       table_dev, n, delta);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_zero_bytes(void* p, size_t bytes, void* stream) {
    CD_ARGCHK(p && bytes > 0 && ((uintptr_t)p & 15) == 0 && (bytes & 3) == 0);
    // An ordinary kernel of this library, NOT hipMemsetAsync: round 6's first build zeroed the flat gradient buffer (21 MB) with the
    // runtime's memset and BASELINE configs[1] -- whose gradients torch's autograd ACCUMULATES into that buffer -- diverged in about half
    // of its runs (clean with ATen's fill, clean before the change: gpurun_out bisection, HISTORY round 6); the HIP engine never
    // noticed, its first gradient contribution overwrites.
    const size_t n16 = bytes / 16;
    size_t blocks = (n16 + cd::kBlock * 4 - 1) / (cd::kBlock * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cd::zero_bytes_kernel, dim3((unsigned)blocks), dim3(cd::kBlock), 0, (hipStream_t)stream, (unsigned*)p, bytes / 4);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_channel_sum(const float* src, int ctot, int coff, int C, int N, int H, int W, float* out, int accumulate,
                   void* stream) {
    CD_ARGCHK(src && out && C > 0 && coff + C <= ctot);
    hipLaunchKernelGGL(cd::channel_sum_kernel, dim3(1, C), dim3(cd::kSumBlock), 0, (hipStream_t)stream, src, ctot,
                       coff, N, H * W, out, accumulate);
    CD_CHECK_LAUNCH();
    return CD_OK;
}

int cd_debug_set_layers_mode(int bits) {
    cd::g_layers_mode = bits;
    return CD_OK;
}

}  // extern "C"
