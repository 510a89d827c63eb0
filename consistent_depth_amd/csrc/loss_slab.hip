// Fused geometric-consistency loss, v3: "evaluate once, reduce slabs" -- the training path.
//
// Same math as loss_fused.hip (v1); different decomposition of the scatter part
// of d loss / d depth (the 4 bilinear taps of the OTHER frame's depth per source pixel):
//
//   pass A  loss_source_kernel   one workgroup = one 32x32 tile of SOURCE pixels of plane (b, j):
//             stage the tap window of frame k = 1-j (tight bounding box of where the tile's valid pixels
//             sample frame k, <= 64x64) in LDS; evaluate every source ONCE (loss partial sums, direct
//             gradient -> plain store); accumulate its 4 tap contributions into an LDS accumulator laid out
//             like the window (64-bit fixed point, ds_add_u64: integer LDS atomics are full rate on gfx950,
//             float ones are ~37x slower, and integer sums are order-independent); write the accumulator
//             out as a float SLAB (plain coalesced stores).
//   pass B  loss_gather_kernel   one workgroup = one 32x32 tile of TARGET pixels of plane (b, k): add the
//             overlapping part of every slab of plane j to the direct gradient (fixed order: bit-reproducible),
//             one plain read-modify-write of the tile.
//
// No global atomics, no zero-initialised gradient, every source evaluated exactly once (v2 evaluates ~2.07x),
// and exactness does not depend on forward/backward flow consistency: a tap is outside its tile's window only
// when the window hit the 64x64 cap (wild flow) or the pixel is masked out (no contribution); those few go
// through the same overflow list / guarded v1 fallback as in v2.
#include "loss_tiles.h"

namespace cd {

// v3 keeps TWO window-shaped arrays in LDS (depth + 64-bit accumulator), so its window cap is smaller than the
// table's (64x64): table windows are centre-cropped to 48x48 on the fly, identically in both passes.
constexpr int V3W = 48;
// Slab layout: row r of the window starts at image column xa0 = win.x0 & ~3 (so that float4-aligned target columns
// are float4-aligned in the slab), fixed row stride SLAB_RS; the <= 3 pad columns either side of the window hold 0.
constexpr int SLAB_RS = V3W + 4;
constexpr int SLAB_STRIDE = V3W * SLAB_RS;  // floats reserved per (plane, tile) slab; only h rows of pw are touched

__device__ __forceinline__ TileWin crop_win(TileWin w) {
    if (w.w > V3W) { w.x0 = (short)(w.x0 + (w.w - V3W) / 2); w.w = (short)V3W; }
    if (w.h > V3W) { w.y0 = (short)(w.y0 + (w.h - V3W) / 2); w.h = (short)V3W; }
    return w;
}

template <int MODE, bool REPROJ>
__global__ __launch_bounds__(kBlock) void loss_source_kernel(
    const float* __restrict__ depth, const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd,
    const float* __restrict__ mask_fwd, const float* __restrict__ mask_bwd, const PairCam* __restrict__ cams,
    const TileWin* __restrict__ wins, int H, int W, int tiles_x, int ntiles, int wstride, float* __restrict__ partial,
    float* __restrict__ grad, float* __restrict__ slabs, Overflow* ovf, unsigned* __restrict__ oidx,
    float* __restrict__ oval, int b0) {
    __shared__ float sA[V3W * V3W];                     // depth of frame k over the window
    __shared__ unsigned long long sW[V3W * V3W];        // scatter accumulator over the same window (2^-40 fixed point)
    __shared__ float red[kBlock / kWave];

    const int j = blockIdx.y, b = b0 + blockIdx.z, tile = blockIdx.x, k = 1 - j;   // slabs are per chunk: indexed by blockIdx.z
    const int HW = H * W;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int X0 = txi * TW, Y0 = tyi * TH;
    const PairCam& cj = cams[b * 2 + j];
    const TileWin win = crop_win(wins[(size_t)b * wstride + (size_t)j * ntiles + tile]);
    const float* __restrict__ v_j = depth + (size_t)(b * 2 + j) * HW;
    const float* __restrict__ v_k = depth + (size_t)(b * 2 + k) * HW;
    const float* __restrict__ fl_j = (j == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
    const float* __restrict__ mk_j = (j == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
    float* __restrict__ g_j = grad + (size_t)(b * 2 + j) * HW;
    const unsigned base_k = (unsigned)(b * 2 + k) * (unsigned)HW;

    // ---- per-pixel inputs of this thread's 4 rows: issued first so their latency hides behind the window staging
    constexpr int ROWS_PER_IT = kBlock / TW, ITERS = TH / ROWS_PER_IT;
    const int lx = threadIdx.x & (TW - 1), ly0 = threadIdx.x / TW;
    float in_v[ITERS], in_m[ITERS], in_fx[ITERS], in_fy[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int x = X0 + lx, y = Y0 + ly0 + it * ROWS_PER_IT;
        const bool valid = x < W && y < H;
        const int p = valid ? y * W + x : 0;
        in_v[it] = v_j[p];
        in_m[it] = valid ? mk_j[p] : 0.f;
        in_fx[it] = fl_j[p];
        in_fy[it] = fl_j[HW + p];
    }

    // ---- stage the window, clear the accumulator
    const int wn = (int)win.w * (int)win.h;
    const float inv_ww = win.w > 0 ? 1.f / (float)win.w : 0.f;
    for (int i = threadIdx.x; i < wn; i += kBlock) {
        const int r = (int)(((float)i + 0.5f) * inv_ww), c = i - r * win.w;
        sA[r * V3W + c] = to_depth<MODE>(v_k[(win.y0 + r) * W + win.x0 + c]);
        sW[r * V3W + c] = 0ull;
    }
    __syncthreads();

    // ---- evaluate every source pixel of the tile once
    float acc_r = 0.f, acc_d = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int ly = ly0 + it * ROWS_PER_IT;
        const int x = X0 + lx, y = Y0 + ly;
        const bool valid = x < W && y < H;
        const int p = valid ? y * W + x : 0;
        const float d = to_depth<MODE>(in_v[it]);
        const float m = in_m[it];
        const float fx = in_fx[it], fy = in_fy[it];
        const float xf = (float)x, yf = (float)y;
        const float r0 = (xf - cj.cx_r) * cj.ifx_r, r1 = -(yf - cj.cy_r) * cj.ify_r;
        const float a0 = cj.M[0] * r0 + cj.M[1] * r1 - cj.M[2];
        const float a1 = cj.M[3] * r0 + cj.M[4] * r1 - cj.M[5];
        const float a2 = cj.M[6] * r0 + cj.M[7] * r1 - cj.M[8];
        const float X = d * a0 + cj.c[0], Y = d * a1 + cj.c[1], Z = d * a2 + cj.c[2];
        const float iZ = __builtin_amdgcn_rcpf(Z);
        float g = 0.f;
        if (REPROJ) {
            const float mx = xf + fx, my = yf + fy;
            const float ex = (cj.cx_t - cj.fx_t * X * iZ) - mx, ey = (cj.cy_t + cj.fy_t * Y * iZ) - my;
            const float e2 = ex * ex + ey * ey;
            const float ie = e2 > 0.f ? __builtin_amdgcn_rsqf(e2) : 0.f;  // subgradient 0 at e = 0
            acc_r += valid ? m * (e2 * ie) : 0.f;   // multiply, not select: 0*inf = NaN exactly like the reference
            const float dpx = cj.fx_t * iZ * (X * a2 * iZ - a0), dpy = cj.fy_t * iZ * (a1 - Y * a2 * iZ);
            g += cj.gr * m * (ex * dpx + ey * dpy) * ie;
        }
        const Taps t = tap_coords(xf, yf, fx, fy, cj.sx, cj.sy, W, H);
        const int ra = t.ya - win.y0, ca = t.xa - win.x0, dyb = t.yb - t.ya, dxb = t.xb - t.xa;
        const bool inside = (unsigned)ra < (unsigned)max(win.h - dyb, 0) && (unsigned)ca < (unsigned)max(win.w - dxb, 0);
        const int i00 = inside ? ra * V3W + ca : 0;
        float d00, d01, d10, d11;
        const bool fast = __all(inside || !valid);   // the whole wave's taps are inside the window: the common case
        if (fast) {
            d00 = sA[i00]; d01 = sA[i00 + dxb]; d10 = sA[i00 + dyb * V3W]; d11 = sA[i00 + dyb * V3W + dxb];
        } else {
            const int rb = ra + dyb, cb = ca + dxb;
            const bool ina = (unsigned)ra < (unsigned)win.h, inb = (unsigned)rb < (unsigned)win.h;
            const bool inca = (unsigned)ca < (unsigned)win.w, incb = (unsigned)cb < (unsigned)win.w;
            d00 = (ina && inca) ? sA[ra * V3W + ca] : to_depth<MODE>(v_k[t.ya * W + t.xa]);
            d01 = (ina && incb) ? sA[ra * V3W + cb] : to_depth<MODE>(v_k[t.ya * W + t.xb]);
            d10 = (inb && inca) ? sA[rb * V3W + ca] : to_depth<MODE>(v_k[t.yb * W + t.xa]);
            d11 = (inb && incb) ? sA[rb * V3W + cb] : to_depth<MODE>(v_k[t.yb * W + t.xb]);
        }
        const float zs = -(d00 * t.w00 + d01 * t.w01 + d10 * t.w10 + d11 * t.w11);
        const float izs = __builtin_amdgcn_rcpf(zs);
        const float dd = iZ - izs;
        acc_d += valid ? m * fabsf(dd) : 0.f;
        const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
        const float gm = cj.gb * m * sg;
        g -= gm * a2 * iZ * iZ;
        const float gz = gm * izs * izs;
        if (valid) g_j[p] = g * depth_jac<MODE>(d);   // direct term; pass B adds the scatter term
        // scatter term of the 4 taps of frame k: LDS window accumulator (or the overflow list outside the window)
        const float c00 = -gz * t.w00 * depth_jac<MODE>(d00), c01 = -gz * t.w01 * depth_jac<MODE>(d01);
        const float c10 = -gz * t.w10 * depth_jac<MODE>(d10), c11 = -gz * t.w11 * depth_jac<MODE>(d11);
        if (fast) {
            if (m != 0.f) {   // valid lanes only have m != 0
                atomicAdd(&sW[i00], to_fixed(c00));
                atomicAdd(&sW[i00 + dxb], to_fixed(c01));
                atomicAdd(&sW[i00 + dyb * V3W], to_fixed(c10));
                atomicAdd(&sW[i00 + dyb * V3W + dxb], to_fixed(c11));
            }
        } else {
            const int xs[4] = {t.xa, t.xb, t.xa, t.xb}, ys[4] = {t.ya, t.ya, t.yb, t.yb};
            const float cs[4] = {c00, c01, c10, c11};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rq = ys[q] - win.y0, cq = xs[q] - win.x0;
                const bool inq = (unsigned)rq < (unsigned)win.h && (unsigned)cq < (unsigned)win.w;
                const bool live = m != 0.f && cs[q] != 0.f;
                if (live && inq) atomicAdd(&sW[rq * V3W + cq], to_fixed(cs[q]));
                ovf_push(live && !inq, ovf, oidx, oval, base_k + (unsigned)(ys[q] * W + xs[q]), cs[q]);   // convergent
            }
        }
    }
    __syncthreads();

    // ---- the window accumulator leaves as a float slab (float4 stores, target-aligned rows)
    float* __restrict__ slab = slabs + ((size_t)(blockIdx.z * 2 + j) * ntiles + tile) * SLAB_STRIDE;
    const int xa0 = win.x0 & ~3, lpad = win.x0 - xa0;
    const int pw4 = (((win.x0 + win.w + 3) & ~3) - xa0) >> 2;
    const int nq = pw4 * win.h;
    const float inv_pw4 = pw4 > 0 ? 1.f / (float)pw4 : 0.f;
    for (int i = threadIdx.x; i < nq; i += kBlock) {
        const int r = (int)(((float)i + 0.5f) * inv_pw4), q = i - r * pw4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = q * 4 + e - lpad;
            v[e] = (unsigned)c < (unsigned)win.w ? from_fixed(sW[r * V3W + c]) : 0.f;
        }
        *reinterpret_cast<float4*>(slab + r * SLAB_RS + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    acc_r = block_sum(acc_r, red);
    acc_d = block_sum(acc_d, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((size_t)(b * 2 + j) * ntiles + tile) * 2;
        o[0] = acc_r;
        o[1] = acc_d;
    }
}

// pass B, general form (any W, any number of tiles): grad[b, k, T] += sum over the source tiles s of plane j of
// slab_s restricted to T.  Which slabs overlap T is found in parallel (one thread per source tile -> bitmask in LDS); the overlapping
// ones are then visited in increasing s (fixed order -> bit-reproducible sums), each contributing a rectangle of
// distinct elements of the LDS tile accumulator.
__global__ __launch_bounds__(kBlock) void loss_gather_kernel(const TileWin* __restrict__ wins,
                                                             const float* __restrict__ slabs, int H, int W,
                                                             int tiles_x, int ntiles, int wstride, float* __restrict__ grad, int b0) {
    __shared__ float sAcc[TH * TW];
    __shared__ unsigned sBits[MAXT_LDS / 32];
    __shared__ TileWin sWin[MAXT_LDS];
    const int k = blockIdx.y, b = b0 + blockIdx.z, tile = blockIdx.x, j = 1 - k;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int X0 = txi * TW, Y0 = tyi * TH, X1 = min(X0 + TW, W), Y1 = min(Y0 + TH, H);
    const TileWin* __restrict__ wj = wins + (size_t)b * wstride + (size_t)j * ntiles;
    const float* __restrict__ sl = slabs + (size_t)(blockIdx.z * 2 + j) * ntiles * SLAB_STRIDE;
    for (int i = threadIdx.x; i < TH * TW; i += kBlock) sAcc[i] = 0.f;
    if (threadIdx.x < MAXT_LDS / 32) sBits[threadIdx.x] = 0u;
    __syncthreads();
    const bool use_bits = ntiles <= MAXT_LDS;
    if (use_bits) {
        for (int s = threadIdx.x; s < ntiles; s += kBlock) {
            const TileWin w = crop_win(wj[s]);
            sWin[s] = w;
            const bool hit = min((int)w.x0 + w.w, X1) > max((int)w.x0, X0) && min((int)w.y0 + w.h, Y1) > max((int)w.y0, Y0);
            if (hit) atomicOr(&sBits[s >> 5], 1u << (s & 31));
        }
        __syncthreads();
    }
    const int nwords = use_bits ? (ntiles + 31) / 32 : ntiles;
    for (int wd = 0; wd < nwords; ++wd) {
        unsigned bits = use_bits ? sBits[wd] : 1u;   // block-uniform
        while (bits) {
            const int s = use_bits ? wd * 32 + __ffs((int)bits) - 1 : wd;
            bits &= bits - 1u;
            const TileWin w = use_bits ? sWin[s] : crop_win(wj[s]);
            const int x0 = max((int)w.x0, X0), y0 = max((int)w.y0, Y0);
            const int x1 = min((int)w.x0 + w.w, X1), y1 = min((int)w.y0 + w.h, Y1);
            const int rw = x1 - x0, rh = y1 - y0;
            if (rw <= 0 || rh <= 0) continue;
            const float* __restrict__ src = sl + (size_t)s * SLAB_STRIDE;
            const float inv = 1.f / (float)rw;
            for (int i = threadIdx.x; i < rw * rh; i += kBlock) {
                const int r = (int)(((float)i + 0.5f) * inv), c = i - r * rw;
                const int y = y0 + r, x = x0 + c;
                sAcc[(y - Y0) * TW + (x - X0)] += src[(y - w.y0) * SLAB_RS + (x - (w.x0 & ~3))];   // distinct elements within one slab
            }
            __syncthreads();   // the next slab may touch the same elements from other threads
        }
    }
    float* __restrict__ g = grad + (size_t)(b * 2 + k) * H * W;
    for (int i = threadIdx.x; i < TH * TW; i += kBlock) {
        const int ly = i / TW, lx = i - ly * TW, y = Y0 + ly, x = X0 + lx;
        if (x < W && y < H) g[y * W + x] += sAcc[i];
    }
}

// pass B, float4 form (W % 4 == 0, <= MAXT_LDS tiles): one thread owns one float4 of the 32x32 target tile.  The
// overlapping slabs are found in parallel (bitmask), compacted in increasing s into an LDS list, and every thread
// adds its float4 of each listed slab that covers it: no barriers in the accumulation, fixed order of summation.
__global__ __launch_bounds__(kBlock) void loss_gather4_kernel(const TileWin* __restrict__ wins,
                                                              const float* __restrict__ slabs, int H, int W,
                                                              int tiles_x, int ntiles, int wstride, float* __restrict__ grad, int b0) {
    static_assert(kBlock == TH * TW / 4, "one float4 per thread");
    __shared__ unsigned sBits[MAXT_LDS / 32];
    __shared__ int4 sEnt[MAXT_LDS];   // y0, h, xa0, padded width of the overlapping slabs, in increasing s
    __shared__ int sOff[MAXT_LDS];
    const int k = blockIdx.y, b = b0 + blockIdx.z, tile = blockIdx.x, j = 1 - k;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int X0 = txi * TW, Y0 = tyi * TH, X1 = min(X0 + TW, W), Y1 = min(Y0 + TH, H);
    const TileWin* __restrict__ wj = wins + (size_t)b * wstride + (size_t)j * ntiles;
    const float* __restrict__ sl = slabs + (size_t)(blockIdx.z * 2 + j) * ntiles * SLAB_STRIDE;
    if (threadIdx.x < MAXT_LDS / 32) sBits[threadIdx.x] = 0u;
    __syncthreads();
    for (int s = threadIdx.x; s < ntiles; s += kBlock) {
        const TileWin w = crop_win(wj[s]);
        const int xa0 = w.x0 & ~3, xa1 = (w.x0 + w.w + 3) & ~3;
        if (min(xa1, X1) > max(xa0, X0) && min((int)w.y0 + w.h, Y1) > max((int)w.y0, Y0)) atomicOr(&sBits[s >> 5], 1u << (s & 31));
    }
    __syncthreads();
    const int nwords = (ntiles + 31) >> 5;
    for (int s = threadIdx.x; s < ntiles; s += kBlock) {
        const unsigned word = sBits[s >> 5];
        if (word >> (s & 31) & 1u) {
            int pos = __popc(word & ((1u << (s & 31)) - 1u));
            for (int wd = 0; wd < (s >> 5); ++wd) pos += __popc(sBits[wd]);
            const TileWin w = crop_win(wj[s]);
            const int xa0 = w.x0 & ~3;
            sEnt[pos] = make_int4(w.y0, w.h, xa0, ((w.x0 + w.w + 3) & ~3) - xa0);
            sOff[pos] = s * SLAB_STRIDE;
        }
    }
    int count = 0;
    for (int wd = 0; wd < nwords; ++wd) count += __popc(sBits[wd]);
    __syncthreads();
    const int y = Y0 + (threadIdx.x >> 3), x = X0 + (threadIdx.x & 7) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = 0; e < count; ++e) {
        const int4 en = sEnt[e];
        const int r = y - en.x, c = x - en.z;
        if ((unsigned)r < (unsigned)en.y && (unsigned)c < (unsigned)en.w) {
            const float4 v = *reinterpret_cast<const float4*>(sl + sOff[e] + r * SLAB_RS + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (y < H && x < W) {
        float4* g = reinterpret_cast<float4*>(grad + (size_t)(b * 2 + k) * H * W + (size_t)y * W + x);
        float4 o = *g;
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *g = o;
    }
}

// ---------------------------------------------------------------- host side
static int g_slab_chunk_override = 0;   // pairs per chunk; 0 = size the chunk's slabs to kSlabChunkBytes
constexpr size_t kSlabChunkBytes = 216u << 20;   // 128 pairs at 384x224; chunks are balanced (B=256 -> 2 x 128)
void set_slab_chunk_pairs(int pairs) { g_slab_chunk_override = pairs > 0 ? pairs : 0; }
int slab_chunk_pairs(int H, int W) {
    if (g_slab_chunk_override > 0) return g_slab_chunk_override;
    const size_t per_pair = (size_t)2 * owner_ntiles(H, W) * SLAB_STRIDE * sizeof(float);
    const size_t ch = kSlabChunkBytes / per_pair;
    return ch < 1 ? 1 : (int)ch;
}
size_t slab_floats(int B, int H, int W) {
    return (size_t)min(B, slab_chunk_pairs(H, W)) * 2 * owner_ntiles(H, W) * SLAB_STRIDE;
}

template <int MODE>
static void launch_source_mode(bool reproj, dim3 grid, hipStream_t s, const float* depth, const float* ff,
                               const float* fb, const float* mf, const float* mb, const PairCam* cams,
                               const TileWin* wins, int H, int W, int tx, int nt, int wstride, float* partial, float* grad,
                               float* slabs, Overflow* ovf, unsigned* oidx, float* oval, int b0) {
    if (reproj)
        hipLaunchKernelGGL((loss_source_kernel<MODE, true>), grid, dim3(kBlock), 0, s, depth, ff, fb, mf, mb, cams, wins, H, W,
                           tx, nt, wstride, partial, grad, slabs, ovf, oidx, oval, b0);
    else
        hipLaunchKernelGGL((loss_source_kernel<MODE, false>), grid, dim3(kBlock), 0, s, depth, ff, fb, mf, mb, cams, wins, H, W,
                           tx, nt, wstride, partial, grad, slabs, ovf, oidx, oval, b0);
}

// Enqueues: overflow header reset, [before_main] source pass, gather pass [after_main], overflow apply.
int launch_slab(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb,
                const void* cams, const void* wins, int mode, bool reproj, int B, int H, int W, float* partial,
                float* grad, float* slabs, void* ovf_mem, int ovf_cap, hipStream_t s, void (*before_main)(hipStream_t),
                void (*after_main)(hipStream_t)) {
    const int tx = owner_tiles_x(W), nt = owner_ntiles(H, W);
    const int wstride = (int)(pair_record_bytes(H, W) / sizeof(TileWin));
    Overflow* ovf = (Overflow*)ovf_mem;
    unsigned* oidx = (unsigned*)((char*)ovf_mem + 256);
    float* oval = (float*)(oidx + ovf_cap);
    if (hipMemsetAsync(ovf, 0, sizeof(Overflow), s) != hipSuccess) return CD_ERR_LAUNCH;
    if (hipMemsetD32Async((hipDeviceptr_t)&ovf->cap, ovf_cap, 1, s) != hipSuccess) return CD_ERR_LAUNCH;
    if (before_main) before_main(s);
    // Pairs are processed in balanced chunks (pass A then pass B per chunk) so the slab scratch is bounded for any B.
    // (Measured: chunk size does not change the rate as long as each launch has >= ~10k workgroups.)
    const int ch_max = slab_chunk_pairs(H, W), nch = (B + ch_max - 1) / ch_max, ch = (B + nch - 1) / nch;
    for (int b0 = 0; b0 < B; b0 += ch) {
        const dim3 grid(nt, 2, min(ch, B - b0));
        if (mode == CD_DEPTH_EXP)
            launch_source_mode<CD_DEPTH_EXP>(reproj, grid, s, depth, ff, fb, mf, mb, (const PairCam*)cams, (const TileWin*)wins, H, W, tx, nt, wstride, partial, grad, slabs, ovf, oidx, oval, b0);
        else if (mode == CD_DEPTH_RECIPROCAL)
            launch_source_mode<CD_DEPTH_RECIPROCAL>(reproj, grid, s, depth, ff, fb, mf, mb, (const PairCam*)cams, (const TileWin*)wins, H, W, tx, nt, wstride, partial, grad, slabs, ovf, oidx, oval, b0);
        else
            launch_source_mode<CD_DEPTH_IDENTITY>(reproj, grid, s, depth, ff, fb, mf, mb, (const PairCam*)cams, (const TileWin*)wins, H, W, tx, nt, wstride, partial, grad, slabs, ovf, oidx, oval, b0);
        if (W % 4 == 0 && nt <= MAXT_LDS && reinterpret_cast<uintptr_t>(grad) % 16 == 0)
            hipLaunchKernelGGL(loss_gather4_kernel, grid, dim3(kBlock), 0, s, (const TileWin*)wins, slabs, H, W, tx, nt, wstride, grad, b0);
        else
            hipLaunchKernelGGL(loss_gather_kernel, grid, dim3(kBlock), 0, s, (const TileWin*)wins, slabs, H, W, tx, nt, wstride, grad, b0);
    }
    if (after_main) after_main(s);
    if (hipGetLastError() != hipSuccess) return CD_ERR_LAUNCH;
    launch_overflow_apply(ovf_mem, ovf_cap, grad, s);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // namespace cd
