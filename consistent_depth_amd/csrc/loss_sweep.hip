// Fused geometric-consistency loss, v4: the ROW SWEEP -- one workgroup per frame pair, every input byte read once,
// every gradient byte written once, scatter through 64-bit integer LDS atomics into sliding row rings.
// Design, plan and the per-thread phase functions: loss_sweep_core.h (shared with the host emulation of the CPU tests).
//
// Replaces (reference, /root/reference): loss/consistency_loss.py:98-253 + the utils/geometry.py chain :9-128,201-208 and
// the autograd backward of all of it -- for batches large enough to give every CU a pair (bench.py's roofline launch,
// validation-sized batches); small batches (the B = 4 training step) stay on the tile kernels of loss_slab.hip, which
// have 2 * B * tiles workgroups to spread.
#include "loss_tiles.h"
#include "loss_sweep_core.h"

namespace cd {

using namespace sweep;

static int g_sweep_pxt = 0;   // cd_debug_set_loss_sweep: pixels per thread (0 = rule below)

int sweep_pxt(int H, int W) {
    if (g_sweep_pxt > 0) return g_sweep_pxt;
    (void)H;
    // two columns per thread fill 7/8 of the 512 lanes of a frame at W = 224 and keep G (rows per item) small, which
    // leaves the ring's rows to the flow's vertical spread
    return W <= 512 ? 2 : 4;
}
void set_sweep_pxt(int pxt) { g_sweep_pxt = pxt > 0 && pxt <= kMaxPXT ? pxt : 0; }

Geo sweep_geo(int H, int W) { return make_geo(H, W, sweep_pxt(H, W)); }

// per-pair record of the "tile windows" blob: [TileWin wins[2 * ntiles]] [PlanHeader + Item[max_items]] (16-byte aligned parts)
// The record size must NOT depend on the (mutable, debug) pixels-per-thread choice: blobs are cached by the callers
// (PairStore) and every variant derives its window stride from it -- room for the largest plan of any supported choice.
size_t pair_record_bytes(int H, int W) {
    const size_t wins = align_up(sizeof(TileWin) * 2 * (size_t)owner_ntiles(H, W), 16);
    size_t plan = 0;
    for (int pxt = 1; pxt <= kMaxPXT; pxt *= 2) {
        const Geo g = make_geo(H, W, pxt);
        if (g.ok && g.max_items <= 2560) { const size_t n = align_up(plan_bytes(g), 16); plan = n > plan ? n : plan; }
    }
    return wins + plan;
}
static size_t plan_offset(int H, int W) { return align_up(sizeof(TileWin) * 2 * (size_t)owner_ntiles(H, W), 16); }

// ---------------------------------------------------------------- plan (dataset constant: flows and masks only)
constexpr int kFanWords = 16384;      // fan-in counters of one band of target rows (64 KB)
constexpr int kPlanItemsLds = 2560;   // Item scratch of the planner (8 B each); plans longer than this are not made (-> v3)

__global__ __launch_bounds__(kBlock) void sweep_plan_kernel(const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd,
                                                            const float* __restrict__ mask_fwd, const float* __restrict__ mask_bwd,
                                                            const Geo g, char* __restrict__ blob, size_t stride, size_t plan_off) {
    __shared__ int lo_i[2 * kMaxGroups], hi_i[2 * kMaxGroups];
    __shared__ short lo_s[2 * kMaxGroups], hi_s[2 * kMaxGroups], suf[2 * (kMaxGroups + 1)];
    __shared__ Item items[kPlanItemsLds];
    __shared__ int fan[kFanWords];
    const int b = blockIdx.x, HW = g.H * g.W, NG = g.NG;
    for (int i = threadIdx.x; i < 2 * NG; i += kBlock) { lo_i[i] = kNoRow; hi_i[i] = -1; }
    __syncthreads();
    for (int f = 0; f < 2; ++f) {
        const float* fl = (f == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
        const float* mk = (f == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
        for (int p = threadIdx.x; p < HW; p += kBlock) {
            if (mk[p] != 0.f) {
                const int y = p / g.W, x = p - y * g.W;
                int ya, yb;
                tap_rows((float)x, (float)y, fl[p], fl[HW + p], g.W, g.H, &ya, &yb);
                atomicMin(&lo_i[f * NG + y / g.G], ya);
                atomicMax(&hi_i[f * NG + y / g.G], yb);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NG; i += kBlock) { lo_s[i] = (short)lo_i[i]; hi_s[i] = (short)hi_i[i]; }
    __syncthreads();
    // FAN-IN: how many valid sources add to one target pixel (its 4 taps count a source once each).  The 32-bit accumulators
    // (loss_math.h) cannot wrap while (fan-in + 1) * LIMIT < 2^31.  Counted band by band of target rows in LDS; the sources are
    // re-scanned per band (a once-per-dataset kernel).
    const int cw = g.W + 1, band = kFanWords / cw;
    int fmax = 0;
    for (int f = 0; f < 2; ++f) {
        const float* fl = (f == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
        const float* mk = (f == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
        for (int r0 = 0; r0 <= g.H; r0 += band) {
            for (int i = threadIdx.x; i < band * cw; i += kBlock) fan[i] = 0;
            __syncthreads();
            for (int p = threadIdx.x; p < HW; p += kBlock) {
                if (mk[p] != 0.f) {
                    const int y = p / g.W, x = p - y * g.W;
                    int xa, ya;
                    tap_targets((float)x, (float)y, fl[p], fl[HW + p], g.W, g.H, &xa, &ya);
                    for (int dy = 0; dy < 2; ++dy) {
                        const int rr = ya + dy - r0;
                        if ((unsigned)rr < (unsigned)band) { atomicAdd(&fan[rr * cw + xa], 1); atomicAdd(&fan[rr * cw + xa + 1], 1); }
                    }
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < band * cw; i += kBlock) fmax = fan[i] > fmax ? fan[i] : fmax;
            __syncthreads();
        }
    }
    lo_i[threadIdx.x] = fmax;          // (lo_i is free now: block-wide max through it)
    __syncthreads();
    if (threadIdx.x == 0) {
        int fan_in = 0;
        for (int i = 0; i < kBlock; ++i) fan_in = lo_i[i] > fan_in ? lo_i[i] : fan_in;
        PlanHeader* ph = reinterpret_cast<PlanHeader*>(blob + (size_t)b * stride + plan_off);
        int n = plan_items(g, lo_s, hi_s, suf, items);
        if (n > 0 && fan_in > SWEEP_MAX_FAN_IN) n = -3;     // a 32-bit accumulator could wrap: no plan, the exact fallback path
        if (n > 0) expand_plan(g, items, n, reinterpret_cast<PlanItem*>(ph + 1));
        ph->n_items = n; ph->G = g.G; ph->R = g.R; ph->PXT = g.PXT;
        ph->fan_in = fan_in; ph->limit = sweep_limit_scaled(fan_in); ph->pad[0] = ph->pad[1] = 0;
    }
}

int launch_sweep_plan(const float* ff, const float* fb, const float* mf, const float* mb, int B, int H, int W, void* blob,
                      hipStream_t s) {
    const Geo g = sweep_geo(H, W);
    if (!g.ok || g.max_items > kPlanItemsLds) return CD_OK;
    hipLaunchKernelGGL(sweep_plan_kernel, dim3(B), dim3(kBlock), 0, s, ff, fb, mf, mb, g, (char*)blob, pair_record_bytes(H, W),
                       plan_offset(H, W));
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- accumulator units (per launch: they depend on the depths)
// PairPrep (non-null intr): the per-(pair, direction) constants are computed HERE, by the pair's own workgroup, instead of by a
// one-workgroup prep_kernel launch in front (loss_api.hip) -- the same code on the same inputs (fbar by the same block_sum over the
// same 256 threads: identical bits), one launch less on the path of every gradient launch.
struct PairPrep { const float* intr; const float* extr; const float* mask_sum; float lambda_r, lambda_b; };

template <int MODE>
__global__ __launch_bounds__(kUnitGrid * kUnitGrid) void sweep_units_kernel(const float* __restrict__ depth, const float* __restrict__ ff,
                                                                            const float* __restrict__ fb, const float* __restrict__ mf,
                                                                            const float* __restrict__ mb, PairCam* __restrict__ cams, int H, int W,
                                                                            const PairPrep pp, int B) {
    constexpr int NS = kUnitGrid * kUnitGrid;
    static_assert(NS == kBlock, "fbar is reduced exactly like prep_kernel does");
    __shared__ float sd[2][NS], ss[2][NS];
    __shared__ int sn[2][NS];
    const int b = blockIdx.x, t = threadIdx.x, HW = H * W;
    if (pp.intr != nullptr) {
        __shared__ float lds[kBlock / kWave];
        __shared__ float fbar_s[2];
        for (int k = 0; k < 2; ++k) {      // = prep_kernel (loss_api.hip)
            float acc = 0.f;
            for (int bb = threadIdx.x; bb < B; bb += kBlock) acc += pp.intr[(bb * 2 + k) * 4 + 0] + pp.intr[(bb * 2 + k) * 4 + 1];
            acc = block_sum(acc, lds);
            if (threadIdx.x == 0) fbar_s[k] = acc / (2.f * (float)B);
            __syncthreads();
        }
        if (t == 0) prep_pair(pp.intr + b * 8, pp.extr + b * 24, pp.mask_sum + b * 2, fbar_s, pp.lambda_r, pp.lambda_b, B, H, W, cams + b * 2);
        __syncthreads();      // the pair's constants are visible to the workgroup
    }
    for (int j = 0; j < 2; ++j) {
        const UnitSample u = unit_sample_at<MODE>(cams + b * 2, depth + (size_t)b * 2 * HW, ff + (size_t)b * 2 * HW, fb + (size_t)b * 2 * HW,
                                                  mf + (size_t)b * HW, mb + (size_t)b * HW, H, W, j, t);
        sd[j][t] = u.direct; ss[j][t] = u.scatter; sn[j][t] = u.valid;
    }
    __syncthreads();
    if (t == 0) {   // 256 samples: summed in index order by one thread -- the order of the host emulation, bit-reproducible
        float D[2] = {0.f, 0.f}, S[2] = {0.f, 0.f};
        int n[2] = {0, 0};
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < NS; ++i) { D[j] += sd[j][i]; S[j] += ss[j][i]; n[j] += sn[j][i]; }
        units_from_samples(cams + b * 2, D, S, n);
    }
}

static int launch_sweep_units(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, PairCam* cams, int mode,
                              int B, int H, int W, const PairPrep& pp, hipStream_t s) {
    const dim3 grid(B), block(kUnitGrid * kUnitGrid);
    if (mode == CD_DEPTH_EXP) hipLaunchKernelGGL(sweep_units_kernel<CD_DEPTH_EXP>, grid, block, 0, s, depth, ff, fb, mf, mb, cams, H, W, pp, B);
    else if (mode == CD_DEPTH_RECIPROCAL) hipLaunchKernelGGL(sweep_units_kernel<CD_DEPTH_RECIPROCAL>, grid, block, 0, s, depth, ff, fb, mf, mb, cams, H, W, pp, B);
    else hipLaunchKernelGGL(sweep_units_kernel<CD_DEPTH_IDENTITY>, grid, block, 0, s, depth, ff, fb, mf, mb, cams, H, W, pp, B);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

// ---------------------------------------------------------------- the sweep
struct DevEnv {
    Overflow* ovf; unsigned* oidx; float* oval;
    __device__ __forceinline__ static void add32(unsigned* p, int v) { atomicAdd(p, (unsigned)v); }   // ds_add_u32, no return
    __device__ __forceinline__ static bool any(bool x) { return __any(x) != 0; }
    __device__ __forceinline__ void push(bool need, unsigned idx, float v) { ovf_push(need, ovf, oidx, oval, idx, v); }
    __device__ __forceinline__ void degenerate() { ovf->degenerate = 1; }
};

struct SweepShape { Geo g; size_t stride, plan_off; };

__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// SG: 0 = the geometry arrives at run time (kernel argument); 1 = the BASELINE shape 384 x 224 at 2 pixels per thread as
// COMPILE-TIME constants: ring strides, rows per pass, image size fold into immediates (the kernel is short of scalar registers:
// ~30 wave-uniform camera constants, 10 pointers and the plan records live next to them) -- same code, same results.
constexpr int kStaticH = 384, kStaticW = 224, kStaticPXT = 2;
static_assert(kStagePasses == 2, "the service wave's quad count assumes SMAX = 2 passes of RP = 4 rows at the static geometry");
constexpr int kStaticSvcQuads = (2 * 4 * (kStaticW / 4) + kSvcLanes - 1) / kSvcLanes;
// what the compile-time instantiation assumes about its geometry beyond H / W / PXT (the service wave's row assignment and the fast
// source pass): checked at launch against make_geo's result -- a change of kFrameThreads, kStagePasses or make_geo that breaks one of
// them sends the BASELINE shape to the run-time-geometry instantiation instead of computing wrong gradients
static bool static_geometry_holds(const Geo& c) {
    return svc_geometry_ok(c) && svc_quads(c) == kStaticSvcQuads && c.RP == 4 && c.SMAX == kStagePasses * c.RP && fast_geometry_ok(c) &&
           kStaticH % 4 == 0;
}

template <int MODE, bool REPROJ, int PXT, int SG>
__global__ __launch_bounds__(kThreads) void loss_sweep_kernel(
    const float* __restrict__ depth, const float* __restrict__ ff, const float* __restrict__ fb, const float* __restrict__ mf,
    const float* __restrict__ mb, const PairCam* __restrict__ cams, const char* __restrict__ blob, float* __restrict__ partial,
    float* __restrict__ grad, Overflow* ovf, unsigned* __restrict__ oidx, float* __restrict__ oval, const SweepShape sh) {
    extern __shared__ __align__(16) unsigned smem[];   // [2][ring] accumulators, [2][ring] depths, reduction scratch
    const Geo g = SG == 1 ? make_geo(kStaticH, kStaticW, kStaticPXT) : sh.g;
    const int b = blockIdx.x, HW = g.H * g.W, ring = ring_rows(g) * g.RW;
    const int f = uni((int)(threadIdx.x / kFrameThreads)), k = 1 - f;   // whole waves serve one frame: everything derived from f is scalar
    View v;
    v.H = g.H; v.W = g.W; v.R = g.R; v.RW = g.RW; v.RP = g.RP; v.G = g.G; v.CG = g.CG; v.HW = (unsigned)HW;
    const float* dpair = depth + (size_t)b * 2 * HW;
    v.vj = dpair + (f ? HW : 0); v.vk = dpair + (f ? 0 : HW);
    v.flj = (f ? fb : ff) + (size_t)b * 2 * HW;
    v.mkj = (f ? mb : mf) + (size_t)b * HW;
    v.gradj = grad + (size_t)b * 2 * HW + (f ? HW : 0);
    float* D0 = reinterpret_cast<float*>(smem + 2 * ring);
    v.Aj = smem + (f ? ring : 0); v.Ak = smem + (f ? 0 : ring);
    v.Dj = D0 + (f ? ring : 0); v.Dk = D0 + (f ? 0 : ring);
    float* red = D0 + 2 * ring;
    {   // the direction's constants, once, into scalar registers
        const PairCam& pc = cams[b * 2 + f];
        Cam c = make_cam(pc);
        float* cf = reinterpret_cast<float*>(&c);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Cam) / sizeof(float)); ++i) cf[i] = uni(cf[i]);
        v.cj = c;
        v.unit_k_s = uni(cams[b * 2 + k].unit) * (1.f / SWEEP_FX_ONE_F);
    }
    v.gbj = (unsigned)b * 2u * (unsigned)HW + (f ? (unsigned)HW : 0u);
    v.gbk = (unsigned)b * 2u * (unsigned)HW + (f ? 0u : (unsigned)HW);
    const PlanHeader* ph = reinterpret_cast<const PlanHeader*>(blob + (size_t)b * sh.stride + sh.plan_off);
    const PlanItem* __restrict__ items = reinterpret_cast<const PlanItem*>(ph + 1);
    const int n_items = ph->n_items;
    v.limit = uni(ph->limit);
    if (n_items <= 0 || ph->G != g.G || ph->R != g.R || ph->PXT != g.PXT) {
        // no plan (the planner's item backstop), or one made for another geometry (cd_debug_set_loss_sweep changed after the
        // blob was cached): this pair cannot be swept -- raise the degenerate flag, the guarded exact v1 pass recomputes the
        // gradient and the loss of the launch (loss_api.hip).  Uniform per workgroup: no barrier has been passed yet.
        if (threadIdx.x == 0) ovf->degenerate = 1;
        if ((threadIdx.x & (kFrameThreads - 1)) == 0) {
            float* o = partial + (size_t)(b * 2 + f) * 2;
            o[0] = o[1] = 0.f;
        }
        return;
    }
    DevEnv env{ovf, oidx, oval};
    const Lane<PXT> l = make_lane<PXT>(v, (int)threadIdx.x - f * kFrameThreads);
    Regs<PXT> r;
    init_regs<PXT>(r);
    for (int i = threadIdx.x; i < 2 * ring; i += kThreads) smem[i] = 0u;
    const int init_hi = init_stage_hi(g);
    for (int lo = 0; lo < init_hi; lo += kStagePasses * g.RP) {     // prologue: the initial window [0, R)
        const int hi = min(lo + kStagePasses * g.RP, init_hi);
        float sv[kStagePasses][PXT];
        load_stage<PXT>(v, l, lo, hi, sv);
        r.bad = !stage_rows<MODE, PXT>(v, l, lo, hi, sv) || r.bad;
    }
    // One barrier per item.  During item t three things run side by side, on disjoint ring rows by construction of the plan:
    //   rows [fl_lo, fl_hi) -- which no source of item t or later touches -- leave ring j (accumulator -> gradient row),
    //   rows [s_lo, s_hi) enter ring j in the slots just vacated (their depth was loaded during item t - 1),
    //   the source rows of item t are evaluated against rows staged BEFORE item t (Rec::nv).
    // Software pipeline: the depth rows entering at item t + 1 are loaded during item t (r.sv is free once item t's rows are
    // staged).
    // A row group is evaluated in one or two PASSES of RP rows (Geo::G, kGroupPasses); the flow / mask of a pass are loaded
    // while the pass before it is evaluated, through two register sets that alternate explicitly (no copies): pass 0 reads set
    // A (loaded during the previous item), pass 1 set B (loaded during pass 0).  The record of the next item is read one item
    // ahead: loading it at the top of its own item exposed the scalar-load latency (0.344 -> 0.374 ms at 256 pairs).
    // The frame's SERVICE wave (loss_sweep_core.h: the eighth wave of each frame has no source pixels at W = 224 and takes over the
    // rows that enter and leave the ring; compile-time geometry only -- the run-time-geometry build keeps every thread on its columns).
    constexpr bool SVC = SG == 1 && PXT == kStaticPXT;
    constexpr bool SRC_STAGES = SVC;       // the sources stage their own columns, the service wave only flushes (see below)
    if (SVC && (int)threadIdx.x - f * kFrameThreads >= g.RP * g.CG) {      // wave-uniform
        constexpr int NQ = kStaticSvcQuads;     // kStagePasses * RP rows of W / 4 quads (launch_sweep_inst checks it against the geometry)
        const int sl = (int)threadIdx.x - f * kFrameThreads - g.RP * g.CG;
        SvcRegs<NQ> q;
#pragma unroll
        for (int i = 0; i < NQ; ++i) q.v[i][0] = q.v[i][1] = q.v[i][2] = q.v[i][3] = 0.f;
        Rec me = items[0].f[f];
        __syncthreads();
        // Round 5: the rows ENTERING the rings are staged by the source waves again (two pixels per lane: their share of a frame's
        // rows costs a source lane ~12 instructions per item); the service wave keeps the rows LEAVING them.  With both on one wave the
        // service wave was the critical path of every item once the fast source pass had halved the sources' work: measured at 256
        // pairs, sources alone 0.181 ms, service waves alone 0.209 ms -- a single wave issues an instruction every ~4-5 cycles and
        // each of its quads is a latency chain (load -> exp -> LDS write; LDS read -> convert -> store).
        for (int it = 0; it < n_items; ++it) {
            if (!SRC_STAGES) r.bad = !svc_stage<MODE, NQ>(v, sl, me.s_lo, me.s_hi, q) || r.bad;
            const bool more = it + 1 < n_items;
            const Rec nx = items[more ? it + 1 : it].f[f];
            if (!SRC_STAGES) svc_load<NQ>(v, sl, nx.s_lo, more ? nx.s_hi : nx.s_lo, q);
            svc_flush<NQ>(v, sl, me.fl_lo, me.fl_hi);
            __syncthreads();
            me = nx;
        }
    } else {
    Inputs<PXT> inA, inB;
    const bool two = kGroupPasses > 1 && uni((int)(g.G > g.RP)) != 0;
    Rec me = items[0].f[f];
    int wk = items[0].f[k].w, nvk = items[0].f[k].nv;
    // the fast source pass (loss_sweep_core.h): compile-time geometry whose source waves have a row for every lane in every item
    constexpr bool FAST = SVC && PXT == 2 && kStaticH % 4 == 0;
    LaneF<PXT> lf;
    CamF cf;
    if constexpr (FAST) {
        lf = make_lanef<PXT>(v, l);
        cf = make_camf(v.cj);
        float* cff = reinterpret_cast<float*>(&cf);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(CamF) / sizeof(float)); ++i) cff[i] = uni(cff[i]);
        load_inputs_all<PXT>(v, lf, me.p > 0 ? me.p : 0, inA);
    } else load_inputs<PXT>(v, l, me.p, 0, inA);
    __syncthreads();
    if (!two) {
        // ONE pass per item (the default): the flow / mask of item t + 1 are requested at the TOP of item t and consumed one whole item
        // later, through two register sets that alternate explicitly (the loop is unrolled by two: no copies, and nothing the compiler
        // can fold back into "load right before the first use").  Round 4 issued them at the END of item t -- right before the barrier
        // -- and waited for them at the top of item t + 1: every wave of the workgroup then sat out one full memory latency per item
        // at the same time (all of them have just passed the barrier), with nothing left on the CU to cover it.
        auto item = [&](int it, const Inputs<PXT>& cur, Inputs<PXT>& nxt) {
            const bool more = it + 1 < n_items;
            const int nt = more ? it + 1 : it;
            const Rec nx = items[nt].f[f];
            const int nwk = items[nt].f[k].w, nnvk = items[nt].f[k].nv;
            if constexpr (FAST) load_inputs_all<PXT>(v, lf, nx.p > 0 ? nx.p : 0, nxt);      // (an item without a group, and the last one: row 0, unused)
            else load_inputs<PXT>(v, l, more ? nx.p : -1, 0, nxt);
            if (SRC_STAGES) {      // the depth rows entering now were requested during the previous item; request the next ones
                r.bad = !stage_rows<MODE, PXT>(v, l, me.s_lo, me.s_hi, r.sv) || r.bad;
                load_stage_nosel<PXT>(v, l, nx.s_lo, more ? nx.s_hi : nx.s_lo, r.sv);
            } else if (!SVC) {
                r.bad = !stage_rows<MODE, PXT>(v, l, me.s_lo, me.s_hi, r.sv) || r.bad;
                load_stage<PXT>(v, l, nx.s_lo, more ? nx.s_hi : nx.s_lo, r.sv);
                flush_rows<PXT>(v, l, me.fl_lo, me.fl_hi);
            }
            if constexpr (FAST) {
                if (me.p >= 0) process_rows_fast<MODE, REPROJ, PXT>(v, cf, env, r, l, lf, cur, me.p, wk, nvk);     // wave-uniform branch
            } else process_rows<MODE, REPROJ, PXT>(v, env, r, l, cur, me.p, 0, wk, nvk);
            __syncthreads();
            me = nx; wk = nwk; nvk = nnvk;
        };
        for (int it = 0; it < n_items; it += 2) {
            item(it, inA, inB);
            if (it + 1 < n_items) item(it + 1, inB, inA);
        }
    } else {
    for (int it = 0; it < n_items; ++it) {
        if (!SVC) r.bad = !stage_rows<MODE, PXT>(v, l, me.s_lo, me.s_hi, r.sv) || r.bad;
        const bool more = it + 1 < n_items;
        const int nt = more ? it + 1 : it;
        const Rec nx = items[nt].f[f];
        const int nwk = items[nt].f[k].w, nnvk = items[nt].f[k].nv;
        if (!SVC) load_stage<PXT>(v, l, nx.s_lo, more ? nx.s_hi : nx.s_lo, r.sv);
        load_inputs<PXT>(v, l, me.p, 1, inB);
        if (!SVC) flush_rows<PXT>(v, l, me.fl_lo, me.fl_hi);
        process_rows<MODE, REPROJ, PXT>(v, env, r, l, inA, me.p, 0, wk, nvk);
        load_inputs<PXT>(v, l, more ? nx.p : -1, 0, inA);
        process_rows<MODE, REPROJ, PXT>(v, env, r, l, inB, me.p, 1, wk, nvk);
        __syncthreads();
        me = nx; wk = nwk; nvk = nnvk;
    }
    }
    }
    if (env.any(r.bad) && (threadIdx.x & (kWave - 1)) == 0) env.degenerate();
    // loss partial sums: one (reprojection, disparity) pair per (pair, direction)
    float ar = wave_sum((float)r.acc_r), ad = wave_sum((float)r.acc_d);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) { red[wid * 2] = ar; red[wid * 2 + 1] = ad; }
    __syncthreads();
    if ((threadIdx.x & (kFrameThreads - 1)) == 0) {
        const int w0 = f * (kFrameThreads / kWave);
        float sr = 0.f, sd = 0.f;
        for (int i = 0; i < kFrameThreads / kWave; ++i) { sr += red[(w0 + i) * 2]; sd += red[(w0 + i) * 2 + 1]; }
        float* o = partial + (size_t)(b * 2 + f) * 2;
        o[0] = sr; o[1] = sd;
    }
}

struct SweepArgs {
    const float* depth; const float* ff; const float* fb; const float* mf; const float* mb;
    const PairCam* cams; const char* blob; float* partial; float* grad; Overflow* ovf; unsigned* oidx; float* oval;
    SweepShape sh;
};

template <int MODE, bool REPROJ, int PXT, int SG>
static int launch_sweep_sg(const SweepArgs& a, int B, size_t lds, hipStream_t s) {
    static bool configured = false;   // raise the dynamic-LDS limit of this instantiation once
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_sweep_kernel<MODE, REPROJ, PXT, SG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes) != hipSuccess)
            return CD_ERR_LAUNCH;
        configured = true;
    }
    hipLaunchKernelGGL((loss_sweep_kernel<MODE, REPROJ, PXT, SG>), dim3(B), dim3(kThreads), lds, s, a.depth, a.ff, a.fb, a.mf, a.mb,
                       a.cams, a.blob, a.partial, a.grad, a.ovf, a.oidx, a.oval, a.sh);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

static bool g_sweep_static_geo = true;   // CD_AMD_SWEEP_STATIC_GEO=0: always the run-time-geometry instantiation (A/B measurements)

template <int MODE, bool REPROJ, int PXT>
static int launch_sweep_inst(const SweepArgs& a, int B, size_t lds, hipStream_t s) {
    if (PXT == kStaticPXT && a.sh.g.H == kStaticH && a.sh.g.W == kStaticW && g_sweep_static_geo) {
        const Geo c = make_geo(kStaticH, kStaticW, kStaticPXT);
        if (memcmp(&c, &a.sh.g, sizeof(Geo)) == 0 && static_geometry_holds(c)) return launch_sweep_sg<MODE, REPROJ, PXT, PXT == kStaticPXT ? 1 : 0>(a, B, lds, s);
    }
    return launch_sweep_sg<MODE, REPROJ, PXT, 0>(a, B, lds, s);
}

template <int MODE, bool REPROJ>
static int launch_sweep_pxt(const SweepArgs& a, int B, size_t lds, hipStream_t s) {
    switch (a.sh.g.PXT) {
        case 1: return launch_sweep_inst<MODE, REPROJ, 1>(a, B, lds, s);
        case 2: return launch_sweep_inst<MODE, REPROJ, 2>(a, B, lds, s);
        case 4: return launch_sweep_inst<MODE, REPROJ, 4>(a, B, lds, s);
    }
    return CD_ERR_UNSUPPORTED;
}

template <int MODE>
static int launch_sweep_mode(bool reproj, const SweepArgs& a, int B, size_t lds, hipStream_t s) {
    return reproj ? launch_sweep_pxt<MODE, true>(a, B, lds, s) : launch_sweep_pxt<MODE, false>(a, B, lds, s);
}

bool sweep_supported(int H, int W) {
    const Geo g = sweep_geo(H, W);
    return g.ok != 0 && g.max_items <= kPlanItemsLds;
}

// The rule of the default dispatch (loss_api.hip): the sweep needs a pair per CU to fill the chip and a ring tall enough for
// the flow's vertical spread (R = 30 rows at W = 224; 17 at W = 384: the tile kernels keep those).
bool sweep_preferred(int B, int H, int W) {
    const Geo g = sweep_geo(H, W);
    return sweep_supported(H, W) && g.R >= 24 && B >= 96;
}

// Enqueues: overflow header reset, [before_main] sweep [after_main], overflow apply.  Partial sums: partial[(b*2+k)*2 + {0,1}].
static const bool g_sweep_env_read = [] {
    const char* e = getenv("CD_AMD_SWEEP_STATIC_GEO");
    if (e && e[0] == '0') g_sweep_static_geo = false;
    return true;
}();

int launch_sweep(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, const void* cams,
                 const void* blob, int mode, bool reproj, int B, int H, int W, float* partial, float* grad, void* ovf_mem,
                 int ovf_cap, hipStream_t s, void (*before_main)(hipStream_t), void (*after_main)(hipStream_t), const float* prep_intr,
                 const float* prep_extr, const float* prep_mask_sum, float lambda_r, float lambda_b) {
    const Geo g = sweep_geo(H, W);
    if (!sweep_supported(H, W)) return CD_ERR_UNSUPPORTED;
    Overflow* ovf = (Overflow*)ovf_mem;
    unsigned* oidx = (unsigned*)((char*)ovf_mem + 256);
    float* oval = (float*)(oidx + ovf_cap);
    if (hipMemsetAsync(ovf, 0, sizeof(Overflow), s) != hipSuccess) return CD_ERR_LAUNCH;
    if (hipMemsetD32Async((hipDeviceptr_t)&ovf->cap, ovf_cap, 1, s) != hipSuccess) return CD_ERR_LAUNCH;
    SweepArgs prm{depth, ff, fb, mf, mb, (const PairCam*)cams, (const char*)blob, partial, grad, ovf, oidx, oval,
                  SweepShape{g, pair_record_bytes(H, W), plan_offset(H, W)}};
    const size_t lds = ring_lds_bytes(g);
    const PairPrep pp{prep_intr, prep_extr, prep_mask_sum, lambda_r, lambda_b};     // prep_intr == nullptr: `cams` is already filled (prep_kernel)
    if (launch_sweep_units(depth, ff, fb, mf, mb, (PairCam*)const_cast<void*>(cams), mode, B, H, W, pp, s) != CD_OK) return CD_ERR_LAUNCH;
    if (before_main) before_main(s);
    int rc;
    if (mode == CD_DEPTH_EXP) rc = launch_sweep_mode<CD_DEPTH_EXP>(reproj, prm, B, lds, s);
    else if (mode == CD_DEPTH_RECIPROCAL) rc = launch_sweep_mode<CD_DEPTH_RECIPROCAL>(reproj, prm, B, lds, s);
    else rc = launch_sweep_mode<CD_DEPTH_IDENTITY>(reproj, prm, B, lds, s);
    if (after_main) after_main(s);
    if (rc != CD_OK) return rc;
    launch_overflow_apply(ovf_mem, ovf_cap, grad, s);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

}  // namespace cd
