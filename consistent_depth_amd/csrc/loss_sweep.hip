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
#include "loss_v1_pixel.h"

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
    // the fingerprint of the flows / masks this plan belongs to (loss_sweep_core.h, plan_fingerprint_term)
    if (threadIdx.x == 0) hi_i[0] = 0;
    __syncthreads();
    {
        unsigned h = 0u;
        for (int i = threadIdx.x; i < 2 * kUnitGrid * kUnitGrid; i += kBlock) {
            const int j = i / (kUnitGrid * kUnitGrid);
            int sx, sy;
            unit_sample_xy(g.H, g.W, i - j * kUnitGrid * kUnitGrid, &sx, &sy);
            const float* fl = (j == 0 ? flow_fwd : flow_bwd) + (size_t)b * 2 * HW;
            const float* mk = (j == 0 ? mask_fwd : mask_bwd) + (size_t)b * HW;
            const int p = sy * g.W + sx;
            h += plan_fingerprint_term(i, mk[p], fl[p], fl[HW + p]);
        }
        atomicAdd(reinterpret_cast<unsigned*>(&hi_i[0]), h);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int fan_in = 0;
        for (int i = 0; i < kBlock; ++i) fan_in = lo_i[i] > fan_in ? lo_i[i] : fan_in;
        PlanHeader* ph = reinterpret_cast<PlanHeader*>(blob + (size_t)b * stride + plan_off);
        int n = plan_items(g, lo_s, hi_s, suf, items);
        if (n > 0 && fan_in > SWEEP_MAX_FAN_IN) n = -3;     // a 32-bit accumulator could wrap: no plan, the exact fallback path
        if (n > 0) expand_plan(g, items, n, reinterpret_cast<PlanItem*>(ph + 1), lo_s, hi_s);
        ph->n_items = n; ph->G = g.G; ph->R = g.R; ph->PXT = g.PXT;
        ph->fan_in = fan_in; ph->limit = sweep_limit_scaled(fan_in); ph->fingerprint = (unsigned)hi_i[0]; ph->pad = 0;
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

// ---------------------------------------------------------------- per-pair constants and accumulator units
// Computed by the pair's own workgroup in the sweep kernel's prologue (pair_constants below): round 2-4 launched them in front
// (prep_kernel, then sweep_units_kernel: 18.6 us of latency chains per gradient call) -- the same code on the same inputs (fbar by
// the same sum over the same 256 threads, the unit samples summed in index order by one thread: identical bits).
struct PairPrep { const float* intr; const float* extr; const float* mask_sum; float lambda_r, lambda_b; };

// ---------------------------------------------------------------- the sweep
// Round 5: the call is ONE kernel.  A workgroup owns its pair from the per-pair constants to the pair's loss: nothing it needs is
// produced by another workgroup, so
//   * its overflow entries live in the pair's own SEGMENT of the list (count in LDS, no global atomic per push) and are applied by the
//     workgroup itself after its last row has left the rings;
//   * a pair that cannot be swept (no plan, a degenerate depth, a full segment) is recomputed by its own workgroup in the EXACT mode
//     (exact_pair: the v1 per-pixel body with global atomics on the pair's two planes) -- no launch-wide fallback pass;
//   * the pair's reprojection / disparity loss is written by the workgroup, the batch mean by whichever workgroup finishes last.
// Round 4's call was 2 memsets + 7 kernels (units, sweep, overflow apply, guarded zero, guarded v1, two finalize kernels): 319 us
// for 276 us of sweep (profiles/rocprofv3_loss_sweep_b256_r04.txt).
struct WgState {           // in LDS (the `red` scratch behind the rings)
    PairCam cam[2];
    float wave_part[2 * kThreads / kWave];
    unsigned ovf_n;        // pushes attempted by this workgroup
    int redo;              // the pair must be recomputed in the exact mode
    int is_last;
    float fbar[2];
    float red4[8];
    float prep_in[34];     // the pair's intrinsics [2][4], extrinsics [2][3][4], mask sums [2] (pair_constants)
    unsigned fingerprint;  // of the flows / masks of THIS call at the unit-sample positions (cf. PlanHeader::fingerprint)
};
static_assert(sizeof(WgState) <= kLdsReserve, "WgState must fit the LDS reserve behind the rings");

struct DevEnv {
    WgState* st; unsigned* oidx; float* oval; int seg_cap;     // oidx / oval: the pair's segment
    __device__ __forceinline__ static void add32(unsigned* p, int v) { atomicAdd(p, (unsigned)v); }   // ds_add_u32, no return
    __device__ __forceinline__ static bool any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
    __device__ __forceinline__ static unsigned long long vote(bool x) { return __builtin_amdgcn_ballot_w64(x); }   // lane mask (scalar)
    __device__ __forceinline__ static bool any_vote(unsigned long long m) { return m != 0ull; }
    // wave-aggregated append: ONE returning LDS atomic per wave and call; all lanes of the wave must call it (convergent)
    __device__ __forceinline__ void push(bool need, unsigned idx, float v) {
        const unsigned long long mask = __ballot(need);
        if (mask == 0ull) return;  // wave-uniform
        const int lane = threadIdx.x & (kWave - 1);
        const int leader = __ffsll((long long)mask) - 1;
        unsigned base = 0u;
        if (lane == leader) base = atomicAdd(&st->ovf_n, (unsigned)__popcll(mask));
        base = (unsigned)__shfl((int)base, leader, kWave);
        if (need) {
            const unsigned i = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            if (i < (unsigned)seg_cap) { oidx[i] = idx; oval[i] = v; }
            else st->redo = 1;
        }
    }
    __device__ __forceinline__ void degenerate() { st->redo = 1; }
};

struct SweepShape { Geo g; size_t stride, plan_off; };

__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// SG: 0 = the geometry arrives at run time (kernel argument); 1 = the BASELINE shape 384 x 224 at 2 pixels per thread as
// COMPILE-TIME constants: ring strides, rows per pass, image size fold into immediates (the kernel is short of scalar registers:
// ~30 wave-uniform camera constants, 10 pointers and the plan records live next to them) -- same code, same results;
// 2 = 1 with non-temporal loads of the depth rows (round 6: launches of more than ~290 pairs, launch_sweep_inst).
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

struct SweepOut { float* reproj; float* disp; float* total; WorkspaceHeader* hdr; };     // per-pair losses [B], batch mean [1], the workspace header (finished-pairs counter)

// fbar (batch mean of (fx + fy) / 2 of the ref frames, consistency_loss.py:178), the pair's PairCam[2] (prep_pair) and the accumulator
// units (units_from_samples) -> st.cam; scratch: 24 floats of LDS.  All kThreads threads call both parts.
// Part 1, at the very top of the kernel -- BEFORE the first window's depth rows are requested: every global read whose address is
// known at entry (the unit samples' own pixels, the batch's focal lengths, the pair's intrinsics / extrinsics / mask sums).  Measured
// (tools/exp/sweep_times.py): requested behind the window's 57 KB, the samples' two dependent latencies (flow -> tap positions ->
// sampled depths; `vmcnt` retires in order) ended 10 us after kernel entry.
struct PairReq { UnitLoads ul; float f0[2], f1[2], pin; int sx, sy; };
__device__ __forceinline__ PairReq pair_constants_request(const PairPrep& pp, const float* __restrict__ depth_p, const float* __restrict__ ff,
                                                          const float* __restrict__ fb, const float* __restrict__ mf,
                                                          const float* __restrict__ mb, int b, int B, int H, int W) {
    constexpr int NS = kUnitGrid * kUnitGrid;
    const int t = threadIdx.x, HW = H * W;
    PairReq q{};
    if (t < 2 * NS) {
        const int j = t / NS;
        unit_sample_xy(H, W, t % NS, &q.sx, &q.sy);
        unit_sample_load_own(q.ul, depth_p + (j ? HW : 0), depth_p + (j ? 0 : HW), j ? fb : ff, j ? mb : mf, H, W, q.sx, q.sy);
    }
    if (t < kBlock && t < B)
        for (int k = 0; k < 2; ++k) { q.f0[k] = pp.intr[(t * 2 + k) * 4 + 0]; q.f1[k] = pp.intr[(t * 2 + k) * 4 + 1]; }
    if (t >= 2 * NS && t < 2 * NS + 34) {      // intr [8], extr [24], mask sums [2] of the pair
        const int i = t - 2 * NS;
        q.pin = i < 8 ? pp.intr[b * 8 + i] : (i < 32 ? pp.extr[b * 24 + (i - 8)] : pp.mask_sum[b * 2 + (i - 32)]);
    }
    return q;
}
// Part 2: the arithmetic (round 6: no second read step -- the unit samples need one memory latency, loss_sweep_core.h).
template <int MODE>
__device__ __forceinline__ void pair_constants(WgState& st, float* scratch, const PairPrep& pp, PairReq& q, const float* __restrict__ depth_p,
                                               int B, int H, int W) {
    constexpr int NS = kUnitGrid * kUnitGrid;
    static_assert(NS == kBlock, "fbar is reduced like prep_kernel does: 256 threads, 4 wave sums added in order");
    const int t = threadIdx.x, lane = t & (kWave - 1), wid = t / kWave;
    const int HW = H * W;
    if (t >= 2 * NS && t < 2 * NS + 34) st.prep_in[t - 2 * NS] = q.pin;
    const float* f0 = q.f0; const float* f1 = q.f1;
    const UnitLoads& ul = q.ul;
    const int sx = q.sx, sy = q.sy;
    float acc[2] = {0.f, 0.f};
    if (t < kBlock)
        for (int k = 0; k < 2; ++k) {
            acc[k] += f0[k] + f1[k];
            for (int bb = t + kBlock; bb < B; bb += kBlock) acc[k] += pp.intr[(bb * 2 + k) * 4 + 0] + pp.intr[(bb * 2 + k) * 4 + 1];
        }
    acc[0] = wave_sum(acc[0]); acc[1] = wave_sum(acc[1]);
    if (t < kBlock && lane == 0) { st.red4[wid] = acc[0]; st.red4[4 + wid] = acc[1]; }
    __syncthreads();
    if (t == 0) {
        for (int k = 0; k < 2; ++k)
            st.fbar[k] = (((0.f + st.red4[4 * k]) + st.red4[4 * k + 1]) + st.red4[4 * k + 2] + st.red4[4 * k + 3]) / (2.f * (float)B);
        prep_pair(st.prep_in, st.prep_in + 8, st.prep_in + 32, st.fbar, pp.lambda_r, pp.lambda_b, B, H, W, st.cam);
    }
    __syncthreads();
    // The unit samples: 256 per direction, one per thread of waves 0..7; their sums by wave shuffles, then the 4 wave sums of a
    // direction in wave order by one thread -- a fixed order (bit-reproducible; the host emulation sums in index order: the estimate
    // is floored to a power of two, any estimate gives a correct gradient).
    UnitSample u;
    u.direct = u.scatter = 0.f; u.valid = 0;
    if (t < 2 * NS) u = unit_sample_eval<MODE>(st.cam[t / NS], ul, sx, sy);
    const float wd = wave_sum(u.direct), ws = wave_sum(u.scatter), wn = wave_sum((float)u.valid);
    // the call's fingerprint (plan_fingerprint_term over the same 2 x 256 samples; waves 0..7 hold them): an integer sum, any order
    unsigned fpw = t < 2 * NS ? plan_fingerprint_term(t, ul.m, ul.fx, ul.fy) : 0u;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) fpw += (unsigned)__shfl_xor((int)fpw, o, kWave);
    if (t < 2 * NS && lane == 0) {
        scratch[wid * 3] = wd; scratch[wid * 3 + 1] = ws; scratch[wid * 3 + 2] = wn;
        reinterpret_cast<unsigned*>(scratch)[3 * (2 * NS / kWave) + wid] = fpw;
    }
    __syncthreads();
    if (t == 0) {
        unsigned fp = 0u;
        for (int w = 0; w < 2 * NS / kWave; ++w) fp += reinterpret_cast<const unsigned*>(scratch)[3 * (2 * NS / kWave) + w];
        st.fingerprint = fp;
        float D[2] = {0.f, 0.f}, S[2] = {0.f, 0.f};
        int n[2] = {0, 0};
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < NS / kWave; ++i) {
                const int w = j * (NS / kWave) + i;
                D[j] += scratch[w * 3]; S[j] += scratch[w * 3 + 1]; n[j] += (int)scratch[w * 3 + 2];
            }
        units_from_samples(st.cam, D, S, n);
    }
    __syncthreads();
}

// The EXACT mode of one pair, by its own workgroup: both gradient planes are zeroed and every source goes through the v1 per-pixel
// body (global atomics, ~4 % of the HBM roofline: for the pairs the sweep cannot take -- no plan, a depth that is not a positive
// finite number, a full overflow segment).  Returns the loss partial sums of direction f in (sr, sd) on thread f * kFrameThreads.
template <int MODE, bool REPROJ>
__device__ __forceinline__ void exact_pair(WgState& st, const float* __restrict__ dpair, const float* __restrict__ ffp, const float* __restrict__ fbp,
                                           const float* __restrict__ mfp, const float* __restrict__ mbp, float* gpair, int H, int W,
                                           float& sr, float& sd) {
    const int HW = H * W, t = threadIdx.x;
    float4* g4 = reinterpret_cast<float4*>(gpair);       // (planes are 16-byte aligned and HW * 4 % 16 == 0: run_loss's sweep_aligned)
    for (int i = t; i < 2 * HW / 4; i += kThreads) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();       // (waits for the stores' acknowledgements: the atomics below reach the same L2 lines after them)
    const int f = t / kFrameThreads, lt = t - f * kFrameThreads;
    const PairCam cam = st.cam[f];
    const float* v_ref = dpair + (f ? HW : 0); const float* v_tgt = dpair + (f ? 0 : HW);
    const float* fl = f ? fbp : ffp; const float* mk = f ? mbp : mfp;
    float* g_ref = gpair + (f ? HW : 0); float* g_tgt = gpair + (f ? 0 : HW);
    float acc_r = 0.f, acc_d = 0.f;
    for (int p = lt; p < HW; p += kFrameThreads) {
        const int y = p / W, x = p - y * W;
        v1_pixel<true, MODE, REPROJ, true>(cam, v_tgt, v_ref[p], fl[p], fl[HW + p], mk[p], (float)x, (float)y, p, H, W, g_ref, g_tgt, acc_r, acc_d);
    }
    acc_r = wave_sum(acc_r); acc_d = wave_sum(acc_d);
    const int lane = t & (kWave - 1), wid = t / kWave;
    if (lane == 0) { st.wave_part[wid * 2] = acc_r; st.wave_part[wid * 2 + 1] = acc_d; }
    __syncthreads();
    sr = sd = 0.f;
    if (lt == 0) {
        const int w0 = f * (kFrameThreads / kWave);
        for (int i = 0; i < kFrameThreads / kWave; ++i) { sr += st.wave_part[(w0 + i) * 2]; sd += st.wave_part[(w0 + i) * 2 + 1]; }
    }
}

template <int MODE, bool REPROJ, int PXT, int SG>
__global__ __launch_bounds__(kThreads) void loss_sweep_kernel(
    const float* __restrict__ depth, const float* __restrict__ ff, const float* __restrict__ fb, const float* __restrict__ mf,
    const float* __restrict__ mb, PairCam* __restrict__ cams, const char* __restrict__ blob, float* __restrict__ grad,
    unsigned* __restrict__ oidx_all, float* __restrict__ oval_all, int seg_cap, const SweepShape sh, const PairPrep pp, int B,
    const SweepOut out) {
    // [2][ring] depths, [2][ring] accumulators, WgState.  Depths FIRST: a source's four depth taps are ds_read2_b32 (offsets of at most
    // 255 words: {0, 1} and {RW, RW + 1} fit), its accumulator adds ds_add_u32 with 16-bit byte offsets -- so ONE address register,
    // the tap's depth address, serves all eight (accumulators first cost a v_add per read pair)
    extern __shared__ __align__(16) unsigned smem[];
    const Geo g = SG >= 1 ? make_geo(kStaticH, kStaticW, kStaticPXT) : sh.g;
    constexpr bool NTD = SG == 2;        // non-temporal depth rows (launches whose depth planes cannot stay in the Infinity Cache)
    const int b = blockIdx.x, HW = g.H * g.W, ring = ring_rows(g) * g.RW;
    const int f = uni((int)(threadIdx.x / kFrameThreads)), k = 1 - f;   // whole waves serve one frame: everything derived from f is scalar
    View v;
    v.H = g.H; v.W = g.W; v.R = g.R; v.RW = g.RW; v.RP = g.RP; v.G = g.G; v.CG = g.CG; v.HW = (unsigned)HW;
    const float* dpair = depth + (size_t)b * 2 * HW;
    v.vj = dpair + (f ? HW : 0); v.vk = dpair + (f ? 0 : HW);
    v.flj = (f ? fb : ff) + (size_t)b * 2 * HW;
    v.mkj = (f ? mb : mf) + (size_t)b * HW;
    v.gradj = grad + (size_t)b * 2 * HW + (f ? HW : 0);
    float* D0 = reinterpret_cast<float*>(smem);
    unsigned* A0 = smem + 2 * ring;
    v.Aj = A0 + (f ? ring : 0); v.Ak = A0 + (f ? 0 : ring);
    v.Dj = D0 + (f ? ring : 0); v.Dk = D0 + (f ? 0 : ring);
    WgState& st = *reinterpret_cast<WgState*>(smem + 4 * ring);
    if (threadIdx.x == 0) { st.ovf_n = 0u; st.redo = 0; st.is_last = 0; }
    // ---- the pair constants' reads first (PairReq), then the accumulators are cleared and the depth rows of the first window requested
    // -- all of it BEFORE anything is computed: the latencies overlap
    PairReq req = pair_constants_request(pp, dpair, ff + (size_t)b * 2 * HW, fb + (size_t)b * 2 * HW, mf + (size_t)b * HW, mb + (size_t)b * HW,
                                         b, B, g.H, g.W);
    constexpr int kInitBatches = 4;
    v.cj = Cam{};
    const Lane<PXT> l0 = make_lane<PXT>(v, (int)threadIdx.x - f * kFrameThreads);      // (only its row / column fields are used here)
    for (int i = threadIdx.x; i < 2 * ring; i += kThreads) A0[i] = 0u;
    const int init_hi = init_stage_hi(g);
    const bool init_batched = init_hi <= kInitBatches * kStagePasses * g.RP;
    float sv0[kInitBatches][kStagePasses][PXT];
#pragma unroll
    for (int j = 0; j < kInitBatches; ++j) {
        const int lo = j * kStagePasses * g.RP;
        load_stage_nosel<PXT, NTD>(v, l0, lo < init_hi ? lo : 0, init_batched ? min(lo + kStagePasses * g.RP, init_hi) : 0, sv0[j]);
    }
    // ... and the flow / mask of the first item's source rows (fast pass; consumed at the top of the loop, ~15 us from here)
    const PlanHeader* ph = reinterpret_cast<const PlanHeader*>(blob + (size_t)b * sh.stride + sh.plan_off);
    const PlanItem* __restrict__ items = reinterpret_cast<const PlanItem*>(ph + 1);
    Inputs<PXT> in0;
    {
        const int p0 = ph->n_items > 0 ? uni(items[0].f[f].p) : 0;
        // (lanes without source pixels -- the service wave -- read the group's first pixels: in bounds, never used)
        load_inputs_goff<PXT>(v, l0.on ? (l0.rrW + l0.x0) << 2 : 0u, min(max(p0, 0), g.H - 1), in0);
    }
    // ---- the pair's constants
    pair_constants<MODE>(st, st.wave_part, pp, req, dpair, B, g.H, g.W);
    if (threadIdx.x < 2 * (int)(sizeof(PairCam) / sizeof(float)))      // (kept in the workspace: debugging, the tile kernels' format)
        reinterpret_cast<float*>(cams + b * 2)[threadIdx.x] = reinterpret_cast<const float*>(st.cam)[threadIdx.x];
    {   // the direction's constants, once, into scalar registers
        Cam c = make_cam(st.cam[f]);
        float* cf = reinterpret_cast<float*>(&c);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Cam) / sizeof(float)); ++i) cf[i] = uni(cf[i]);
        v.cj = c;
        v.unit_k_s = uni(st.cam[k].unit) * (1.f / SWEEP_FX_ONE_F);
    }
    v.gbj = (unsigned)b * 2u * (unsigned)HW + (f ? (unsigned)HW : 0u);
    v.gbk = (unsigned)b * 2u * (unsigned)HW + (f ? 0u : (unsigned)HW);
    const int n_items = ph->n_items;
    v.limit = uni(ph->limit);
    // no plan (the planner's item backstop, a fan-in beyond the accumulators' range), or one made for another geometry
    // (cd_debug_set_loss_sweep changed after the blob was cached): this pair cannot be swept -> the exact mode below.  Workgroup-uniform.
    const bool has_plan = !(n_items <= 0 || ph->G != g.G || ph->R != g.R || ph->PXT != g.PXT);
    // ... or one made from OTHER flows / masks (a stale or foreign tile_windows blob): its items still describe a valid order of the
    // rows, but their `inw` promises are not believed -- the general source pass clamps and votes (round 5's build computed silently
    // wrong gradients there).  Wave-uniform.
    const bool trust_inw = uni((int)(ph->fingerprint == st.fingerprint)) != 0;
    unsigned* oidx = oidx_all + (size_t)b * seg_cap;
    float* oval = oval_all + (size_t)b * seg_cap;
    __syncthreads();       // (pair_constants' scratch is free: the rings can be cleared)
    Regs<PXT> r;
    init_regs<PXT>(r);
    if (has_plan) {
    DevEnv env{&st, oidx, oval, seg_cap};
    const Lane<PXT> l = make_lane<PXT>(v, (int)threadIdx.x - f * kFrameThreads);
    if (init_batched) {                  // prologue: the initial window [0, R), from the rows requested at the top of the kernel
#pragma unroll
        for (int j = 0; j < kInitBatches; ++j) {
            const int lo = j * kStagePasses * g.RP;
            if (lo < init_hi) r.bad = !stage_rows<MODE, PXT>(v, l, lo, min(lo + kStagePasses * g.RP, init_hi), sv0[j]) || r.bad;
        }
    } else {
        for (int lo = 0; lo < init_hi; lo += kStagePasses * g.RP) {
            const int hi = min(lo + kStagePasses * g.RP, init_hi);
            float sv[kStagePasses][PXT];
            load_stage<PXT>(v, l, lo, hi, sv);
            r.bad = !stage_rows<MODE, PXT>(v, l, lo, hi, sv) || r.bad;
        }
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
    constexpr bool SVC = SG >= 1 && PXT == kStaticPXT;
#ifndef CD_SWEEP_SRC_STAGES
#define CD_SWEEP_SRC_STAGES 1
#endif
    constexpr bool SRC_STAGES = SVC && CD_SWEEP_SRC_STAGES != 0;       // the sources stage their own columns, the service wave only flushes (see below)
    if (SVC && (int)threadIdx.x - f * kFrameThreads >= g.RP * g.CG) {      // wave-uniform
        constexpr int NQ = kStaticSvcQuads;     // kStagePasses * RP rows of W / 4 quads (launch_sweep_inst checks it against the geometry)
        const int sl = (int)threadIdx.x - f * kFrameThreads - g.RP * g.CG;
        SvcRegs<NQ> q;
        const SvcLane<NQ> slc = make_svc_lane<NQ>(v, sl);
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
            if (!SRC_STAGES) r.bad = !svc_stage<MODE, NQ>(v, slc, sl, me.s_lo, me.s_hi, q) || r.bad;
            const bool more = it + 1 < n_items;
            const Rec nx = items[more ? it + 1 : it].f[f];
            if (!SRC_STAGES) svc_load<NQ>(v, slc, sl, nx.s_lo, more ? nx.s_hi : nx.s_lo, q);
            svc_flush<NQ>(v, slc, sl, me.fl_lo, me.fl_hi);
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
        inA = in0;       // (requested at the top of the kernel)
    } else load_inputs<PXT>(v, l, me.p, 0, inA);
    __syncthreads();
    if (!two) {
        // ONE pass per item (the default): the flow / mask of item t + 1 are requested at the TOP of item t and consumed one whole item
        // later, through two register sets that alternate explicitly (the loop is unrolled by two: no copies, and nothing the compiler
        // can fold back into "load right before the first use").  Round 4 issued them at the END of item t -- right before the barrier
        // -- and waited for them at the top of item t + 1: every wave of the workgroup then sat out one full memory latency per item
        // at the same time (all of them have just passed the barrier), with nothing left on the CU to cover it.
        // The plan records are scalar loads: the record of item t + 2 is requested at the top of item t and first used at the top of
        // item t + 1 (the loads of item t + 1's inputs need its row right away: a record read at the top of its own use exposed a
        // scalar-cache latency per item on every wave at once).
        Rec nx = items[n_items > 1 ? 1 : 0].f[f];
        int nwk = items[n_items > 1 ? 1 : 0].f[k].w, nnvk = items[n_items > 1 ? 1 : 0].f[k].nv;
        auto item = [&](int it, const Inputs<PXT>& cur, Inputs<PXT>& nxt) {
            const bool more = it + 1 < n_items;
            const int t2 = it + 2 < n_items ? it + 2 : n_items - 1;
            const Rec n2 = items[t2].f[f];
            const int n2wk = items[t2].f[k].w, n2nvk = items[t2].f[k].nv;
            if constexpr (FAST) load_inputs_all<PXT>(v, lf, nx.p > 0 ? nx.p : 0, nxt);      // (an item without a group, and the last one: row 0, unused)
            else load_inputs<PXT>(v, l, more ? nx.p : -1, 0, nxt);
            if (SRC_STAGES) {      // the depth rows entering now were requested during the previous item; request the next ones
                r.bad = !stage_rows<MODE, PXT, false>(v, l, me.s_lo, me.s_hi, r.sv) || r.bad;     // (pad columns: written by the prologue, constant)
                load_stage_nosel<PXT, NTD>(v, l, nx.s_lo, more ? nx.s_hi : nx.s_lo, r.sv);
            } else if (!SVC) {
                r.bad = !stage_rows<MODE, PXT>(v, l, me.s_lo, me.s_hi, r.sv) || r.bad;
                load_stage<PXT>(v, l, nx.s_lo, more ? nx.s_hi : nx.s_lo, r.sv);
                flush_rows<PXT>(v, l, me.fl_lo, me.fl_hi);
            }
            if constexpr (FAST) {
                if (me.p >= 0) process_rows_fast<MODE, REPROJ, PXT>(v, cf, env, r, l, lf, cur, me.p, wk, nvk, me.inw != 0 && trust_inw);     // wave-uniform branch
            } else process_rows<MODE, REPROJ, PXT>(v, env, r, l, cur, me.p, 0, wk, nvk);
            __syncthreads();
            me = nx; wk = nwk; nvk = nnvk;
            nx = n2; nwk = n2wk; nnvk = n2nvk;
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
    // loss partial sums: one (reprojection, disparity) pair per wave
    {
        const float ar = wave_sum((float)loss_sum_r<PXT>(r)), ad = wave_sum((float)loss_sum_d<PXT>(r));
        const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
        if (lane == 0) { st.wave_part[wid * 2] = ar; st.wave_part[wid * 2 + 1] = ad; }
    }
    }                      // has_plan
    // (No fence: the gradient rows this workgroup stored and the atomics it may add to them below go through the same L1 -> L2 path of
    // this CU, and the barrier waits for the stores' acknowledgements.  An agent-scope fence here writes back the XCD's whole L2 --
    // megabytes of freshly written gradient -- and cost ~100 us per workgroup when every thread issued one: 0.232 -> 0.332 ms.)
    __syncthreads();
    float sr = 0.f, sd = 0.f;       // the pair's loss partial sums of direction f, on thread f * kFrameThreads
    if (!has_plan || st.redo != 0) {        // workgroup-uniform (read after the barrier)
        __syncthreads();
        exact_pair<MODE, REPROJ>(st, dpair, ff + (size_t)b * 2 * HW, fb + (size_t)b * 2 * HW, mf + (size_t)b * HW, mb + (size_t)b * HW,
                                 grad + (size_t)b * 2 * HW, g.H, g.W, sr, sd);
    } else {
        // the pair's overflow entries (taps outside the rings, values beyond the fixed-point range): written by this workgroup, applied by it
        const unsigned n = st.ovf_n;
        for (unsigned i = threadIdx.x; i < n; i += kThreads) atomic_add_f32(grad + oidx[i], oval[i]);
        if ((threadIdx.x & (kFrameThreads - 1)) == 0) {
            const int w0 = f * (kFrameThreads / kWave);
            for (int i = 0; i < kFrameThreads / kWave; ++i) { sr += st.wave_part[(w0 + i) * 2]; sd += st.wave_part[(w0 + i) * 2 + 1]; }
        }
    }
    // ---- the pair's losses (what finalize_pairs_kernel does for the tile kernels: fixed order, fp64), then the batch mean by the
    // workgroup that finishes last (finalize_total_kernel's order: 256 strided fp64 sums, LDS tree)
    if ((threadIdx.x & (kFrameThreads - 1)) == 0) { st.red4[f * 2] = sr; st.red4[f * 2 + 1] = sd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double rr[2], qq[2];
        for (int kk = 0; kk < 2; ++kk) {
            rr[kk] = (double)st.red4[kk * 2] * (double)st.cam[kk].invS;
            qq[kk] = (double)st.cam[kk].fbar * ((double)st.red4[kk * 2 + 1] * (double)st.cam[kk].invS);
        }
        // Hand-off to the workgroup that finishes last, WITHOUT fences (a release fence writes back every dirty line of the XCD's L2):
        // write-through (sc1) stores of the two numbers, drained, then the device-scope counter; the reader uses sc1 loads
        // (MI355X_MICROARCH.md, inter-workgroup visibility: "sc0 sc1 stores and loads both sides").
        __hip_atomic_store(out.reproj + b, pp.lambda_r > 0.f ? (float)((double)pp.lambda_r * (rr[0] + rr[1]) * 0.5) : 0.f, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(out.disp + b, pp.lambda_b > 0.f ? (float)((double)pp.lambda_b * (qq[0] + qq[1]) * 0.5) : 0.f, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // The finished-pairs counter lives in the workspace header and is zero BETWEEN calls: cd_consistency_loss_workspace_init zeroes it
        // once, the workgroup that finishes last puts it back to zero (round 6: the per-call 4-byte fill was a dispatch of its own,
        // 5-10 us in front of a 200 us kernel).  A workspace that was never initialised carries no magic word: every workgroup then
        // reports a NaN mean loss instead of leaving a stale one behind.
        const unsigned magic = __hip_atomic_load(&out.hdr->magic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st.is_last = __hip_atomic_fetch_add(&out.hdr->finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(B - 1) ? 1 : 0;
        if (magic != kWorkspaceMagic) { out.total[0] = __builtin_nanf(""); st.is_last = 0; }
    }
    __syncthreads();
    if (st.is_last != 0) {
        double* ld = reinterpret_cast<double*>(smem);
        if (threadIdx.x < kBlock) {
            double acc = 0.0;
            for (int bb = threadIdx.x; bb < B; bb += kBlock)
                acc += (double)__hip_atomic_load(out.reproj + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                       (double)__hip_atomic_load(out.disp + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ld[threadIdx.x] = acc;
        }
        __syncthreads();
        for (int s2 = kBlock / 2; s2 > 0; s2 >>= 1) {
            if ((int)threadIdx.x < s2) ld[threadIdx.x] += ld[threadIdx.x + s2];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out.total[0] = (float)(ld[0] / (double)B);
            __hip_atomic_store(&out.hdr->finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // every workgroup of this launch has counted: ready for the next call
        }
    }
}

struct SweepArgs {
    const float* depth; const float* ff; const float* fb; const float* mf; const float* mb;
    PairCam* cams; const char* blob; float* grad; unsigned* oidx; float* oval; int seg_cap;
    SweepShape sh; PairPrep pp; int B; SweepOut out;
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
                       a.cams, a.blob, a.grad, a.oidx, a.oval, a.seg_cap, a.sh, a.pp, a.B, a.out);
    return hipGetLastError() == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

static bool g_sweep_static_geo = true;   // CD_AMD_SWEEP_STATIC_GEO=0: always the run-time-geometry instantiation (A/B measurements)

template <int MODE, bool REPROJ, int PXT>
static int launch_sweep_inst(const SweepArgs& a, int B, size_t lds, hipStream_t s) {
    if (PXT == kStaticPXT && a.sh.g.H == kStaticH && a.sh.g.W == kStaticW && g_sweep_static_geo) {
        const Geo c = make_geo(kStaticH, kStaticW, kStaticPXT);
        if (memcmp(&c, &a.sh.g, sizeof(Geo)) == 0 && static_geometry_holds(c)) {
            // depth rows: default cache policy while the launch's depth planes (2 per pair) fit the 256 MB Infinity Cache with room to
            // spare (a 256-pair call: 176 MB), non-temporal beyond (loss_sweep_core.h, CD_SWEEP_NT)
            if ((size_t)B * 2 * kStaticH * kStaticW * sizeof(float) > (size_t)200 << 20)
                return launch_sweep_sg<MODE, REPROJ, PXT, PXT == kStaticPXT ? 2 : 0>(a, B, lds, s);
            return launch_sweep_sg<MODE, REPROJ, PXT, PXT == kStaticPXT ? 1 : 0>(a, B, lds, s);
        }
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

// Enqueues: [before] reset of the finished-pairs counter, the sweep [after].  The per-pair losses, their batch mean and the whole gradient
// are complete when the kernel is: nothing follows it (loss_api.hip).
static const bool g_sweep_env_read = [] {
    const char* e = getenv("CD_AMD_SWEEP_STATIC_GEO");
    if (e && e[0] == '0') g_sweep_static_geo = false;
    return true;
}();

int launch_sweep(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, void* cams,
                 const void* blob, int mode, bool reproj, int B, int H, int W, float* grad, void* ovf_mem, int ovf_cap, hipStream_t s,
                 void (*before)(hipStream_t), void (*after)(hipStream_t), const float* intr, const float* extr, const float* mask_sum,
                 float lambda_r, float lambda_b, float* reproj_out, float* disp_out, float* total_out, WorkspaceHeader* hdr) {
    const Geo g = sweep_geo(H, W);
    if (!sweep_supported(H, W) || !intr || !extr || !mask_sum) return CD_ERR_UNSUPPORTED;
    unsigned* oidx = (unsigned*)((char*)ovf_mem + 256);
    float* oval = (float*)(oidx + ovf_cap);
    if (before) before(s);
    SweepArgs prm{depth, ff, fb, mf, mb, (PairCam*)cams, (const char*)blob, grad, oidx, oval, ovf_cap / B,
                  SweepShape{g, pair_record_bytes(H, W), plan_offset(H, W)}, PairPrep{intr, extr, mask_sum, lambda_r, lambda_b}, B,
                  SweepOut{reproj_out, disp_out, total_out, hdr}};
    const size_t lds = ring_lds_bytes(g);
    int rc;
    if (mode == CD_DEPTH_EXP) rc = launch_sweep_mode<CD_DEPTH_EXP>(reproj, prm, B, lds, s);
    else if (mode == CD_DEPTH_RECIPROCAL) rc = launch_sweep_mode<CD_DEPTH_RECIPROCAL>(reproj, prm, B, lds, s);
    else rc = launch_sweep_mode<CD_DEPTH_IDENTITY>(reproj, prm, B, lds, s);
    if (after) after(s);
    return rc;
}

}  // namespace cd
