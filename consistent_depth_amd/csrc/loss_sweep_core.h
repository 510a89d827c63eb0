// Row-sweep formulation (v4) of the fused geometric-consistency loss + gradient: geometry, the per-pair schedule
// ("plan") and the per-thread phase functions.  Host/device neutral: hipcc compiles it into loss_sweep.hip, g++ into
// tests/emul (sequential execution of the same phase functions -- checks plans, ring indexing and flush logic on the
// CPU; never part of the product path).
//
// Why a sweep.  The gradient has a SCATTER part: source pixel (j, y, x) adds to the 4 bilinear taps of frame k = 1 - j
// at (y + flow_y, x + flow_x).  v1 scatters with global atomics (memory-side on MI355X: 4 % of the HBM roofline), v2/v3
// give every 32x32 tile a workgroup and pay for the tile halos with re-evaluation (v2, 2.07x) or with slabs written to
// and re-read from memory by a second pass (v3, 1.6-2.6x the algorithmic traffic, 22 % of the roofline).  Here ONE
// workgroup owns ONE pair and walks both frames top to bottom in lock step, a few rows per step ("item"):
//
//   * each frame has a RING of R image rows in LDS: depth (fp32, head already applied) and a 64-bit fixed-point
//     gradient accumulator.  Ring f is the sampling source AND the scatter target of the sources of frame 1-f, and it
//     also receives the direct gradient term of frame f's own sources;
//   * per item: rows that no later source can touch leave the ring (accumulator -> float -> ONE coalesced store of
//     the finished gradient row), new rows enter (depth loaded ONCE, exp/reciprocal applied once), then the
//     G source rows of each frame are evaluated: 5 LDS reads and 5 ds_add_u64 per pixel, no global atomics;
//   * every input byte is read once and every gradient byte written once: HBM traffic = the algorithmic 10*H*W*4 bytes
//     per pair, in row order (fully coalesced);
//   * the two frames need not advance at the same rate: a per-pair PLAN (dataset constant, like the tile windows: it
//     depends only on flows and masks) lists, per item, which row group of which frame is processed and where the two
//     ring windows start, so that the taps of the valid sources fall inside the rings (global vertical offsets between
//     the frames, zoom, slow drifts are absorbed by the schedule; only the vertical SPREAD inside one item must fit);
//   * anything that does not fit (wild flow, |value| beyond the fixed-point range, NaN/inf) goes through the same
//     overflow list + guarded v1 fallback as v2/v3: exact for ANY input, fast for consistent flow.
//
// Integer accumulation is order independent: the gradient is bit-reproducible run to run.
#pragma once
#include "loss_math.h"

namespace cd {
namespace sweep {

constexpr int kThreads = 1024;       // one workgroup per pair = 16 waves = one CU at 4 waves per SIMD
constexpr int kFrameThreads = 512;   // threads [0, 512) serve frame 0, [512, 1024) frame 1 (whole waves per frame)
constexpr int kStagePasses = 2;      // at most SMAX = kStagePasses * RP rows enter / leave a ring per item
// A row group (the source rows of one item) is evaluated in up to this many passes of RP rows.  Two passes halve the barriers
// and the per-item scalar work, but the second inlined copy of the evaluation code pushes the kernel over its register budget:
// measured at 256 pairs of 384x224 (profiles/loss_sweep_variants_r03.txt) 0.344 ms with 1 pass, 0.361 ms with 2 (0.374 with
// the two passes sharing one non-unrolled copy).  The code path stays (CPU emulation tests run it: -DCD_SWEEP_GROUP_PASSES=2).
#ifndef CD_SWEEP_GROUP_PASSES
#define CD_SWEEP_GROUP_PASSES 1
#endif
constexpr int kGroupPasses = CD_SWEEP_GROUP_PASSES;
constexpr int kMaxGroups = 1024;     // row groups per frame the planner handles
constexpr int kLdsBytes = 160 * 1024;
constexpr int kLdsReserve = 1024;    // reduction scratch etc.
constexpr int kMaxPXT = 4;

struct Geo {
    int H, W;
    int PXT;        // ADJACENT pixels per thread: columns PXT * cg + i, i < PXT -- one 4*PXT-byte load / store / LDS access per plane and
                    // row (W is a multiple of PXT), and the pixels of a thread sit in consecutive registers (packed fp32 instructions)
    int CG;         // column groups = W / PXT
    int RP;         // image rows per pass of the 512 threads of a frame = kFrameThreads / CG
    int G;          // source rows per item and frame (<= kGroupPasses * RP)
    int RW;         // ring row stride in elements: W + pad column(s) (weight-0 taps land there), a multiple of PXT
    int R;          // ring rows per frame
    int NG;         // row groups per frame = ceil(H / G)
    int SMAX;       // max ring-window advance per item
    int max_items;  // capacity of a plan
    int ok;         // this geometry is supported by the sweep kernel (else: v3)
};

CD_HD Geo make_geo(int H, int W, int pxt) {
    Geo g;
    memset(&g, 0, sizeof(g));
    g.H = H; g.W = W; g.PXT = pxt;
    if (H < 2 || W < 2 || H > 16384 || (pxt != 1 && pxt != 2 && pxt != 4) || W % pxt != 0) return g;
    g.CG = W / pxt;
    if (g.CG > kFrameThreads) return g;
    g.RP = kFrameThreads / g.CG;
    if (g.RP > 16) g.RP = 16;
    g.RW = (W + pxt) / pxt * pxt;                // >= W + 1
    // fp32 depth + 32-bit accumulator per element; one row more than R is kept: slot R is a COPY of slot 0 (see ring_rows below)
    int R = (kLdsBytes - kLdsReserve) / (2 * g.RW * 8) - 1;
    if (R > 512) R = 512;
    if (R < 1) return g;
    // a POWER OF TWO: image row r lives in slot r & (R - 1) -- one AND per ring address instead of a window-relative offset
    // with a wrap (the rings are direct mapped; windows only say which rows are resident)
    while (R & (R - 1)) R &= R - 1;
    while (R >= 2 * (H + 2) && R > 16) R >>= 1;   // rows 0 .. H (H = the pad row under the image) never need more
    g.R = R;
    if (R < 12) return g;
    // rows per item: as many as two passes of the threads can take (half the barriers and per-item scalar work of one pass),
    // as long as a third of the ring stays free for the taps' spread; small images: fewer
    const int gmax = (R - 8) / 3;
    g.G = kGroupPasses * g.RP < gmax ? kGroupPasses * g.RP : gmax;
    if (g.G < 1) return g;
    g.SMAX = kStagePasses * g.RP;
    if (g.SMAX > R - g.G - 8) g.SMAX = R - g.G - 8;   // leave room for the taps' spread
    g.NG = (H + g.G - 1) / g.G;
    if (g.SMAX < g.G || g.NG > kMaxGroups) return g;  // windows must be able to follow the sources: else v3
    g.max_items = 4 * g.NG + 3 * ((H + 1 + R) / g.SMAX + 2) + 8;   // groups (+ a forced slide each, worst case) + window-only items
    g.ok = 1;
    return g;
}

// Rows a ring occupies in LDS: the R direct-mapped slots plus slot R, which MIRRORS slot 0 -- the depth of the row in slot 0 is written
// to both, accumulator words of slot R are added to slot 0's when the row leaves -- so that "the ring row under slot s" is slot s + 1 for
// EVERY s < R: the four bilinear taps of a source are one address and the constant offsets {0, 1, RW, RW + 1} (process_rows_fast).
CD_HD int ring_rows(const Geo& g) { return g.R + 1; }
CD_HD size_t ring_lds_bytes(const Geo& g) { return (size_t)2 * ring_rows(g) * g.RW * 8 + kLdsReserve; }

// ---------------------------------------------------------------- plan
struct Item {
    short p[2];   // first source row of frame f processed in this item (-1: none)
    short w[2];   // first image row held by ring f during this item (rows [w, w + R))
};

constexpr short kNoRow = 32767;

// lo / hi: [2][NG] first / last ring row (of frame 1-f) touched by the taps of the VALID sources of each row group of
// frame f (lo = kNoRow, hi = -1: no valid source).  suf: scratch [2][NG + 1].  Returns the number of items (<= max_items)
// or -1.  Guarantees (checked by tests/test_sweep_plan_cpu.py):
//   * every row group of both frames is processed exactly once, in increasing order;
//   * windows never move back and advance by at most SMAX rows per item;
//   * the G source rows of a processed group lie inside their own frame's window AND below the previous item's window top
//     (own depth and the direct term live in the ring; rows entering during an item are not used by it);  the taps lie
//     inside the other ring, below its previous top, whenever the flow allows it (otherwise: overflow path);
//   * the last items move both windows to H, i.e. every gradient row has left the rings.
CD_HD int plan_items(const Geo& g, const short* lo, const short* hi, short* suf, Item* items) {
    const int NG = g.NG, G = g.G, R = g.R, H = g.H, SMAX = g.SMAX;
    for (int f = 0; f < 2; ++f) {
        suf[f * (NG + 1) + NG] = kNoRow;
        for (int i = NG - 1; i >= 0; --i) {
            const short a = lo[f * NG + i], b = suf[f * (NG + 1) + i + 1];
            suf[f * (NG + 1) + i] = a < b ? a : b;
        }
    }
    int p[2] = {0, 0}, w[2] = {0, 0}, n = 0;
    while (p[0] < NG || p[1] < NG) {
        if (n >= g.max_items - 2) return -1;
        int target[2];
        bool ready[2] = {false, false};
        for (int f = 0; f < 2; ++f) {
            const int k = 1 - f;
            const int own = p[f] < NG ? p[f] * G : H + 1;                       // next own source row: must stay in ring f
            const int need = p[k] < NG ? (int)suf[k * (NG + 1) + p[k]] : (int)kNoRow;   // lowest row a later source of k samples
            int t = own < need ? own : need;
            if (t > H + 1) t = H + 1;
            target[f] = w[f] > t ? w[f] : t;
        }
        // An item's rows enter and leave the rings WHILE its sources are evaluated (one barrier per item), so a group can only
        // use rows staged before: rows below the PREVIOUS window's top w[.] + R (and at or above the new bases target[.]).
        for (int f = 0; f < 2; ++f) {
            const int k = 1 - f;
            if (p[f] >= NG) continue;
            const bool own_fit = p[f] * G + G <= w[f] + R;
            const bool taps_fit = (int)hi[f * NG + p[f]] < w[k] + R;
            ready[f] = own_fit && taps_fit;
        }
        bool slide_only = false;
        if (!ready[0] && !ready[1]) {
            if (target[0] > w[0] || target[1] > w[1]) slide_only = true;     // let the windows catch up first
            else {
                // windows are stuck and no group fits: force the frame that is further behind.  Its own rows must fit (make
                // room by dropping the lowest rows of its ring); whatever its taps miss goes through the overflow list.
                const int f = (p[0] < NG && (p[1] >= NG || p[0] <= p[1])) ? 0 : 1;
                const int base = p[f] * G + G - R;
                if (base > w[f]) { target[f] = base; slide_only = true; }
                else ready[f] = true;
            }
        }
        bool big = false;
        for (int f = 0; f < 2; ++f) big = big || (target[f] - w[f] > SMAX);
        if (big || slide_only) {   // slide in steps the staging registers can feed
            for (int f = 0; f < 2; ++f) w[f] = target[f] < w[f] + SMAX ? target[f] : w[f] + SMAX;
            Item it; it.p[0] = it.p[1] = -1; it.w[0] = (short)w[0]; it.w[1] = (short)w[1];
            items[n++] = it;
            continue;
        }
        w[0] = target[0]; w[1] = target[1];
        Item it;
        it.p[0] = ready[0] ? (short)(p[0] * G) : (short)-1;
        it.p[1] = ready[1] ? (short)(p[1] * G) : (short)-1;
        it.w[0] = (short)w[0]; it.w[1] = (short)w[1];
        items[n++] = it;
        if (ready[0]) ++p[0];
        if (ready[1]) ++p[1];
    }
    while (w[0] < H || w[1] < H) {   // tail: the remaining rows leave the rings
        if (n >= g.max_items) return -1;
        for (int f = 0; f < 2; ++f) if (w[f] < H) w[f] = H < w[f] + SMAX ? H : w[f] + SMAX;
        Item it; it.p[0] = it.p[1] = -1; it.w[0] = (short)w[0]; it.w[1] = (short)w[1];
        items[n++] = it;
    }
    return n;
}

// ---------------------------------------------------------------- expanded plan: what the kernel reads
// One record per item and frame with everything the phases need as plain ints (a wave reads its frame's record with one
// 32-byte scalar load per item, plus w / ws of the other frame): no window bookkeeping inside the kernel.
struct Rec {
    int w, ws;         // first image row held by the ring during this item, and its slot
    int nv;            // rows [w, w + nv) can be used by this item's sources: staged BEFORE it (= previous window top - w)
    int p;             // first source row of the group processed in this item (-1: none)
    int fl_lo, fl_hi;  // rows [fl_lo, fl_hi) leave the ring at the start of this item ...
    int fl_slot;       // ... fl_lo sits in this slot
    int s_lo, s_hi;    // rows [s_lo, s_hi) enter the ring during this item (row H = the pad row)
    int inw;           // 1: EVERY tap of every valid source of this item's group lies in rows of the other ring this item can use
                       // (the planner knows the groups' tap-row bounds: the fast source pass then needs neither clamp nor vote)
    int pad[2];
};
static_assert(sizeof(Rec) == 48, "Rec layout");
struct PlanItem { Rec f[2]; };
// fan_in: max sources per target pixel; limit: sweep_limit_scaled(fan_in); fingerprint: plan_fingerprint_term summed over the 2 x 256
// unit-sample positions of the flows / masks the plan was made from (0 in plans no device kernel made: the host emulation's)
struct PlanHeader { int n_items, G, R, PXT; int fan_in; float limit; unsigned fingerprint; int pad; };
static_assert(sizeof(PlanHeader) == 32, "PlanHeader layout");
CD_HD size_t plan_bytes(const Geo& g) { return sizeof(PlanHeader) + sizeof(PlanItem) * (size_t)g.max_items; }

CD_HD int wrap_slot(int s, int R) { return s >= R ? s - R : s; }
CD_HD unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
CD_HD unsigned wrapu(unsigned s, unsigned R) { return umin(s, s - R); }   // s in [0, 2R): s - R wraps to a huge number when s < R
#if defined(__HIP_DEVICE_COMPILE__)
CD_HD unsigned mad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }   // v_mad_u32_u24 (operands < 2^24)
#else
CD_HD unsigned mad24(unsigned a, unsigned b, unsigned c) { return a * b + c; }
#endif

// rows [0, init_hi) are staged by the kernel's prologue (window of item 0 = [0, R))
CD_HD int init_stage_hi(const Geo& g) { return g.R < g.H + 1 ? g.R : g.H + 1; }

// lo / hi: the planner's tap-row bounds (plan_items); without them no item claims `inw`
CD_HD void expand_plan(const Geo& g, const Item* items, int n, PlanItem* out, const short* lo = nullptr, const short* hi = nullptr) {
    int wprev[2] = {0, 0}, wsprev[2] = {0, 0}, staged[2];
    staged[0] = staged[1] = init_stage_hi(g);
    for (int t = 0; t < n; ++t) {
        for (int f = 0; f < 2; ++f) {
            Rec r;
            r.w = items[t].w[f];
            r.ws = wrap_slot(wsprev[f] + (r.w - wprev[f]), g.R);
            r.p = items[t].p[f];
            r.nv = wprev[f] + g.R - r.w;
            r.inw = 0; r.pad[0] = r.pad[1] = 0;
            r.fl_lo = wprev[f];
            r.fl_hi = r.w < g.H ? r.w : g.H;
            if (r.fl_hi < r.fl_lo) r.fl_hi = r.fl_lo;
            r.fl_slot = wsprev[f];
            r.s_lo = staged[f];
            const int top = r.w + g.R < g.H + 1 ? r.w + g.R : g.H + 1;
            r.s_hi = top > r.s_lo ? top : r.s_lo;
            staged[f] = r.s_hi;
            wprev[f] = r.w; wsprev[f] = r.ws;
            out[t].f[f] = r;
        }
        if (lo != nullptr && hi != nullptr)
            for (int f = 0; f < 2; ++f) {
                Rec& r = out[t].f[f];
                const Rec& o = out[t].f[1 - f];
                if (r.p < 0) continue;
                const int gi = f * g.NG + r.p / g.G;
                // the fast pass addresses rows ya and ya + 1 (hi holds ya + 1): both inside [o.w, o.w + o.nv).  A group without a
                // valid source (lo = kNoRow, hi = -1) qualifies: its lanes are all mask-0 lanes, for which any resident row does.
                r.inw = (o.nv >= 2 && (int)lo[gi] >= o.w && (int)hi[gi] <= o.w + o.nv - 1) ? 1 : 0;
            }
    }
}

// ---------------------------------------------------------------- accumulator units from the data
// The 32-bit accumulator of ring j resolves 2^-20 U_j and accepts sources of up to 64 U_j (loss_math.h), so U_j has to sit near
// the typical gradient magnitude of plane j -- which depends on the depths (1/z^2 factors), the baseline and the lambdas, not
// only on the per-pair constants prep_pair knows.  It is ESTIMATED per launch from a 16 x 16 grid of sample sources per
// direction: the mean |direct term| of direction j plus the mean |scatter| of direction k = 1 - j (what lands in ring j), rounded
// down to a power of two (scaling by U is then exact).  Any estimate gives a correct gradient -- a poor one costs resolution
// (too large) or overflow-list traffic (too small); non-finite or empty estimates keep prep_pair's rule.
constexpr int kUnitGrid = 16;

struct UnitSample { float direct, scatter; int valid; };

// one sample source (x, y) of direction j, in two steps: what it READS (mask, flow, its own depth and the OTHER frame's depth at the
// same pixel) and what it COMPUTES from them with the pair's PairCam.  The kernel issues the reads of all samples at its very top, before
// the pair's constants exist.  vj / vk: the raw depth planes of its frame / the other frame.
// Round 6: the scatter magnitude used to come from the four depths the sample's FLOW points at -- a second, dependent memory latency
// per workgroup behind 57 KB of window rows (`vmcnt` retires in order: the constants were ready 13 us after kernel entry, 8 % of a
// 256-pair call).  The estimate only fixes a power of two: the other frame's depth at the sample's own pixel stands in for the sampled
// depth (the flow moves it by a few pixels of a smooth map), one latency, no tap arithmetic.
struct UnitLoads { float m, fx, fy, vj, vk0; };
CD_HD void unit_sample_load_own(UnitLoads& l, const float* vj, const float* vk, const float* fl, const float* mk, int H, int W, int x, int y) {
    const int HW = H * W, p = y * W + x;
    l.m = mk[p]; l.fx = fl[p]; l.fy = fl[HW + p]; l.vj = vj[p]; l.vk0 = vk[p];
}
CD_HD UnitLoads unit_sample_load(const float* vj, const float* vk, const float* fl, const float* mk, int H, int W, int x, int y) {
    UnitLoads l;
    unit_sample_load_own(l, vj, vk, fl, mk, H, W, x, y);
    return l;
}
template <int MODE>
CD_HD UnitSample unit_sample_eval(const PairCam& c, const UnitLoads& l, int x, int y) {
    UnitSample u;
    u.direct = u.scatter = 0.f; u.valid = 0;
    const float m = l.m;
    if (m == 0.f) return u;
    const float d = to_depth<MODE>(l.vj), dk = to_depth<MODE>(l.vk0);
    const float r0 = ((float)x - c.cx_r) * c.ifx_r, r1 = -((float)y - c.cy_r) * c.ify_r;
    const float a0 = c.M[0] * r0 + c.M[1] * r1 - c.M[2], a1 = c.M[3] * r0 + c.M[4] * r1 - c.M[5], a2 = c.M[6] * r0 + c.M[7] * r1 - c.M[8];
    const float X = d * a0 + c.c[0], Y = d * a1 + c.c[1], Z = d * a2 + c.c[2];
    const float iZ = 1.f / Z;
    const float ex = (c.cx_t - c.fx_t * X * iZ) - ((float)x + l.fx), ey = (c.cy_t + c.fy_t * Y * iZ) - ((float)y + l.fy);
    const float e2 = ex * ex + ey * ey;
    const float ie = e2 > 0.f ? 1.f / sqrtf(e2) : 0.f;
    const float dpx = c.fx_t * iZ * (X * a2 * iZ - a0), dpy = c.fy_t * iZ * (a1 - Y * a2 * iZ);
    const float izs = 1.f / dk;       // (|1 / zs| with zs = -(sampled depth) ~ -dk)
    const float direct = (fabsf(c.gr * m * (ex * dpx + ey * dpy) * ie) + fabsf(c.gb * m * a2 * iZ * iZ)) * fabsf(depth_jac<MODE>(d));
    const float scat = fabsf(c.gb * m * izs * izs) * fabsf(depth_jac<MODE>(dk));
    if (!(direct < INFINITY) || !(scat < INFINITY)) return u;   // NaN / inf: not a sample (such inputs end on the exact paths anyway)
    u.direct = direct; u.scatter = scat; u.valid = 1;
    return u;
}

CD_HD float pow2_floor(float x) {   // largest power of two <= x (x a positive normal number)
    unsigned b;
    memcpy(&b, &x, 4);
    b &= 0x7f800000u;
    float r;
    memcpy(&r, &b, 4);
    return r;
}

// sums over the samples of both directions -> cams[0..1].{unit, dr, db, sc}
CD_HD void units_from_samples(PairCam* cams, const float* sum_direct, const float* sum_scatter, const int* n) {
    float U[2];
    for (int j = 0; j < 2; ++j) {
        const int k = 1 - j;
        const float est = (n[j] > 0 ? sum_direct[j] / (float)n[j] : 0.f) + (n[k] > 0 ? sum_scatter[k] / (float)n[k] : 0.f);
        U[j] = (est > 1e-30f && est < 1e30f) ? pow2_floor(est) : cams[j].unit;
    }
    for (int j = 0; j < 2; ++j) {
        const int k = 1 - j;
        cams[j].unit = U[j];
        cams[j].dr = cams[j].gr / U[j];
        cams[j].db = cams[j].gb / U[j];
        cams[j].sc = cams[j].gb / U[k];
    }
}

// the sample of grid point t (< kUnitGrid^2) of direction j of one pair (depth_p: [2][HW] raw depths; fwd / bwd flow and mask planes of the pair)
template <int MODE>
CD_HD UnitSample unit_sample_at(const PairCam* cams, const float* depth_p, const float* ff, const float* fb, const float* mf, const float* mb,
                                int H, int W, int j, int t) {
    const int iy = t / kUnitGrid, ix = t - iy * kUnitGrid;
    const int x = (2 * ix + 1) * W / (2 * kUnitGrid), y = (2 * iy + 1) * H / (2 * kUnitGrid);
    const int HW = H * W;
    return unit_sample_eval<MODE>(cams[j], unit_sample_load(depth_p + (j ? HW : 0), depth_p + (j ? 0 : HW), j ? fb : ff, j ? mb : mf, H, W, x, y), x, y);
}
// (the kernel's two steps of unit_sample_at)
// A plan's statements (Rec::inw: "every tap of this row group's valid sources lies in resident rows") hold for the flows and masks it
// was made from.  Round 5's fast source pass TRUSTS inw (no clamp, no vote): a blob that belongs to other flows / masks would give
// silently wrong gradients.  The plan therefore carries a fingerprint -- the sum (mod 2^32: order-free) of this term over the 2 x 256
// positions the sweep kernel samples anyway for its accumulator units -- which the sweep recomputes from the flows / masks it is
// given; on a mismatch no item's inw is believed (the general pass: slower, exact).  i = direction * 256 + sample index.
CD_HD unsigned plan_fingerprint_term(int i, float m, float fx, float fy) {
    union { float f; unsigned u; } a, b, c;
    a.f = m; b.f = fx; c.f = fy;
    unsigned h = (a.u * 0x9E3779B1u) ^ (b.u * 0x85EBCA77u) ^ ((c.u << 13) | (c.u >> 19));
    h = (h ^ (unsigned)i) * 0xC2B2AE3Du;
    return h ^ (h >> 15);
}

CD_HD void unit_sample_xy(int H, int W, int t, int* x, int* y) {
    const int iy = t / kUnitGrid, ix = t - iy * kUnitGrid;
    *x = (2 * ix + 1) * W / (2 * kUnitGrid); *y = (2 * iy + 1) * H / (2 * kUnitGrid);
}

// ---------------------------------------------------------------- execution state
struct Cam {   // the fields of PairCam the sweep uses, copied once into registers
    float M[9], c[3], ifx_r, ify_r, cx_r, cy_r, fx_t, fy_t, cx_t, cy_t, sx, sy;
    float drs, dbs, scs;   // dr, db, sc of PairCam times 2^20: the algebra produces fixed-point-ready values (exact scaling)
    float unit_s;          // unit * 2^-20: accumulator integer -> gradient, and pre-scaled value -> gradient (overflow list)
};
CD_HD Cam make_cam(const PairCam& p) {
    Cam c;
    for (int i = 0; i < 9; ++i) c.M[i] = p.M[i];
    for (int i = 0; i < 3; ++i) c.c[i] = p.c[i];
    c.ifx_r = p.ifx_r; c.ify_r = p.ify_r; c.cx_r = p.cx_r; c.cy_r = p.cy_r;
    c.fx_t = p.fx_t; c.fy_t = p.fy_t; c.cx_t = p.cx_t; c.cy_t = p.cy_t;
    c.sx = p.sx; c.sy = p.sy;
    c.drs = p.dr * SWEEP_FX_ONE_F; c.dbs = p.db * SWEEP_FX_ONE_F; c.scs = p.sc * SWEEP_FX_ONE_F;
    c.unit_s = p.unit * (1.f / SWEEP_FX_ONE_F);
    return c;
}

// What one wave works with: its own frame j (whose source rows it evaluates, whose ring rows it flushes and stages)
// and the other frame k (whose ring its sources sample and scatter into).  Wave-uniform.
struct View {
    int H, W, R, RW, RP, G, CG;
    unsigned HW;
    const float* vj;           // raw depth plane of frame j
    const float* vk;           // ... of frame k (slow path only)
    const float* flj;          // flow of direction j: dx plane, dy plane at + HW
    const float* mkj;          // mask of direction j
    float* gradj;              // gradient plane of frame j (w.r.t. the raw depth input)
    float* Dj; float* Dk;      // LDS depth rings [R][RW]
    unsigned* Aj; unsigned* Ak;   // LDS accumulator rings [R][RW]: 32-bit fixed point (loss_math.h)
    Cam cj;                    // direction j
    float unit_k_s;            // accumulator unit of ring k, times 2^-20
    float limit;               // bound of the |.| sum of a source's pre-scaled contributions (PlanHeader::limit)
    unsigned gbj, gbk;         // element index of the two gradient planes in the whole gradient tensor (overflow list)
};

// PXT floats / accumulator words of one thread and row: ONE aligned memory or LDS access
template <int N> struct alignas(4 * N) VecF { float v[N]; };
template <int N> struct alignas(4 * N) VecU { unsigned v[N]; };

template <int PXT> struct Lane {   // per-thread constants
    int rr;                 // row inside a pass
    unsigned rrW;           // rr * W
    bool on;                // takes part at all (rr < RP and its column group exists); whole waves are off at W = 224 (448 of 512 lanes)
    unsigned x0;            // its columns: x0 + i, i < PXT
    float a0[PXT], a1[PXT], a2[PXT];      // M[0], M[3], M[6] * r0(x): the column part of a = M (r0, r1, -1)
};
template <int PXT> CD_HD Lane<PXT> make_lane(const View& v, int lt /* thread index inside its frame's 512 */) {
    Lane<PXT> l;
    l.rr = lt / v.CG;
    const int cg = lt - l.rr * v.CG;
    l.rrW = (unsigned)(l.rr * v.W);
    l.on = l.rr < v.RP;
    l.x0 = (unsigned)(cg * PXT);
    for (int i = 0; i < PXT; ++i) {
        const float r0 = ((float)(l.x0 + i) - v.cj.cx_r) * v.cj.ifx_r;
        l.a0[i] = v.cj.M[0] * r0; l.a1[i] = v.cj.M[3] * r0; l.a2[i] = v.cj.M[6] * r0;
    }
    return l;
}

// global load / store of the PXT pixels of a thread at a 32-bit BYTE offset from a wave-uniform base: the address is base
// (SGPRs) + offset (one VGPR), one dwordxPXT instruction (rows start at multiples of 4 * PXT bytes: W % PXT == 0, planes 16-byte aligned)
template <int N> CD_HD VecF<N> ldgv(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const VecF<N>*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <int N> CD_HD void stgv(float* base, unsigned byte_off, const VecF<N>& v) {
    *reinterpret_cast<VecF<N>*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// ... the same for planes that are touched ONCE per call (flow, mask: read; gradient: written): non-temporal accesses keep them from
// displacing the depth rows / plan records in L2 and the Infinity Cache.  (The host emulation has no cache policy.)
// Measured (round 6, profiles/loss_sweep_variants_r06.txt; whole call, 256 / 1024 pairs, fraction of 8 TB/s): default policy 0.546 / 0.552,
// flow + mask loads nt 0.553 / 0.554, gradient stores nt 0.560 / 0.569, both 0.616 / 0.578, depth rows too 0.588 / 0.602 -- the depth
// planes of a 256-pair call (176 MB) survive in the 256 MB Infinity Cache when nothing else allocates there, those of 1024 pairs do
// not: the depth policy is a template argument of the kernel (NTD), chosen by the launch size.
#ifndef CD_SWEEP_NT
#define CD_SWEEP_NT 3      // bit 0: flow / mask loads, bit 1: gradient stores, bit 2: depth rows of every instantiation (A/B builds)
#endif
template <int N> CD_HD VecF<N> ldgv_nt(const float* base, unsigned byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float vt __attribute__((ext_vector_type(N)));
    const vt t = __builtin_nontemporal_load(reinterpret_cast<const vt*>(reinterpret_cast<const char*>(base) + byte_off));
    VecF<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    return r;
#else
    return ldgv<N>(base, byte_off);
#endif
}
template <int N> CD_HD void stgv_nt(float* base, unsigned byte_off, const VecF<N>& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float vt __attribute__((ext_vector_type(N)));
    vt t;
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = v.v[i];
    __builtin_nontemporal_store(t, reinterpret_cast<vt*>(reinterpret_cast<char*>(base) + byte_off));
#else
    stgv<N>(base, byte_off, v);
#endif
}
template <int N> CD_HD VecF<N> ldg_in(const float* base, unsigned byte_off) { return (CD_SWEEP_NT & 1) ? ldgv_nt<N>(base, byte_off) : ldgv<N>(base, byte_off); }
template <int N, bool NTD = false> CD_HD VecF<N> ldg_depth(const float* base, unsigned byte_off) { return ((CD_SWEEP_NT & 4) || NTD) ? ldgv_nt<N>(base, byte_off) : ldgv<N>(base, byte_off); }
template <int N> CD_HD void stg_grad(float* base, unsigned byte_off, const VecF<N>& v) { if (CD_SWEEP_NT & 2) stgv_nt<N>(base, byte_off, v); else stgv<N>(base, byte_off, v); }

template <int PXT> struct Inputs { float fx[PXT], fy[PXT], m[PXT]; };   // flow and mask of the PXT source pixels of one pass

template <int PXT> struct Regs {
    float sv[kStagePasses][PXT];         // raw depth of the rows entering the ring
    double acc_r, acc_d;                 // loss partial sums of this thread (its frame = direction) ...
    float pend_r, pend_d;                // ... plus what the fast source pass of the previous item left (added in straight-line code at the
                                         // top of the next pass: an fp64 add behind the pass's slow-path branch costs register copies per item)
    bool bad;                            // a staged depth was not a positive finite number (see stage_rows)
};
template <int PXT> CD_HD void init_regs(Regs<PXT>& r) {
    for (int i = 0; i < PXT; ++i)
        for (int s = 0; s < kStagePasses; ++s) r.sv[s][i] = 0.f;
    r.acc_r = r.acc_d = 0.0;
    r.pend_r = r.pend_d = 0.f;
    r.bad = false;
}
// the thread's loss partial sums (what the kernel's epilogue reads)
template <int PXT> CD_HD double loss_sum_r(const Regs<PXT>& r) { return r.acc_r + (double)r.pend_r; }
template <int PXT> CD_HD double loss_sum_d(const Regs<PXT>& r) { return r.acc_d + (double)r.pend_d; }

// does pass q of the row group at p give this thread a source row?  (p < 0: the item has no group for this frame)
template <int PXT> CD_HD bool pass_row_ok(const View& v, const Lane<PXT>& l, int p, int q) {
    const int ro = q * v.RP + l.rr;
    return l.on && p >= 0 && ro < v.G && p + ro < v.H;
}

// inputs of pass q of the source rows [p, p + G) of the wave's frame.  Unconditional loads on clamped offsets (a load under a
// divergent branch makes hipcc wait for it at once), the values of inactive threads are zeroed afterwards.
template <int PXT> CD_HD void load_inputs(const View& v, const Lane<PXT>& l, int p, int q, Inputs<PXT>& in) {
    const bool ok = pass_row_ok<PXT>(v, l, p, q);
    const unsigned off = ok ? ((unsigned)(p + q * v.RP) * (unsigned)v.W + l.rrW + l.x0) << 2 : 0u;
    const VecF<PXT> a = ldg_in<PXT>(v.flj, off), b = ldg_in<PXT>(v.flj + v.HW, off), mm = ldg_in<PXT>(v.mkj, off);
#pragma unroll
    for (int i = 0; i < PXT; ++i) { in.fx[i] = ok ? a.v[i] : 0.f; in.fy[i] = ok ? b.v[i] : 0.f; in.m[i] = ok ? mm.v[i] : 0.f; }
}

// raw depth of the rows [s_lo, s_hi) that enter the wave's ring
template <int PXT> CD_HD void load_stage(const View& v, const Lane<PXT>& l, int s_lo, int s_hi, float (*sv)[PXT]) {
#pragma unroll
    for (int s = 0; s < kStagePasses; ++s) {
        const int row = s_lo + s * v.RP + l.rr;
        const bool ok = l.on && row < s_hi && row < v.H;
        const unsigned off = ok ? ((unsigned)(s_lo + s * v.RP) * (unsigned)v.W + l.rrW + l.x0) << 2 : 0u;
        const VecF<PXT> a = ldg_depth<PXT>(v.vj, off);
#pragma unroll
        for (int i = 0; i < PXT; ++i) sv[s][i] = ok ? a.v[i] : 0.f;
    }
}

// ... the same without anything to select after the loads (clamped addresses; stage_rows only reads the values of rows it stages): the
// compiler has no reason to wait for them before the stage of the NEXT item
template <int PXT, bool NTD = false> CD_HD void load_stage_nosel(const View& v, const Lane<PXT>& l, int s_lo, int s_hi, float (*sv)[PXT]) {
#pragma unroll
    for (int s = 0; s < kStagePasses; ++s) {
        // an ordinary item moves its window by RP rows: the passes beyond the first are skipped by a WAVE-UNIFORM branch (the rows of
        // the tail / slide-only items need them); stage_rows never reads the registers of a pass without rows
        if (s > 0 && s_hi - s_lo <= s * v.RP) continue;
        const int row = s_lo + s * v.RP + l.rr;
        const int rc = row < v.H ? row : v.H - 1;       // (a lane beyond s_hi reads a row it does not stage: in bounds, never used)
        const unsigned off = mad24((unsigned)rc, (unsigned)v.W << 2, l.x0 << 2);
        const VecF<PXT> a = ldg_depth<PXT, NTD>(v.vj, off);
#pragma unroll
        for (int i = 0; i < PXT; ++i) sv[s][i] = a.v[i];
    }
}

// (keeps a rare wave-uniform branch a branch: without it the compiler folds the branch into per-lane selects on the hot path)
#if defined(__HIP_DEVICE_COMPILE__)
#define CD_KEEP_BRANCH() asm volatile("" ::: "memory")
#else
#define CD_KEEP_BRANCH() ((void)0)
#endif
// rows [s_lo, s_hi) enter the ring (their values are in sv; row r -> slot r & (R - 1)).  Returns false if a staged depth
// is not a positive finite number (such an input takes the exact v1 path: see process_rows, "lenient").
// PADS = false: the pad column(s) are left alone -- their value (1) never changes once the prologue has written every slot.
template <int MODE, int PXT, bool PADS = true>
CD_HD bool stage_rows(const View& v, const Lane<PXT>& l, int s_lo, int s_hi, const float (*sv)[PXT]) {
    bool good = true;
#pragma unroll
    for (int s = 0; s < kStagePasses; ++s) {
        const int row = s_lo + s * v.RP + l.rr;
        if (l.on && row < s_hi) {
            const unsigned base = (unsigned)((row & (v.R - 1)) * v.RW);
            VecF<PXT> d;
#pragma unroll
            for (int i = 0; i < PXT; ++i) d.v[i] = to_depth<MODE>(sv[s][i]);
            if (s_hi > v.H) {                   // wave-uniform, the last item(s) only: row H is the pad row -- finite depth, only ever sampled with weight 0
                CD_KEEP_BRANCH();
#pragma unroll
                for (int i = 0; i < PXT; ++i) d.v[i] = row < v.H ? d.v[i] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < PXT; ++i) good = good && (d.v[i] > 0.f && d.v[i] < INFINITY);
            *reinterpret_cast<VecF<PXT>*>(&v.Dj[base + l.x0]) = d;
            if (base == 0u) *reinterpret_cast<VecF<PXT>*>(&v.Dj[(unsigned)(v.R * v.RW) + l.x0]) = d;      // slot R mirrors slot 0
            if (PADS && l.x0 == 0u)
                for (int c = v.W; c < v.RW; ++c) {
                    v.Dj[base + (unsigned)c] = 1.f;   // the pad column(s)
                    if (base == 0u) v.Dj[(unsigned)(v.R * v.RW + c)] = 1.f;
                }
        }
    }
    return good;
}

// rows [lo, hi) leave the ring: accumulator -> gradient row (one plain store), accumulator
// cleared for the next tenant
template <int PXT>
CD_HD void flush_rows(const View& v, const Lane<PXT>& l, int lo, int hi) {
    const float unit = v.cj.unit_s;
#pragma unroll
    for (int s = 0; s < kStagePasses; ++s) {
        const int row = lo + s * v.RP + l.rr;
        if (l.on && row < hi) {
            const unsigned base = (unsigned)((row & (v.R - 1)) * v.RW);
            VecU<PXT>* ap = reinterpret_cast<VecU<PXT>*>(&v.Aj[base + l.x0]);
            VecU<PXT> n = *ap;
            VecU<PXT> z;
            VecF<PXT> g;
#pragma unroll
            for (int i = 0; i < PXT; ++i) z.v[i] = 0u;
            if (base == 0u) {      // what the fast pass added to the mirror of slot 0
                VecU<PXT>* mp = reinterpret_cast<VecU<PXT>*>(&v.Aj[(unsigned)(v.R * v.RW) + l.x0]);
                const VecU<PXT> n2 = *mp;
#pragma unroll
                for (int i = 0; i < PXT; ++i) n.v[i] += n2.v[i];
                *mp = z;
            }
#pragma unroll
            for (int i = 0; i < PXT; ++i) g.v[i] = (float)(int)n.v[i] * unit;
            *ap = z;
            stg_grad<PXT>(v.gradj, ((unsigned)(lo + s * v.RP) * (unsigned)v.W + l.rrW + l.x0) << 2, g);
            if (l.x0 == 0u)
                for (int c = v.W; c < v.RW; ++c) {
                    v.Aj[base + (unsigned)c] = 0u;
                    if (base == 0u) v.Aj[(unsigned)(v.R * v.RW + c)] = 0u;
                }
        }
    }
}

// ---------------------------------------------------------------- service wave (round 4)
// At W = 224 with 2 pixels per thread a pass of the 512 threads of a frame covers 4 rows x 112 column groups = 448 threads: the
// eighth wave of each frame has no source pixels, and -- waves of a workgroup go to the SIMDs round robin -- both idle waves sit on
// the SAME SIMD: three SIMDs carry four working waves, one carries two, and the kernel is bound by vector-ALU issue on the full
// ones.  The rows entering and leaving a ring (load_stage / stage_rows / flush_rows above: the depth head's exp, the fixed-point ->
// float conversion, the gradient store; ~12 % of a working thread's instructions) need no per-thread state of the sources at all,
// so the idle wave of each frame takes them over as the frame's SERVICE wave: quads of 4 adjacent columns, quad lane + 64 i of the
// rows [lo, hi) (at most SMAX rows: NQ = ceil(SMAX * W / 256) quads per lane), 16-byte global accesses, 8-byte LDS accesses (a ring
// row is RW = W + 2 floats: 8-byte aligned).  Same per-element arithmetic, same ring contents, same gradient bits.
constexpr int kSvcLanes = 64;
template <int NQ> struct SvcRegs { float v[NQ][4]; };
CD_HD bool svc_geometry_ok(const Geo& g) {     // a whole idle wave per frame, rows that split into aligned quads
    return g.ok && g.W % 4 == 0 && g.RW % 2 == 0 && (g.RP * g.CG) % kSvcLanes == 0 && kFrameThreads - g.RP * g.CG >= kSvcLanes;
}
CD_HD int svc_quads(const Geo& g) { return (g.SMAX * (g.W / 4) + kSvcLanes - 1) / kSvcLanes; }

// Everything of a quad that depends only on the lane (its row inside a run of rows, its first column, its byte offset inside the run)
// is computed once per kernel:
template <int NQ> struct SvcLane { unsigned rq[NQ], cw[NQ], go[NQ]; };   // quad lane + 64 i of a run of rows: row, first column (words), byte offset
template <int NQ> CD_HD SvcLane<NQ> make_svc_lane(const View& v, int lane) {
    SvcLane<NQ> s;
    const int QW = v.W >> 2;
    for (int i = 0; i < NQ; ++i) {
        const int qi = lane + kSvcLanes * i, rr = qi / QW;
        s.rq[i] = (unsigned)rr; s.cw[i] = 4u * (unsigned)(qi - rr * QW); s.go[i] = 16u * (unsigned)qi;
    }
    return s;
}
// raw depth of the rows [s_lo, s_hi) that enter the ring (cf. load_stage): requested one item ahead.  The quads of a run are cut off by a
// WAVE-UNIFORM bound (an ordinary item moves 4 of the SMAX = 8 rows); a lane beyond the run's last quad reads that last quad (an
// unconditional load on a clamped offset: a load under a per-lane condition is waited for at once).  Row H (the pad row) has no data.
template <int NQ> CD_HD void svc_load(const View& v, const SvcLane<NQ>& sl, int lane, int s_lo, int s_hi, SvcRegs<NQ>& q) {
    const int nq = ((s_hi < v.H ? s_hi : v.H) - s_lo) * (v.W >> 2);
    const float* srow = v.vj + (size_t)s_lo * (size_t)v.W;      // (wave-uniform)
    (void)lane;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (i * kSvcLanes >= nq) break;                          // wave-uniform
        const VecF<4> a = ldg_depth<4>(srow, umin(sl.go[i], 16u * (unsigned)(nq - 1)));
#pragma unroll
        for (int e = 0; e < 4; ++e) q.v[i][e] = a.v[e];
    }
}

// the rows enter the ring (cf. stage_rows; returns false if a staged depth is not a positive finite number).  The pad column(s) are
// not written: their value (1) is constant once the prologue has written every slot.
template <int MODE, int NQ> CD_HD bool svc_stage(const View& v, const SvcLane<NQ>& sl, int lane, int s_lo, int s_hi, const SvcRegs<NQ>& q) {
    const int nq = (s_hi - s_lo) * (v.W >> 2);
    const bool mirror = ((-s_lo) & (v.R - 1)) < s_hi - s_lo;      // wave-uniform: a row of slot 0 is among [s_lo, s_hi)
    bool good = true;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (i * kSvcLanes >= nq) break;                           // wave-uniform
        if (lane + kSvcLanes * i < nq) {
            const unsigned base = mad24(((unsigned)s_lo + sl.rq[i]) & (unsigned)(v.R - 1), (unsigned)v.RW, sl.cw[i]);
            float d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = to_depth<MODE>(q.v[i][e]);
            if (s_hi > v.H) {                                     // wave-uniform, the last item(s): row H is the pad row
                CD_KEEP_BRANCH();
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = (unsigned)s_lo + sl.rq[i] < (unsigned)v.H ? d[e] : 1.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) good = good && (d[e] > 0.f && d[e] < INFINITY);
            VecF<2> d0, d1;
            d0.v[0] = d[0]; d0.v[1] = d[1]; d1.v[0] = d[2]; d1.v[1] = d[3];
            *reinterpret_cast<VecF<2>*>(&v.Dj[base]) = d0;
            *reinterpret_cast<VecF<2>*>(&v.Dj[base + 2]) = d1;
            if (mirror) {
                CD_KEEP_BRANCH();
                if (base < (unsigned)v.RW) {                      // slot R mirrors slot 0
                    *reinterpret_cast<VecF<2>*>(&v.Dj[(unsigned)(v.R * v.RW) + base]) = d0;
                    *reinterpret_cast<VecF<2>*>(&v.Dj[(unsigned)(v.R * v.RW) + base + 2]) = d1;
                }
            }
        }
    }
    return good;
}
template <int NQ> CD_HD void svc_load(const View& v, int lane, int s_lo, int s_hi, SvcRegs<NQ>& q) {      // (the host emulation: no per-kernel state)
    svc_load<NQ>(v, make_svc_lane<NQ>(v, lane), lane, s_lo, s_hi, q);
}
template <int MODE, int NQ> CD_HD bool svc_stage(const View& v, int lane, int s_lo, int s_hi, const SvcRegs<NQ>& q) {
    return svc_stage<MODE, NQ>(v, make_svc_lane<NQ>(v, lane), lane, s_lo, s_hi, q);
}

// rows [lo, hi) leave the ring (cf. flush_rows).  Round 5: ONE wave per frame does this for everybody, so it is bound by how fast a
// single wave issues instructions and by the latency chain LDS read -> convert -> store.  Measured at the end of round 5 (256 pairs):
// with the sources doing nothing but their loads the call still took 0.194 of 0.208 ms -- this wave WAS the other critical path.
// Hence: everything that depends only on the lane (a quad's row inside the run, its column, its offset in the gradient rows) is
// computed once per kernel (SvcLane); the quads of an item are cut off by a WAVE-UNIFORM bound (an ordinary item moves 4 of the
// SMAX = 8 rows: 4 of 7 quads -- the old per-lane test ran all 7 with an empty exec mask); whether a row of slot 0 is among the rows
// (its mirror slot has to be folded in) is scalar arithmetic on [lo, hi); all LDS reads of the call are issued before the first use.
// The pad column(s) of the accumulators are not touched: a tap in a pad column carries weight 0, its fixed-point addend is 0.
template <int NQ> CD_HD void svc_flush(const View& v, const SvcLane<NQ>& sl, int lane, int lo, int hi) {
    const int nq = (hi - lo) * (v.W >> 2);
    const float unit = v.cj.unit_s;
    float* grow = v.gradj + (size_t)lo * (size_t)v.W;            // (wave-uniform)
    VecU<2> n0[NQ], n1[NQ];
    unsigned base[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (i * kSvcLanes >= nq) break;                           // wave-uniform
        base[i] = mad24(((unsigned)lo + sl.rq[i]) & (unsigned)(v.R - 1), (unsigned)v.RW, sl.cw[i]);
        if (lane + kSvcLanes * i < nq) {
            n0[i] = *reinterpret_cast<const VecU<2>*>(&v.Aj[base[i]]); n1[i] = *reinterpret_cast<const VecU<2>*>(&v.Aj[base[i] + 2]);
        }
    }
    VecU<2> z; z.v[0] = z.v[1] = 0u;
    if (((-lo) & (v.R - 1)) < hi - lo) {      // wave-uniform: a row of slot 0 is among [lo, hi) (once in R rows) -- what the fast pass added to its mirror
        CD_KEEP_BRANCH();
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (i * kSvcLanes >= nq) break;
            if (lane + kSvcLanes * i < nq && base[i] < (unsigned)v.RW) {
                VecU<2>* m0 = reinterpret_cast<VecU<2>*>(&v.Aj[(unsigned)(v.R * v.RW) + base[i]]);
                VecU<2>* m1 = reinterpret_cast<VecU<2>*>(&v.Aj[(unsigned)(v.R * v.RW) + base[i] + 2]);
                const VecU<2> k0 = *m0, k1 = *m1;
                n0[i].v[0] += k0.v[0]; n0[i].v[1] += k0.v[1]; n1[i].v[0] += k1.v[0]; n1[i].v[1] += k1.v[1];
                *m0 = z; *m1 = z;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (i * kSvcLanes >= nq) break;
        if (lane + kSvcLanes * i < nq) {
            *reinterpret_cast<VecU<2>*>(&v.Aj[base[i]]) = z; *reinterpret_cast<VecU<2>*>(&v.Aj[base[i] + 2]) = z;
            VecF<4> g;
            g.v[0] = (float)(int)n0[i].v[0] * unit; g.v[1] = (float)(int)n0[i].v[1] * unit;
            g.v[2] = (float)(int)n1[i].v[0] * unit; g.v[3] = (float)(int)n1[i].v[1] * unit;
            stg_grad<4>(grow, sl.go[i], g);
        }
    }
}
template <int NQ> CD_HD void svc_flush(const View& v, int lane, int lo, int hi) {      // (the host emulation: no per-kernel state)
    svc_flush<NQ>(v, make_svc_lane<NQ>(v, lane), lane, lo, hi);
}

// Evaluate pass q of the source rows [p, p + G) of the wave's frame j: loss partials, direct gradient -> ring j, the 4 tap
// contributions -> ring k.  Closed form: SURVEY.md appendix A.1 (= oracle/cd_oracle_body.inc).
// Env supplies what differs between the GPU and the host emulation:
//   env.add32(p, v)   32-bit LDS atomic add            env.any(x)   wave vote
//   env.push(need, idx, v)   wave-aggregated append to the overflow list (gradient element idx += v)
// The pixels of a thread go through the stages TOGETHER, two at a time (coordinates -> tap reads -> algebra -> atomics):
// independent dependency chains per wave, one wave vote per stage instead of per pixel, and -- the two pixels being adjacent
// registers -- packed fp32 instructions for the algebra.
//
// "Lenient" lanes.  A source with mask == 0 scatters nothing and adds m * |...| = 0 to the loss -- unless a sampled depth is
// zero, negative or not finite (0 * inf = NaN in the reference).  Such sources are mostly the ones whose flow points out of
// the frame, i.e. whose taps are far from the ring.  Instead of sending their whole wave down the exact slow path, they
// sample a resident row (any finite positive depth gives the same 0), and stage_rows watches the ONLY inputs for which this
// could differ from the reference: if any depth of the pair is not a positive finite number, the kernel raises the fallback
// flag and the exact v1 pass recomputes gradient and loss (loss_api.hip).
// (process_rows_sums: the pass itself, its loss partial sums returned; process_rows adds them to the thread's fp64 sums)
template <int MODE, bool REPROJ, int PXT, class Env>
CD_HD void process_rows_sums(const View& v, Env& env, const Lane<PXT>& l, const Inputs<PXT>& in, int p, int q, int wk, int nvk,
                             float& sum_r, float& sum_d) {
    const int y = p + q * v.RP + l.rr;
    const bool rowok = pass_row_ok<PXT>(v, l, p, q);
    const Cam& cj = v.cj;
    const float yf = (float)y;
    const float r1 = -(yf - cj.cy_r) * cj.ify_r;
    const float B0 = cj.M[1] * r1 - cj.M[2], B1 = cj.M[4] * r1 - cj.M[5], B2 = cj.M[7] * r1 - cj.M[8];
    const unsigned own = rowok ? mad24((unsigned)(y & (v.R - 1)), (unsigned)v.RW, l.x0) : 0u;
    const int R = v.R, RW = v.RW, W = v.W, H = v.H;
    VecF<PXT> dv;
    if (rowok) dv = *reinterpret_cast<const VecF<PXT>*>(&v.Dj[own]);
    // at most 2 pixels share the staged registers (a 1024-thread workgroup has 128 VGPRs per lane); PXT = 4 runs two batches
    constexpr int NB = PXT < 2 ? PXT : 2;
    sum_r = 0.f; sum_d = 0.f;
#pragma unroll
    for (int b0 = 0; b0 < PXT; b0 += NB) {
        // ---- stage 0: own depth, sampling coordinates, ring addresses
        unsigned i0[NB], i1[NB];
        float d[NB];
        Taps tp[NB];
        bool need_slow_rd = false;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            d[i] = rowok ? dv.v[b0 + i] : 1.f;
            tp[i] = tap_coords((float)(l.x0 + b0 + i), yf, in.fx[b0 + i], in.fy[b0 + i], cj.sx, cj.sy, W, H);
            // Ring addressing uses the UNCLIPPED neighbours (xa + 1, ya + 1): the pad column / pad row hold a finite depth and
            // the clipped tap's weight is exactly 0 there, so no min() is needed on the fast path.
            const bool inside = (unsigned)(tp[i].ya - wk) < (unsigned)(nvk - 1);      // rows ya, ya + 1 both in [wk, wk + nvk); nvk >= 1
            const bool lenient = !rowok || in.m[b0 + i] == 0.f;
            need_slow_rd = need_slow_rd || (!inside && !lenient);
            const unsigned ra = (unsigned)(inside ? tp[i].ya : wk);       // outside (lenient lanes): the window's first row, always staged
            i0[i] = mad24(ra & (unsigned)(R - 1), (unsigned)RW, (unsigned)tp[i].xa);
            i1[i] = mad24((ra + 1u) & (unsigned)(R - 1), (unsigned)RW, (unsigned)tp[i].xa);
        }
        const bool slow_rd = __builtin_expect(env.any(need_slow_rd), 0);   // (hot path laid out contiguously)

        // ---- stage 1: the 4 depth taps of frame k
        float d00[NB], d01[NB], d10[NB], d11[NB];
        if (!slow_rd) {
#pragma unroll
            for (int i = 0; i < NB; ++i) { d00[i] = v.Dk[i0[i]]; d01[i] = v.Dk[i0[i] + 1]; d10[i] = v.Dk[i1[i]]; d11[i] = v.Dk[i1[i] + 1]; }
        } else {   // some valid source of the wave samples outside the ring: every tap decides for itself (LDS or global)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool exact = rowok && in.m[b0 + i] != 0.f;
                auto tap = [&](int rq, int cq) -> float {
                    const int rel = rq - wk;
                    if ((unsigned)rel < (unsigned)nvk) return v.Dk[(rq & (R - 1)) * RW + cq];
                    return exact ? to_depth<MODE>(v.vk[(unsigned)(rq * W + cq)]) : v.Dk[(wk & (R - 1)) * RW + cq];
                };
                d00[i] = tap(tp[i].ya, tp[i].xa); d01[i] = tap(tp[i].ya, tp[i].xb);
                d10[i] = tap(tp[i].yb, tp[i].xa); d11[i] = tap(tp[i].yb, tp[i].xb);
            }
        }

        // ---- stage 2: the algebra
        float gd[NB], c00[NB], c01[NB], c10[NB], c11[NB];
        bool need_slow = false;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float xf = (float)(l.x0 + b0 + i);
            const float m = in.m[b0 + i], fx = in.fx[b0 + i], fy = in.fy[b0 + i];      // 0 for inactive lanes
            const float a0 = l.a0[b0 + i] + B0, a1 = l.a1[b0 + i] + B1, a2 = l.a2[b0 + i] + B2;
            const float X = d[i] * a0 + cj.c[0], Y = d[i] * a1 + cj.c[1], Z = d[i] * a2 + cj.c[2];
            const float iZ = cd_rcp(Z);
            float gdi = 0.f;   // direct term, in units of ring j
            if (REPROJ) {
                const float mx = xf + fx, my = yf + fy;
                const float ex = (cj.cx_t - cj.fx_t * X * iZ) - mx, ey = (cj.cy_t + cj.fy_t * Y * iZ) - my;
                const float e2 = ex * ex + ey * ey;
                const float ie = e2 > 0.f ? cd_rsq(e2) : 0.f;       // subgradient 0 at e = 0
                sum_r += rowok ? m * (e2 * ie) : 0.f;               // multiply, not select: 0*inf = NaN exactly like the reference
                const float dpx = cj.fx_t * iZ * (X * a2 * iZ - a0), dpy = cj.fy_t * iZ * (a1 - Y * a2 * iZ);
                gdi = cj.drs * m * (ex * dpx + ey * dpy) * ie;
            }
            // z of the sampled point = -(sum of the weighted taps); 1 / z through ONE reciprocal of the positive sum
            const float zsum = d00[i] * tp[i].w00 + d01[i] * tp[i].w01 + d10[i] * tp[i].w10 + d11[i] * tp[i].w11;   // = -z of the sampled point
            const float izs = -cd_rcp(zsum);
            const float dd = iZ - izs;
            sum_d += rowok ? m * fabsf(dd) : 0.f;
            const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
            const float ms = m * sg;
            gdi -= cj.dbs * ms * a2 * iZ * iZ;
            gdi *= depth_jac<MODE>(d[i]);
            const float gz = cj.scs * ms * izs * izs;           // scatter scale, in units of ring k (times 2^20)
            c00[i] = -gz * tp[i].w00 * depth_jac<MODE>(d00[i]); c01[i] = -gz * tp[i].w01 * depth_jac<MODE>(d01[i]);
            c10[i] = -gz * tp[i].w10 * depth_jac<MODE>(d10[i]); c11[i] = -gz * tp[i].w11 * depth_jac<MODE>(d11[i]);
            gd[i] = gdi;
            // inside the fixed-point range?  (a sum, not a max: NaN must fail the test.)  The four tap contributions are -gz w_ij jac(d_ij)
            // with w_ij >= 0: for the exp head (jac(d) = d > 0) their |.| sum IS |gz| * zsum, for the identity head |gz| -- one
            // multiply instead of four |.| and three adds per pixel; the reciprocal head keeps the explicit sum.  (The test only routes
            // a source to the fast or to the exact slow path: a last-bit difference in it cannot change a result.)
            const float csum = MODE == kDepthExp ? fabsf(gz) * zsum : (MODE == kDepthIdentity ? fabsf(gz)
                               : fabsf(c00[i]) + fabsf(c01[i]) + fabsf(c10[i]) + fabsf(c11[i]));
            const bool fits = csum + fabsf(gdi) <= v.limit;
            need_slow = need_slow || (rowok && !fits);
        }
        const bool slow = __builtin_expect(slow_rd || env.any(need_slow), 0);

        // ---- stage 3: 5 integer LDS atomics per pixel
        if (!slow) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (rowok) {
                    env.add32(&v.Aj[own + (unsigned)(b0 + i)], sweep_scaled_to_fixed(gd[i]));
                    if (in.m[b0 + i] != 0.f) {
                        env.add32(&v.Ak[i0[i]], sweep_scaled_to_fixed(c00[i])); env.add32(&v.Ak[i0[i] + 1], sweep_scaled_to_fixed(c01[i]));
                        env.add32(&v.Ak[i1[i]], sweep_scaled_to_fixed(c10[i])); env.add32(&v.Ak[i1[i] + 1], sweep_scaled_to_fixed(c11[i]));
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool dfit = fabsf(gd[i]) <= v.limit;
                if (rowok && dfit) env.add32(&v.Aj[own + (unsigned)(b0 + i)], sweep_scaled_to_fixed(gd[i]));
                env.push(rowok && !dfit, v.gbj + (unsigned)(y * W) + l.x0 + (unsigned)(b0 + i), gd[i] * cj.unit_s);
                const bool a = rowok;
                auto scatter = [&](int rq, int cq, float cv) {
                    const int rel = rq - wk;
                    const bool live = a && cv != 0.f;                        // NaN != 0: it propagates like in the reference
                    const bool in_ring = (unsigned)rel < (unsigned)nvk && fabsf(cv) <= v.limit;
                    if (live && in_ring) env.add32(&v.Ak[(rq & (R - 1)) * RW + cq], sweep_scaled_to_fixed(cv));
                    env.push(live && !in_ring, v.gbk + (unsigned)(rq * W + cq), cv * v.unit_k_s);
                };
                scatter(tp[i].ya, tp[i].xa, c00[i]); scatter(tp[i].ya, tp[i].xb, c01[i]);
                scatter(tp[i].yb, tp[i].xa, c10[i]); scatter(tp[i].yb, tp[i].xb, c11[i]);
            }
        }
    }
}
template <int MODE, bool REPROJ, int PXT, class Env>
CD_HD void process_rows(const View& v, Env& env, Regs<PXT>& r, const Lane<PXT>& l, const Inputs<PXT>& in, int p, int q, int wk, int nvk) {
    float sum_r, sum_d;
    process_rows_sums<MODE, REPROJ, PXT>(v, env, l, in, p, q, wk, nvk, sum_r, sum_d);
    r.acc_r += (double)sum_r; r.acc_d += (double)sum_d;
}

// ---------------------------------------------------------------- the fast source pass (round 5)
// process_rows above is the GENERAL pass: any geometry, any lane may be without a row, every decision per lane.  At 256 pairs it ran
// at 145 vector instructions per pixel-direction, a third of them compares / selects guarding cases that cannot occur when
//   * every lane of a source wave has a row in every pass (H a multiple of RP, G = RP: `rowok` is the wave-uniform p >= 0), and
//   * the ring keeps a COPY of slot 0 behind slot R - 1 (slot R; View::dup): the row under ring row s is always at s + 1, i.e. the four
//     taps of a source are ONE address + the immediates {0, 1, RW, RW + 1} (ds_read2_b32 / ds_add_u32 offsets) -- no second wrap.
// process_rows_fast is the same closed form written for that case, in "target pixel" units -- the camera constants
// arrive premultiplied by the target intrinsics (CamF): X' = fx_t X, Y' = fy_t Y, so
//     ex = (cx_t - mx) - X'/Z        ey = (cy_t - my) + Y'/Z        d ex/dd = (t a2 - a0')/Z     d ey/dd = (a1' - u a2)/Z   (t = X'/Z, u = Y'/Z)
// and without selects on the hot path: 1/|e| = rsq(max(e2, 1e-30)) (e2 is 0 or >= ulp(coordinate)^2 ~ 1e-11: the max only replaces the
// reference's "subgradient 0 at e = 0" select), sign(dd) = med3(dd * 2^126, -1, 1) (exactly -1 / 0 / 1: dd is 0 or a normal number),
// the tap row clamped into the resident window by one med3 (sources outside are either mask-0 "lenient" lanes -- any resident row
// gives their exact 0 -- or raise the wave's slow vote).  Whenever a wave votes "slow" (a valid source outside the ring, a value beyond
// the fixed-point range, NaN) NOTHING of the fast pass is kept and the general pass runs for that wave and pass: the exact paths
// (overflow list, global taps) live in one place.  Fast and general passes differ in the last bits of the algebra (both are within
// the fp32 class of the oracle; tests/test_sweep_cpu.py runs the goldens through both).
struct CamF {
    float kx, ky, kz;                      // a'(x, p + rr) = LaneF::a(x, rr) + p * k  (k = -b1 / fy_r: a' is linear in the row)
    float cX, cY, cZ;                      // fx_t c0, fy_t c1, c2
    float cx_t, cy_t, sx, sy;
    float drs, dbs, scs;
};
CD_HD CamF make_camf(const Cam& c) {
    CamF f;
    f.kx = -(c.fx_t * c.M[1]) * c.ify_r; f.ky = -(c.fy_t * c.M[4]) * c.ify_r; f.kz = -c.M[7] * c.ify_r;
    f.cX = c.fx_t * c.c[0]; f.cY = c.fy_t * c.c[1]; f.cZ = c.c[2];
    f.cx_t = c.cx_t; f.cy_t = c.cy_t; f.sx = c.sx; f.sy = c.sy;
    f.drs = c.drs; f.dbs = c.dbs; f.scs = c.scs;
    return f;
}
template <int PXT> struct LaneF {
    float ax[PXT], ay[PXT], az[PXT];   // a' = (fx_t a0, fy_t a1, a2) of the lane's pixels in row rr (the group at p = 0)
    float xf[PXT];                     // the columns as floats
    float rrf;                         // rr as a float
    unsigned own;                      // rr * RW + x0: element index of the lane's own pixels in ring row 0
    unsigned goff;                     // (rr * W + x0) * 4: byte offset of the lane's pixels inside a row group
    int rr;
};
template <int PXT> CD_HD LaneF<PXT> make_lanef(const View& v, const Lane<PXT>& l) {
    LaneF<PXT> f;
    for (int i = 0; i < PXT; ++i) {
        const float x = (float)(l.x0 + i);
        const float r0 = (x - v.cj.cx_r) * v.cj.ifx_r;
        const float r1 = (v.cj.cy_r - (float)l.rr) * v.cj.ify_r;
        f.ax[i] = cd_fma(v.cj.fx_t * v.cj.M[0], r0, cd_fma(v.cj.fx_t * v.cj.M[1], r1, -(v.cj.fx_t * v.cj.M[2])));
        f.ay[i] = cd_fma(v.cj.fy_t * v.cj.M[3], r0, cd_fma(v.cj.fy_t * v.cj.M[4], r1, -(v.cj.fy_t * v.cj.M[5])));
        f.az[i] = cd_fma(v.cj.M[6], r0, cd_fma(v.cj.M[7], r1, -v.cj.M[8]));
        f.xf[i] = x;
    }
    f.rr = l.rr;
    f.rrf = (float)l.rr;
    f.own = (unsigned)(l.rr * v.RW) + l.x0;
    f.goff = (l.rrW + l.x0) << 2;
    return f;
}
// does this geometry allow the fast pass?  (every lane of a source wave has a row in every item that has a group at all)
CD_HD bool fast_geometry_ok(const Geo& g) { return g.ok && g.PXT == 2 && g.G == g.RP && g.H % g.RP == 0 && g.RP * g.CG <= kFrameThreads; }

// flow / mask of the source rows [p, p + RP): unconditional loads (p >= 0; the caller clamps) -- nothing to select afterwards, so the
// compiler has no reason to wait for them before their first use one item later
template <int PXT> CD_HD void load_inputs_goff(const View& v, unsigned goff, int p, Inputs<PXT>& in) {      // (goff = LaneF::goff)
    const unsigned off = (unsigned)p * ((unsigned)v.W << 2) + goff;
    const VecF<PXT> a = ldg_in<PXT>(v.flj, off), b = ldg_in<PXT>(v.flj + v.HW, off), mm = ldg_in<PXT>(v.mkj, off);
#pragma unroll
    for (int i = 0; i < PXT; ++i) { in.fx[i] = a.v[i]; in.fy[i] = b.v[i]; in.m[i] = mm.v[i]; }
}
template <int PXT> CD_HD void load_inputs_all(const View& v, const LaneF<PXT>& lf, int p, Inputs<PXT>& in) {
    const unsigned off = (unsigned)p * ((unsigned)v.W << 2) + lf.goff;
    const VecF<PXT> a = ldg_in<PXT>(v.flj, off), b = ldg_in<PXT>(v.flj + v.HW, off), mm = ldg_in<PXT>(v.mkj, off);
#pragma unroll
    for (int i = 0; i < PXT; ++i) { in.fx[i] = a.v[i]; in.fy[i] = b.v[i]; in.m[i] = mm.v[i]; }
}

#if defined(__HIP_DEVICE_COMPILE__)
// (VOP3 reads ONE scalar register on gfx9: the upper bound arrives in a vector register, copied once per item)
CD_HD int cd_med3i(int a, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(lo), "v"(hi)); return r; }
#else
CD_HD int cd_med3i(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }
#endif
CD_HD float cd_sign(float x) { return cd_clamp(x * 8.5070592e37f /* 2^126 */, -1.f, 1.f); }

// Evaluate the source rows [p, p + RP) (p >= 0: wave-uniform) of the wave's frame j.  wk / nvk: first resident row of ring k and the
// number of rows usable by this item (Rec::w, Rec::nv of the other frame).  inw (wave-uniform, Rec::inw): the planner guarantees that
// the taps of every valid source of these rows are usable rows of ring k -- no clamp, no vote (a mask-0 lane reads whatever slot
// its row number maps to: a resident depth, finite and positive, and its contributions are exact zeros like everywhere else).
template <int MODE, bool REPROJ, int PXT, class Env>
CD_HD void process_rows_fast(const View& v, const CamF& cf, Env& env, Regs<PXT>& r, const Lane<PXT>& l, const LaneF<PXT>& lf,
                             const Inputs<PXT>& in, int p, int wk, int nvk, bool inw) {
    static_assert(PXT == 2, "the fast pass handles the two adjacent pixels of a lane as one batch");
    const int R = v.R, RW = v.RW, W = v.W, H = v.H;
    r.acc_r += (double)r.pend_r; r.acc_d += (double)r.pend_d;          // the previous fast pass's sums (see Regs)
    const float pf = (float)p;
    const float yf = pf + lf.rrf;                                      // (exact: small integers)
    const unsigned own = lf.own + (unsigned)((p & (R - 1)) * RW);      // rows of a group share p's slot run: (p + rr) & (R - 1) = (p & (R - 1)) + rr
    const VecF<PXT> dv = *reinterpret_cast<const VecF<PXT>*>(&v.Dj[own]);
    // ---- stage 0: sampling coordinates, ring address of the upper-left tap
    float tx[PXT], ty[PXT], mx[PXT], my[PXT];
    unsigned i0[PXT];
    int ya[PXT], xa[PXT];
    unsigned long long slow = 0ull;      // lane masks of the votes (scalar registers: a bool merged across the branches below becomes 0 / 1 selects)
#pragma unroll
    for (int i = 0; i < PXT; ++i) {
        mx[i] = cd_fadd(lf.xf[i], in.fx[i]); my[i] = cd_fadd(yf, in.fy[i]);
        const float ix = cd_clamp(cd_fma(mx[i], cf.sx, -0.5f), 0.f, (float)(W - 1));
        const float iy = cd_clamp(cd_fma(my[i], cf.sy, -0.5f), 0.f, (float)(H - 1));
        tx[i] = cd_fract(ix); ty[i] = cd_fract(iy);
        xa[i] = (int)ix; ya[i] = (int)iy;
    }
    if (inw) {
#pragma unroll
        for (int i = 0; i < PXT; ++i) i0[i] = mad24((unsigned)ya[i] & (unsigned)(R - 1), (unsigned)RW, (unsigned)xa[i]);
    } else {
        const int hi = wk + (nvk >= 2 ? nvk - 2 : 0);                  // last row whose lower neighbour is usable too (scalar)
        const bool window = nvk >= 2;
#pragma unroll
        for (int i = 0; i < PXT; ++i) {
            const int ra = cd_med3i(ya[i], wk, hi);
            slow |= env.vote((!window || ra != ya[i]) && in.m[i] != 0.f);
            i0[i] = mad24((unsigned)ra & (unsigned)(R - 1), (unsigned)RW, (unsigned)xa[i]);
        }
    }
    // ---- stage 1: the 4 depth taps of frame k (the row under slot s is slot s + 1: View::dup)
    float d00[PXT], d01[PXT], d10[PXT], d11[PXT];
#pragma unroll
    for (int i = 0; i < PXT; ++i) { d00[i] = v.Dk[i0[i]]; d01[i] = v.Dk[i0[i] + 1]; d10[i] = v.Dk[i0[i] + (unsigned)RW]; d11[i] = v.Dk[i0[i] + (unsigned)RW + 1]; }
    // ---- stage 2: the algebra
    float gd[PXT], c00[PXT], c01[PXT], c10[PXT], c11[PXT], er[PXT], ed[PXT];
#pragma unroll
    for (int i = 0; i < PXT; ++i) {
        const float d = dv.v[i], m = in.m[i];
        const float a0 = cd_fma(pf, cf.kx, lf.ax[i]), a1 = cd_fma(pf, cf.ky, lf.ay[i]), a2 = cd_fma(pf, cf.kz, lf.az[i]);
        const float X = cd_fma(d, a0, cf.cX), Y = cd_fma(d, a1, cf.cY), Z = cd_fma(d, a2, cf.cZ);
        const float iZ = cd_rcp(Z);
        float g1 = 0.f;      // direct term before the depth head's jacobian, in units of ring j (times 2^20)
        er[i] = 0.f;
        if (REPROJ) {
            const float t = X * iZ, u = Y * iZ;
            const float ex = (cf.cx_t - mx[i]) - t, ey = (cf.cy_t - my[i]) + u;
            const float e2 = cd_fma(ey, ey, ex * ex);
            const float ie = cd_rsq(fmaxf(e2, 1e-30f));
            er[i] = e2 * ie;
            const float q = cd_fma(ey, cd_fma(-u, a2, a1), ex * cd_fma(t, a2, -a0));
            g1 = (m * cf.drs) * ((ie * iZ) * q);
        }
        const float omx = 1.f - tx[i], omy = 1.f - ty[i];
        const float w00 = omx * omy, w01 = tx[i] * omy, w10 = omx * ty[i], w11 = tx[i] * ty[i];
        // exp head: jac(d) = d, so the weighted taps p_ij = w_ij d_ij serve the sampled depth AND the scatter (c_ij = -gz p_ij)
        const float p00 = w00 * d00[i], p01 = w01 * d01[i], p10 = w10 * d10[i], p11 = w11 * d11[i];
        const float zsum = (p00 + p01) + (p10 + p11);             // = -z of the sampled point
        const float izp = cd_rcp(zsum);                           // = -1 / z
        const float dd = iZ + izp;
        ed[i] = fabsf(dd);
        const float ms = m * cd_sign(dd);
        const float g = cd_fma(-(ms * cf.dbs), a2 * (iZ * iZ), g1);
        gd[i] = g * depth_jac<MODE>(d);
        const float ngz = -((ms * cf.scs) * (izp * izp));         // -(scatter scale), in units of ring k (times 2^20)
        float csum;
        if (MODE == kDepthExp) {
            c00[i] = ngz * p00; c01[i] = ngz * p01; c10[i] = ngz * p10; c11[i] = ngz * p11;
            csum = fabsf(ngz) * zsum;
        } else {
            c00[i] = ngz * w00 * depth_jac<MODE>(d00[i]); c01[i] = ngz * w01 * depth_jac<MODE>(d01[i]);
            c10[i] = ngz * w10 * depth_jac<MODE>(d10[i]); c11[i] = ngz * w11 * depth_jac<MODE>(d11[i]);
            csum = MODE == kDepthIdentity ? fabsf(ngz) : fabsf(c00[i]) + fabsf(c01[i]) + fabsf(c10[i]) + fabsf(c11[i]);
        }
        slow |= env.vote(!(csum + fabsf(gd[i]) <= v.limit));              // (NaN fails the test)
    }
    float sum_r = 0.f, sum_d = 0.f;
    if (__builtin_expect(env.any_vote(slow), 0)) {     // the exact paths: the general pass redoes this wave's pass from scratch
        process_rows_sums<MODE, REPROJ, PXT>(v, env, l, in, p, 0, wk, nvk, sum_r, sum_d);
    } else {
    // ---- stage 3: 5 integer LDS atomics per pixel, loss partial sums
#pragma unroll
    for (int i = 0; i < PXT; ++i) {
        env.add32(&v.Aj[own + (unsigned)i], sweep_scaled_to_fixed(gd[i]));
        if (in.m[i] != 0.f) {
            env.add32(&v.Ak[i0[i]], sweep_scaled_to_fixed(c00[i])); env.add32(&v.Ak[i0[i] + 1], sweep_scaled_to_fixed(c01[i]));
            env.add32(&v.Ak[i0[i] + (unsigned)RW], sweep_scaled_to_fixed(c10[i])); env.add32(&v.Ak[i0[i] + (unsigned)RW + 1], sweep_scaled_to_fixed(c11[i]));
        }
        sum_r = cd_fma(in.m[i], er[i], sum_r);           // multiply, not select: 0 * inf = NaN exactly like the reference
        sum_d = cd_fma(in.m[i], ed[i], sum_d);
    }
    }
    r.pend_r = sum_r; r.pend_d = sum_d;
}

// tap-row bounds of one source pixel for the planner (the same tap_coords as the kernels: identical rows)
CD_HD void tap_rows(float xf, float yf, float fx, float fy, int W, int H, int* ya, int* yb) {
    const Taps t = tap_coords(xf, yf, fx, fy, (float)W / (float)(W - 1), (float)H / (float)(H - 1), W, H);
    *ya = t.ya; *yb = t.ya + 1;   // the unclipped neighbour row: what the fast path addresses (row H = the pad row)
}

// the 4 target pixels (unclipped neighbours: column W / row H are the pad column / row) a valid source adds to
CD_HD void tap_targets(float xf, float yf, float fx, float fy, int W, int H, int* xa, int* ya) {
    const Taps t = tap_coords(xf, yf, fx, fy, (float)W / (float)(W - 1), (float)H / (float)(H - 1), W, H);
    *xa = t.xa; *ya = t.ya;
}

}  // namespace sweep
}  // namespace cd
