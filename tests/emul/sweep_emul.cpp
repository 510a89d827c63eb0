// TEST INFRASTRUCTURE ONLY -- sequential host execution of the row-sweep loss kernel's phase functions
// (consistent_depth_amd/csrc/loss_sweep_core.h, the very code hipcc compiles into loss_sweep.hip).
//
// The GPU kernel is: prologue; per item { flush + stage | barrier | prefetch next ; process | barrier }; epilogue.
// Between two barriers the 1024 threads are independent except for LDS atomics, so running them one after the other on
// the host is a valid execution.  This catches plan, ring-index, window and flush mistakes without a GPU; it is built by
// tests/emul/build.py with g++ and loaded only by tests/test_sweep_*_cpu.py.  Nothing under consistent_depth_amd/ links it.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "loss_sweep_core.h"

using namespace cd;
using namespace cd::sweep;

namespace {

int g_order = 0;   // 0: an item's sources run before its rows enter / leave, 1: after (see run_pair)
int g_fast = 0;    // 1: source rows go through process_rows_fast (round 5) wherever the geometry allows it, else the general pass
int g_inw = 1;     // 0: no item claims Rec::inw (the fast pass then clamps and votes everywhere: results must not change)
int g_service = 0; // 1: rows enter / leave through the frame's service wave (svc_* of loss_sweep_core.h) instead of every thread's own columns
constexpr int kEmulNQ = 16;

struct HostEnv {
    std::vector<unsigned>* idx;
    std::vector<float>* val;
    bool force_slow;
    long* n_slow;
    long* n_push;
    bool* degen;
    void degenerate() { *degen = true; }
    static void add32(unsigned* p, int v) { *p += (unsigned)v; }
    bool any(bool x) {
        if (x) ++*n_slow;
        return x || force_slow;
    }   // every thread is its own "wave"
    static unsigned long long vote(bool x) { return x ? 1ull : 0ull; }
    bool any_vote(unsigned long long m) { return any(m != 0ull); }
    void push(bool need, unsigned i, float v) {
        if (need) { idx->push_back(i); val->push_back(v); ++*n_push; }
    }
};

// bounds of the valid sources' tap rows per row group (what sweep_plan_kernel computes on the GPU)
void group_bounds(const Geo& g, const float* flow, const float* mask, short* lo, short* hi) {
    const int H = g.H, W = g.W;
    for (int gi = 0; gi < g.NG; ++gi) { lo[gi] = kNoRow; hi[gi] = -1; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int p = y * W + x;
            if (mask[p] == 0.f) continue;
            int ya, yb;
            tap_rows((float)x, (float)y, flow[p], flow[H * W + p], W, H, &ya, &yb);
            const int gi = y / g.G;
            if (ya < lo[gi]) lo[gi] = (short)ya;
            if (yb > hi[gi]) hi[gi] = (short)yb;
        }
}

// max number of valid sources adding to one target pixel (the fan-in count of sweep_plan_kernel), both directions
int fan_in_of(const Geo& g, const float* ff, const float* fb, const float* mf, const float* mb) {
    const int H = g.H, W = g.W, cw = W + 1;
    int fmax = 0;
    for (int f = 0; f < 2; ++f) {
        std::vector<int> cnt((size_t)(H + 1) * cw, 0);
        const float* fl = f ? fb : ff;
        const float* mk = f ? mb : mf;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int p = y * W + x;
                if (mk[p] == 0.f) continue;
                int xa, ya;
                tap_targets((float)x, (float)y, fl[p], fl[H * W + p], W, H, &xa, &ya);
                for (int dy = 0; dy < 2; ++dy) { ++cnt[(ya + dy) * cw + xa]; ++cnt[(ya + dy) * cw + xa + 1]; }
            }
        for (int c : cnt) fmax = c > fmax ? c : fmax;
    }
    return fmax;
}

template <int MODE, bool REPROJ, int PXT>
int run_pair(const Geo& g, const float* depth_p, const float* ff, const float* fb, const float* mf, const float* mb,
             const PairCam* cams, const Item* raw_items, int n_items, unsigned gbase0, float* grad_p, float* partial,
             HostEnv& env) {
    const float limit = sweep_limit_scaled(fan_in_of(g, ff, fb, mf, mb));
    const int H = g.H, W = g.W, HW = H * W, ring = ring_rows(g) * g.RW;
    std::vector<float> D(2 * (size_t)ring, 0.f);
    std::vector<unsigned> A(2 * (size_t)ring, 0u);
    std::vector<PlanItem> items(n_items);
    {   // the planner's tap-row bounds: which items may claim "every valid tap is a usable row" (Rec::inw)
        std::vector<short> lo(2 * g.NG), hi(2 * g.NG);
        group_bounds(g, ff, mf, lo.data(), hi.data());
        group_bounds(g, fb, mb, lo.data() + g.NG, hi.data() + g.NG);
        if (g_inw) expand_plan(g, raw_items, n_items, items.data(), lo.data(), hi.data());
        else expand_plan(g, raw_items, n_items, items.data());
    }
    View vw[2];
    for (int f = 0; f < 2; ++f) {
        View& v = vw[f];
        v.H = H; v.W = W; v.R = g.R; v.RW = g.RW; v.RP = g.RP; v.G = g.G; v.CG = g.CG; v.HW = (unsigned)HW;
        v.vj = depth_p + (f ? HW : 0); v.vk = depth_p + (f ? 0 : HW);
        v.flj = f ? fb : ff; v.mkj = f ? mb : mf;
        v.gradj = grad_p + (f ? HW : 0);
        v.Aj = A.data() + (f ? ring : 0); v.Ak = A.data() + (f ? 0 : ring);
        v.Dj = D.data() + (f ? ring : 0); v.Dk = D.data() + (f ? 0 : ring);
        v.cj = make_cam(cams[f]); v.unit_k_s = cams[1 - f].unit * (1.f / SWEEP_FX_ONE_F);
        v.gbj = gbase0 + (f ? HW : 0); v.gbk = gbase0 + (f ? 0 : HW);
        v.limit = limit;
    }
    std::vector<Regs<PXT>> regs(kThreads);
    std::vector<Lane<PXT>> lanes(kThreads);
    std::vector<LaneF<PXT>> lanesf(kThreads);
    std::vector<int> fr(kThreads);
    const bool fast = g_fast && PXT == 2 && fast_geometry_ok(g);
    if (g_fast && !fast) { fprintf(stderr, "emul: geometry has no fast pass\n"); return -8; }
    CamF camf[2] = {make_camf(vw[0].cj), make_camf(vw[1].cj)};
    for (int t = 0; t < kThreads; ++t) {
        fr[t] = t / kFrameThreads;
        lanes[t] = make_lane<PXT>(vw[fr[t]], t - fr[t] * kFrameThreads);
        lanesf[t] = make_lanef<PXT>(vw[fr[t]], lanes[t]);
        init_regs<PXT>(regs[t]);
    }
    // prologue: the initial window [0, R) of both rings
    const int init_hi = init_stage_hi(g);
    for (int lo = 0; lo < init_hi; lo += kStagePasses * g.RP) {
        const int hi = lo + kStagePasses * g.RP < init_hi ? lo + kStagePasses * g.RP : init_hi;
        for (int t = 0; t < kThreads; ++t) {
            float sv[kStagePasses][PXT];
            load_stage<PXT>(vw[fr[t]], lanes[t], lo, hi, sv);
            regs[t].bad = !stage_rows<MODE, PXT>(vw[fr[t]], lanes[t], lo, hi, sv) || regs[t].bad;
        }
    }
    // the service-wave form of the rows entering / leaving the rings (loss_sweep_core.h; the GPU uses it for its compile-time geometry)
    const bool service = g_service && svc_geometry_ok(g) && svc_quads(g) <= kEmulNQ;
    if (g_service && !service) { fprintf(stderr, "emul: geometry has no service wave\n"); return -7; }
    const int svc0 = g.RP * g.CG;
    std::vector<SvcRegs<kEmulNQ>> svc[2];
    for (int f = 0; f < 2; ++f) {
        svc[f].resize(kSvcLanes);
        for (auto& q : svc[f]) memset(&q, 0, sizeof(q));
    }
    int wlast[2] = {0, 0};
    for (int it = 0; it < n_items; ++it) {
        for (int f = 0; f < 2; ++f) {
            const Rec& me = items[it].f[f];
            if (me.w < wlast[f] || me.w - wlast[f] > g.SMAX) { fprintf(stderr, "emul: window step out of range\n"); return -2; }
            if (me.s_hi - me.s_lo > kStagePasses * g.RP || me.fl_hi - me.fl_lo > kStagePasses * g.RP) { fprintf(stderr, "emul: stage/flush range too long\n"); return -3; }
            if (me.p >= 0 && (me.p < me.w || me.p + g.G > me.w + me.nv)) { fprintf(stderr, "emul: own rows outside the usable part of the ring\n"); return -4; }
            wlast[f] = me.w;
        }
        // ONE phase per item on the GPU: rows enter, rows leave and the sources are evaluated side by side.  The plan makes the
        // three touch disjoint rows, so any order is a valid execution; the emulation runs the sources FIRST (`order` 0) or
        // LAST (1): a plan that let them depend on this item's entering / leaving rows would give different results.
        const bool more = it + 1 < n_items;
        std::vector<Regs<PXT>> nxt;
        std::vector<SvcRegs<kEmulNQ>> svc_nxt[2];
        if (more) {      // the kernel loads the next item's entering rows while this item runs
            nxt = regs;
            for (int t = 0; t < kThreads; ++t) {
                const Rec& nx = items[it + 1].f[fr[t]];
                if (!service) load_stage<PXT>(vw[fr[t]], lanes[t], nx.s_lo, nx.s_hi, nxt[t].sv);
                else if (g_service == 2) load_stage_nosel<PXT>(vw[fr[t]], lanes[t], nx.s_lo, nx.s_hi, nxt[t].sv);
            }
            if (service)
                for (int f = 0; f < 2; ++f) {
                    svc_nxt[f] = svc[f];
                    for (int sl = 0; sl < kSvcLanes; ++sl) svc_load<kEmulNQ>(vw[f], sl, items[it + 1].f[f].s_lo, items[it + 1].f[f].s_hi, svc_nxt[f][sl]);
                }
        }
        const int passes = g.G > g.RP ? 2 : 1;
        for (int pass = 0; pass < 2; ++pass) {
            if ((pass == 0) == (g_order == 0)) {
                for (int q = 0; q < passes; ++q)
                    for (int t = 0; t < kThreads; ++t) {
                        const int f = fr[t];
                        const Rec& me = items[it].f[f];
                        const Rec& ot = items[it].f[1 - f];
                        Inputs<PXT> in;
                        if constexpr (PXT == 2) {
                            if (fast) {      // the kernel's source waves: lanes with a row (the others are the service wave), items with a group
                                if (lanes[t].on && me.p >= 0) {
                                    load_inputs_all<PXT>(vw[f], lanesf[t], me.p, in);
                                    process_rows_fast<MODE, REPROJ, PXT>(vw[f], camf[f], env, regs[t], lanes[t], lanesf[t], in, me.p, ot.w, ot.nv, me.inw != 0);
                                }
                                continue;
                            }
                        }
                        load_inputs<PXT>(vw[f], lanes[t], me.p, q, in);
                        process_rows<MODE, REPROJ, PXT>(vw[f], env, regs[t], lanes[t], in, me.p, q, ot.w, ot.nv);
                    }
            } else {
                if (service) {     // the frame's service wave does what the threads below do for their own columns
                    for (int f = 0; f < 2; ++f)
                        for (int sl = 0; sl < kSvcLanes; ++sl) {
                            const Rec& me = items[it].f[f];
                            if (g_service == 1) regs[f * kFrameThreads + svc0 + sl].bad = !svc_stage<MODE, kEmulNQ>(vw[f], sl, me.s_lo, me.s_hi, svc[f][sl]) || regs[f * kFrameThreads + svc0 + sl].bad;
                            svc_flush<kEmulNQ>(vw[f], sl, me.fl_lo, me.fl_hi);
                        }
                    if (g_service == 2)     // round 5's split: rows ENTER through every thread's own columns, LEAVE through the service wave
                        for (int t = 0; t < kThreads; ++t) {
                            const Rec& me = items[it].f[fr[t]];
                            regs[t].bad = !stage_rows<MODE, PXT, false>(vw[fr[t]], lanes[t], me.s_lo, me.s_hi, regs[t].sv) || regs[t].bad;   // (as the kernel: no pad-column writes)
                        }
                } else
                for (int t = 0; t < kThreads; ++t) {
                    const Rec& me = items[it].f[fr[t]];
                    regs[t].bad = !stage_rows<MODE, PXT>(vw[fr[t]], lanes[t], me.s_lo, me.s_hi, regs[t].sv) || regs[t].bad;
                    flush_rows<PXT>(vw[fr[t]], lanes[t], me.fl_lo, me.fl_hi);
                }
            }
        }
        if (more) {
            for (int t = 0; t < kThreads; ++t)
                for (int i = 0; i < PXT; ++i)
                    for (int s = 0; s < kStagePasses; ++s) regs[t].sv[s][i] = nxt[t].sv[s][i];
            if (service) { svc[0] = svc_nxt[0]; svc[1] = svc_nxt[1]; }
        }
    }
    for (int f = 0; f < 2; ++f)
        if (wlast[f] < H) { fprintf(stderr, "emul: rows left in ring %d\n", f); return -5; }
    for (size_t i = 0; i < A.size(); ++i)
        if (A[i] != 0u) { fprintf(stderr, "emul: accumulator not drained at %zu\n", i); return -6; }
    double ar[2] = {0, 0}, ad[2] = {0, 0};
    for (int t = 0; t < kThreads; ++t) { ar[fr[t]] += (float)loss_sum_r<PXT>(regs[t]); ad[fr[t]] += (float)loss_sum_d<PXT>(regs[t]); if (regs[t].bad) env.degenerate(); }
    for (int f = 0; f < 2; ++f) { partial[f * 2] = (float)ar[f]; partial[f * 2 + 1] = (float)ad[f]; }
    return 0;
}

template <int MODE, int PXT>
int run_pair_r(bool reproj, const Geo& g, const float* depth_p, const float* ff, const float* fb, const float* mf,
               const float* mb, const PairCam* cams, const Item* items, int n_items, unsigned gbase0, float* grad_p,
               float* partial, HostEnv& env) {
    return reproj ? run_pair<MODE, true, PXT>(g, depth_p, ff, fb, mf, mb, cams, items, n_items, gbase0, grad_p, partial, env)
                  : run_pair<MODE, false, PXT>(g, depth_p, ff, fb, mf, mb, cams, items, n_items, gbase0, grad_p, partial, env);
}

template <int PXT>
int run_pair_m(int mode, bool reproj, const Geo& g, const float* depth_p, const float* ff, const float* fb, const float* mf,
               const float* mb, const PairCam* cams, const Item* items, int n_items, unsigned gbase0, float* grad_p,
               float* partial, HostEnv& env) {
    if (mode == kDepthExp) return run_pair_r<kDepthExp, PXT>(reproj, g, depth_p, ff, fb, mf, mb, cams, items, n_items, gbase0, grad_p, partial, env);
    if (mode == kDepthReciprocal) return run_pair_r<kDepthReciprocal, PXT>(reproj, g, depth_p, ff, fb, mf, mb, cams, items, n_items, gbase0, grad_p, partial, env);
    return run_pair_r<kDepthIdentity, PXT>(reproj, g, depth_p, ff, fb, mf, mb, cams, items, n_items, gbase0, grad_p, partial, env);
}

}  // namespace

extern "C" {

int sweep_emul_fan_in(const int* geo, const float* ff, const float* fb, const float* mf, const float* mb) {
    Geo g; memcpy(&g, geo, sizeof(g));
    return fan_in_of(g, ff, fb, mf, mb);
}

void sweep_emul_set_order(int order) { g_order = order ? 1 : 0; }
void sweep_emul_set_service(int on) { g_service = on; }   // 1: rows enter and leave through the service wave; 2: they leave through it
void sweep_emul_set_fast(int on) { g_fast = on ? 1 : 0; }
void sweep_emul_set_inw(int on) { g_inw = on ? 1 : 0; }
// Rec::inw of every item: out[n_items][2]
int sweep_emul_inw(const int* geo, const float* ff, const float* fb, const float* mf, const float* mb, int* out) {
    Geo g; memcpy(&g, geo, sizeof(g));
    std::vector<short> lo(2 * g.NG), hi(2 * g.NG), suf(2 * (g.NG + 1));
    group_bounds(g, ff, mf, lo.data(), hi.data());
    group_bounds(g, fb, mb, lo.data() + g.NG, hi.data() + g.NG);
    std::vector<Item> items(g.max_items);
    const int n = plan_items(g, lo.data(), hi.data(), suf.data(), items.data());
    if (n <= 0) return n;
    std::vector<PlanItem> ex(n);
    expand_plan(g, items.data(), n, ex.data(), lo.data(), hi.data());
    for (int i = 0; i < n; ++i) { out[i * 2] = ex[i].f[0].inw; out[i * 2 + 1] = ex[i].f[1].inw; }
    return n;
}

// geometry as the kernel would choose it: out[0..11] = the Geo fields; ring_rows > 0 overrides R (to force tiny rings)
int sweep_emul_geo(int H, int W, int pxt, int ring_rows, int* out) {
    Geo g = make_geo(H, W, pxt);
    if (ring_rows > 0 && (ring_rows & (ring_rows - 1))) return 0;   // rings are direct mapped: a power of two
    if (ring_rows > 0 && g.CG > 0) {   // same rules as make_geo with a smaller ring
        g.R = ring_rows;
        g.G = kGroupPasses * g.RP < (g.R - 8) / 3 ? kGroupPasses * g.RP : (g.R - 8) / 3;
        g.SMAX = kStagePasses * g.RP;
        if (g.SMAX > g.R - g.G - 8) g.SMAX = g.R - g.G - 8;
        g.ok = g.G >= 1 && g.SMAX >= g.G;
        g.NG = g.G >= 1 ? (H + g.G - 1) / g.G : 0;
        g.max_items = g.ok ? 4 * g.NG + 3 * ((H + 1 + g.R) / g.SMAX + 2) + 8 : 0;
    }
    memcpy(out, &g, sizeof(g));
    return g.ok;
}

// plan of one pair from its flows / masks; items_out: [max_items][4] shorts (p0, p1, w0, w1).  Returns n_items or < 0.
int sweep_emul_plan(const int* geo, const float* ff, const float* fb, const float* mf, const float* mb, short* items_out,
                    short* lo_out, short* hi_out) {
    Geo g; memcpy(&g, geo, sizeof(g));
    std::vector<short> lo(2 * g.NG), hi(2 * g.NG), suf(2 * (g.NG + 1));
    group_bounds(g, ff, mf, lo.data(), hi.data());
    group_bounds(g, fb, mb, lo.data() + g.NG, hi.data() + g.NG);
    std::vector<Item> items(g.max_items);
    const int n = plan_items(g, lo.data(), hi.data(), suf.data(), items.data());
    for (int i = 0; i < n; ++i) { items_out[i * 4] = items[i].p[0]; items_out[i * 4 + 1] = items[i].p[1]; items_out[i * 4 + 2] = items[i].w[0]; items_out[i * 4 + 3] = items[i].w[1]; }
    if (lo_out) memcpy(lo_out, lo.data(), sizeof(short) * 2 * g.NG);
    if (hi_out) memcpy(hi_out, hi.data(), sizeof(short) * 2 * g.NG);
    return n;
}

// the whole loss + gradient through the emulated sweep kernel.  stats[0] = lanes that asked for the slow path,
// stats[1] = overflow-list entries, stats[2] = total items, stats[3] = 1 if a
// degenerate depth was met (the product then recomputes everything with the exact v1 kernel; the emulation's result is not
// meaningful in that case).
int sweep_emul_loss(const float* depth, const float* ff, const float* fb, const float* mf, const float* mb, const float* intr,
                    const float* extr, float lambda_r, float lambda_b, int mode, int B, int H, int W, int pxt, int ring_rows,
                    int force_slow, float* reproj, float* disp, float* total, float* grad, long* stats) {
    int gi[16];
    if (!sweep_emul_geo(H, W, pxt, ring_rows, gi)) return -1;
    Geo g; memcpy(&g, gi, sizeof(g));
    const int HW = H * W;
    std::vector<float> msum(B * 2);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 2; ++k) {
            const float* m = (k == 0 ? mf : mb) + (size_t)b * HW;
            float s = 0.f;
            for (int p = 0; p < HW; ++p) s += m[p];
            msum[b * 2 + k] = s;
        }
    float fbar[2];
    for (int k = 0; k < 2; ++k) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += intr[(b * 2 + k) * 4] + intr[(b * 2 + k) * 4 + 1];
        fbar[k] = acc / (2.f * (float)B);
    }
    std::vector<PairCam> cams(B * 2);
    for (int b = 0; b < B; ++b) {
        prep_pair(intr + b * 8, extr + b * 24, msum.data() + b * 2, fbar, lambda_r, lambda_b, B, H, W, cams.data() + b * 2);
        // sweep_units_kernel: the accumulator units from a 16 x 16 grid of sample sources per direction, summed in index order
        float D[2] = {0.f, 0.f}, S[2] = {0.f, 0.f};
        int n[2] = {0, 0};
        for (int j = 0; j < 2; ++j)
            for (int t = 0; t < kUnitGrid * kUnitGrid; ++t) {
                const float* dp = depth + (size_t)b * 2 * HW;
                const float* ffb = ff + (size_t)b * 2 * HW; const float* fbb = fb + (size_t)b * 2 * HW;
                const float* mfb = mf + (size_t)b * HW; const float* mbb = mb + (size_t)b * HW;
                UnitSample u;
                if (mode == kDepthExp) u = unit_sample_at<kDepthExp>(cams.data() + b * 2, dp, ffb, fbb, mfb, mbb, H, W, j, t);
                else if (mode == kDepthReciprocal) u = unit_sample_at<kDepthReciprocal>(cams.data() + b * 2, dp, ffb, fbb, mfb, mbb, H, W, j, t);
                else u = unit_sample_at<kDepthIdentity>(cams.data() + b * 2, dp, ffb, fbb, mfb, mbb, H, W, j, t);
                D[j] += u.direct; S[j] += u.scatter; n[j] += u.valid;
            }
        units_from_samples(cams.data() + b * 2, D, S, n);
    }
    std::vector<unsigned> oidx;
    std::vector<float> oval;
    long n_slow = 0, n_push = 0, n_items_total = 0;
    bool degen = false;
    HostEnv env{&oidx, &oval, force_slow != 0, &n_slow, &n_push, &degen};
    std::vector<float> partial(B * 4);
    std::vector<short> items_raw((size_t)g.max_items * 4);
    for (int b = 0; b < B; ++b) {
        const float* ffb = ff + (size_t)b * 2 * HW; const float* fbb = fb + (size_t)b * 2 * HW;
        const float* mfb = mf + (size_t)b * HW; const float* mbb = mb + (size_t)b * HW;
        const int n = sweep_emul_plan(gi, ffb, fbb, mfb, mbb, items_raw.data(), nullptr, nullptr);
        if (n <= 0) return -10;
        if (fan_in_of(g, ffb, fbb, mfb, mbb) > SWEEP_MAX_FAN_IN) return -11;   // the product takes the exact fallback path for such a pair
        n_items_total += n;
        std::vector<Item> items(n);
        for (int i = 0; i < n; ++i) { items[i].p[0] = items_raw[i * 4]; items[i].p[1] = items_raw[i * 4 + 1]; items[i].w[0] = items_raw[i * 4 + 2]; items[i].w[1] = items_raw[i * 4 + 3]; }
        int rc;
        const bool rp = lambda_r > 0.f;
        float* gp = grad + (size_t)b * 2 * HW;
        const float* dp = depth + (size_t)b * 2 * HW;
        const unsigned gb0 = (unsigned)((size_t)b * 2 * HW);
        if (pxt == 1) rc = run_pair_m<1>(mode, rp, g, dp, ffb, fbb, mfb, mbb, cams.data() + b * 2, items.data(), n, gb0, gp, partial.data() + b * 4, env);
        else if (pxt == 2) rc = run_pair_m<2>(mode, rp, g, dp, ffb, fbb, mfb, mbb, cams.data() + b * 2, items.data(), n, gb0, gp, partial.data() + b * 4, env);
        else rc = run_pair_m<4>(mode, rp, g, dp, ffb, fbb, mfb, mbb, cams.data() + b * 2, items.data(), n, gb0, gp, partial.data() + b * 4, env);
        if (rc != 0) return rc;
    }
    for (size_t i = 0; i < oidx.size(); ++i) grad[oidx[i]] += oval[i];   // overflow_apply_kernel
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {   // finalize_pairs_kernel / finalize_total_kernel
        double r[2], q[2];
        for (int k = 0; k < 2; ++k) {
            r[k] = (double)partial[b * 4 + k * 2] * (double)cams[b * 2 + k].invS;
            q[k] = (double)cams[b * 2 + k].fbar * ((double)partial[b * 4 + k * 2 + 1] * (double)cams[b * 2 + k].invS);
        }
        reproj[b] = lambda_r > 0.f ? (float)((double)lambda_r * (r[0] + r[1]) * 0.5) : 0.f;
        disp[b] = lambda_b > 0.f ? (float)((double)lambda_b * (q[0] + q[1]) * 0.5) : 0.f;
        tot += (double)reproj[b] + (double)disp[b];
    }
    total[0] = (float)(tot / (double)B);
    if (stats) { stats[0] = n_slow; stats[1] = n_push; stats[2] = n_items_total; stats[3] = degen ? 1 : 0; }
    return 0;
}

}  // extern "C"
