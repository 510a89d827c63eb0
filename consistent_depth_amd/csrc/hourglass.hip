// The Mannequin-Challenge hourglass depth CNN behind a C handle: plan, buffers, forward and explicit backward in C++, so a host
// without Python can run the whole fine-tuning step through the C ABI
//     cd_hourglass_forward -> cd_consistency_loss_fwd_bwd -> cd_hourglass_backward -> cd_adam_step_flat.
//
// Replaces (reference, /root/reference): monodepth/mannequin_challenge_model.py:52-69 (netG.forward on the un-vendored
// hourglass, architecture SURVEY.md appendix A.3) and the autograd backward of depth_fine_tuning.py:282.
// Same plan as consistent_depth_amd/monodepth/hourglass_engine.py (the Python orchestration of the same kernels, which adds
// side streams and timed launch shapes): every inception is ONE buffer [m1|m2|m3 | b0|o1|o2|o3], its four branch-entry 1x1
// convolutions are one convolution, activations are kept as x_hat (pre-ReLU) and consumers apply ReLU / the stem's affine on
// load, all filters are re-packed by one launch per forward, all weight gradients leave their per-workgroup slices by one
// launch per backward; BatchNorm never passes over an activation (cd_bn_finalize: consumers apply relu(raw * scale + shift) on
// load).  Single stream; capture it in a hipGraph for launch-bound hosts.
//
// Parameters live in ONE flat buffer laid out like consistent_depth_amd.optimizer.FlatAdam does it: the tensors of
// HourglassModel.named_parameters() in order, each starting on a 64-float boundary -- so cd_adam_step_flat updates them in
// one launch and a data-parallel host all-reduces cd_hourglass_grads() in one call.
#include <string.h>

#include <vector>

#include "cd_common.h"

namespace cd {
namespace hg {

constexpr float kEps = 1e-5f, kMomentum = 0.1f;

struct IncCfg { int cin, a0, k[3], mid[3], out[3]; };
static IncCfg inception_cfg(const char* kind) {   // monodepth/hourglass.py INCEPTION
    struct Row { const char* name; IncCfg c; };
    static const Row rows[] = {
        {"A", {128, 16, {3, 7, 11}, {32, 32, 32}, {16, 16, 16}}},  {"A2", {128, 16, {3, 7, 11}, {64, 64, 64}, {16, 16, 16}}},
        {"B", {128, 32, {3, 5, 7}, {32, 32, 32}, {32, 32, 32}}},   {"B2", {128, 32, {3, 5, 7}, {64, 64, 64}, {32, 32, 32}}},
        {"C", {128, 32, {3, 7, 11}, {64, 64, 64}, {32, 32, 32}}},  {"D", {128, 64, {3, 5, 7}, {32, 32, 32}, {64, 64, 64}}},
        {"E", {256, 64, {3, 5, 7}, {32, 32, 32}, {64, 64, 64}}},   {"F", {256, 64, {3, 7, 11}, {64, 64, 64}, {64, 64, 64}}},
        {"G", {256, 32, {3, 5, 7}, {32, 32, 32}, {32, 32, 32}}},
    };
    for (const Row& r : rows) if (strcmp(r.name, kind) == 0) return r.c;
    return IncCfg{};
}

struct Act {   // channels [coff, coff + C) of buf; consumers apply relu(v * scale + shift) on load when flagged
    float* buf = nullptr; int ctot = 0, coff = 0, C = 0, H = 0, W = 0;
    bool relu = false; const float* scale = nullptr; const float* shift = nullptr;
    float* gbuf = nullptr;          // d loss / d (activated value), same geometry as buf
    bool grad_written = false;
};

struct ConvP { int ks, cin, cout; size_t w, b; };   // a convolution's parameters: offsets into the flat buffer
// Descriptor tables are filled before the flat buffers exist: a parameter is recorded as its byte offset dressed up as a
// pointer and rebased once the buffers are allocated (cd_hourglass_create).
static inline float* off_ptr(size_t float_off) { return reinterpret_cast<float*>(static_cast<uintptr_t>(float_off * sizeof(float))); }
static inline size_t ptr_off(const float* p) { return static_cast<size_t>(reinterpret_cast<uintptr_t>(p) / sizeof(float)); }
struct ParamInfo { size_t off; int shape[4]; int ndim; };

struct Unit {          // conv [+ BatchNorm + ReLU]
    ConvP cv{}; int src = -1;                       // index of the source Act
    float* dst = nullptr; int dst_ctot = 0, dst_coff = 0, H = 0, W = 0;
    bool bn = false, affine = false, bn_fused = false;
    size_t gamma = 0, beta = 0;                     // affine BatchNorm parameters (the stem)
    double* stats = nullptr; float* mi = nullptr;   // [16][ctot][2] doubles, [ctot][2] floats of the DESTINATION buffer
    float* sc = nullptr; float* sh = nullptr;       // [ctot] apply-on-load BatchNorm of the destination buffer (cd_bn_finalize)
    float* rm = nullptr; float* rv = nullptr;       // running statistics (C floats each), engine-owned
    float* pk = nullptr; float* pkT = nullptr;      // packed filters (forward, dgrad twin)
    float* wgrad_ws = nullptr; double* sums = nullptr;
    int gact = -1;                                  // Act whose gbuf holds d/d(activated output) (channels at g_coff)
    float* gbuf = nullptr; int g_ctot = 0, g_coff = 0;
};

struct Inception {
    IncCfg c{}; int H = 0, W = 0, M = 0, Co = 0, ctot_entry = 0;
    int src = -1, out = -1;                         // Acts
    float* P = nullptr; float* Pg = nullptr;        // [N][M + Co][H][W]
    double* stats = nullptr; float* mi = nullptr;
    float* sc = nullptr; float* sh = nullptr;       // [M + Co] scale / shift of P: consumers apply relu(raw * scale + shift) while loading
    ConvP entry[4];                                 // m1, m2, m3, b0 (member convolutions of the fused 1x1)
    float* filt = nullptr; float* filtT = nullptr; float* bias = nullptr;   // fused filter (fwd / dgrad), fused bias [ctot_entry]
    float* rm_entry = nullptr; float* rv_entry = nullptr; float* rm_out = nullptr; float* rv_out = nullptr;
    float* wgrad_ws = nullptr; double* sums_entry = nullptr; double* sums_out = nullptr;
    Unit branch[3];
    int mid[3];                                     // Acts of the mid activations
};

enum Kind { kConv, kInception, kPool, kChannels };
struct Node {
    Kind kind = kConv;
    Unit unit;                 // kConv (stem / head)
    Inception inc;             // kInception
    int src = -1, out = -1;    // kPool: Acts
    std::vector<Node> flat, up;   // kChannels
    int lo = -1, hi = -1, x = -1;
};

}  // namespace hg
}  // namespace cd

using namespace cd::hg;

struct cd_hourglass {
    int N = 0, H = 0, W = 0;
    int conv_arith = 0;   // cd_get_conv_arith() at creation: the packed weight-gradient layouts of the plan belong to that mode
    std::vector<void*> allocs;
    std::vector<Act> acts;
    std::vector<Node> steps;
    std::vector<ParamInfo> params;
    size_t n_param_floats = 0, n_bn_floats = 0;
    float* flat_param = nullptr; float* flat_grad = nullptr;
    float* x_in = nullptr; float* pred = nullptr; float* dpred = nullptr;
    // tables / arenas
    std::vector<cd_pack_desc> pack_host; cd_pack_desc* pack_dev = nullptr;
    std::vector<cd_unpack_desc> unpack_host; cd_unpack_desc* unpack_dev = nullptr;
    double* stats_arena = nullptr; size_t stats_doubles = 0, stats_used = 0;
    double* sums_arena = nullptr; size_t sums_doubles = 0, sums_used = 0;
    struct BnMap { float* rm; float* rv; int C; };   // BatchNorm modules in state_dict order -> engine storage
    std::vector<BnMap> bns;
    struct BiasCopy { size_t src; float* dst; int n; };   // member biases -> fused bias vectors (ONE gather launch per forward)
    std::vector<BiasCopy> bias_copies;
    size_t* bias_src_dev = nullptr; float** bias_dst_dev = nullptr; int* bias_n_dev = nullptr;
    bool ok = true;

    template <class T> T* alloc(size_t n, bool zero = false) {
        void* p = nullptr;
        if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
        if (zero && hipMemset(p, 0, (n ? n : 1) * sizeof(T)) != hipSuccess) ok = false;
        allocs.push_back(p);
        return static_cast<T*>(p);
    }
    size_t add_param(int a, int b = 0, int c = 0, int d = 0) {
        ParamInfo p{};
        p.off = n_param_floats;
        p.shape[0] = a; p.shape[1] = b; p.shape[2] = c; p.shape[3] = d;
        p.ndim = d ? 4 : (c ? 3 : (b ? 2 : 1));
        size_t n = (size_t)a * (b ? b : 1) * (c ? c : 1) * (d ? d : 1);
        n_param_floats += (n + 63) / 64 * 64;
        params.push_back(p);
        return p.off;
    }
    ConvP add_conv(int cin, int cout, int ks) {
        ConvP c{ks, cin, cout, 0, 0};
        c.w = add_param(cout, cin, ks, ks);
        c.b = add_param(cout);
        return c;
    }
    int new_act(float* buf, int ctot, int coff, int C, int h, int w, bool relu, bool needs_grad) {
        Act a;
        a.buf = buf; a.ctot = ctot; a.coff = coff; a.C = C; a.H = h; a.W = w; a.relu = relu;
        if (needs_grad) a.gbuf = alloc<float>((size_t)N * ctot * h * w);
        acts.push_back(a);
        return (int)acts.size() - 1;
    }
    float* new_buf(int C, int h, int w) { return alloc<float>((size_t)N * C * h * w); }
};

namespace cd {
namespace hg {

// ---------------------------------------------------------------- plan construction (first pass: parameters, in module order)
struct Builder {
    cd_hourglass& e;
    explicit Builder(cd_hourglass& eng) : e(eng) {}

    double* take_stats(int channels) {
        double* p = e.stats_arena + e.stats_used;
        e.stats_used += (size_t)CD_BN_STAT_SLOTS * channels * 2;
        return p;
    }
    double* take_sums(int channels) {
        double* p = e.sums_arena + e.sums_used;
        e.sums_used += (size_t)2 * channels;
        return p;
    }
    float* packed(int OC, int IC, int ks) {   // zeroed once: padding elements are never written
        return e.alloc<float>(cd_conv2d_packed_weight_floats(OC, IC, ks, 0), true);
    }
    void pack_src(const ConvP& c, float* dst, bool transposed, int OC, int IC, int oc_off, int ic_off) {
        cd_pack_desc d{off_ptr(c.w), dst, c.cout, c.cin, c.ks, transposed ? 1 : 0, OC, IC, oc_off, ic_off};
        e.pack_host.push_back(d);
    }
    void unpack(float* ws, const ConvP& c, int cin, int ks, int cout_total, int row0, int h, int w) {
        int cob = 0, cib = 0, splits = 0;
        cd_conv2d_wgrad_plan(cout_total, cin, ks, e.N, h, w, &cob, &cib, &splits);
        const int cig = (cin + cib - 1) / cib;
        const int stride = ((cout_total + cob - 1) / cob) * cig * ks * ks * cob * cib;
        cd_unpack_desc d{ws, off_ptr(c.w), cin, ks, cob, cib, cig, row0, c.cout, 1, splits, stride};
        e.unpack_host.push_back(d);
    }

    int inception(std::vector<Node>& steps, const char* kind, int x, int H, int W) {
        Node n;
        n.kind = kInception;
        Inception& I = n.inc;
        I.c = inception_cfg(kind);
        I.H = H; I.W = W; I.src = x;
        // parameters in nn.Module order: convs.0 (a0), then per branch [1x1 mid, k x k out]
        ConvP a0 = e.add_conv(I.c.cin, I.c.a0, 1);
        ConvP m[3], o[3];
        for (int i = 0; i < 3; ++i) { m[i] = e.add_conv(I.c.cin, I.c.mid[i], 1); o[i] = e.add_conv(I.c.mid[i], I.c.out[i], I.c.k[i]); }
        I.M = I.c.mid[0] + I.c.mid[1] + I.c.mid[2];
        I.Co = I.c.a0 + I.c.out[0] + I.c.out[1] + I.c.out[2];
        I.ctot_entry = I.M + I.c.a0;
        const int ctot = I.M + I.Co;
        I.P = e.new_buf(ctot, H, W);
        I.Pg = e.new_buf(ctot, H, W);
        I.stats = take_stats(ctot);
        I.mi = e.alloc<float>((size_t)ctot * 2, true);
        I.sc = e.alloc<float>((size_t)ctot, true); I.sh = e.alloc<float>((size_t)ctot, true);
        I.entry[0] = m[0]; I.entry[1] = m[1]; I.entry[2] = m[2]; I.entry[3] = a0;
        I.filt = packed(I.ctot_entry, I.c.cin, 1);
        I.filtT = packed(I.c.cin, I.ctot_entry, 1);
        I.bias = e.alloc<float>(I.ctot_entry);
        int off = 0;
        for (int i = 0; i < 4; ++i) {
            pack_src(I.entry[i], I.filt, false, I.ctot_entry, I.c.cin, off, 0);
            pack_src(I.entry[i], I.filtT, true, I.c.cin, I.ctot_entry, 0, off);
            e.bias_copies.push_back({I.entry[i].b, I.bias + off, I.entry[i].cout});
            off += I.entry[i].cout;
        }
        const int Cout3 = I.c.out[0] + I.c.out[1] + I.c.out[2];
        I.rm_entry = e.alloc<float>(I.ctot_entry, true); I.rv_entry = e.alloc<float>(I.ctot_entry);
        I.rm_out = e.alloc<float>(Cout3, true); I.rv_out = e.alloc<float>(Cout3);
        I.wgrad_ws = e.alloc<float>(cd_conv2d_wgrad_workspace_floats(I.ctot_entry, I.c.cin, 1));
        I.sums_entry = take_sums(I.ctot_entry);
        // BatchNorm modules in state_dict order: convs.0.1 (a0), convs.i.1 (mid i), convs.i.4 (out i)
        e.bns.push_back({I.rm_entry + I.M, I.rv_entry + I.M, I.c.a0});
        int moff = 0, ooff = 0;
        for (int i = 0; i < 3; ++i) {
            e.bns.push_back({I.rm_entry + moff, I.rv_entry + moff, I.c.mid[i]});
            e.bns.push_back({I.rm_out + ooff, I.rv_out + ooff, I.c.out[i]});
            moff += I.c.mid[i]; ooff += I.c.out[i];
        }
        // fused gradient rows -> member weights
        off = 0;
        for (int i = 0; i < 4; ++i) { unpack(I.wgrad_ws, I.entry[i], I.c.cin, 1, I.ctot_entry, off, H, W); off += I.entry[i].cout; }
        // the three k x k units; their BatchNorm sums are consecutive (one joint backward launch)
        I.sums_out = nullptr;
        moff = 0; ooff = I.M + I.c.a0;
        for (int i = 0; i < 3; ++i) {
            Unit& u = I.branch[i];
            u.cv = o[i];
            I.mid[i] = e.new_act(I.P, ctot, moff, I.c.mid[i], H, W, true, false);
            e.acts[I.mid[i]].gbuf = I.Pg;
            e.acts[I.mid[i]].scale = I.sc + moff; e.acts[I.mid[i]].shift = I.sh + moff;
            u.src = I.mid[i];
            u.dst = I.P; u.dst_ctot = ctot; u.dst_coff = ooff; u.H = H; u.W = W;
            u.bn = true; u.bn_fused = true; u.stats = I.stats; u.mi = I.mi; u.sc = I.sc; u.sh = I.sh;
            u.pk = packed(u.cv.cout, u.cv.cin, u.cv.ks); u.pkT = packed(u.cv.cin, u.cv.cout, u.cv.ks);
            pack_src(u.cv, u.pk, false, u.cv.cout, u.cv.cin, 0, 0);
            pack_src(u.cv, u.pkT, true, u.cv.cin, u.cv.cout, 0, 0);
            u.wgrad_ws = e.alloc<float>(cd_conv2d_wgrad_workspace_floats(u.cv.cout, u.cv.cin, u.cv.ks));
            u.sums = take_sums(u.cv.cout);
            if (i == 0) I.sums_out = u.sums;
            u.gbuf = I.Pg; u.g_ctot = ctot; u.g_coff = ooff;
            unpack(u.wgrad_ws, u.cv, u.cv.cin, u.cv.ks, u.cv.cout, 0, H, W);
            moff += I.c.mid[i]; ooff += I.c.out[i];
        }
        I.out = e.new_act(I.P, ctot, I.M, I.Co, H, W, true, false);
        e.acts[I.out].gbuf = I.Pg;
        e.acts[I.out].scale = I.sc + I.M; e.acts[I.out].shift = I.sh + I.M;
        steps.push_back(n);
        return I.out;
    }

    int channels(std::vector<Node>& steps, int level, int x, int H, int W);

    int sequence(std::vector<Node>& steps, const std::vector<const char*>& items, int x, int& H, int& W) {
        for (const char* it : items) {
            if (strcmp(it, "pool") == 0) {
                const Act s = e.acts[x];
                Node n; n.kind = kPool; n.src = x;
                n.out = e.new_act(e.new_buf(s.C, H / 2, W / 2), s.C, 0, s.C, H / 2, W / 2, false, true);
                steps.push_back(n);
                x = n.out; H /= 2; W /= 2;
            } else if (strcmp(it, "up") == 0) {
                // fused with the residual add by the caller
            } else if (it[0] == '#') {
                x = channels(steps, it[1] - '0', x, H, W);
            } else {
                x = inception(steps, it, x, H, W);
            }
        }
        return x;
    }
};

int Builder::channels(std::vector<Node>& steps, int level, int x, int H, int W) {
    // monodepth/hourglass.py CHANNELS: (list[0], list[1]); the side ending in "up" is the low-resolution one
    static const std::vector<const char*> L0[5] = {{}, {"E", "E"}, {"E", "F"}, {"pool", "B", "D", "#2", "E", "G", "up"}, {"pool", "B", "B", "#3", "B2", "A", "up"}};
    static const std::vector<const char*> L1[5] = {{}, {"pool", "E", "E", "E", "up"}, {"pool", "E", "E", "#1", "E", "F", "up"}, {"B", "C"}, {"A2"}};
    Node n;
    n.kind = kChannels;
    n.x = x;
    const bool up_first = strcmp(L0[level].back(), "up") == 0;
    // parameters are registered in module order (list[0] before list[1]) whichever side it is
    int h0 = H, w0 = W, h1 = H, w1 = W;
    const int r0 = sequence(up_first ? n.up : n.flat, L0[level], x, h0, w0);
    const int r1 = sequence(up_first ? n.flat : n.up, L1[level], x, h1, w1);
    n.lo = up_first ? r0 : r1;
    n.hi = up_first ? r1 : r0;
    const Act hi = e.acts[n.hi];
    n.out = e.new_act(e.new_buf(hi.C, H, W), hi.C, 0, hi.C, H, W, false, true);
    steps.push_back(n);
    return n.out;
}

// ---------------------------------------------------------------- small kernels of the engine
__global__ void bias_gather_kernel(const float* __restrict__ flat, const size_t* __restrict__ src, float* const* __restrict__ dst,
                                   const int* __restrict__ n) {
    const size_t s = src[blockIdx.x];
    float* d = dst[blockIdx.x];
    for (int i = threadIdx.x; i < n[blockIdx.x]; i += blockDim.x) d[i] = flat[s + i];
}
// eval mode: statistics synthesised from the running mean / variance in slot 0 (the BatchNorm kernels sum the slots)
__global__ void eval_stats_kernel(double* __restrict__ stats, int ctot, int coff, int C, const float* __restrict__ rm,
                                  const float* __restrict__ rv, double count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = (double)rm[c], v = (double)rv[c];
    for (int s = 0; s < CD_BN_STAT_SLOTS; ++s) {
        stats[((size_t)s * ctot + coff + c) * 2] = s == 0 ? m * count : 0.0;
        stats[((size_t)s * ctot + coff + c) * 2 + 1] = s == 0 ? (v + m * m) * count : 0.0;
    }
}

struct Runner {
    cd_hourglass& e;
    hipStream_t s;
    bool training;
    int rc = CD_OK;
    void chk(int r) { if (rc == CD_OK && r != CD_OK) rc = r; }
    bool grad_mode(int a) { const bool acc = e.acts[a].grad_written; e.acts[a].grad_written = true; return acc; }

    // BatchNorm WITHOUT a pass over the activation: the statistics become the (scale, shift) the consumers apply while loading
    void bn_forward(int ctot, int coff, int C, double* stats, float* mi, float* rm, float* rv, float* sc, float* sh, const float* gamma,
                    const float* beta, int H, int W) {
        const double count = (double)e.N * H * W;
        if (training) {
            chk(cd_bn_finalize(stats, ctot, coff, C, count, kEps, gamma, beta, rm, rv, kMomentum, mi, sc, sh, s));
        } else {
            hipLaunchKernelGGL(eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, stats, ctot, coff, C, rm, rv, count);
            chk(cd_bn_finalize(stats, ctot, coff, C, count, kEps, gamma, beta, nullptr, nullptr, kMomentum, mi, sc, sh, s));
        }
    }
    void unit_forward(Unit& u) {
        const Act& a = e.acts[u.src];
        chk(cd_conv2d_fwd(a.buf, a.ctot, a.coff, u.cv.cin, u.pk, e.flat_param + u.cv.b, a.scale, a.shift, a.relu ? 1 : 0, u.dst, u.dst_ctot,
                          u.dst_coff, u.cv.cout, (u.bn && training) ? u.stats : nullptr, 0, e.N, u.H, u.W, u.cv.ks, s));
        if (u.bn && !u.bn_fused)
            bn_forward(u.dst_ctot, u.dst_coff, u.cv.cout, u.stats, u.mi, u.rm, u.rv, u.sc, u.sh, u.affine ? e.flat_param + u.gamma : nullptr,
                       u.affine ? e.flat_param + u.beta : nullptr, u.H, u.W);
    }
    void unit_backward(Unit& u) {
        const Act& a = e.acts[u.src];
        if (u.bn_fused) {
        } else if (u.bn) {
            chk(cd_bn_relu_bwd(u.gbuf, u.g_ctot, u.g_coff, u.dst, u.dst_ctot, u.dst_coff, u.cv.cout, u.affine ? e.flat_param + u.gamma : nullptr,
                               u.affine ? e.flat_param + u.beta : nullptr, u.mi, u.sc, u.sh, u.sums, 1,
                               u.affine ? e.flat_grad + u.gamma : nullptr, u.affine ? e.flat_grad + u.beta : nullptr, e.N, u.H, u.W, s));
        } else {
            chk(cd_channel_sum(u.gbuf, u.g_ctot, u.g_coff, u.cv.cout, e.N, u.H, u.W, e.flat_grad + u.cv.b, 1, s));
        }
        chk(cd_conv2d_wgrad(a.buf, a.ctot, a.coff, u.cv.cin, a.scale, a.shift, a.relu ? 1 : 0, u.gbuf, u.g_ctot, u.g_coff, u.cv.cout, nullptr,
                            4, u.wgrad_ws, e.N, u.H, u.W, u.cv.ks, s));
        if (a.gbuf)
            chk(cd_conv2d_fwd(u.gbuf, u.g_ctot, u.g_coff, u.cv.cout, u.pkT, nullptr, nullptr, nullptr, 0, a.gbuf, a.ctot, a.coff, u.cv.cin,
                              nullptr, grad_mode(u.src) ? 1 : 0, e.N, u.H, u.W, u.cv.ks, s));
    }
    void forward(std::vector<Node>& steps) {
        for (Node& n : steps) {
            if (n.kind == kConv) {
                unit_forward(n.unit);
            } else if (n.kind == kInception) {
                Inception& I = n.inc;
                const Act& a = e.acts[I.src];
                const int ctot = I.M + I.Co;
                chk(cd_conv2d_fwd(a.buf, a.ctot, a.coff, I.c.cin, I.filt, I.bias, a.scale, a.shift, a.relu ? 1 : 0, I.P, ctot, 0, I.ctot_entry,
                                  training ? I.stats : nullptr, 0, e.N, I.H, I.W, 1, s));
                bn_forward(ctot, 0, I.ctot_entry, I.stats, I.mi, I.rm_entry, I.rv_entry, I.sc, I.sh, nullptr, nullptr, I.H, I.W);           // [m1|m2|m3|b0]
                for (Unit& u : I.branch) unit_forward(u);
                bn_forward(ctot, I.ctot_entry, I.Co - I.c.a0, I.stats, I.mi, I.rm_out, I.rv_out, I.sc, I.sh, nullptr, nullptr, I.H, I.W);    // [o1|o2|o3]
            } else if (n.kind == kPool) {
                const Act& a = e.acts[n.src];
                const Act& y = e.acts[n.out];
                chk(cd_avgpool2_fwd(a.buf, a.ctot, a.coff, a.scale, a.shift, a.relu ? 1 : 0, y.buf, y.ctot, 0, a.C, e.N, a.H, a.W, s));
            } else {
                forward(n.flat);
                forward(n.up);
                const Act& lo = e.acts[n.lo];
                const Act& hi = e.acts[n.hi];
                const Act& o = e.acts[n.out];
                chk(cd_upsample2x_add_fwd(lo.buf, lo.ctot, lo.coff, lo.scale, lo.shift, lo.relu ? 1 : 0, hi.buf, hi.ctot, hi.coff, hi.scale,
                                          hi.shift, hi.relu ? 1 : 0, o.buf, o.ctot, 0, lo.C, e.N, lo.H, lo.W, s));
            }
        }
    }
    void backward(std::vector<Node>& steps) {
        for (size_t i = steps.size(); i-- > 0;) {
            Node& n = steps[i];
            if (n.kind == kConv) {
                unit_backward(n.unit);
            } else if (n.kind == kInception) {
                Inception& I = n.inc;
                const Act& a = e.acts[I.src];
                const int ctot = I.M + I.Co;
                // the concat output's gradient is complete: k x k convolutions first (they fill the gradient of the mid
                // activations), then the fused entry convolution
                chk(cd_bn_relu_bwd(I.Pg, ctot, I.ctot_entry, I.P, ctot, I.ctot_entry, I.Co - I.c.a0, nullptr, nullptr, I.mi, I.sc, I.sh,
                                   I.sums_out, 1, nullptr, nullptr, e.N, I.H, I.W, s));
                for (Unit& u : I.branch) unit_backward(u);
                chk(cd_bn_relu_bwd(I.Pg, ctot, 0, I.P, ctot, 0, I.ctot_entry, nullptr, nullptr, I.mi, I.sc, I.sh, I.sums_entry, 1, nullptr,
                                   nullptr, e.N, I.H, I.W, s));
                chk(cd_conv2d_wgrad(a.buf, a.ctot, a.coff, I.c.cin, a.scale, a.shift, a.relu ? 1 : 0, I.Pg, ctot, 0, I.ctot_entry, nullptr, 4,
                                    I.wgrad_ws, e.N, I.H, I.W, 1, s));
                if (a.gbuf)
                    chk(cd_conv2d_fwd(I.Pg, ctot, 0, I.ctot_entry, I.filtT, nullptr, nullptr, nullptr, 0, a.gbuf, a.ctot, a.coff, I.c.cin, nullptr,
                                      grad_mode(I.src) ? 1 : 0, e.N, I.H, I.W, 1, s));
            } else if (n.kind == kPool) {
                const Act& a = e.acts[n.src];
                const Act& y = e.acts[n.out];
                chk(cd_avgpool2_bwd(y.gbuf, y.ctot, 0, a.gbuf, a.ctot, a.coff, a.C, e.N, a.H, a.W, grad_mode(n.src) ? 1 : 0, s));
            } else {
                const Act& lo = e.acts[n.lo];
                const Act& hi = e.acts[n.hi];
                const Act& o = e.acts[n.out];
                chk(cd_add_slice(o.gbuf, o.ctot, 0, hi.gbuf, hi.ctot, hi.coff, hi.C, e.N, hi.H, hi.W, grad_mode(n.hi) ? 1 : 0, s));
                chk(cd_upsample2x_bwd(o.gbuf, o.ctot, 0, lo.gbuf, lo.ctot, lo.coff, lo.C, e.N, lo.H, lo.W, grad_mode(n.lo) ? 1 : 0, s));
                backward(n.flat);
                backward(n.up);
            }
        }
    }
};

}  // namespace hg
}  // namespace cd

extern "C" {

int cd_hourglass_create(int N, int H, int W, cd_hourglass** out) {
    if (!out || N <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) return CD_ERR_INVALID_ARG;
    cd_hourglass* e = new cd_hourglass();
    e->N = N; e->H = H; e->W = W;
    e->conv_arith = cd_get_conv_arith();
    // pass 0: sizes that the builder needs up front (arenas are sized generously: 16384 statistic channels like the Python engine)
    e->stats_doubles = (size_t)16384 * CD_BN_STAT_SLOTS * 2;
    e->stats_arena = e->alloc<double>(e->stats_doubles, true);
    e->sums_doubles = (size_t)2 * 16384;
    e->sums_arena = e->alloc<double>(e->sums_doubles, true);
    // the parameter layout is known before any buffer is: build with placeholder bases, then rebase the descriptors
    cd::hg::Builder b(*e);
    e->x_in = e->new_buf(3, H, W);
    const int a_in = e->new_act(e->x_in, 3, 0, 3, H, W, false, false);
    // stem: seq.0 (conv 7x7 3 -> 128), seq.1 (BatchNorm2d affine), ReLU
    Node stem;
    stem.kind = kConv;
    Unit& su = stem.unit;
    su.cv = e->add_conv(3, 128, 7);
    su.gamma = e->add_param(128); su.beta = e->add_param(128);
    su.src = a_in;
    su.dst = e->new_buf(128, H, W); su.dst_ctot = 128; su.dst_coff = 0; su.H = H; su.W = W;
    su.bn = true; su.affine = true;
    su.stats = b.take_stats(128);
    su.mi = e->alloc<float>(256, true);
    su.sc = e->alloc<float>(128, true); su.sh = e->alloc<float>(128, true);
    su.rm = e->alloc<float>(128, true); su.rv = e->alloc<float>(128);
    e->bns.push_back({su.rm, su.rv, 128});
    su.pk = b.packed(128, 3, 7); su.pkT = nullptr;
    su.wgrad_ws = e->alloc<float>(cd_conv2d_wgrad_workspace_floats(128, 3, 7));
    su.sums = b.take_sums(128);
    const int a_stem = e->new_act(su.dst, 128, 0, 128, H, W, true, true);
    su.gbuf = e->acts[a_stem].gbuf; su.g_ctot = 128; su.g_coff = 0;
    e->steps.push_back(stem);
    const int feat = b.channels(e->steps, 4, a_stem, H, W);
    // uncertainty_layer.0 (unused head: parameters exist, never evaluated), pred_layer
    (void)e->add_conv(64, 1, 3);
    Node head;
    head.kind = kConv;
    Unit& hu = head.unit;
    hu.cv = e->add_conv(64, 1, 3);
    hu.src = feat;
    e->pred = e->new_buf(1, H, W); e->dpred = e->new_buf(1, H, W);
    hu.dst = e->pred; hu.dst_ctot = 1; hu.dst_coff = 0; hu.H = H; hu.W = W;
    hu.pk = b.packed(1, 64, 3); hu.pkT = b.packed(64, 1, 3);
    hu.wgrad_ws = e->alloc<float>(cd_conv2d_wgrad_workspace_floats(1, 64, 3));
    hu.gbuf = e->dpred; hu.g_ctot = 1; hu.g_coff = 0;
    e->steps.push_back(head);
    // now the flat parameter / gradient buffers exist: rebase everything that points into them
    e->flat_param = e->alloc<float>(e->n_param_floats, true);
    e->flat_grad = e->alloc<float>(e->n_param_floats, true);
    {
        Node& st = e->steps.front();
        st.unit.stats = su.stats;
        e->acts[a_stem].scale = st.unit.sc;      // gamma * invstd, beta - gamma * mean * invstd: written by cd_bn_finalize
        e->acts[a_stem].shift = st.unit.sh;
        cd_pack_desc d{off_ptr(st.unit.cv.w), st.unit.pk, 128, 3, 7, 0, 128, 3, 0, 0};
        e->pack_host.push_back(d);
        Node& hd = e->steps.back();
        cd_pack_desc d1{off_ptr(hd.unit.cv.w), hd.unit.pk, 1, 64, 3, 0, 1, 64, 0, 0};
        cd_pack_desc d2{off_ptr(hd.unit.cv.w), hd.unit.pkT, 1, 64, 3, 1, 64, 1, 0, 0};
        e->pack_host.push_back(d1);
        e->pack_host.push_back(d2);
        int cob = 0, cib = 0, splits = 0;
        cd_conv2d_wgrad_plan(128, 3, 7, N, H, W, &cob, &cib, &splits);
        int cig = (3 + cib - 1) / cib;
        cd_unpack_desc u1{st.unit.wgrad_ws, off_ptr(st.unit.cv.w), 3, 7, cob, cib, cig, 0, 128, 1, splits,
                          ((128 + cob - 1) / cob) * cig * 49 * cob * cib};
        e->unpack_host.push_back(u1);
        cd_conv2d_wgrad_plan(1, 64, 3, N, H, W, &cob, &cib, &splits);
        cig = (64 + cib - 1) / cib;
        cd_unpack_desc u2{hd.unit.wgrad_ws, off_ptr(hd.unit.cv.w), 64, 3, cob, cib, cig, 0, 1, 1, splits,
                          ((1 + cob - 1) / cob) * cig * 9 * cob * cib};
        e->unpack_host.push_back(u2);
    }
    // descriptors created by the builder hold (nullptr + offset) pointers: add the real bases
    for (cd_pack_desc& d : e->pack_host) d.w = e->flat_param + ptr_off(d.w);
    for (cd_unpack_desc& d : e->unpack_host) d.dw = e->flat_grad + ptr_off(d.dw);
    {
        std::vector<size_t> bs; std::vector<float*> bd; std::vector<int> bn;
        for (const cd_hourglass::BiasCopy& b : e->bias_copies) { bs.push_back(b.src); bd.push_back(b.dst); bn.push_back(b.n); }
        e->bias_src_dev = e->alloc<size_t>(bs.size()); e->bias_dst_dev = e->alloc<float*>(bd.size()); e->bias_n_dev = e->alloc<int>(bn.size());
        if (e->ok && (hipMemcpy(e->bias_src_dev, bs.data(), bs.size() * sizeof(size_t), hipMemcpyHostToDevice) != hipSuccess ||
                      hipMemcpy(e->bias_dst_dev, bd.data(), bd.size() * sizeof(float*), hipMemcpyHostToDevice) != hipSuccess ||
                      hipMemcpy(e->bias_n_dev, bn.data(), bn.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess))
            e->ok = false;
    }
    e->pack_dev = e->alloc<cd_pack_desc>(e->pack_host.size());
    e->unpack_dev = e->alloc<cd_unpack_desc>(e->unpack_host.size());
    if (e->ok) {
        if (hipMemcpy(e->pack_dev, e->pack_host.data(), e->pack_host.size() * sizeof(cd_pack_desc), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e->unpack_dev, e->unpack_host.data(), e->unpack_host.size() * sizeof(cd_unpack_desc), hipMemcpyHostToDevice) != hipSuccess)
            e->ok = false;
    }
    e->n_bn_floats = 0;
    for (const cd_hourglass::BnMap& m : e->bns) e->n_bn_floats += 2 * (size_t)m.C;
    if (!e->ok || e->stats_used > e->stats_doubles || e->sums_used > e->sums_doubles) {
        for (void* p : e->allocs) (void)hipFree(p);
        delete e;
        return CD_ERR_LAUNCH;
    }
    // running_var starts at 1 like nn.BatchNorm2d (running_mean at 0: zero-initialised above)
    {
        std::vector<float> ones;
        for (const cd_hourglass::BnMap& m : e->bns) {
            ones.assign(m.C, 1.f);
            (void)hipMemcpy(m.rv, ones.data(), sizeof(float) * m.C, hipMemcpyHostToDevice);
        }
    }
    *out = e;
    return CD_OK;
}

int cd_hourglass_destroy(cd_hourglass* e) {
    if (!e) return CD_ERR_INVALID_ARG;
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
    return CD_OK;
}

size_t cd_hourglass_param_floats(const cd_hourglass* e) { return e ? e->n_param_floats : 0; }
size_t cd_hourglass_bn_floats(const cd_hourglass* e) { return e ? e->n_bn_floats : 0; }
int cd_hourglass_param_count(const cd_hourglass* e) { return e ? (int)e->params.size() : 0; }
int cd_hourglass_param_info(const cd_hourglass* e, int index, size_t* offset, int* shape4) {
    if (!e || index < 0 || index >= (int)e->params.size() || !offset || !shape4) return CD_ERR_INVALID_ARG;
    *offset = e->params[index].off;
    for (int i = 0; i < 4; ++i) shape4[i] = e->params[index].shape[i];
    return CD_OK;
}
float* cd_hourglass_params(cd_hourglass* e) { return e ? e->flat_param : nullptr; }
float* cd_hourglass_grads(cd_hourglass* e) { return e ? e->flat_grad : nullptr; }

int cd_hourglass_load_state(cd_hourglass* e, const float* params_flat, const float* bn_flat, void* stream) {
    if (!e || !params_flat) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(e->flat_param, params_flat, sizeof(float) * e->n_param_floats, hipMemcpyDefault, s) != hipSuccess) return CD_ERR_LAUNCH;
    if (bn_flat) {
        size_t off = 0;
        for (const cd_hourglass::BnMap& m : e->bns) {
            if (hipMemcpyAsync(m.rm, bn_flat + off, sizeof(float) * m.C, hipMemcpyDefault, s) != hipSuccess) return CD_ERR_LAUNCH;
            if (hipMemcpyAsync(m.rv, bn_flat + off + m.C, sizeof(float) * m.C, hipMemcpyDefault, s) != hipSuccess) return CD_ERR_LAUNCH;
            off += 2 * (size_t)m.C;
        }
    }
    return CD_OK;
}

int cd_hourglass_save_state(cd_hourglass* e, float* params_flat, float* bn_flat, void* stream) {
    if (!e || (!params_flat && !bn_flat)) return CD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (params_flat && hipMemcpyAsync(params_flat, e->flat_param, sizeof(float) * e->n_param_floats, hipMemcpyDefault, s) != hipSuccess)
        return CD_ERR_LAUNCH;
    if (bn_flat) {
        size_t off = 0;
        for (const cd_hourglass::BnMap& m : e->bns) {
            if (hipMemcpyAsync(bn_flat + off, m.rm, sizeof(float) * m.C, hipMemcpyDefault, s) != hipSuccess) return CD_ERR_LAUNCH;
            if (hipMemcpyAsync(bn_flat + off + m.C, m.rv, sizeof(float) * m.C, hipMemcpyDefault, s) != hipSuccess) return CD_ERR_LAUNCH;
            off += 2 * (size_t)m.C;
        }
    }
    return CD_OK;
}

/* n floats, any mix of host / device pointers (hipMemcpyDefault), stream ordered: lets a binding without its own HIP runtime
 * handle read the engine-owned buffers (cd_hourglass_params / cd_hourglass_grads). */
int cd_copy_f32(const float* src, float* dst, size_t n, void* stream) {
    if (!src || !dst) return CD_ERR_INVALID_ARG;
    return hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDefault, (hipStream_t)stream) == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

int cd_hourglass_zero_grad(cd_hourglass* e, void* stream) {
    if (!e) return CD_ERR_INVALID_ARG;
    return hipMemsetAsync(e->flat_grad, 0, sizeof(float) * e->n_param_floats, (hipStream_t)stream) == hipSuccess ? CD_OK : CD_ERR_LAUNCH;
}

int cd_hourglass_forward(cd_hourglass* e, const float* images, float* pred, int training, void* stream) {
    if (!e || !images || !pred) return CD_ERR_INVALID_ARG;
    if (cd_get_conv_arith() != e->conv_arith) return CD_ERR_INVALID_ARG;   // the mode changed under the handle (its launch shapes and
                                                                           // weight-gradient plans belong to the old one): re-create it
    hipStream_t s = (hipStream_t)stream;
    const size_t px = (size_t)e->N * e->H * e->W;
    if (hipMemcpyAsync(e->x_in, images, sizeof(float) * 3 * px, hipMemcpyDeviceToDevice, s) != hipSuccess) return CD_ERR_LAUNCH;
    if (hipMemsetAsync(e->stats_arena, 0, sizeof(double) * e->stats_used, s) != hipSuccess) return CD_ERR_LAUNCH;
    int rc = cd_conv2d_pack_weights_table(e->pack_dev, (int)e->pack_host.size(), s);
    if (rc != CD_OK) return rc;
    hipLaunchKernelGGL(cd::hg::bias_gather_kernel, dim3((unsigned)e->bias_copies.size()), dim3(64), 0, s, e->flat_param, e->bias_src_dev,
                       e->bias_dst_dev, e->bias_n_dev);   // the fused 1x1 biases follow the parameters
    cd::hg::Runner r{*e, s, training != 0};
    r.forward(e->steps);
    if (r.rc != CD_OK) return r.rc;
    if (hipMemcpyAsync(pred, e->pred, sizeof(float) * px, hipMemcpyDeviceToDevice, s) != hipSuccess) return CD_ERR_LAUNCH;
    return CD_OK;
}

int cd_hourglass_backward(cd_hourglass* e, const float* dpred, void* stream) {
    if (!e || !dpred) return CD_ERR_INVALID_ARG;
    if (cd_get_conv_arith() != e->conv_arith) return CD_ERR_INVALID_ARG;   // the mode changed under the handle: re-create it
    hipStream_t s = (hipStream_t)stream;
    const size_t px = (size_t)e->N * e->H * e->W;
    if (hipMemcpyAsync(e->dpred, dpred, sizeof(float) * px, hipMemcpyDeviceToDevice, s) != hipSuccess) return CD_ERR_LAUNCH;
    if (hipMemsetAsync(e->sums_arena, 0, sizeof(double) * e->sums_used, s) != hipSuccess) return CD_ERR_LAUNCH;
    for (Act& a : e->acts) a.grad_written = false;   // first gradient contribution overwrites, later ones accumulate
    cd::hg::Runner r{*e, s, true};
    r.backward(e->steps);
    if (r.rc != CD_OK) return r.rc;
    return cd_conv2d_wgrad_unpack_table(e->unpack_dev, (int)e->unpack_host.size(), s);   // every weight gradient in one launch
}

}  // extern "C"
