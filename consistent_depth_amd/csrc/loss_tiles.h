// Tile / window machinery shared by the gradient kernels of the consistency loss
// (loss_slab.hip: v3 single evaluation + slab reduce; loss_tiles.hip builds the table).
#pragma once
#include "loss_common.h"

namespace cd {

constexpr int TW = 32, TH = 32;            // owned tile
constexpr int WMAXW = 64, WMAXH = 64;      // cap of the staged window of the other frame
constexpr int EXPAND_V2 = 2;               // v2 only: px slack around the table window (beyond it: overflow list)
constexpr int MAXT_LDS = 512;              // window table of one plane kept in LDS up to this many tiles
constexpr int SBW = TW + 2, SBH = TH + 2;  // own tile + 1 px halo

struct TileWin { short x0, y0, w, h; };    // window in the OTHER frame's pixel grid (w*h may be 0)

struct Overflow {          // global overflow list (workspace)
    int count;             // number of pushes attempted
    int cap;               // capacity of idx/val
    int fallback;          // set by overflow_apply when count > cap or `degenerate` is set
    int degenerate;        // set by the row-sweep kernel: a depth of the batch is not a positive finite number, the exact v1
                           // pass must produce gradient AND loss (loss_sweep_core.h, "lenient" lanes)
};

// ---------------------------------------------------------------- fixed-point scatter accumulator
constexpr double FX_ONE = 1099511627776.0;           // 2^40
constexpr double FX_MAGIC = 6755399441055744.0;      // 1.5 * 2^52: adding it rounds to an integer in the low mantissa bits
constexpr float FX_LIMIT = 2047.f;                   // saturation (|x| * 2^40 must stay below 2^51)

__device__ __forceinline__ unsigned long long to_fixed(float c) {
    c = fminf(fmaxf(c, -FX_LIMIT), FX_LIMIT);
    const double d = __fma_rn((double)c, FX_ONE, FX_MAGIC);
    return (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(FX_MAGIC));
}
__device__ __forceinline__ float from_fixed(unsigned long long v) {
    return (float)((double)(long long)v * (1.0 / FX_ONE));
}


__device__ __forceinline__ bool in_win(const TileWin& w, int x, int y) {
    return (unsigned)(x - w.x0) < (unsigned)w.w && (unsigned)(y - w.y0) < (unsigned)w.h;
}

// v2 scans a margin around the predicted window (slack for forward/backward flow inconsistency); every user
// of the table applies the SAME deterministic expansion, so "will the owner see me" and the owner's scan agree.
__device__ __forceinline__ TileWin expand_win(TileWin w, int e, int W, int H) {
    if (w.w == 0 || w.h == 0 || e == 0) return w;
    int x0 = max((int)w.x0 - e, 0), y0 = max((int)w.y0 - e, 0);
    int x1 = min((int)w.x0 + w.w - 1 + e, W - 1), y1 = min((int)w.y0 + w.h - 1 + e, H - 1);
    int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
    if (ww > WMAXW) { x0 += (ww - WMAXW) / 2; ww = WMAXW; }
    if (wh > WMAXH) { y0 += (wh - WMAXH) / 2; wh = WMAXH; }
    TileWin r; r.x0 = (short)x0; r.y0 = (short)y0; r.w = (short)ww; r.h = (short)wh;
    return r;
}

// Wave-aggregated append to the overflow list: ONE returning atomic per wave per call (a same-address
// returning atomic costs ~11 ns on this chip, so per-lane pushes would serialise).  Must be called by all
// lanes of the wave (convergent); `need` selects the lanes that append.
__device__ __forceinline__ void ovf_push(bool need, Overflow* ovf, unsigned* oidx, float* oval, unsigned idx, float v) {
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return;  // wave-uniform
    const int lane = threadIdx.x & (kWave - 1);
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&ovf->count, (int)__popcll(mask));
    base = __shfl(base, leader, kWave);
    if (need) {
        const int i = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
        if (i < ovf->cap) { oidx[i] = idx; oval[i] = v; }
    }
}

}  // namespace cd
