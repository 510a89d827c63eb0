// TEST INFRASTRUCTURE ONLY.  Host check of the weight-gradient kernel's staging map (consistent_depth_amd/csrc/wgrad_stage_map.h,
// the code wgrad_split.hip compiles for the device): for every block shape, walk all (thread, slot) pairs and compare with the
// CONSUMER's view of the LDS tile -- the fragment addresses ws_wave reads (wgrad_split.hip):
//   dY: word 3 SPX + split * SPD + co * PSD + y * 16 + pixel / 2          (co < 16 COT, y < TY, pixel < 32)
//   X : word           split * SPX + ci * PSX + row * 24 + pixel / 2      (ci < 16, row < ROWS, pixel < 48)
// Every pixel pair of both tiles must be written by exactly one (thread, slot), nothing outside the three plane sets.
#include <cstdio>
#include <vector>
#include "wgrad_stage_map.h"

template <int KS, int COT> static int check() {
    using Cfg = cd::WsCfg<KS, COT>;
    const int words = 3 * (Cfg::SPX + Cfg::SPD);
    std::vector<int> hits(words, 0);
    int bad = 0, dead = 0;
    for (int t = 0; t < Cfg::NT; ++t)
        for (int j = 0; j < Cfg::NQ; ++j) {
            const cd::WsQuad q = cd::ws_stage_quad<KS, COT>(t, j);
            if (!q.live) { ++dead; continue; }
            const int rows = q.is_dy ? Cfg::TY : Cfg::ROWS, quads = q.is_dy ? 8 : 12, chans = q.is_dy ? 16 * COT : 16;
            if (q.c < 0 || q.c >= chans || q.r < 0 || q.r >= rows || q.q < 0 || q.q >= quads) { ++bad; continue; }
            // the consumer's address of pixel pair 2q (and 2q + 1) of (channel c, row r)
            const int expect = q.is_dy ? 3 * Cfg::SPX + q.c * Cfg::PSD + q.r * 16 + (4 * q.q) / 2 : q.c * Cfg::PSX + q.r * 24 + (4 * q.q) / 2;
            if (q.lds_word != expect || q.split_words != (q.is_dy ? Cfg::SPD : Cfg::SPX) || (q.lds_word & 1)) ++bad;   // (8-byte stores)
            for (int sp = 0; sp < 3; ++sp)
                for (int w = 0; w < 2; ++w) {
                    const int a = q.lds_word + sp * q.split_words + w;
                    if (a < 0 || a >= words) ++bad; else ++hits[a];
                }
        }
    // coverage: every word a fragment read can touch is written once; the pad words of the planes are never written
    long covered = 0;
    for (int sp = 0; sp < 3; ++sp) {
        for (int c = 0; c < 16 * COT; ++c)
            for (int y = 0; y < Cfg::TY; ++y)
                for (int w = 0; w < 16; ++w) { const int a = 3 * Cfg::SPX + sp * Cfg::SPD + c * Cfg::PSD + y * 16 + w; if (hits[a] != 1) ++bad; ++covered; }
        for (int c = 0; c < 16; ++c)
            for (int r = 0; r < Cfg::ROWS; ++r)
                for (int w = 0; w < 24; ++w) { const int a = sp * Cfg::SPX + c * Cfg::PSX + r * 24 + w; if (hits[a] != 1) ++bad; ++covered; }
    }
    long total = 0;
    for (int a = 0; a < words; ++a) total += hits[a];
    if (total != covered) ++bad;
    // the dead slots are exactly the ones beyond the last X quad, and the allocation has room for the 32 affine floats behind the planes
    if (dead != Cfg::NQ * Cfg::NT - Cfg::QT) ++bad;
    if ((size_t)words * 4 + 128 != Cfg::LDS || Cfg::LDS > 160 * 1024) ++bad;
    std::printf("KS=%d COT=%d NT=%d NQ=%d (dY slots %d) words=%d LDS=%zu dead=%d bad=%d\n", KS, COT, Cfg::NT, Cfg::NQ, Cfg::JDY, words, Cfg::LDS, dead, bad);
    return bad;
}

int main() {
    int bad = 0;
    bad += check<3, 1>(); bad += check<3, 2>(); bad += check<5, 1>(); bad += check<7, 1>(); bad += check<11, 1>();
    return bad == 0 ? 0 : 1;
}
