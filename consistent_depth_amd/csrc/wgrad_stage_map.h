// Tile geometry and the staging map of the split-bf16 weight-gradient kernel (wgrad_split.hip): which element of the dY / X tile
// a thread stages in its j-th slot and where it lands in LDS.  Plain C++ (no HIP headers) so that the host test
// tests/emul/wgrad_map_emul.cpp compiles THIS code with g++ and checks it against the consumer's view of the tile (the fragment
// addresses of ws_wave): every element staged exactly once, every address inside the allocation.
#pragma once
#include <cstddef>

#if defined(__HIPCC__)
#define WS_HD __host__ __device__
#else
#define WS_HD
#endif

namespace cd {

// image-tile rows per work item of the split-bf16 weight-gradient kernel (16 x 16 channels per block, 32-pixel rows)
WS_HD constexpr int wgrad_split_tile_rows(int ks) { return ks == 11 ? 8 : 6; }

WS_HD constexpr int ws_pad(int words) { return words + ((4 - words % 8) + 8) % 8; }   // == 4 (mod 8): 16 channel planes tile the 64 banks

template <int KS, int COT = 1> struct WsCfg {
    static constexpr int TY = wgrad_split_tile_rows(KS);
    static constexpr int NW = KS == 11 ? 8 : 4;               // waves per block (k = 11: 8 x 16 taps = 64 accumulator registers each)
    static constexpr int WPS = NW / COT;                      // waves per 16 x 16 sub-tile (COT output-channel groups per block)
    static constexpr int P = (KS - 1) / 2, TAPS = KS * KS, TPW = (TAPS + WPS - 1) / WPS;
    static constexpr int ROWS = TY + KS - 1;
    static constexpr int XW = 48;                              // pixels per LDS row of X: [X0 - 8, X0 + 40)
    static constexpr int PSX = ws_pad(ROWS * XW / 2);          // 32-bit words per channel plane
    static constexpr int PSD = ws_pad(TY * 32 / 2);
    static constexpr int SPX = 16 * PSX, SPD = 16 * COT * PSD;   // words per split plane set; LDS words: [3][SPX] X planes, then [3][SPD] dY planes
    static constexpr size_t LDS = (size_t)3 * (SPX + SPD) * 4 + 128;   // + scale / shift of the block's 16 input channels
    // staging: one element ("quad") = 4 consecutive pixels of one channel row; QDY quads of dY, then QX quads of X with its halo
    static constexpr int NT = NW * 64;
    static constexpr int QDY = 16 * COT * TY * 8, QX = 16 * ROWS * 12, QT = QDY + QX, NQ = (QT + NT - 1) / NT, JDY = QDY / NT;
    static_assert(QDY % NT == 0, "dY quads fill whole rounds of the block: slot j is dY for j < JDY, X otherwise -- for every thread");
    static_assert(NQ * 4 <= 64, "one keep bit per fetched pixel");
};

// Slot j of thread t: channel c of the tile (dY: 0 .. 16 COT - 1, X: 0 .. 15), tile row r, quad q (pixels 4q .. 4q + 3 of the
// row: dY rows have 8 quads, X rows 12), and the LDS word of the quad's first pixel pair in split plane 0 (the other two planes
// are `split_words` further each).  !live: the slot has no element (beyond the last X quad); its fetch is clamped, nothing is written.
struct WsQuad { bool live, is_dy; int c, r, q, lds_word, split_words; };

template <int KS, int COT> WS_HD constexpr WsQuad ws_stage_quad(int t, int j) {
    using Cfg = WsCfg<KS, COT>;
    const bool is_dy = j < Cfg::JDY;
    int i = t + j * Cfg::NT - (is_dy ? 0 : Cfg::QDY);
    const int rows = is_dy ? Cfg::TY : Cfg::ROWS, quads = is_dy ? 8 : 12;
    const bool live = is_dy || i < Cfg::QX;
    if (!live) i = Cfg::QX - 1;
    const int c = i / (rows * quads), rem = i - c * (rows * quads), r = rem / quads, q = rem - r * quads;
    const int word = is_dy ? 3 * Cfg::SPX + c * Cfg::PSD + r * 16 + 2 * q : c * Cfg::PSX + r * 24 + 2 * q;
    return WsQuad{live, is_dy, c, r, q, word, is_dy ? Cfg::SPD : Cfg::SPX};
}

}  // namespace cd
