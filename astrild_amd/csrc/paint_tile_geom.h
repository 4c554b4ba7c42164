// Tile geometry of the LDS-tiled paint, shared with the FFT z pass that can fold the halo
// records of the column walk while it loads the grid rows (fft_tile.hip).
#pragma once
#include "paint_window.h"

namespace ast {

#ifndef TILE_TX
#define TILE_TX 8
#define TILE_TY 8
#define TILE_TZ 32
#endif
constexpr int TX = TILE_TX, TY = TILE_TY, TZ = TILE_TZ;   // owned cells per tile

// Pitch (elements) of a halo-record z line: n.  (Every line then starts at a multiple of 4 KB for fp32 and n = 1024, like
// the tile segments of the paint workspace before their strides were skewed - but here a pitch of n + 32 measured no
// better: step 12.79 / 12.86 against 12.73 / 12.81 ms, scripts/micro/ab_rec_skew.sh.  -DREC_LINE_SKEW=32 for A/B builds.)
#ifndef REC_LINE_SKEW
#define REC_LINE_SKEW 0
#endif
__host__ __device__ inline size_t rec_pitch(size_t n) { return n + REC_LINE_SKEW; }

// halo ring of one (LX x LY) plane of the LDS tile: the cells a column deposits for its x / y
// neighbours, kept as z lines of the column's halo record [column][ring cell][z]
template <int W> struct RingMap {
    static constexpr int HL = Window<W>::LO, HH = W - 1 - Window<W>::LO, H = HL + HH;
    static constexpr int LX = TX + W - 1, LY = TY + W - 1;
    static constexpr int COUNT = H * LY + TX * H;
    __device__ static inline bool owned(int v, int t) { return v >= HL && v < HL + t; }
    __device__ static inline int to_h(int v, int t) { return v < HL ? v : v - t; }
    __device__ static inline int cell(int a, int b) {        // (a, b) outside the owned block
        if (!owned(a, TX)) return to_h(a, TX) * LY + b;
        return H * LY + (a - HL) * H + to_h(b, TY);
    }
};

// The record lines that end in the owned grid line (x, y) of a periodic (nx_alloc = n) grid: up to 3,
// in the fixed order the fold adds them (dx outer, dy inner).  Returns their count.
// x_periodic = false: a slab buffer of ntx tile rows (x = buffer plane): neighbours past either end of the buffer do not
// exist (what they would have brought arrives with the ghost planes).
template <typename T, int W>
__device__ inline int halo_sources(const T* rec, int x, int y, int n, int ntx, int nty, const T* (&src)[3], bool x_periodic = true) {
    constexpr int LO = Window<W>::LO;
    using RM = RingMap<W>;
    const int tx = x / TX, ao = x % TX, ty = y / TY, bo = y % TY;
    int ns = 0;
    // three scalars filled by selects: a `src[ns++] = ...` with a run-time ns puts the array into scratch memory
    // (32 bytes per lane written and read back by every thread of the z pass)
    const T *s0 = nullptr, *s1 = nullptr, *s2 = nullptr;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            if (dx == 0 && dy == 0) continue;
            const int a = ao + LO - dx * TX, b = bo + LO - dy * TY;   // this cell in the neighbour's LDS frame
            if (a < 0 || a >= RM::LX || b < 0 || b >= RM::LY) continue;
            if (!x_periodic && (tx + dx < 0 || tx + dx >= ntx)) continue;
            const int ntx_ = x_periodic ? wrap1(tx + dx, ntx) : tx + dx, nty_ = wrap1(ty + dy, nty);
            const T* p = rec + ((size_t)(ntx_ * nty + nty_) * RM::COUNT + RM::cell(a, b)) * rec_pitch((size_t)n);
            s0 = ns == 0 ? p : s0;
            s1 = ns == 1 ? p : s1;
            s2 = ns == 2 ? p : s2;
            ++ns;
        }
    }
    src[0] = s0;
    src[1] = s1;
    src[2] = s2;
    ns = ns > 3 ? 3 : ns;
    return ns;
}

}  // namespace ast
