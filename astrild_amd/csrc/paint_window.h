// pmesh window conventions shared by the direct and the LDS-tiled paint.
//
// Cell index and sub-cell offset are always float64 (x * n/L of a float32
// position would lose ~1e-4 cell at n = 1024); the separable weights are then
// evaluated in the grid's own precision R.  pmesh renormalises the weights by
// their sum, which is 1 to within 2 ulp analytically; that division is not
// repeated here.
#pragma once
#include <hip/hip_runtime.h>

namespace ast {

template <int W> struct Window;

template <> struct Window<1> {              // NGP: support 1, left 0, shift 0.5
    static constexpr int LO = 0;
    template <typename R>
    __device__ static inline void eval(double s, long long& i0, R* w) {
        i0 = (long long)floor(s + 0.5);
        w[0] = (R)1;
    }
};

template <> struct Window<2> {              // CIC: leftmost = floor(s)
    static constexpr int LO = 0;            // base cell = leftmost + LO
    template <typename R>
    __device__ static inline void eval(double s, long long& i0, R* w) {
        const double fl = floor(s);
        const R f = (R)(s - fl);
        i0 = (long long)fl;
        w[0] = (R)1 - f;
        w[1] = f;
    }
};

template <> struct Window<3> {              // TSC: centre = floor(s + 1/2), leftmost = centre - 1
    static constexpr int LO = 1;
    template <typename R>
    __device__ static inline void eval(double s, long long& i0, R* w) {
        const double ic = floor(s + 0.5);
        const R d = (R)(s - ic);
        i0 = (long long)ic - 1;
        const R hm = (R)0.5 - d, hp = (R)0.5 + d;
        w[0] = (R)0.5 * (hm * hm);
        w[1] = (R)0.75 - d * d;
        w[2] = (R)0.5 * (hp * hp);
    }
};

// base cell only (tile keys); the weights are dead code here
template <int W>
__device__ inline long long base_cell(double s) {
    return W == 3 ? (long long)floor(s + 0.5) : (W == 2 ? (long long)floor(s) : (long long)floor(s + 0.5));
}

// periodic wrap; in-box particles take the branch-free fast path (a 64-bit
// modulo is a ~100-instruction software routine on the GPU)
__device__ inline int wrap(long long i, int n) {
    if ((unsigned long long)i < (unsigned long long)n) return (int)i;
    if (i < 0 && i >= -(long long)n) return (int)(i + n);
    if (i >= n && i < 2ll * n) return (int)(i - n);
    long long r = i % n;
    return (int)(r < 0 ? r + n : r);
}

}  // namespace ast
