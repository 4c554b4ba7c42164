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

// periodic wrap of an index that is at most one period out of range
__device__ inline int wrap1(int i, int n) {
    if (i < 0) i += n;
    if (i >= n) i -= n;
    return i;
}

// Window<W>: support W, base cell = the cell the particle is assigned to
// (CIC: lower corner floor(s); NGP/TSC: nearest grid point floor(s + 1/2)),
// stencil = base - LO .. base - LO + W - 1.
template <int W> struct Window;

template <> struct Window<1> {
    static constexpr int LO = 0;
    template <typename R> __device__ static inline void weights(double, R* w) { w[0] = (R)1; }
};
template <> struct Window<2> {
    static constexpr int LO = 0;
    template <typename R> __device__ static inline void weights(double frac, R* w) {
        const R f = (R)frac;              // in [0, 1)
        w[0] = (R)1 - f;
        w[1] = f;
    }
};
template <> struct Window<3> {
    static constexpr int LO = 1;
    template <typename R> __device__ static inline void weights(double frac, R* w) {
        const R d = (R)frac;              // in [-1/2, 1/2)
        const R hm = (R)0.5 - d, hp = (R)0.5 + d;
        w[0] = (R)0.5 * (hm * hm);
        w[1] = (R)0.75 - d * d;
        w[2] = (R)0.5 * (hp * hp);
    }
};

// s = x * n/L  ->  base cell wrapped into [0, n) and the offset s - floor(..).
// Branch-free and call-free on purpose: an earlier version kept the rare far-out-of-box case
// in a noinline helper, and the call sites alone cost the deposit kernels ~20 SGPRs, SGPR
// spills and a wave of occupancy.  The period is removed in double (exact for |s| < 2^52),
// with one correction step either way for non-power-of-two n.
template <int W>
__device__ inline int locate(double s, int n, double& frac) {
    const double fl = floor(W == 2 ? s : s + 0.5);
    frac = s - fl;
    const double dn = (double)n, inv_dn = 1.0 / dn;      // loop invariant: hoisted by the compiler
    double r = fl - floor(fl * inv_dn) * dn;
    r = r >= dn ? r - dn : r;
    r = r < 0.0 ? r + dn : r;
    return (int)r;
}

// Fast variant for callers that can test the result: base cell minus `origin` (origin in
// [0, n)), reduced by at most one period either way.  Exactly (cell - origin) mod n for positions
// less than one box length outside the box; anything further out comes back >= n as unsigned
// and the caller falls back to locate().  The double -> int conversion saturates.
template <int W>
__device__ inline int locate_rel(double s, int n, int origin, double& frac) {
    const double fl = floor(W == 2 ? s : s + 0.5);
    frac = s - fl;
    int l = (int)fl - origin;
    l = (int)min((unsigned)l, (unsigned)(l + n));
    l = (int)min((unsigned)l, (unsigned)(l - n));
    return l;
}

}  // namespace ast
