// Double-precision transforms written by hand.  Part 1: column and row transforms of the zero-padded lens convolution;
// part 2 (end of the file): the 3-D power spectrum of a float64 grid (ast_fft64_power_3d), which reuses part 1's row
// kernel and three-stage scheme.
//
// Column transforms of the zero-padded lens convolution (kappa -> alpha / phi, lensing_funcs.c:85-115 with
// fft_convolve.c:60-90) in double precision, hand-written because the half that is zero can be skipped and rocFFT's
// strided 8192-point double transform moves its data at 1.1 TB/s (0.98 ms per pass over the 537 MB half spectrum).
//
// A column of L = N1 * N2 points (8192 = 64 * 128) does not fit LDS with enough neighbours for whole cache lines
// (16 columns x 8192 x 16 B = 2 MB), so it is transformed in TWO in-place passes over the array (four-step FFT):
//   pass A   for every n2: the N1-point transform over n1 of x[N2 n1 + n2] (rows N2 apart), times W_L^{n2 k1},
//            stored at row N2 k1 + n2;
//   pass B   for every k1: the N2-point transform over n2 of the N2 CONSECUTIVE rows N2 k1 + n2, stored at row
//            N2 k1 + k2.
// Frequency k = k1 + N1 k2 therefore ends up at row N2 k1 + k2: a fixed permutation of the rows.  Nothing downstream
// needs the natural order - the spectra are only multiplied point by point with kernel spectra that went through the
// same two passes, and the inverse (pass B backwards, conjugate twiddle, pass A backwards) undoes it.
// A convolution (ast_lens_cols_convolve) runs forward pass B, the product(s) and inverse pass B as ONE pass
// (col_mid_kernel): they touch the same N2 rows x 16 columns, and the spectrum is then never written to memory.
// Zero padding: the forward pass A reads the first `in_points` of its N1 points only (rows >= Nc hold zeros that are
// never stored either), the last inverse pass writes the first `out_points` only (the corner that is kept).
// Each workgroup works on 16 adjacent columns (256-byte row pieces): R1 x R2 register FFTs with one LDS exchange,
// the scheme of fft_tile.hip's strided pass in double.
#include "../../include/astrild_hip.h"
#include "ast_common.h"
#include "paint_tile_geom.h"
#include <cmath>
#include <mutex>
#include <vector>

#define LENS_FWD(call)                 \
    do {                               \
        int rc_ = (call);              \
        if (rc_ != AST_OK) return rc_; \
    } while (0)

namespace {

// e^{-2 pi i r / 16}
__device__ constexpr double kC16[16] = {1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0,
                                        -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                        -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                        0.38268343236508977, 0.70710678118654752, 0.92387953251128674};
__device__ constexpr double kS16[16] = {0.0, -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                        -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                        0.38268343236508977, 0.70710678118654752, 0.92387953251128674, 1.0,
                                        0.92387953251128674, 0.70710678118654752, 0.38268343236508977};

constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

__device__ inline double2 cmul(double2 a, double2 w) {
    return make_double2(fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x));
}

// e^{-2 pi i r / 32} for odd r (the even ones are the 16th roots above): cos and sin of pi r / 16
__device__ constexpr double kC32odd[8] = {0.98078528040323044, 0.83146961230254524, 0.55557023301960222, 0.19509032201612827,
                                          -0.19509032201612827, -0.55557023301960222, -0.83146961230254524, -0.98078528040323044};
__device__ constexpr double kS32odd[8] = {-0.19509032201612827, -0.55557023301960222, -0.83146961230254524, -0.98078528040323044,
                                          -0.98078528040323044, -0.83146961230254524, -0.55557023301960222, -0.19509032201612827};

__device__ inline float2 cmul(float2 a, float2 w) {
    return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x));
}
template <typename V> struct ScalarOf;
template <> struct ScalarOf<double2> { typedef double type; };
template <> struct ScalarOf<float2> { typedef float type; };

// forward FFT of R <= 32 points in registers, decimation in frequency: X[k] ends up in v[bitrev(k)]
// (V = double2, or float2 for the single-precision passes of cubes of side 2048 at the end of this file)
template <int R, typename V>
__device__ inline void fft_reg(V (&v)[R]) {
    typedef typename ScalarOf<V>::type S;
#pragma unroll
    for (int h = R / 2; h >= 1; h /= 2) {
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int j = 0; j < h; ++j) {
                const V a = v[blk + j], b = v[blk + j + h];
                v[blk + j] = V(a.x + b.x, a.y + b.y);
                const V d = V(a.x - b.x, a.y - b.y);
                const int t = j * (16 / h);                  // W_{2h}^j = W_32^{j * 32 / (2h)}, t < 16
                if (t == 0) v[blk + j + h] = d;
                else if (t == 8) v[blk + j + h] = V(d.y, -d.x);          // * (-i)
                else if (t % 2 == 0) v[blk + j + h] = cmul(d, V((S)kC16[t / 2], (S)kS16[t / 2]));
                else v[blk + j + h] = cmul(d, V((S)kC32odd[t / 2], (S)kS32odd[t / 2]));
            }
        }
    }
}

// One pass.  Workgroup (column tile, group g): points j < NP = R1 * R2 at rows row0(g) + j * point_rows, 16 columns.
//   in_points:  points >= in_points are zero and not loaded (a multiple of R2);
//   out_points: points >= out_points are not stored;
//   twist:      multiply output point j by W_L^{+-(g j)} (the four-step twiddle; tw_big has L entries);
//   mul:        multiply every loaded value by mul[same index] (the convolution's kernel spectrum, fused into the first
//               inverse pass), the result goes to `out` (may equal `in`).
// INV: conj(FFT(conj(.))), the twist conjugated.
template <int R1, int R2, int C, bool INV>
__global__ void __launch_bounds__(C * (R1 > R2 ? R1 : R2))
col_pass_kernel(const double2* __restrict__ in, const double2* __restrict__ mul, double2* __restrict__ out, size_t pitch,
                int ncols, size_t group_rows, size_t point_rows, int in_points, int out_points, int twist,
                const double2* __restrict__ tw_big, int big_len) {
    constexpr int NP = R1 * R2;
    constexpr int NT = C * (R1 > R2 ? R1 : R2);
    __shared__ double2 Y[NP * C];
    __shared__ double2 tw[NP];                            // e^{-2 pi i m / NP}
    for (int i = threadIdx.x; i < NP; i += NT) tw[i] = tw_big[(size_t)i * (big_len / NP)];
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    const size_t c0 = (size_t)blockIdx.x * C;
    const int g = blockIdx.y;
    const bool col_ok = c0 + c < (size_t)ncols;
    const size_t base = (size_t)g * group_rows * pitch + min(c0 + c, (size_t)ncols - 1);
    const size_t pstride = point_rows * pitch;
    if (sub < R2) {                                       // stage 1: task (c, n2 = sub)
        double2 v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            v[n1] = make_double2(0.0, 0.0);
            if (n1 * R2 < in_points) {                    // uniform: in_points is a multiple of R2
                const size_t idx = base + (size_t)(n1 * R2 + sub) * pstride;
                v[n1] = in[idx];
                if (mul) v[n1] = cmul(v[n1], mul[idx]);
            }
        }
        if (INV) {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1].y = -v[n1].y;
        }
        fft_reg<R1>(v);
        __syncthreads();                                  // the twiddle table is in LDS
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            double2 y = v[bitrev(k1, ilog2(R1))];
            if (k1 != 0) y = cmul(y, tw[sub * k1]);
            Y[(sub * R1 + k1) * C + c] = y;
        }
    } else {
        __syncthreads();
    }
    __syncthreads();
    if (sub < R1) {                                       // stage 2: task (c, k1 = sub)
        double2 u[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) u[n2] = Y[(n2 * R1 + sub) * C + c];
        fft_reg<R2>(u);
        if (col_ok) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int j = sub + R1 * k2;              // output point
                if (j >= out_points) continue;
                double2 x = u[bitrev(k2, ilog2(R2))];
                if (twist) x = cmul(x, tw_big[(size_t)((g * j) % big_len)]);     // W_L^{g j}; INV: conjugated below with x
                if (INV) x.y = -x.y;
                out[base + (size_t)j * pstride] = x;
            }
        }
    }
}

// The middle of a column convolution in one pass: the forward transform's second pass (group g = k1, points n2 at
// consecutive rows), the product with NMUL kernel spectra and the inverse's first pass (back to n2, times conj W_L^{k1 n2})
// all work on the same NP rows x C columns, so the spectrum itself is never written: `in` holds pass A's output, out_m
// what the inverse's last pass (pass A backwards) reads.  Forward as NP = RA * RB (RA-point transforms in registers, LDS
// exchange, RB-point transforms), which leaves thread (c, sub < RA) with the points sub + RA k2 - exactly the input
// layout of an inverse split the other way round (RB-point transforms first), so the products are formed in place.
// measured at 8192 rows, two kernels: three workgroups per CU without prefetching the second kernel spectrum 0.585 ms,
// with it (34 registers spilled) 0.629; two per CU with the prefetch and nothing spilled 0.690
#ifndef MID_PREFETCH
#define MID_PREFETCH 0
#endif
#ifndef MID_WAVES
#define MID_WAVES 3
#endif
template <int RA, int RB, int C, int NMUL>
__global__ void __launch_bounds__(C * (RA > RB ? RA : RB)) __attribute__((amdgpu_waves_per_eu(MID_WAVES)))
col_mid_kernel(const double2* __restrict__ in, const double2* __restrict__ mul0, const double2* __restrict__ mul1,
               double2* __restrict__ out0, double2* __restrict__ out1, size_t pitch, int ncols,
               const double2* __restrict__ tw_big, int big_len) {
    constexpr int NP = RA * RB;
    constexpr int NT = C * (RA > RB ? RA : RB);
    __shared__ double2 Y[NP * C];
    __shared__ double2 tw[NP];                            // e^{-2 pi i m / NP}
    for (int i = threadIdx.x; i < NP; i += NT) tw[i] = tw_big[(size_t)i * (big_len / NP)];
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    const size_t c0 = (size_t)blockIdx.x * C;
    const int g = blockIdx.y;
    const bool col_ok = c0 + c < (size_t)ncols;
    // addresses: a uniform row base per point plus ONE 32-bit lane offset (host check: pitch < 2^24)
    const size_t gbase = (size_t)g * NP * pitch;
    const uint32_t voff = (uint32_t)sub * (uint32_t)pitch + min((uint32_t)c0 + (uint32_t)c, (uint32_t)ncols - 1u);
    double2 m[RB];                                        // the kernel spectrum at this thread's points (sub < RA)
    auto fetch_mul = [&](const double2* __restrict__ mul) {
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) m[k2] = (mul + gbase + (size_t)(RA * k2) * pitch)[voff];
    };
    // (every barrier of this kernel is reached by ALL threads outside any branch: ADVICE r3 - the <16, 8> shape used to
    // execute two different s_barrier instructions in the two arms of `if (sub < RB)`)
    const bool task1 = sub < RB;                          // forward stage 1: task (c, n2 = sub)
    double2 v[RA];
    if (task1) {
#pragma unroll
        for (int n1 = 0; n1 < RA; ++n1) v[n1] = (in + gbase + (size_t)(n1 * RB) * pitch)[voff];
    }
    if (sub < RA) fetch_mul(mul0);                        // in flight across the exchange
    if (task1) fft_reg<RA>(v);
    __syncthreads();                                      // the twiddle table is in LDS
    if (task1) {
#pragma unroll
        for (int k1 = 0; k1 < RA; ++k1) {
            double2 y = v[bitrev(k1, ilog2(RA))];
            if (k1 != 0) y = cmul(y, tw[sub * k1]);
            Y[(sub * RA + k1) * C + c] = y;
        }
    }
    __syncthreads();
    double2 X[RB];                                        // sub < RA: spectrum point sub + RA k2 in X[bitrev(k2)]
    if (sub < RA) {
#pragma unroll
        for (int n2 = 0; n2 < RB; ++n2) X[n2] = Y[(n2 * RA + sub) * C + c];
        fft_reg<RB>(X);
    }
#pragma unroll
    for (int which = 0; which < NMUL; ++which) {
        double2* __restrict__ out = which ? out1 : out0;
        __syncthreads();                                  // Y has been read by everyone
        if (!MID_PREFETCH && which && sub < RA) fetch_mul(mul1);
        if (sub < RA) {                                   // inverse stage 1: task (c, n2' = sub), points n1' RA + sub, n1' = k2
            double2 w[RB];
#pragma unroll
            for (int k2 = 0; k2 < RB; ++k2) {
                w[k2] = cmul(X[bitrev(k2, ilog2(RB))], m[k2]);
                w[k2].y = -w[k2].y;
            }
            if (MID_PREFETCH && which + 1 < NMUL) fetch_mul(mul1);
            fft_reg<RB>(w);
#pragma unroll
            for (int k1 = 0; k1 < RB; ++k1) {
                double2 y = w[bitrev(k1, ilog2(RB))];
                if (k1 != 0) y = cmul(y, tw[sub * k1]);
                Y[(sub * RB + k1) * C + c] = y;
            }
        }
        __syncthreads();
        if (sub < RB) {                                   // inverse stage 2: task (c, k1' = sub)
            double2 u[RA];
#pragma unroll
            for (int n2 = 0; n2 < RA; ++n2) u[n2] = Y[(n2 * RB + sub) * C + c];
            fft_reg<RA>(u);
            if (col_ok) {
#pragma unroll
                for (int k2 = 0; k2 < RA; ++k2) {
                    const int j = sub + RB * k2;          // output point n2
                    double2 x = cmul(u[bitrev(k2, ilog2(RA))], tw_big[(size_t)((g * j) % big_len)]);
                    x.y = -x.y;
                    (out + gbase + (size_t)(RB * k2) * pitch)[voff] = x;
                }
            }
        }
    }
}

// e^{-2 pi i m / len}, m < len, per (device, len)
__global__ void tw_table_kernel(double2* out, int len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double sn, cs;
    sincospi(-2.0 * (double)i / (double)len, &sn, &cs);
    out[i] = make_double2(cs, sn);
}
struct TwCache {
    std::mutex m;
    struct E { int dev, len; double2* d; };
    std::vector<E> tabs;
    const double2* get(int len, hipStream_t s) {
        std::lock_guard<std::mutex> lock(m);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& t : tabs) if (t.dev == dev && t.len == len) return t.d;
        double2* d = nullptr;
        if (hipMalloc(&d, (size_t)len * sizeof(double2)) != hipSuccess) return nullptr;
        tw_table_kernel<<<(len + 255) / 256, 256, 0, s>>>(d, len);
        // once per (device, length): the table must be complete before ANY stream reads it (the pointer is handed
        // to later callers without an event)
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(d); return nullptr; }
        tabs.push_back({dev, len, d});
        return d;
    }
} g_tw;

// ------------------------------------------------------------------ row transforms of the padded convolution
// Row r < nc of the padded (2nc)^2 array holds nc reals followed by nc zeros.  Its real-to-complex transform of length
// L = 2nc is the M = nc point complex transform of z[j] = x[2j] + i x[2j+1] (z[j] = 0 for j >= M / 2) followed by the
// even/odd split; M = RA RB RC points (4096 = 16 16 16, 2048 = 16 16 8, 1024 = 16 8 8, 512 = 8 8 8, 256 = 8 8 4,
// 128 = 8 4 4) go through three register FFTs with two exchanges through one LDS line (M + M / 8 double2: 74 KB at
// M = 4096, two workgroups per CU), one row per workgroup:
//   j = RB RC a + RC b + c,  k = alpha + RA beta + RA RB gamma
//   stage 1 (thread j' = RC b + c):  y[alpha][j'] = W_M^{j' alpha} sum_a z[RB RC a + j'] W_RA^{a alpha}   (a < RA / 2 only: the rest is zero)
//   stage 2 (thread (alpha, c)):     y'[alpha][beta][c] = W_{RB RC}^{c beta} sum_b y[alpha][RC b + c] W_RB^{b beta}
//   stage 3 (thread (alpha, beta)):  Z[alpha + RA beta + RA RB gamma] = sum_c y'[alpha][beta][c] W_RC^{c gamma}
// rocFFT's batched 1-D plans did these rows at 2.2 TB/s and needed the zero half in memory, a pad kernel before and a
// crop kernel after; here kappa is read as it is (134 MB per map) and the inverse stores the kept, scaled corner.
__device__ inline int lds_pad(int i) { return i + (i >> 3); }

template <int RA, int RB, int RC>
struct RowGeo {
    static constexpr int M = RA * RB * RC;
    static constexpr int T1 = RB * RC, T2 = RA * RC, T3 = RA * RB;
    static constexpr int NT = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
    static constexpr int RMAX = RA > RB ? (RA > RC ? RA : RC) : (RB > RC ? RB : RC);
};

// On entry (threads t < T1): va[a] = input[RB RC a + t].  On exit (threads t < T3, t = alpha * RB + beta): vc[bitrev(gamma)] =
// output[alpha + RA beta + RA RB gamma].
template <int RA, int RB, int RC, typename V>
__device__ inline void three_stage(V (&va)[RA], V (&vc)[RC], V* Y, const V* __restrict__ twM, int t) {
    using G = RowGeo<RA, RB, RC>;
    if (t < G::T1) {
        fft_reg<RA>(va);
#pragma unroll
        for (int al = 0; al < RA; ++al) {
            V y = va[bitrev(al, ilog2(RA))];
            if (al != 0) y = cmul(y, twM[t * al]);
            Y[lds_pad(al * G::T1 + t)] = y;
        }
    }
    __syncthreads();
    if (t < G::T2) {
        const int al = t / RC, c = t % RC;
        V vb[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) vb[b] = Y[lds_pad(al * G::T1 + RC * b + c)];
        fft_reg<RB>(vb);
        // the slots this thread read are the slots it writes: no barrier in between
#pragma unroll
        for (int be = 0; be < RB; ++be) {
            V y = vb[bitrev(be, ilog2(RB))];
            if (be != 0 && c != 0) y = cmul(y, twM[RA * c * be]);
            Y[lds_pad(al * G::T1 + RC * be + c)] = y;
        }
    }
    __syncthreads();
    if (t < G::T3) {
#pragma unroll
        for (int c = 0; c < RC; ++c) vc[c] = Y[lds_pad(t * RC + c)];      // t = alpha * RB + beta: RC consecutive (padded) slots
        fft_reg<RC>(vc);
    }
}

// ZPAD: the row holds M reals followed by M zeros that are not in memory (the lens plan); otherwise 2 M reals (the rows
// of a 3-D grid: the z pass of the double-precision power spectrum).  in_pitch: reals between rows; the spectrum is
// multiplied by `scale`.
// FOLDW = 2 / 3 (CIC / TSC, grids only): the grid is what a deferred-fold paint left and `rec` its halo records; the up to
// three record lines that end in a border row are added as the row is loaded - records first, then onto the row, the
// order of the paint's own fold kernel (as in fft_tile.hip's float row pass).
// TIN = float: the rows are read from a single-precision grid and widened on load (the double transform of an fp32 grid
// without a 2x copy of it: cubes the fp32 tile passes do not cover - 128^3, 2048^3; no FOLDW).
template <typename TIN> struct PairOf { typedef double2 type; };
template <> struct PairOf<float> { typedef float2 type; };
__device__ inline double2 widen(double2 v) { return v; }
__device__ inline double2 widen(float2 v) { return make_double2((double)v.x, (double)v.y); }

template <int RA, int RB, int RC, bool ZPAD, int FOLDW = 0, typename TIN = double>
__global__ void __launch_bounds__((RowGeo<RA, RB, RC>::NT))
lens_rows_forward_kernel(const TIN* __restrict__ kappa, size_t in_pitch, double2* __restrict__ spec, size_t pitch,
                         const double2* __restrict__ twM, const double2* __restrict__ twL, double scale,
                         const double* __restrict__ rec = nullptr) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int M = G::M;
    static_assert(FOLDW == 0 || sizeof(TIN) == 8, "halo records are folded in the grid's own precision: double rows only");
    extern __shared__ double2 Y[];
    const int t = threadIdx.x;
    const size_t row = blockIdx.x;
    const typename PairOf<TIN>::type* z = reinterpret_cast<const typename PairOf<TIN>::type*>(kappa + row * in_pitch);        // packed pairs: M / 2 (ZPAD) or M
    double2 va[RA], vc[RC];
    // the halo sources of this row first (integer work, the same for the whole workgroup: one row per workgroup), so
    // that the row's loads and the records' loads are all in flight together
    const double* src[3] = {nullptr, nullptr, nullptr};
    int ns = 0;
    if (FOLDW != 0) {
        constexpr int W = FOLDW != 0 ? FOLDW : 2;
        const int ng = 2 * M;
        ns = ast::halo_sources<double, W>(rec, (int)(row / ng), (int)(row % ng), ng, ng / ast::TX, ng / ast::TY, src);
    }
    if (t < G::T1) {
#pragma unroll
        for (int a = 0; a < RA; ++a) va[a] = (!ZPAD || a < RA / 2) ? widen(z[G::T1 * a + t]) : make_double2(0.0, 0.0);
        if (FOLDW != 0 && ns > 0) {
            double2 h0[RA], h1[RA], h2[RA];
#pragma unroll
            for (int a = 0; a < RA; ++a) h0[a] = reinterpret_cast<const double2*>(src[0])[G::T1 * a + t];
            if (ns > 1) {
#pragma unroll
                for (int a = 0; a < RA; ++a) h1[a] = reinterpret_cast<const double2*>(src[1])[G::T1 * a + t];
            }
            if (ns > 2) {
#pragma unroll
                for (int a = 0; a < RA; ++a) h2[a] = reinterpret_cast<const double2*>(src[2])[G::T1 * a + t];
            }
            // records first, then onto the row: the fold kernel's order
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                double2 h = h0[a];
                if (ns > 1) { h.x += h1[a].x; h.y += h1[a].y; }
                if (ns > 2) { h.x += h2[a].x; h.y += h2[a].y; }
                va[a].x += h.x;
                va[a].y += h.y;
            }
        }
    }
    three_stage<RA, RB, RC>(va, vc, Y, twM, t);
    // the untangling twiddles of this thread's outputs, in flight across the exchange below
    constexpr int IT = (M / 2) / G::NT;
    static_assert(IT * G::NT == M / 2, "row length and thread count");
    double2 wk[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) wk[it] = twL[t + it * G::NT];
    __syncthreads();                                      // everyone has read its stage-3 inputs: Y becomes Z[k]
    if (t < G::T3) {
#pragma unroll
        for (int ga = 0; ga < RC; ++ga) Y[lds_pad((t / RB) + RA * (t % RB) + RA * RB * ga)] = vc[bitrev(ga, ilog2(RC))];
    }
    __syncthreads();
    double2* orow = spec + row * pitch;
    auto emit = [&](int k, double2 w) {
        const double2 zk = Y[lds_pad(k)];
        const double2 zm = Y[lds_pad(M - k)];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));      // (Zk + conj Zm) / 2
        const double2 o = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));      // (Zk - conj Zm) / 2
        const double2 tt = cmul(o, w);                                                  // w^k o, w = e^{-2 pi i / L}
        orow[k] = make_double2((e.x + tt.y) * scale, (e.y - tt.x) * scale);             // e - i t
        orow[M - k] = make_double2((e.x - tt.y) * scale, (-e.y - tt.x) * scale);        // conj(e + i t)
    };
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int k = t + it * G::NT;
        if (k == 0) {
            const double2 zk = Y[lds_pad(0)];
            orow[0] = make_double2((zk.x + zk.y) * scale, 0.0);
            orow[M] = make_double2((zk.x - zk.y) * scale, 0.0);
        } else {
            emit(k, wk[it]);
        }
    }
    if (t == 0) emit(M / 2, twL[M / 2]);
}

// rows of the product spectrum -> the kept corner: out[row][0 .. nc) = scale * (first nc reals of the length-2nc C2R)
template <int RA, int RB, int RC>
__global__ void __launch_bounds__((RowGeo<RA, RB, RC>::NT))
lens_rows_inverse_kernel(const double2* __restrict__ spec, size_t pitch, int nc, double scale, double* __restrict__ out,
                         const double2* __restrict__ twM, const double2* __restrict__ twL) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int M = G::M;
    extern __shared__ double2 Y[];
    const int t = threadIdx.x;
    const size_t row = blockIdx.x;
    const double2* xrow = spec + row * pitch;
    // Z[k] = (X[k] + conj X[M-k]) / 2 + i w^{-k} (X[k] - conj X[M-k]) / 2, conjugated for the conj-FFT-conj inverse
    auto emit = [&](int k, double2 xk, double2 xm, double2 w) {
        const double2 e = make_double2(0.5 * (xk.x + xm.x), 0.5 * (xk.y - xm.y));
        const double2 d = make_double2(0.5 * (xk.x - xm.x), 0.5 * (xk.y + xm.y));
        const double2 tt = make_double2(fma(d.x, w.x, d.y * w.y), fma(d.y, w.x, -d.x * w.y));       // conj(w^k) d
        // Z[k] = e + i t;  Z[M-k] follows from the same formula with k -> M - k:
        //   e' = conj e, d' = -conj d, w^{-(M-k)} = -conj(w^{-k})  ->  t' = conj t,  Z[M-k] = conj(e) + i conj(t)
        if (k < M) Y[lds_pad(k)] = make_double2(e.x - tt.y, -(e.y + tt.x));                         // conj(Z[k])
        if (k != 0 && k != M / 2) Y[lds_pad(M - k)] = make_double2(e.x + tt.y, -(tt.x - e.y));      // conj(Z[M-k]) = e - i t
    };
    // the whole row in flight before the first use: (M / 2) / NT pairs per thread, k = M / 2 by thread 0
    constexpr int IT = (M / 2) / G::NT;
    static_assert(IT * G::NT == M / 2, "row length and thread count");
    {
        double2 xk[IT], xm[IT], wk[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int k = t + it * G::NT;
            xk[it] = xrow[k];
            xm[it] = xrow[M - k];
            wk[it] = twL[k];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) emit(t + it * G::NT, xk[it], xm[it], wk[it]);
        if (t == 0) emit(M / 2, xrow[M / 2], xrow[M / 2], twL[M / 2]);
    }
    __syncthreads();
    double2 va[RA], vc[RC];
    if (t < G::T1) {
#pragma unroll
        for (int a = 0; a < RA; ++a) va[a] = Y[lds_pad(G::T1 * a + t)];
    }
    __syncthreads();                                      // the line is overwritten by the stages
    three_stage<RA, RB, RC>(va, vc, Y, twM, t);
    __syncthreads();
    // z[j], j = alpha + RA beta + RA RB gamma < M / 2, i.e. gamma < RC / 2: the kept half
    if (t < G::T3) {
#pragma unroll
        for (int ga = 0; ga < RC / 2; ++ga) Y[lds_pad((t / RB) + RA * (t % RB) + RA * RB * ga)] = vc[bitrev(ga, ilog2(RC))];
    }
    __syncthreads();
    double2* orow = reinterpret_cast<double2*>(out + row * (size_t)nc);
    const double s2 = 2.0 * scale;
    for (int j = t; j < M / 2; j += G::NT) {
        const double2 zz = Y[lds_pad(j)];                 // conj of z[j]
        orow[j] = make_double2(zz.x * s2, -zz.y * s2);    // x[2j] = 2 Re z, x[2j+1] = 2 Im z
    }
}

template <int R1, int R2, bool INV>
int launch_pass(const double2* in, const double2* mul, double2* out, size_t pitch, int ncols, int groups, size_t group_rows,
                size_t point_rows, int in_points, int out_points, int twist, const double2* tw_big, int big_len, hipStream_t s) {
    constexpr int C = 16, NT = C * (R1 > R2 ? R1 : R2);
    const dim3 grid((unsigned)((ncols + C - 1) / C), (unsigned)groups);
    col_pass_kernel<R1, R2, C, INV><<<grid, NT, 0, s>>>(in, mul, out, pitch, ncols, group_rows, point_rows, in_points, out_points,
                                                       twist, tw_big, big_len);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// L = N1 * N2 with N1 = R1a * R2a (pass A, rows N2 apart) and N2 = R1b * R2b (pass B, consecutive rows)
struct Split { int n1, n2; };
bool split_of(size_t len, Split& sp) {
    switch (len) {
        case 16384: sp = {128, 128}; return true;
        case 8192: sp = {64, 128}; return true;
        case 4096: sp = {64, 64}; return true;
        case 2048: sp = {32, 64}; return true;
        case 1024: sp = {32, 32}; return true;
        case 512: sp = {16, 32}; return true;
        case 256: sp = {16, 16}; return true;
        default: return false;
    }
}

template <bool INV>
int pass_of(int np, const double2* in, const double2* mul, double2* out, size_t pitch, int ncols, int groups, size_t group_rows,
            size_t point_rows, int in_points, int out_points, int twist, const double2* tw_big, int big_len, hipStream_t s) {
    switch (np) {
        case 128: return launch_pass<8, 16, INV>(in, mul, out, pitch, ncols, groups, group_rows, point_rows, in_points, out_points, twist, tw_big, big_len, s);
        case 64: return launch_pass<8, 8, INV>(in, mul, out, pitch, ncols, groups, group_rows, point_rows, in_points, out_points, twist, tw_big, big_len, s);
        case 32: return launch_pass<4, 8, INV>(in, mul, out, pitch, ncols, groups, group_rows, point_rows, in_points, out_points, twist, tw_big, big_len, s);
        default: return launch_pass<4, 4, INV>(in, mul, out, pitch, ncols, groups, group_rows, point_rows, in_points, out_points, twist, tw_big, big_len, s);
    }
}

template <int RA, int RB, int NMUL>
int launch_mid(const double2* in, const double2* const* muls, double2* const* outs, size_t pitch, int ncols, int groups,
               const double2* tw_big, int big_len, hipStream_t s) {
    constexpr int C = 16, NT = C * (RA > RB ? RA : RB);
    const dim3 grid((unsigned)((ncols + C - 1) / C), (unsigned)groups);
    col_mid_kernel<RA, RB, C, NMUL><<<grid, NT, 0, s>>>(in, muls[0], muls[NMUL - 1], outs[0], outs[NMUL - 1], pitch, ncols, tw_big,
                                                        big_len);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

template <int NMUL>
int mid_of(int np, const double2* in, const double2* const* muls, double2* const* outs, size_t pitch, int ncols, int groups,
           const double2* tw_big, int big_len, hipStream_t s) {
    switch (np) {
        case 128: return launch_mid<16, 8, NMUL>(in, muls, outs, pitch, ncols, groups, tw_big, big_len, s);
        case 64: return launch_mid<8, 8, NMUL>(in, muls, outs, pitch, ncols, groups, tw_big, big_len, s);
        case 32: return launch_mid<4, 8, NMUL>(in, muls, outs, pitch, ncols, groups, tw_big, big_len, s);
        default: return launch_mid<4, 4, NMUL>(in, muls, outs, pitch, ncols, groups, tw_big, big_len, s);
    }
}

}  // namespace

extern "C" int ast_lens_cols_supported(size_t len) {
    Split sp;
    return split_of(len, sp) ? 1 : 0;
}

// Forward column transform, in place, of data_d[len][pitch] (complex double; `ncols` <= pitch columns used): rows >=
// nonzero_rows are taken as zero and never read (nonzero_rows = len: a full transform; otherwise len / 2).  The result
// is in the permuted row order described at the top of this file.
extern "C" int ast_lens_cols_forward(void* data, size_t len, size_t pitch, size_t ncols, size_t nonzero_rows, void* stream) {
    AST_CHECK_ARG(data != nullptr && ncols >= 1 && ncols <= pitch);
    Split sp;
    AST_CHECK_ARG(split_of(len, sp));
    AST_CHECK_ARG(nonzero_rows == len || nonzero_rows == len / 2);
    hipStream_t s = ast::as_stream(stream);
    const double2* tw = g_tw.get((int)len, s);
    if (!tw) { ast::set_error("ast_lens_cols_forward: twiddle table allocation failed"); return AST_ERR_HIP; }
    double2* d = (double2*)data;
    AST_PROF("lens.cols_fwd", s);
    // pass A: group = n2 (first row n2), points n1 at rows N2 apart; rows >= len / 2 are points n1 >= N1 / 2
    int rc = pass_of<false>(sp.n1, d, nullptr, d, pitch, (int)ncols, sp.n2, 1, (size_t)sp.n2,
                            nonzero_rows == len ? sp.n1 : sp.n1 / 2, sp.n1, 1, tw, (int)len, s);
    if (rc != AST_OK) return rc;
    // pass B: group = k1 (first row N2 k1), points n2 at consecutive rows
    return pass_of<false>(sp.n2, d, nullptr, d, pitch, (int)ncols, sp.n1, (size_t)sp.n2, 1, sp.n2, sp.n2, 0, tw, (int)len, s);
}

// Inverse column transform (unnormalised) of spec_d * mul_d (point by point; mul_d may be NULL) into out_d (may equal
// spec_d when mul_d is NULL), all [len][pitch] in the permuted order; only the first keep_rows rows of the result are
// written (keep_rows = len or len / 2), in natural order.
extern "C" int ast_lens_cols_inverse(const void* spec, const void* mul, void* out, size_t len, size_t pitch, size_t ncols,
                                     size_t keep_rows, void* stream) {
    AST_CHECK_ARG(spec != nullptr && out != nullptr && ncols >= 1 && ncols <= pitch);
    AST_CHECK_ARG(mul == nullptr || (mul != out && spec != out));
    Split sp;
    AST_CHECK_ARG(split_of(len, sp));
    AST_CHECK_ARG(keep_rows == len || keep_rows == len / 2);
    hipStream_t s = ast::as_stream(stream);
    const double2* tw = g_tw.get((int)len, s);
    if (!tw) { ast::set_error("ast_lens_cols_inverse: twiddle table allocation failed"); return AST_ERR_HIP; }
    AST_PROF("lens.cols_inv", s);
    double2* o = (double2*)out;
    // pass B backwards: group = k1, points k2 at consecutive rows -> n2, times conj W_L^{k1 n2}
    int rc = pass_of<true>(sp.n2, (const double2*)spec, (const double2*)mul, o, pitch, (int)ncols, sp.n1, (size_t)sp.n2, 1, sp.n2,
                           sp.n2, 1, tw, (int)len, s);
    if (rc != AST_OK) return rc;
    // pass A backwards: group = n2, points k1 at rows N2 apart -> n1; rows >= len / 2 are points n1 >= N1 / 2
    return pass_of<true>(sp.n1, o, nullptr, o, pitch, (int)ncols, sp.n2, 1, (size_t)sp.n2, sp.n1,
                         keep_rows == len ? sp.n1 : sp.n1 / 2, 0, tw, (int)len, s);
}

// out_m = the first keep_rows rows of IFFT_cols(FFT_cols(data) * mul_m) (unnormalised, natural order), m < nmul <= 2:
// forward pass A in place on data_d, then ONE pass that finishes the forward transform, forms the products and starts
// the inverses (col_mid_kernel: the spectrum is never written), then pass A backwards on each out_m.  mul_m are spectra
// in the permuted order ast_lens_cols_forward leaves.
extern "C" int ast_lens_cols_convolve(void* data, size_t len, size_t pitch, size_t ncols, size_t nonzero_rows,
                                      const void* const* muls, void* const* outs, int nmul, size_t keep_rows, void* stream) {
    AST_CHECK_ARG(data != nullptr && muls != nullptr && outs != nullptr && ncols >= 1 && ncols <= pitch);
    AST_CHECK_ARG(nmul == 1 || nmul == 2);
    AST_CHECK_ARG(pitch < ((size_t)1 << 24));             // col_mid_kernel: 16 rows of lane offset in 32-bit BYTES
    for (int i = 0; i < nmul; ++i) AST_CHECK_ARG(muls[i] != nullptr && outs[i] != nullptr && outs[i] != data && outs[i] != muls[i]);
    AST_CHECK_ARG(nmul == 1 || outs[0] != outs[1]);
    Split sp;
    AST_CHECK_ARG(split_of(len, sp));
    AST_CHECK_ARG(nonzero_rows == len || nonzero_rows == len / 2);
    AST_CHECK_ARG(keep_rows == len || keep_rows == len / 2);
    hipStream_t s = ast::as_stream(stream);
    const double2* tw = g_tw.get((int)len, s);
    if (!tw) { ast::set_error("ast_lens_cols_convolve: twiddle table allocation failed"); return AST_ERR_HIP; }
    double2* d = (double2*)data;
    {
        AST_PROF("lens.cols_fwd", s);
        LENS_FWD(pass_of<false>(sp.n1, d, nullptr, d, pitch, (int)ncols, sp.n2, 1, (size_t)sp.n2,
                               nonzero_rows == len ? sp.n1 : sp.n1 / 2, sp.n1, 1, tw, (int)len, s));
    }
    {
        AST_PROF("lens.cols_mid", s);
        const double2* const* m = (const double2* const*)muls;
        double2* const* o = (double2* const*)outs;
        LENS_FWD(nmul == 2 ? mid_of<2>(sp.n2, d, m, o, pitch, (int)ncols, sp.n1, tw, (int)len, s)
                          : mid_of<1>(sp.n2, d, m, o, pitch, (int)ncols, sp.n1, tw, (int)len, s));
    }
    AST_PROF("lens.cols_inv", s);
    for (int i = 0; i < nmul; ++i) {
        double2* o = (double2*)outs[i];
        LENS_FWD(pass_of<true>(sp.n1, o, nullptr, o, pitch, (int)ncols, sp.n2, 1, (size_t)sp.n2, sp.n1,
                              keep_rows == len ? sp.n1 : sp.n1 / 2, 0, tw, (int)len, s));
    }
    return AST_OK;
}

extern "C" int ast_lens_rows_supported(size_t nc) {
    return nc == 8192 || nc == 4096 || nc == 2048 || nc == 1024 || nc == 512 || nc == 256 || nc == 128;
}

namespace {
template <int RA, int RB, int RC, bool ZPAD = true, int FOLDW = 0, typename TIN = double>
int rows_forward_launch(const TIN* kappa, size_t nrows, double2* spec, size_t pitch, const double2* twM, const double2* twL, hipStream_t s,
                        size_t in_pitch = 0, double scale = 1.0, const double* rec = nullptr) {
    using G = RowGeo<RA, RB, RC>;
    const size_t lds = (size_t)(G::M + G::M / 8) * sizeof(double2);
    static ast::PerDeviceOnce once;
    if (once.need() && lds > 48 * 1024) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lens_rows_forward_kernel<RA, RB, RC, ZPAD, FOLDW, TIN>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
    AST_CHECK_ARG(nrows < 0x7fffffffull);
    lens_rows_forward_kernel<RA, RB, RC, ZPAD, FOLDW, TIN><<<(unsigned)nrows, G::NT, lds, s>>>(kappa, in_pitch ? in_pitch : (size_t)G::M, spec, pitch,
                                                                                             twM, twL, scale, rec);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
template <int RA, int RB, int RC>
int rows_inverse_launch(const double2* spec, size_t pitch, size_t nc, double scale, double* out, const double2* twM, const double2* twL,
                        hipStream_t s) {
    using G = RowGeo<RA, RB, RC>;
    const size_t lds = (size_t)(G::M + G::M / 8) * sizeof(double2);
    static ast::PerDeviceOnce once;
    if (once.need() && lds > 48 * 1024) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lens_rows_inverse_kernel<RA, RB, RC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
    lens_rows_inverse_kernel<RA, RB, RC><<<(unsigned)nc, G::NT, lds, s>>>(spec, pitch, (int)nc, scale, out, twM, twL);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
}  // namespace

// spec_d[r][0 .. nc] (row pitch `pitch` complex) = R2C of length 2 nc of (kappa_d[r][0 .. nc), nc zeros), r < nc.
extern "C" int ast_lens_rows_forward(const double* kappa, size_t nc, void* spec, size_t pitch, void* stream) {
    AST_CHECK_ARG(kappa != nullptr && spec != nullptr && ast_lens_rows_supported(nc) && pitch >= nc + 1);
    AST_CHECK_ARG(((uintptr_t)kappa & 15) == 0);
    hipStream_t s = ast::as_stream(stream);
    const double2* twM = g_tw.get((int)nc, s);
    const double2* twL = g_tw.get((int)(2 * nc), s);
    if (!twM || !twL) { ast::set_error("ast_lens_rows_forward: twiddle table allocation failed"); return AST_ERR_HIP; }
    AST_PROF("lens.rows_fwd", s);
    double2* o = (double2*)spec;
    switch (nc) {
        case 8192: return rows_forward_launch<32, 16, 16>(kappa, nc, o, pitch, twM, twL, s);      // the reference's npix default (sky_array.py:266)
        case 4096: return rows_forward_launch<16, 16, 16>(kappa, nc, o, pitch, twM, twL, s);
        case 2048: return rows_forward_launch<16, 16, 8>(kappa, nc, o, pitch, twM, twL, s);
        case 1024: return rows_forward_launch<16, 8, 8>(kappa, nc, o, pitch, twM, twL, s);
        case 512: return rows_forward_launch<8, 8, 8>(kappa, nc, o, pitch, twM, twL, s);
        case 256: return rows_forward_launch<8, 8, 4>(kappa, nc, o, pitch, twM, twL, s);
        default: return rows_forward_launch<8, 4, 4>(kappa, nc, o, pitch, twM, twL, s);
    }
}

// spec_d[r][0 .. nc] = R2C of length 2 nc of the FULL row in_d[r][0 .. 2 nc) (no zero padding), r < nrows: the rows of the
// convolution kernels (lensing_funcs.c:45-83 fills all of the padded array), so that a plan needs no rocFFT at all.
extern "C" int ast_lens_rows_forward_full(const double* in, size_t nc, size_t nrows, void* spec, size_t pitch, void* stream) {
    AST_CHECK_ARG(in != nullptr && spec != nullptr && ast_lens_rows_supported(nc) && pitch >= nc + 1 && nrows >= 1);
    AST_CHECK_ARG(((uintptr_t)in & 15) == 0);
    hipStream_t s = ast::as_stream(stream);
    const double2* twM = g_tw.get((int)nc, s);
    const double2* twL = g_tw.get((int)(2 * nc), s);
    if (!twM || !twL) { ast::set_error("ast_lens_rows_forward_full: twiddle table allocation failed"); return AST_ERR_HIP; }
    AST_PROF("lens.rows_fwd", s);
    double2* o = (double2*)spec;
    switch (nc) {
        case 8192: return rows_forward_launch<32, 16, 16, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
        case 4096: return rows_forward_launch<16, 16, 16, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
        case 2048: return rows_forward_launch<16, 16, 8, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
        case 1024: return rows_forward_launch<16, 8, 8, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
        case 512: return rows_forward_launch<8, 8, 8, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
        case 256: return rows_forward_launch<8, 8, 4, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
        default: return rows_forward_launch<8, 4, 4, false>(in, nrows, o, pitch, twM, twL, s, 2 * nc);
    }
}

// out_d[r][0 .. nc) = scale * (the first nc reals of the length-2nc C2R of spec_d[r][0 .. nc]), r < nc (unnormalised C2R).
extern "C" int ast_lens_rows_inverse(const void* spec, size_t pitch, size_t nc, double scale, double* out, void* stream) {
    AST_CHECK_ARG(spec != nullptr && out != nullptr && ast_lens_rows_supported(nc) && pitch >= nc + 1);
    AST_CHECK_ARG(((uintptr_t)out & 15) == 0);
    hipStream_t s = ast::as_stream(stream);
    const double2* twM = g_tw.get((int)nc, s);
    const double2* twL = g_tw.get((int)(2 * nc), s);
    if (!twM || !twL) { ast::set_error("ast_lens_rows_inverse: twiddle table allocation failed"); return AST_ERR_HIP; }
    AST_PROF("lens.rows_inv", s);
    const double2* x = (const double2*)spec;
    switch (nc) {
        case 8192: return rows_inverse_launch<32, 16, 16>(x, pitch, nc, scale, out, twM, twL, s);
        case 4096: return rows_inverse_launch<16, 16, 16>(x, pitch, nc, scale, out, twM, twL, s);
        case 2048: return rows_inverse_launch<16, 16, 8>(x, pitch, nc, scale, out, twM, twL, s);
        case 1024: return rows_inverse_launch<16, 8, 8>(x, pitch, nc, scale, out, twM, twL, s);
        case 512: return rows_inverse_launch<8, 8, 8>(x, pitch, nc, scale, out, twM, twL, s);
        case 256: return rows_inverse_launch<8, 8, 4>(x, pitch, nc, scale, out, twM, twL, s);
        default: return rows_inverse_launch<8, 4, 4>(x, pitch, nc, scale, out, twM, twL, s);
    }
}

// ------------------------------------------------------------------ 3-D power spectrum of a double-precision grid
// FFTPower of an in-memory grid in the reference's own dtype (power_spectrum_3d.py:183-224 on float64 arrays): z rows by
// the row kernel above (no zero padding), y and x passes by col3_kernel - the three-stage scheme with the transform axis
// strided: a workgroup takes 4 or 8 adjacent k_z columns (64- or 128-byte row pieces) of all N = RA RB RC rows, one LDS
// line per column, ONE pass per axis where rocFFT's 3-D plan moves 2.1x the bytes.
// The x pass does not store: it adds w |delta_k|^2 of its modes to the workgroup's LDS shell table (integer rule or
// nbodykit's float64 edge rule, ast_common.h), written out as a row of `partial` and reduced in a fixed order.
template <int RA, int RB, int RC, int C, bool POWER>
__global__ void __launch_bounds__((C * RowGeo<RA, RB, RC>::NT))
col3_kernel(double2* __restrict__ data, const double2* __restrict__ twM, size_t elem_stride, size_t ncols, size_t batch_stride,
            unsigned tiles_per_batch, double scale, double* __restrict__ partial, double kf_rule, int prune2 = 0) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int N = G::M, NB = N / 2 - 1, LINE = N + N / 8;
    extern __shared__ double2 lds64[];
    double* shell = reinterpret_cast<double*>(lds64 + C * LINE);          // [NB + 1] when POWER
    const int c = threadIdx.x % C, t = threadIdx.x / C;
    // C * 16 B < 128 B (N = 1024: four columns): a workgroup touches HALF of every 128-byte line, the workgroup of the next
    // tile the other half.  Workgroups go to the 8 XCDs round robin, so those two ran behind different L2s and every line
    // crossed HBM twice (x pass: 8.6 GB read in 3.3 ms = 2.6 TB/s apparent).  COL3_XCD_PAIR: consecutive tiles are mapped
    // to consecutive launch slots of ONE XCD, so the second one finds the line in that L2.  (All tiles of a batch row on one
    // XCD, as fft_tile.hip's strided passes do for arrays of the natural pitch, measured the same here: the pitch of the
    // power pipeline's scratch spectrum is a whole number of lines, only the two halves of a line are shared.)
#ifndef COL3_XCD_PAIR
#define COL3_XCD_PAIR 1
#endif
    unsigned bid = blockIdx.x;
    const size_t row_pitch = elem_stride < batch_stride ? elem_stride : batch_stride;
    if ((row_pitch & 7) != 0 && (gridDim.x / tiles_per_batch) % 8u == 0u) {
        // rows that are NOT a whole number of lines (ast_fft64_r2c_3d's contiguous n / 2 + 1): a tile's pieces straddle into
        // the lines of the tiles on either side, so all tiles of a batch row follow each other on one XCD (batch b on XCD
        // b mod 8), as in fft_tile.hip's strided passes
        const unsigned xcd = bid % 8, slot = bid / 8;
        bid = ((slot / tiles_per_batch) * 8 + xcd) * tiles_per_batch + slot % tiles_per_batch;
    } else if (COL3_XCD_PAIR && C * sizeof(double2) < 128) {
        const unsigned full = gridDim.x / 16 * 16;
        if (bid < full) {
            const unsigned xcd = bid % 8, slot = bid / 8;
            bid = ((slot >> 1) * 8 + xcd) * 2 + (slot & 1);
        }
    }
    const unsigned tile = bid % tiles_per_batch, b = bid / tiles_per_batch;
    const size_t c0 = (size_t)tile * C;
    const bool col_ok = c0 + c < ncols;
    // FORWARD PRUNING (prune2 = (N/2)^2 on the power pipeline's two strided passes, as in fft_tile.hip): FFTPower keeps
    // |m| < N/2 (power_spectrum_3d.py:189-195; the edge itself stays for the float64 rule), so a row (k_y, tile from k_z0)
    // with k_y^2 + k_z0^2 > (N/2)^2 is not stored by the y pass and its tile is left out by the binning pass (its row of
    // `partial` is zeros) - 21.5 % of the half plane, with C = 4 or 8 columns cut close to the circle.
    if (POWER && prune2 > 0) {
        const int ky = (int)b > N / 2 ? (int)b - N : (int)b;
        if (ky * ky + (int)(c0 * c0) > prune2) {                            // block-uniform, before any barrier
            for (int i = threadIdx.x; i < NB; i += C * G::NT) partial[(size_t)bid * NB + i] = 0.0;
            return;
        }
    }
    double2* base = data + (size_t)b * batch_stride + min(c0 + c, ncols - 1);
    double2* Y = lds64 + c * LINE;
    if (POWER)
        for (int i = threadIdx.x; i <= NB; i += C * G::NT) shell[i] = 0.0;
    double2 va[RA], vc[RC];
    if (t < G::T1) {
#pragma unroll
        for (int a = 0; a < RA; ++a) va[a] = base[(size_t)(G::T1 * a + t) * elem_stride];
    }
    three_stage<RA, RB, RC>(va, vc, Y, twM, t);
    if (t < G::T3 && col_ok) {
        const int al = t / RB, be = t % RB;
        if (!POWER) {
            const int room = prune2 > 0 ? prune2 - (int)(c0 * c0) : 0x7fffffff;        // rows are k_y: keep k_y^2 <= room
#pragma unroll
            for (int ga = 0; ga < RC; ++ga) {
                double2 x = vc[bitrev(ga, ilog2(RC))];
                x.x *= scale;
                x.y *= scale;
                const int row = al + RA * be + RA * RB * ga, aky = RA * RB * ga >= N / 2 ? N - row : row;
                if (aky * aky <= room) base[(size_t)row * elem_stride] = x;
            }
        } else {
            const int kz = (int)(c0 + c);
            const int ky = (int)b > N / 2 ? (int)b - N : (int)b;
            const int m2yz = ky * ky + kz * kz;
            const double w = (kz > 0 && kz < N / 2) ? 2.0 : 1.0;
#pragma unroll
            for (int ga = 0; ga < RC; ++ga) {
                const double2 x = vc[bitrev(ga, ilog2(RC))];
                const int row = al + RA * be + RA * RB * ga;
                const int kx = row > N / 2 ? row - N : row;
                const int m2 = kx * kx + m2yz;
                // floor(sqrt(m2)) for 0 <= m2 <= 3 * 512^2 from one raw v_sqrt_f32 of m2 + 1/2 (exact: fft_tile.hip's
                // tile_isqrt, checked exhaustively by test_gpu_fft_tile) - a double square root plus two repair loops per
                // mode were a third of this pass
                int r = (int)__builtin_amdgcn_sqrtf((float)m2 + 0.5f);
                if (N > 1024) {                           // (beyond the range of the exhaustive check: one repair step either way)
                    if (r * r > m2) --r;
                    else if ((r + 1) * (r + 1) <= m2) ++r;
                }
                if (kf_rule != 0.0 && r > 0 && r * r == m2) r = ast::float64_edge_norm(r, kx, ky, kz, kf_rule);
                if (r >= 1 && r <= NB) atomicAdd(&shell[r], (x.x * x.x + x.y * x.y) * w);      // shell = r - 1
            }
        }
    }
    if (POWER) {
        __syncthreads();
        const double s2 = scale * scale;
        for (int i = threadIdx.x; i < NB; i += C * G::NT) partial[(size_t)bid * NB + i] = shell[i + 1] * s2;
    }
}

// psum[bin] += pnorm * sum over workgroups of partial[wg][bin], fixed order: REDUCE64 blocks each add a contiguous range
// of workgroup rows (lanes = consecutive bins), then one block adds those.
constexpr int REDUCE64 = 1024;
__global__ void __launch_bounds__(256)
power64_stage1_kernel(const double* __restrict__ partial, size_t nwg, int nb, double* __restrict__ out) {
    const size_t per = (nwg + REDUCE64 - 1) / REDUCE64;
    const size_t w0 = (size_t)blockIdx.x * per, w1 = w0 + per < nwg ? w0 + per : nwg;
    for (int i = threadIdx.x; i < nb; i += 256) {
        double acc = 0.0;
        for (size_t wg = w0; wg < w1; ++wg) acc += partial[wg * nb + i];
        out[(size_t)blockIdx.x * nb + i] = acc;
    }
}
__global__ void __launch_bounds__(256)
power64_stage2_kernel(const double* __restrict__ part, int nb, double pnorm, double* __restrict__ psum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    double acc = 0.0;
    for (int r = 0; r < REDUCE64; ++r) acc += part[(size_t)r * nb + i];
    psum[i] += pnorm * acc;
}

namespace {
// columns per workgroup: 4 at N = 1024 (74 KB of LDS: two workgroups per CU; measured 4.5 + 4.0 ms for the y and x passes
// against 5.4 + 4.5 with 8 columns and one workgroup per CU, 7.3 + 5.8 with 2), 8 below
// (N = 2048: 4 columns = 147 KB, one 1024-thread workgroup per CU; N = 128: 8 columns, 18 KB)
constexpr int col3_columns(size_t n) { return n >= 1024 ? 4 : 8; }
template <int RA, int RB, int RC, bool POWER>
int col3_launch(double2* data, const double2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride, double scale,
                double* partial, double kf_rule, hipStream_t s, int prune2 = 0) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int C = col3_columns(G::M), LINE = G::M + G::M / 8;
    const size_t lds = (size_t)C * LINE * sizeof(double2) + (POWER ? (G::M / 2) * sizeof(double) : 0);
    static ast::PerDeviceOnce once;
    if (once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&col3_kernel<RA, RB, RC, C, POWER>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
    const size_t tiles = (ncols + C - 1) / C;
    AST_CHECK_ARG(tiles * batch < 0x7fffffffull);
    col3_kernel<RA, RB, RC, C, POWER><<<(unsigned)(tiles * batch), C * G::NT, lds, s>>>(data, tw, elem_stride, ncols, batch_stride,
                                                                                        (unsigned)tiles, scale, partial, kf_rule, prune2);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
template <bool POWER>
int col3_dispatch(size_t n, double2* data, const double2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride,
                  double scale, double* partial, double kf_rule, hipStream_t s, int prune2 = 0) {
    if (n == 2048) return col3_launch<16, 16, 8, POWER>(data, tw, elem_stride, ncols, batch, batch_stride, scale, partial, kf_rule, s, prune2);
    if (n == 128) return col3_launch<8, 4, 4, POWER>(data, tw, elem_stride, ncols, batch, batch_stride, scale, partial, kf_rule, s, prune2);
    if (n == 1024) return col3_launch<16, 8, 8, POWER>(data, tw, elem_stride, ncols, batch, batch_stride, scale, partial, kf_rule, s, prune2);
    if (n == 512) return col3_launch<8, 8, 8, POWER>(data, tw, elem_stride, ncols, batch, batch_stride, scale, partial, kf_rule, s, prune2);
    return col3_launch<8, 8, 4, POWER>(data, tw, elem_stride, ncols, batch, batch_stride, scale, partial, kf_rule, s, prune2);
}
}  // namespace

// 128 (BASELINE config A's own size) ... 2048 (the largest cube whose float64 grid and scratch spectrum - 69 GB each - fit one
// MI355X; domain_level is arbitrary in the reference, power_spectrum_3d.py:183-188)
extern "C" int ast_fft64_supported(size_t n) { return n == 128 || n == 256 || n == 512 || n == 1024 || n == 2048; }

// row pitch of the scratch spectrum (complex): n / 2 + 1 rounded up to 8 (128-byte pieces stay line aligned)
static size_t fft64_pitch(size_t n) { return (n / 2 + 1 + 7) / 8 * 8; }
extern "C" size_t ast_fft64_power_scratch_bytes(size_t n) {
    const size_t nzp = fft64_pitch(n), cw = (size_t)col3_columns(n), tiles = (n / 2 + 1 + cw - 1) / cw, nb = n / 2 - 1;
    return n * n * nzp * sizeof(double2) + n * tiles * nb * sizeof(double) + (size_t)REDUCE64 * nb * sizeof(double);
}

// psum_d[shell] += L^3 sum over the shell's modes of w |delta_k|^2, delta_k = rfftn(grid) / n^3, for an (n, n, n) double
// grid: FFTPower(ArrayMesh(grid), mode="1d", kmin = k_F)'s shell sums (power_spectrum_3d.py:183-224) without the
// spectrum's last pass ever reaching HBM.  grid_d is not modified.
static int fft64_power_impl(const double* grid, const double* rec, int window, void* scratch, size_t scratch_bytes, size_t n,
                            double boxsize, int binning, double* psum, void* stream, const float* grid32 = nullptr);
extern "C" int ast_fft64_power_3d(const double* grid, void* scratch, size_t scratch_bytes, size_t n, double boxsize, int binning,
                                  double* psum, void* stream) {
    return fft64_power_impl(grid, nullptr, 0, scratch, scratch_bytes, n, boxsize, binning, psum, stream);
}
// The same for a SINGLE-precision grid, transformed in double (widened as the z pass loads the rows): the route of fp32 cubes
// the fp32 tile passes do not cover (128^3, 2048^3) - instead of a float64 copy of the grid in front of the passes.
extern "C" int ast_fft64_power_3d_f32(const float* grid, void* scratch, size_t scratch_bytes, size_t n, double boxsize, int binning,
                                      double* psum, void* stream) {
    AST_CHECK_ARG(grid != nullptr);
    return fft64_power_impl(nullptr, nullptr, 0, scratch, scratch_bytes, n, boxsize, binning, psum, stream, grid);
}
// The same for a grid painted with AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD: halo_rec_d (ast_paint_tiled_halo) is folded
// into the border rows as the z pass loads them.
extern "C" int ast_fft64_power_3d_halo(const double* grid, const double* halo_rec, int window, void* scratch, size_t scratch_bytes,
                                       size_t n, double boxsize, int binning, double* psum, void* stream) {
    AST_CHECK_ARG(halo_rec != nullptr && (window == AST_WIN_CIC || window == AST_WIN_TSC));
    return fft64_power_impl(grid, halo_rec, window, scratch, scratch_bytes, n, boxsize, binning, psum, stream);
}
static int fft64_power_impl(const double* grid, const double* rec, int window, void* scratch, size_t scratch_bytes, size_t n,
                            double boxsize, int binning, double* psum, void* stream, const float* grid32) {
    AST_CHECK_ARG((grid != nullptr || grid32 != nullptr) && scratch != nullptr && psum != nullptr && boxsize > 0.0);
    AST_CHECK_ARG(ast_fft64_supported(n) && scratch_bytes >= ast_fft64_power_scratch_bytes(n));
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    AST_CHECK_ARG(((uintptr_t)grid & 15) == 0 && ((uintptr_t)grid32 & 7) == 0 && (grid32 == nullptr || rec == nullptr));
    hipStream_t s = ast::as_stream(stream);
    const size_t nz = n / 2 + 1, nzp = fft64_pitch(n), cw = (size_t)col3_columns(n), tiles = (nz + cw - 1) / cw, nb = n / 2 - 1;
    double2* spec = (double2*)scratch;
    double* partial = (double*)((char*)scratch + n * n * nzp * sizeof(double2));
    double* part2 = partial + n * tiles * nb;
    const double2* twH = g_tw.get((int)(n / 2), s);       // the rows' half-length transform
    const double2* twN = g_tw.get((int)n, s);
    if (!twH || !twN) { ast::set_error("ast_fft64_power_3d: twiddle table allocation failed"); return AST_ERR_HIP; }
    const int prune2 = getenv("AST_FFT_NO_PRUNE") ? 0 : (int)((n / 2) * (n / 2));      // what FFTPower drops is neither stored nor binned
    int rc;
    {
        AST_PROF("fft64.rows_r2c", s);
        auto rows = [&](auto ra, auto rb, auto rcc) {
            constexpr int A = decltype(ra)::value, B = decltype(rb)::value, Cc = decltype(rcc)::value;
            if (grid32 != nullptr) return rows_forward_launch<A, B, Cc, false, 0, float>(grid32, n * n, spec, nzp, twH, twN, s, n, 1.0);
            if (rec == nullptr) return rows_forward_launch<A, B, Cc, false, 0>(grid, n * n, spec, nzp, twH, twN, s, n, 1.0);
            if (window == AST_WIN_CIC) return rows_forward_launch<A, B, Cc, false, 2>(grid, n * n, spec, nzp, twH, twN, s, n, 1.0, rec);
            return rows_forward_launch<A, B, Cc, false, 3>(grid, n * n, spec, nzp, twH, twN, s, n, 1.0, rec);
        };
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        using I16 = std::integral_constant<int, 16>;
        if (n == 2048) rc = rows(I16{}, I8{}, I8{});
        else if (n == 128) rc = rows(I4{}, I4{}, I4{});
        else if (n == 1024) rc = rows(I8{}, I8{}, I8{});
        else if (n == 512) rc = rows(I8{}, I8{}, I4{});
        else rc = rows(I8{}, I4{}, I4{});
        if (rc != AST_OK) return rc;
    }
    {
        AST_PROF("fft64.cols", s);
        rc = col3_dispatch<false>(n, spec, twN, nzp, nz, n, n * nzp, 1.0, nullptr, 0.0, s, prune2);       // y, per x plane
        if (rc != AST_OK) return rc;
    }
    const double kf_rule = binning == AST_BIN_FLOAT64 ? 2.0 * M_PI / boxsize : 0.0;
    {
        AST_PROF("fft64.cols_power", s);
        rc = col3_dispatch<true>(n, spec, twN, n * nzp, nz, n, nzp, 1.0 / ((double)n * (double)n * (double)n), partial, kf_rule, s, prune2);
        if (rc != AST_OK) return rc;
    }
    AST_PROF("fft64.shell_reduce", s);
    power64_stage1_kernel<<<REDUCE64, 256, 0, s>>>(partial, n * tiles, (int)nb, part2);
    power64_stage2_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(part2, (int)nb, boxsize * boxsize * boxsize, psum);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// rfftn(grid) * scale of an (n, n, n) float64 grid into spec_d (n, n, n / 2 + 1) complex, contiguous (pmesh's r2c with
// scale = 1 / n^3, power_spectrum_3d.py:189-195): the z rows and two strided passes above, in place on spec_d.
extern "C" int ast_fft64_r2c_3d(const double* grid, void* spec_out, size_t n, double scale, void* stream) {
    AST_CHECK_ARG(grid != nullptr && spec_out != nullptr && (const void*)grid != spec_out && ast_fft64_supported(n));
    AST_CHECK_ARG(((uintptr_t)grid & 15) == 0 && ((uintptr_t)spec_out & 15) == 0);
    hipStream_t s = ast::as_stream(stream);
    const size_t nz = n / 2 + 1;
    double2* spec = (double2*)spec_out;
    const double2* twH = g_tw.get((int)(n / 2), s);
    const double2* twN = g_tw.get((int)n, s);
    if (!twH || !twN) { ast::set_error("ast_fft64_r2c_3d: twiddle table allocation failed"); return AST_ERR_HIP; }
    int rc;
    {
        AST_PROF("fft64.rows_r2c", s);
        if (n == 2048) rc = rows_forward_launch<16, 8, 8, false, 0>(grid, n * n, spec, nz, twH, twN, s, n, 1.0);
        else if (n == 128) rc = rows_forward_launch<4, 4, 4, false, 0>(grid, n * n, spec, nz, twH, twN, s, n, 1.0);
        else if (n == 1024) rc = rows_forward_launch<8, 8, 8, false, 0>(grid, n * n, spec, nz, twH, twN, s, n, 1.0);
        else if (n == 512) rc = rows_forward_launch<8, 8, 4, false, 0>(grid, n * n, spec, nz, twH, twN, s, n, 1.0);
        else rc = rows_forward_launch<8, 4, 4, false, 0>(grid, n * n, spec, nz, twH, twN, s, n, 1.0);
        if (rc != AST_OK) return rc;
    }
    AST_PROF("fft64.cols", s);
    rc = col3_dispatch<false>(n, spec, twN, nz, nz, n, n * nz, 1.0, nullptr, 0.0, s);                      // y, per x plane
    if (rc != AST_OK) return rc;
    return col3_dispatch<false>(n, spec, twN, n * nz, nz, n, nz, scale, nullptr, 0.0, s);                 // x
}

// ================================================================== single-precision passes for cubes of side 2048
// The fp32 tile passes of fft_tile.hip are two register FFTs per axis (R1 x R2 <= 32 x 32 = 1024 points).  A cube of side
// 2048 - the largest whose fp32 grid (34 GB) and scratch spectrum (35 GB) one MI355X holds with room to spare - took the
// double passes above, widened on load: twice the bytes on every pass.  These are the same three-stage passes in float:
// z rows (half-length complex transform + untangling, the grid's mean subtracted as the rows are loaded), y pass, x pass
// fused with FFTPower's shell sums (power_spectrum_3d.py:189-224).  The strided passes take EIGHT k_z columns per workgroup
// (64-byte row pieces, the LDS of the double pass's four) and two columns per thread - one 16-byte access per row piece
// and lane, 1024 threads.  The lowest shells are patched from the double-precision side channel by the caller
// (ast_lowk_modes: fp32 round-off of an O(1) field on shells of a few dozen modes), as in the fp32 tile pipeline.
namespace {
__global__ void tw_table32_kernel(float2* out, int len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double sn, cs;
    sincospi(-2.0 * (double)i / (double)len, &sn, &cs);
    out[i] = make_float2((float)cs, (float)sn);
}
struct TwCache32 {
    std::mutex m;
    struct E { int dev, len; float2* d; };
    std::vector<E> tabs;
    const float2* get(int len, hipStream_t s) {
        std::lock_guard<std::mutex> lock(m);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& t : tabs) if (t.dev == dev && t.len == len) return t.d;
        float2* d = nullptr;
        if (hipMalloc(&d, (size_t)len * sizeof(float2)) != hipSuccess) return nullptr;
        tw_table32_kernel<<<(len + 255) / 256, 256, 0, s>>>(d, len);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(d); return nullptr; }
        tabs.push_back({dev, len, d});
        return d;
    }
} g_tw32;

// one row per workgroup: 2 M reals -> M + 1 complex (lens_rows_forward_kernel without padding, fold or widening)
// FOLDW = 2 / 3 (CIC / TSC): the grid is what a deferred-fold paint left and `rec` its halo records; the up to three record
// lines that end in a border row are added as the row is loaded - records first, then onto the row, the order of the paint's
// own fold kernel (as in the double row kernel above and fft_tile.hip's).
template <int RA, int RB, int RC, int FOLDW = 0>
__global__ void __launch_bounds__((RowGeo<RA, RB, RC>::NT))
rows32_forward_kernel(const float* __restrict__ grid, size_t in_pitch, float2* __restrict__ spec, size_t pitch,
                      const float2* __restrict__ twM, const float2* __restrict__ twL, float scale, float mean,
                      const float* __restrict__ rec = nullptr) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int M = G::M;
    extern __shared__ float2 Y32[];
    const int t = threadIdx.x;
    const size_t row = blockIdx.x;
    const float2* z = reinterpret_cast<const float2*>(grid + row * in_pitch);
    float2 va[RA], vc[RC];
    const float* src[3] = {nullptr, nullptr, nullptr};
    int ns = 0;
    if (FOLDW != 0) {
        constexpr int W = FOLDW != 0 ? FOLDW : 2;
        const int ng = 2 * M;
        ns = ast::halo_sources<float, W>(rec, (int)(row / ng), (int)(row % ng), ng, ng / ast::TX, ng / ast::TY, src);
    }
    if (t < G::T1) {
#pragma unroll
        for (int a = 0; a < RA; ++a) va[a] = z[G::T1 * a + t];
        if (FOLDW != 0 && ns > 0) {
            float2 h[RA];
#pragma unroll
            for (int a = 0; a < RA; ++a) h[a] = reinterpret_cast<const float2*>(src[0])[G::T1 * a + t];
            if (ns > 1) {
#pragma unroll
                for (int a = 0; a < RA; ++a) { const float2 g2 = reinterpret_cast<const float2*>(src[1])[G::T1 * a + t]; h[a].x += g2.x; h[a].y += g2.y; }
            }
            if (ns > 2) {
#pragma unroll
                for (int a = 0; a < RA; ++a) { const float2 g3 = reinterpret_cast<const float2*>(src[2])[G::T1 * a + t]; h[a].x += g3.x; h[a].y += g3.y; }
            }
#pragma unroll
            for (int a = 0; a < RA; ++a) { va[a].x += h[a].x; va[a].y += h[a].y; }
        }
#pragma unroll
        for (int a = 0; a < RA; ++a) {
            va[a].x -= mean;
            va[a].y -= mean;
        }
    }
    three_stage<RA, RB, RC>(va, vc, Y32, twM, t);
    constexpr int IT = (M / 2) / G::NT;
    static_assert(IT * G::NT == M / 2, "row length and thread count");
    float2 wk[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) wk[it] = twL[t + it * G::NT];
    __syncthreads();
    if (t < G::T3) {
#pragma unroll
        for (int ga = 0; ga < RC; ++ga) Y32[lds_pad((t / RB) + RA * (t % RB) + RA * RB * ga)] = vc[bitrev(ga, ilog2(RC))];
    }
    __syncthreads();
    float2* orow = spec + row * pitch;
    auto emit = [&](int k, float2 w) {
        const float2 zk = Y32[lds_pad(k)];
        const float2 zm = Y32[lds_pad(M - k)];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
        const float2 tt = cmul(o, w);
        orow[k] = make_float2((e.x + tt.y) * scale, (e.y - tt.x) * scale);
        orow[M - k] = make_float2((e.x - tt.y) * scale, (-e.y - tt.x) * scale);
    };
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int k = t + it * G::NT;
        if (k == 0) {
            const float2 zk = Y32[lds_pad(0)];
            orow[0] = make_float2((zk.x + zk.y) * scale, 0.0f);
            orow[M] = make_float2((zk.x - zk.y) * scale, 0.0f);
        } else {
            emit(k, wk[it]);
        }
    }
    if (t == 0) emit(M / 2, twL[M / 2]);
}

// strided pass over N = RA RB RC rows, 8 adjacent columns per workgroup, 2 per thread (col3_kernel in float)
constexpr int C32 = 8, CT32 = 2;
template <int RA, int RB, int RC, bool POWER>
__global__ void __launch_bounds__(((C32 / CT32) * RowGeo<RA, RB, RC>::NT))
col32_kernel(float2* __restrict__ data, const float2* __restrict__ twM, size_t elem_stride, size_t ncols, size_t batch_stride,
             unsigned tiles_per_batch, float scale, double* __restrict__ partial, double kf_rule, int prune2) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int N = G::M, NB = N / 2 - 1, LINE = N + N / 8, CS = C32 / CT32, NTH = CS * G::NT;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    extern __shared__ float2 L32[];
    double* shell = reinterpret_cast<double*>(L32 + C32 * LINE);          // [NB + 1] when POWER
    const int cs = threadIdx.x % CS, t = threadIdx.x / CS;
    // a workgroup touches HALF of every 128-byte line, the next tile the other half: both on one XCD (see col3_kernel)
    unsigned bid = blockIdx.x;
    {
        const unsigned full = gridDim.x / 16 * 16;
        if (bid < full) {
            const unsigned xcd = bid % 8, slot = bid / 8;
            bid = ((slot >> 1) * 8 + xcd) * 2 + (slot & 1);
        }
    }
    const unsigned tile = bid % tiles_per_batch, b = bid / tiles_per_batch;
    const size_t c0 = (size_t)tile * C32;
    if (prune2 > 0 && POWER) {
        const int ky = (int)b > N / 2 ? (int)b - N : (int)b;
        if (ky * ky + (int)(c0 * c0) > prune2) {                            // block-uniform, before any barrier
            for (int i = threadIdx.x; i < NB; i += NTH) partial[(size_t)bid * NB + i] = 0.0;
            return;
        }
    }
    // columns c0 + 2 cs, c0 + 2 cs + 1 as ONE 16-byte access (the row pitch is even and the tile starts at a multiple of 8);
    // a pair past the last column re-reads the last valid pair and is never stored or binned
    const size_t cpair = c0 + (size_t)(CT32 * cs);
    const size_t cload = cpair + 1 < ncols + (ncols & 1) ? cpair : ((ncols - 1) & ~(size_t)1);
    float2* base = data + (size_t)b * batch_stride + cload;
    const bool ok0 = cpair < ncols, ok1 = cpair + 1 < ncols;
    float2* Y0 = L32 + (CT32 * cs) * LINE;
    float2* Y1 = Y0 + LINE;
    if (POWER)
        for (int i = threadIdx.x; i <= NB; i += NTH) shell[i] = 0.0;
    float2 va0[RA], va1[RA], vc0[RC], vc1[RC];
    if (t < G::T1) {
#pragma unroll
        for (int a = 0; a < RA; ++a) {
            const vf4 v = *reinterpret_cast<const vf4*>(base + (size_t)(G::T1 * a + t) * elem_stride);
            va0[a] = make_float2(v.x, v.y);
            va1[a] = make_float2(v.z, v.w);
        }
    }
    // the two columns' stages between the same barriers
    if (t < G::T1) {
        fft_reg<RA>(va0);
        fft_reg<RA>(va1);
#pragma unroll
        for (int al = 0; al < RA; ++al) {
            float2 y0 = va0[bitrev(al, ilog2(RA))], y1 = va1[bitrev(al, ilog2(RA))];
            if (al != 0) { const float2 w = twM[t * al]; y0 = cmul(y0, w); y1 = cmul(y1, w); }
            Y0[lds_pad(al * G::T1 + t)] = y0;
            Y1[lds_pad(al * G::T1 + t)] = y1;
        }
    }
    __syncthreads();
    if (t < G::T2) {
        const int al = t / RC, c = t % RC;
#pragma unroll
        for (int j = 0; j < CT32; ++j) {
            float2* Y = j ? Y1 : Y0;
            float2 vb[RB];
#pragma unroll
            for (int bb = 0; bb < RB; ++bb) vb[bb] = Y[lds_pad(al * G::T1 + RC * bb + c)];
            fft_reg<RB>(vb);
#pragma unroll
            for (int be = 0; be < RB; ++be) {
                float2 y = vb[bitrev(be, ilog2(RB))];
                if (be != 0 && c != 0) y = cmul(y, twM[RA * c * be]);
                Y[lds_pad(al * G::T1 + RC * be + c)] = y;
            }
        }
    }
    __syncthreads();
    if (t < G::T3) {
#pragma unroll
        for (int c = 0; c < RC; ++c) { vc0[c] = Y0[lds_pad(t * RC + c)]; vc1[c] = Y1[lds_pad(t * RC + c)]; }
        fft_reg<RC>(vc0);
        fft_reg<RC>(vc1);
        const int al = t / RB, be = t % RB;
        if (!POWER) {
            if (ok0) {
                const int room = prune2 > 0 ? prune2 - (int)(c0 * c0) : 0x7fffffff;        // rows are k_y: keep k_y^2 <= room
#pragma unroll
                for (int ga = 0; ga < RC; ++ga) {
                    const float2 x0 = vc0[bitrev(ga, ilog2(RC))], x1 = vc1[bitrev(ga, ilog2(RC))];
                    const int row = al + RA * be + RA * RB * ga, aky = RA * RB * ga >= N / 2 ? N - row : row;
                    if (aky * aky <= room)
                        *reinterpret_cast<vf4*>(base + (size_t)row * elem_stride) = vf4{x0.x * scale, x0.y * scale, x1.x * scale, x1.y * scale};
                }
            }
        } else {
            const int ky = (int)b > N / 2 ? (int)b - N : (int)b;
#pragma unroll
            for (int j = 0; j < CT32; ++j) {
                if (!(j ? ok1 : ok0)) continue;
                const int kz = (int)cpair + j;
                const int m2yz = ky * ky + kz * kz;
                const float w = (kz > 0 && kz < N / 2) ? 2.0f : 1.0f;
#pragma unroll
                for (int ga = 0; ga < RC; ++ga) {
                    const float2 x = j ? vc1[bitrev(ga, ilog2(RC))] : vc0[bitrev(ga, ilog2(RC))];
                    const int row = al + RA * be + RA * RB * ga;
                    const int kx = row > N / 2 ? row - N : row;
                    const int m2 = kx * kx + m2yz;
                    int r = (int)__builtin_amdgcn_sqrtf((float)m2 + 0.5f);
                    if (r * r > m2) --r;                                      // (N = 2048: beyond the exhaustively checked range)
                    else if ((r + 1) * (r + 1) <= m2) ++r;
                    if (kf_rule != 0.0 && r > 0 && r * r == m2) r = ast::float64_edge_norm(r, kx, ky, kz, kf_rule);
                    // |delta_k|^2 in fp32 (one rounding per mode), accumulated in double - as in fft_tile.hip's binning pass
                    if (r >= 1 && r <= NB) atomicAdd(&shell[r], (double)(fmaf(x.x, x.x, x.y * x.y) * w));
                }
            }
        }
    }
    if (POWER) {
        __syncthreads();
        const double s2 = (double)scale * (double)scale;
        for (int i = threadIdx.x; i < NB; i += NTH) partial[(size_t)bid * NB + i] = shell[i + 1] * s2;
    }
}

template <int RA, int RB, int RC, bool POWER>
int col32_launch(float2* data, const float2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride, float scale,
                 double* partial, double kf_rule, hipStream_t s, int prune2) {
    using G = RowGeo<RA, RB, RC>;
    constexpr int LINE = G::M + G::M / 8;
    const size_t lds = (size_t)C32 * LINE * sizeof(float2) + (POWER ? (G::M / 2) * sizeof(double) : 0);
    static ast::PerDeviceOnce once;
    if (once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&col32_kernel<RA, RB, RC, POWER>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
    const size_t tiles = (ncols + C32 - 1) / C32;
    AST_CHECK_ARG(tiles * batch < 0x7fffffffull);
    col32_kernel<RA, RB, RC, POWER><<<(unsigned)(tiles * batch), (C32 / CT32) * G::NT, lds, s>>>(data, tw, elem_stride, ncols, batch_stride,
                                                                                             (unsigned)tiles, scale, partial, kf_rule, prune2);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
}  // namespace

// ------------------------------------------------------------------ the lowest shells in double, for the passes above
// A density painted from a lattice coarser than the grid carries O(1) power at the lattice's Bragg vectors and next to none
// at low k (1024^3 particles on 2048^3 cells: P(shell 5) = 1.4e-6 of the box volume): the fp32 passes' round-off - white,
// proportional to the field's rms - is 3e-6 of the signal at shell 5 and falls below 1e-6 only from shell 16 on (measured,
// scripts/perf_big32.py).  So the shells |m| < 17 come from double-precision sums over the grid, as the fp32 tile pipeline
// takes its five lowest (fft_tile.hip, MLOW) - with a box of |m_i| <= 17 instead of 6:
//   z  one wave per grid row: lane l holds f[l + 64 j]; per k_z the lane's NJ-term sum times e^{-2 pi i k_z l / n}, the 64
//      lanes' terms added in lane order through LDS                                   -> lowz[x][y][kz]
//   y  one workgroup per x plane, direct sums over y                                  -> lowy[x][ky][kz]
//   x  BIGLOW_XPARTS workgroups, each over its range of x, then the parts in order    -> modes[kx][ky][kz]
//   shells: one thread per shell walks all modes in a fixed order (integer rule, or nbodykit's float64 edge rule).
namespace {
constexpr int BIGLOW = 16;                         // shells 0 .. 15 (|m| in [1, 17))
constexpr int BIGBOX = BIGLOW + 1;                 // modes |m_i| <= 17: a vector of norm exactly 17 may fall into shell 15 (float64 rule)
constexpr int BIGKZ = BIGBOX + 1, BIGK = 2 * BIGBOX + 1;
constexpr int BIGLOW_MODES = BIGK * BIGK * BIGKZ;
constexpr int BIGLOW_XPARTS = 256;

template <int NJ, int FOLDW = 0>
__global__ void __launch_bounds__(256)
biglow_z_kernel(const float* __restrict__ grid, int n, size_t nrows, double mean, double2* __restrict__ out,
                const float* __restrict__ rec = nullptr) {
    __shared__ double part[4][2 * BIGKZ][65];                      // [wave][sum][lane] (65: the 36 readers on different banks)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double twc[BIGKZ], tws[BIGKZ];                                 // the lane's e^{-2 pi i kz lane / n}
#pragma unroll
    for (int kz = 0; kz < BIGKZ; ++kz) sincospi(-2.0 * (double)((kz * lane) % n) / (double)n, &tws[kz], &twc[kz]);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < nrows; row += nwaves) {
        const float* in = grid + row * (size_t)n;
        // the lane's NJ samples z = lane + 64 j through ONE NJ-point register FFT (all NJ outputs for the price of the few
        // needed: 80 butterflies at NJ = 32 against 18 x 32 multiply-adds in a dependent chain, 105 ms -> ... at side 2048)
        float fr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fr[j] = in[lane + 64 * j];
        if (FOLDW != 0) {                                  // the fp32 value the z rows see: records first, then onto the row
            constexpr int W = FOLDW != 0 ? FOLDW : 2;
            const float* src[3];
            const int ns = ast::halo_sources<float, W>(rec, (int)(row / n), (int)(row % n), n, n / ast::TX, n / ast::TY, src);
            if (ns > 0) {
                float h[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) h[j] = src[0][lane + 64 * j];
                if (ns > 1) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) h[j] += src[1][lane + 64 * j];
                }
                if (ns > 2) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) h[j] += src[2][lane + 64 * j];
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) fr[j] += h[j];
            }
        }
        double2 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = make_double2((double)fr[j] - mean, 0.0);
        fft_reg<NJ>(v);
#pragma unroll
        for (int kz = 0; kz < BIGKZ; ++kz) {
            const double2 x = v[bitrev(kz % NJ, ilog2(NJ))];
            part[wave][2 * kz][lane] = x.x * twc[kz] - x.y * tws[kz];
            part[wave][2 * kz + 1][lane] = x.x * tws[kz] + x.y * twc[kz];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 2 * BIGKZ) {                                    // the 64 lanes' terms in a fixed order, eight sums in flight
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int l = 0; l < 64; ++l) acc[l & 7] += part[wave][lane][l];
            reinterpret_cast<double*>(out)[row * (size_t)(2 * BIGKZ) + lane] =
                ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// out[(outer * nparts + part) * BIGK + (k + BIGBOX)][i] = sum over the part's t of in[(outer * nt + t) * inner + i] e^{-2 pi i k (t0 + t) / n}
__global__ void __launch_bounds__(256)
biglow_axis_kernel(const double2* __restrict__ in, int n, int nt, int inner, double2* __restrict__ out) {
    extern __shared__ double2 lds64[];
    double2* tw = lds64;                                            // e^{-2 pi i t / n}
    for (int t = threadIdx.x; t < n; t += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)t / (double)n, &sn, &cs);
        tw[t] = make_double2(cs, sn);
    }
    __syncthreads();
    const size_t outer = blockIdx.x;
    const int nparts = gridDim.y, part = blockIdx.y;
    const int per = (nt + nparts - 1) / nparts, ta = part * per, tb = min(nt, ta + per);
    const int nout = BIGK * inner;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int k = o / inner - BIGBOX, i = o % inner;
        double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};      // four loads in flight; added up in a fixed order
        for (int tq = ta; tq < tb; tq += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = tq + q < tb ? tq + q : tb - 1;          // unconditional load; the duplicate is not added
                const double2 v = in[(outer * nt + t) * inner + i];
                const double2 w = tw[(unsigned)(k * t) & (unsigned)(n - 1)];          // n is a power of two
                if (tq + q < tb) {
                    re[q] += v.x * w.x - v.y * w.y;
                    im[q] += v.x * w.y + v.y * w.x;
                }
            }
        }
        out[((outer * nparts + part) * BIGK + k + BIGBOX) * inner + i] = make_double2((re[0] + re[1]) + (re[2] + re[3]), (im[0] + im[1]) + (im[2] + im[3]));
    }
}

// the y sums of one x plane: lowz[x][y][kz] -> lowy[x][ky][kz].  The plane's rows come through LDS 32 at a time (whole lines,
// read once); a thread owns up to three (ky, kz) outputs.  (The generic kernel above took 5.8 ms here: every term a
// dependent global load.)
__global__ void __launch_bounds__(256)
biglow_y_kernel(const double2* __restrict__ lowz, int n, double2* __restrict__ lowy) {
    extern __shared__ double2 lds64[];
    double2* tw = lds64;                                            // e^{-2 pi i t / n}
    double2* chunk = lds64 + n;                                     // [32][BIGKZ]
    constexpr int TC = 32, NOUT = BIGK * BIGKZ, PER = (NOUT + 255) / 256;
    for (int t = threadIdx.x; t < n; t += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)t / (double)n, &sn, &cs);
        tw[t] = make_double2(cs, sn);
    }
    const double2* plane = lowz + (size_t)blockIdx.x * n * BIGKZ;
    double re[PER], im[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) { re[q] = 0.0; im[q] = 0.0; }
    for (int t0 = 0; t0 < n; t0 += TC) {
        __syncthreads();                                            // (the table; the previous chunk's readers)
        for (int e = threadIdx.x; e < TC * BIGKZ; e += 256) chunk[e] = plane[(size_t)t0 * BIGKZ + e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int o = threadIdx.x + 256 * q;
            if (o < NOUT) {
                const int k = o / BIGKZ - BIGBOX, i = o % BIGKZ;
#pragma unroll 8
                for (int tt = 0; tt < TC; ++tt) {
                    const double2 v = chunk[tt * BIGKZ + i];
                    const double2 w = tw[(unsigned)(k * (t0 + tt)) & (unsigned)(n - 1)];
                    re[q] += v.x * w.x - v.y * w.y;
                    im[q] += v.x * w.y + v.y * w.x;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int o = threadIdx.x + 256 * q;
        if (o < NOUT) lowy[(size_t)blockIdx.x * NOUT + o] = make_double2(re[q], im[q]);
    }
}

__global__ void __launch_bounds__(256)
biglow_parts_kernel(const double2* __restrict__ parts, int nparts, int count, double2* __restrict__ modes) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= count) return;
    double re = 0.0, im = 0.0;
    for (int p = 0; p < nparts; ++p) { re += parts[(size_t)p * count + o].x; im += parts[(size_t)p * count + o].y; }
    modes[o] = make_double2(re, im);
}

// psum[shell] += norm * sum over the box's modes of the shell of w |mode|^2, shells 0 .. BIGLOW - 1; modes[kx][ky][kz].
// 256 threads take the modes in turn, every thread its own 16 sums; thread `shell` then adds the 256 in thread order.
__global__ void __launch_bounds__(256)
biglow_shell_kernel(const double2* __restrict__ modes, double norm, double kf_rule, double* __restrict__ psum) {
    __shared__ double part[BIGLOW][257];
    double acc[BIGLOW];
#pragma unroll
    for (int sh = 0; sh < BIGLOW; ++sh) acc[sh] = 0.0;
    for (int o = threadIdx.x; o < BIGLOW_MODES; o += 256) {
        const int kz = o % BIGKZ, ky = (o / BIGKZ) % BIGK - BIGBOX, kx = o / (BIGKZ * BIGK) - BIGBOX;
        const int m2 = kx * kx + ky * ky + kz * kz;
        if (m2 == 0) continue;
        int r = (int)__builtin_amdgcn_sqrtf((float)m2 + 0.5f);       // m2 <= 3 * 17^2: exact after the repair step
        if (r * r > m2) --r;
        else if ((r + 1) * (r + 1) <= m2) ++r;
        if (kf_rule != 0.0 && r * r == m2) r = ast::float64_edge_norm(r, kx, ky, kz, kf_rule);
        const int shell = r - 1;
        if (shell < 0 || shell >= BIGLOW) continue;
        const double2 v = modes[o];
        const double p = (v.x * v.x + v.y * v.y) * (kz > 0 ? 2.0 : 1.0);
#pragma unroll
        for (int sh = 0; sh < BIGLOW; ++sh) acc[sh] += sh == shell ? p : 0.0;
    }
#pragma unroll
    for (int sh = 0; sh < BIGLOW; ++sh) part[sh][threadIdx.x] = acc[sh];
    __syncthreads();
    if (threadIdx.x < BIGLOW) {
        double total = 0.0;
        for (int t = 0; t < 256; ++t) total += part[threadIdx.x][t];
        psum[threadIdx.x] += norm * total;
    }
}

// the fp32 passes' sums for the shells from `first` on (the ones below come from the double-precision box)
__global__ void __launch_bounds__(256)
power32_stage2_kernel(const double* __restrict__ part, int nb, int first, double pnorm, double* __restrict__ psum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb || i < first) return;
    double acc = 0.0;
    for (int r = 0; r < REDUCE64; ++r) acc += part[(size_t)r * nb + i];
    psum[i] += pnorm * acc;
}

size_t biglow_bytes(size_t n) {
    return (n * n * BIGKZ + n * BIGK * BIGKZ + (size_t)(BIGLOW_XPARTS + 1) * BIGLOW_MODES) * sizeof(double2);
}
}  // namespace

// 2048 is what these passes are for; 256 (covered by the fp32 tile passes in production) is there so that the same kernels
// can be checked against the CPU oracle at a size it finishes in seconds
extern "C" int ast_fft32_big_supported(size_t n) { return n == 2048 || n == 256; }
static size_t fft32_pitch(size_t n) { return (n / 2 + 1 + 15) / 16 * 16; }        // complex64 elements: whole 128-byte lines
extern "C" size_t ast_fft32_big_power_scratch_bytes(size_t n) {
    const size_t nzp = fft32_pitch(n), tiles = (n / 2 + 1 + C32 - 1) / C32, nb = n / 2 - 1;
    return n * n * nzp * sizeof(float2) + n * tiles * nb * sizeof(double) + (size_t)REDUCE64 * nb * sizeof(double) + biglow_bytes(n);
}

// psum_d[shell] += L^3 sum over the shell's modes of w |delta_k|^2, delta_k = rfftn(grid - mean) / n^3, for an (n, n, n)
// float grid of side 2048: all passes in single precision, the shells |m| < 17 from double-precision sums over the grid
// (see the two section headers).  grid_d is not modified.
static int fft32_big_impl(const float* grid, const float* rec, int window, void* scratch, size_t scratch_bytes, size_t n,
                          double boxsize, int binning, double mean, double* psum, void* stream);
extern "C" int ast_fft32_big_power_3d(const float* grid, void* scratch, size_t scratch_bytes, size_t n, double boxsize, int binning,
                                      double mean, double* psum, void* stream) {
    return fft32_big_impl(grid, nullptr, 0, scratch, scratch_bytes, n, boxsize, binning, mean, psum, stream);
}
// The same for a grid painted with AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD: halo_rec_d (ast_paint_tiled_halo) is folded
// into the border rows as the z rows - and the low-k sums - load them.
extern "C" int ast_fft32_big_power_3d_halo(const float* grid, const float* halo_rec, int window, void* scratch, size_t scratch_bytes,
                                           size_t n, double boxsize, int binning, double mean, double* psum, void* stream) {
    AST_CHECK_ARG(halo_rec != nullptr && (window == AST_WIN_CIC || window == AST_WIN_TSC));
    return fft32_big_impl(grid, halo_rec, window, scratch, scratch_bytes, n, boxsize, binning, mean, psum, stream);
}
static int fft32_big_impl(const float* grid, const float* rec, int window, void* scratch, size_t scratch_bytes, size_t n,
                          double boxsize, int binning, double mean, double* psum, void* stream) {
    AST_CHECK_ARG(grid != nullptr && scratch != nullptr && psum != nullptr && boxsize > 0.0);
    AST_CHECK_ARG(ast_fft32_big_supported(n) && scratch_bytes >= ast_fft32_big_power_scratch_bytes(n));
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    AST_CHECK_ARG(((uintptr_t)grid & 7) == 0 && ((uintptr_t)scratch & 15) == 0);
    hipStream_t s = ast::as_stream(stream);
    const size_t nz = n / 2 + 1, nzp = fft32_pitch(n), tiles = (nz + C32 - 1) / C32, nb = n / 2 - 1;
    float2* spec = (float2*)scratch;
    double* partial = (double*)((char*)scratch + n * n * nzp * sizeof(float2));
    double* part2 = partial + n * tiles * nb;
    const float2* twH = g_tw32.get((int)(n / 2), s);
    const float2* twN = g_tw32.get((int)n, s);
    if (!twH || !twN) { ast::set_error("ast_fft32_big_power_3d: twiddle table allocation failed"); return AST_ERR_HIP; }
    const int prune2 = getenv("AST_FFT_NO_PRUNE") ? 0 : (int)((n / 2) * (n / 2));
    AST_CHECK_ARG(n * n < 0x7fffffffull);
    const float inv_ng = (float)(1.0 / ((double)n * (double)n * (double)n));
    const double kf_rule = binning == AST_BIN_FLOAT64 ? 2.0 * M_PI / boxsize : 0.0;
    int rc = AST_OK;
    auto passes = [&](auto ra, auto rb, auto rcc, auto ha, auto hb, auto hc) -> int {        // (RA, RB, RC) of the axis, of the half-length rows
        constexpr int A = decltype(ra)::value, B = decltype(rb)::value, Cc = decltype(rcc)::value;
        constexpr int HA = decltype(ha)::value, HB = decltype(hb)::value, HC = decltype(hc)::value;
        {
            AST_PROF("fft32big.rows_r2c", s);
            using G = RowGeo<HA, HB, HC>;
            const size_t lds = (size_t)(G::M + G::M / 8) * sizeof(float2);
            if (rec == nullptr) rows32_forward_kernel<HA, HB, HC, 0><<<(unsigned)(n * n), G::NT, lds, s>>>(grid, n, spec, nzp, twH, twN, 1.0f, (float)mean);
            else if (window == AST_WIN_CIC) rows32_forward_kernel<HA, HB, HC, 2><<<(unsigned)(n * n), G::NT, lds, s>>>(grid, n, spec, nzp, twH, twN, 1.0f, (float)mean, rec);
            else rows32_forward_kernel<HA, HB, HC, 3><<<(unsigned)(n * n), G::NT, lds, s>>>(grid, n, spec, nzp, twH, twN, 1.0f, (float)mean, rec);
            AST_CHECK_LAUNCH();
        }
        {
            AST_PROF("fft32big.cols", s);
            LENS_FWD((col32_launch<A, B, Cc, false>(spec, twN, nzp, nz, n, n * nzp, 1.0f, nullptr, 0.0, s, prune2)));           // y, per x plane
        }
        AST_PROF("fft32big.cols_power", s);
        return col32_launch<A, B, Cc, true>(spec, twN, n * nzp, nz, n, nzp, inv_ng, partial, kf_rule, s, prune2);
    };
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    if (n == 2048) rc = passes(I16{}, I16{}, I8{}, I16{}, I8{}, I8{});
    else rc = passes(I8{}, I8{}, I4{}, I8{}, I4{}, I4{});
    if (rc != AST_OK) return rc;
    {
        AST_PROF("fft32big.shell_reduce", s);
        power64_stage1_kernel<<<REDUCE64, 256, 0, s>>>(partial, n * tiles, (int)nb, part2);
        power32_stage2_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(part2, (int)nb, BIGLOW, boxsize * boxsize * boxsize, psum);
        AST_CHECK_LAUNCH();
    }
    double2* lowz = (double2*)(part2 + (size_t)REDUCE64 * nb);
    double2* lowy = lowz + n * n * BIGKZ;
    double2* parts = lowy + n * BIGK * BIGKZ;
    double2* modes = parts + (size_t)BIGLOW_XPARTS * BIGLOW_MODES;
    const unsigned zblocks = (unsigned)std::min<size_t>((n * n + 3) / 4, (size_t)256 * 8);
    {
        AST_PROF("fft32big.lowk_z", s);
        auto lowz_launch = [&](auto nj) {
            constexpr int NJ = decltype(nj)::value;
            if (rec == nullptr) biglow_z_kernel<NJ, 0><<<zblocks, 256, 0, s>>>(grid, (int)n, n * n, mean, lowz);
            else if (window == AST_WIN_CIC) biglow_z_kernel<NJ, 2><<<zblocks, 256, 0, s>>>(grid, (int)n, n * n, mean, lowz, rec);
            else biglow_z_kernel<NJ, 3><<<zblocks, 256, 0, s>>>(grid, (int)n, n * n, mean, lowz, rec);
        };
        if (n == 2048) lowz_launch(std::integral_constant<int, 32>{});
        else lowz_launch(std::integral_constant<int, 4>{});
    }
    const size_t lds = n * sizeof(double2);
    {
        AST_PROF("fft32big.lowk_y", s);
        biglow_y_kernel<<<(unsigned)n, 256, lds + 32 * BIGKZ * sizeof(double2), s>>>(lowz, (int)n, lowy);                  // y, per x plane
    }
    {
        AST_PROF("fft32big.lowk_x", s);
        biglow_axis_kernel<<<dim3(1, BIGLOW_XPARTS), 256, lds, s>>>(lowy, (int)n, (int)n, BIGK * BIGKZ, parts);            // x, in parts
    }
    AST_PROF("fft32big.lowk_shells", s);
    biglow_parts_kernel<<<(BIGLOW_MODES + 255) / 256, 256, 0, s>>>(parts, BIGLOW_XPARTS, BIGLOW_MODES, modes);
    const double ng = (double)n * (double)n * (double)n;
    biglow_shell_kernel<<<1, 256, 0, s>>>(modes, boxsize * boxsize * boxsize / (ng * ng), kf_rule, psum);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
