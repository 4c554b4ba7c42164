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
// Zero padding: the forward pass A reads the first `in_points` of its N1 points only (rows >= Nc hold zeros that are
// never stored either), the last inverse pass writes the first `out_points` only (the corner that is kept).
// Each workgroup works on 16 adjacent columns (256-byte row pieces): R1 x R2 register FFTs with one LDS exchange,
// the scheme of fft_tile.hip's strided pass in double.
#include "../../include/astrild_hip.h"
#include "ast_common.h"
#include <cmath>
#include <mutex>
#include <vector>

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

// forward FFT of R <= 16 points in registers, decimation in frequency: X[k] ends up in v[bitrev(k)]
template <int R>
__device__ inline void fft_reg(double2 (&v)[R]) {
#pragma unroll
    for (int h = R / 2; h >= 1; h /= 2) {
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int j = 0; j < h; ++j) {
                const double2 a = v[blk + j], b = v[blk + j + h];
                v[blk + j] = make_double2(a.x + b.x, a.y + b.y);
                const double2 d = make_double2(a.x - b.x, a.y - b.y);
                const int t = j * (8 / h);                   // W_{2h}^j = W_16^{j * 16 / (2h)}
                if (t == 0) v[blk + j + h] = d;
                else if (t == 4) v[blk + j + h] = make_double2(d.y, -d.x);          // * (-i)
                else v[blk + j + h] = cmul(d, make_double2(kC16[t], kS16[t]));
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
        if (hipGetLastError() != hipSuccess) return nullptr;
        tabs.push_back({dev, len, d});
        return d;
    }
} g_tw;

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
