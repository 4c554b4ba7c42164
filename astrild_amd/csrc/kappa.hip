// a-7 / a-8 / a-9: Ray-Ramses kappa-map stack and the per-map pipeline
// (kappa -> deflection / potential, Gaussian smoothing), fp64 like the
// reference.
//
// kappa -> alpha replaces rays/skys/lib_so_cgls/{lensing_funcs,fft_convolve}.c:
// the reference re-plans FFTW, rebuilds both kernels and transforms kappa
// twice on every call (fft_convolve.c:64-65 via lensing_funcs.c:103-104).
// Here a plan caches the three kernel spectra per (Nc, bsz); a call is one
// zero-pad kernel, one r2c, and per output one fused multiply, one c2r and one
// crop+scale kernel, all on (2Nc)^2 arrays resident in HBM.
#include "ast_common.h"
#include <cmath>
#include <mutex>
#include <vector>

namespace {

// ----------------------------------------------------------- kappa stack
// V pixels per thread (16-byte loads), planes four at a time: the four loads are issued before the first add, the
// adds stay in plane order (first = copy, then running +=: numpy's sequence, bit for bit)
#ifndef KSTACK_NT
#define KSTACK_NT 1                // 1: nontemporal loads (every plane is read exactly once)
#endif
#ifndef KSTACK_DEPTH
#define KSTACK_DEPTH 4             // planes whose loads are issued before the first add
#endif
template <typename T, int V, int D = KSTACK_DEPTH>
__global__ void __launch_bounds__(256)
kappa_stack_kernel(const T* const* __restrict__ planes, const double* __restrict__ wnum,
                   const double* __restrict__ wden, int nplanes, size_t count, T* __restrict__ out) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const size_t nvec = count / V;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto load = [&](int p, size_t i) -> vec_t {
        const vec_t* src = reinterpret_cast<const vec_t*>(planes[p]) + i;
        if (KSTACK_NT) return __builtin_nontemporal_load(src);
        return *src;
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        vec_t acc;
        auto term = [&](vec_t v, int p) {
            if (wnum) {
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] = (T)((double)v[k] * wnum[p] / wden[p]);   // quantity * g(x_mid, x_s') / g(x_mid, x_s)
            }
            return v;
        };
        int p = 0;
        for (; p + D <= nplanes; p += D) {
            vec_t v[D];
#pragma unroll
            for (int d = 0; d < D; ++d) v[d] = load(p + d, i);
#pragma unroll
            for (int d = 0; d < D; ++d) acc = (p == 0 && d == 0) ? term(v[d], p + d) : acc + term(v[d], p + d);
        }
        for (; p < nplanes; ++p) {
            const vec_t v = load(p, i);
            acc = p == 0 ? term(v, p) : acc + term(v, p);
        }
        reinterpret_cast<vec_t*>(out)[i] = acc;
    }
}

// ------------------------------------------------ kappa -> alpha / phi
// kernel_alphas_iso / kernel_phi_iso (lensing_funcs.c:45-83, 117-148) in closed
// form: the quarter plane i, j <= Ncc/2 is evaluated, the rest mirrored with
// the reference's parities.  which: 0 = alpha1, 1 = alpha2, 2 = phi.
__global__ void __launch_bounds__(256)
iso_kernel_build(int ncc, double dcell, int which, double* __restrict__ out, double rcut) {
    const size_t total = (size_t)ncc * ncc;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int h = ncc / 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / ncc), j = (int)(idx % ncc);
        const int ii = i <= h ? i : ncc - i, jj = j <= h ? j : ncc - j;
        const double x = (double)ii * dcell + 0.5 * dcell;
        const double y = (double)jj * dcell + 0.5 * dcell;
        const double r = sqrt(x * x + y * y);
        double v = 0.0;
        if (!(r > rcut)) {                                     // (the reference: r > Dcell * Ncc / 2)
            if (which == 0) v = x / (M_PI * r * r);
            else if (which == 1) v = y / (M_PI * r * r);
            else v = 1.0 / M_PI * log(r);
        }
        if (which == 0 && i > h) v = -v;
        if (which == 1 && j > h) v = -v;
        out[idx] = v;
    }
}

// zero_padding (lensing_funcs.c:8-19): kappa into the top-left nc x nc corner of the padded array; the rest of `out` is zero and stays zero (the plan
// owns it, nothing else writes there): 2 x 134 MB moved per map instead of the 671 MB of a full zero_pad pass
__global__ void __launch_bounds__(256)
pad_corner_kernel(const double* __restrict__ in, int nc, double* __restrict__ out) {
    const size_t n2 = 2 * (size_t)nc, total = (size_t)nc * nc;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const size_t i = idx / nc, j = idx % nc;
        out[i * n2 + j] = in[idx];
    }
}

// both products of kappa_to_alphas in one pass: the kappa spectrum is read once
__global__ void __launch_bounds__(256)
cmul2_kernel(const double2* __restrict__ a, const double2* __restrict__ b1, const double2* __restrict__ b2,
             double2* __restrict__ out1, double2* __restrict__ out2, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 x = a[i], y1 = b1[i], y2 = b2[i];
        double2 r1, r2;
        r1.x = x.x * y1.x - x.y * y1.y;      // fft_convolve.c:75-78
        r1.y = x.x * y1.y + x.y * y1.x;
        r2.x = x.x * y2.x - x.y * y2.y;
        r2.y = x.x * y2.y + x.y * y2.x;
        out1[i] = r1;
        out2[i] = r2;
    }
}

__global__ void __launch_bounds__(256)
cmul_kernel(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 x = a[i], y = b[i];
        double2 r;
        r.x = x.x * y.x - x.y * y.y;      // fft_convolve.c:75-78
        r.y = x.x * y.y + x.y * y.x;
        out[i] = r;
    }
}

// corner_matrix (lensing_funcs.c:33-43) fused with out/(nx*ny)*dx*dy (fft_convolve.c:88)
__global__ void __launch_bounds__(256)
crop_scale_kernel(const double* __restrict__ in, int nc, double dsx, double* __restrict__ out) {
    const size_t n2 = 2 * (size_t)nc, total = (size_t)nc * nc;
    const double nn = (double)(n2 * n2);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const size_t i = idx / nc, j = idx % nc;
        out[idx] = in[i * n2 + j] / nn * dsx * dsx;
    }
}

// ------------------------------------------------------ Gaussian smoothing
// exp(-0.5 * l^2 * (2 pi sigma)^2) on the half spectrum, then the 1/npix^2 of irfft2
__global__ void __launch_bounds__(256)
gauss_fft_filter_kernel(double2* __restrict__ spec, int npix, double sigma_px) {
    const int nh = npix / 2 + 1;
    const size_t total = (size_t)npix * nh;
    const double two_pi_s = 2.0 * M_PI * sigma_px;
    const double f2 = two_pi_s * two_pi_s;
    const double inv = 1.0 / ((double)npix * (double)npix);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / nh), j = (int)(idx % nh);
        // fftfreq(n): 0..(n-1)//2 then negative; rfftfreq(n): 0..n//2
        const int fi = i <= (npix - 1) / 2 ? i : i - npix;
        const double ly = (double)fi / (double)npix, lx = (double)j / (double)npix;
        const double l2 = lx * lx + ly * ly;
        const double g = exp(-0.5 * l2 * f2) * inv;
        double2 v = spec[idx];
        v.x *= g;
        v.y *= g;
        spec[idx] = v;
    }
}

// scipy.ndimage.correlate1d with a symmetric kernel, mode="reflect"
// (d c b a | a b c d | d c b a), along `axis` of an npix x npix map.
__device__ inline int reflect_idx(int i, int n) {
    // half-sample symmetric extension, valid for any offset
    const int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - 1 - m;
}
// mode="mirror" (d c b | a b c d | c b a): whole-sample symmetric, the edge pixel is not repeated; period 2n - 2
__device__ inline int mirror_idx(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    int m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - m;
}

__global__ void __launch_bounds__(256)
gauss_real_pass_kernel(const double* __restrict__ in, double* __restrict__ out, int npix, int axis,
                       const double* __restrict__ w, int radius, int mirror) {
    const size_t total = (size_t)npix * npix;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / npix), j = (int)(idx % npix);
        const int c = axis == 0 ? i : j;
        auto at = [&](int k) -> double {
            const int r = mirror ? mirror_idx(k, npix) : reflect_idx(k, npix);
            return axis == 0 ? in[(size_t)r * npix + j] : in[(size_t)i * npix + r];
        };
        // ni_filters.c NI_Correlate1D symmetric branch: centre tap, then pairs
        double acc = at(c) * w[radius];
        for (int k = -radius; k < 0; ++k) acc += (at(c + k) + at(c - k)) * w[k + radius];
        out[idx] = acc;
    }
}

// ------------------------------------------- "gaussianFFT" without the FFT
// Multiplying the spectrum by exp(-(2 pi sigma l)^2 / 2) is the periodic convolution with the sampled Gaussian
// g[d] = exp(-d^2 / 2 sigma^2) / (sigma sqrt(2 pi)) up to the aliases of either: exp(-(pi sigma)^2 / 2 ... ) < 4e-14 of the
// peak for sigma >= 2.5 px in Fourier space, exp(-R^2 / 2 sigma^2) < 2e-16 for the taps beyond R = 8.5 sigma in real
// space.  For 2.5 <= sigma_px <= 18.8 (R <= 160) the smoothing therefore runs as two separable periodic passes through
// LDS tiles - 0.54 GB moved instead of two 4096^2 double transforms - and agrees with the FFT route to 1e-13.
constexpr int GP_RMAX = 160;
// along the rows (contiguous axis): a workgroup produces 1024 consecutive pixels of one row, 4 per thread with a
// rolling register window (one LDS read per tap for 4 outputs)
__global__ void __launch_bounds__(256)
gauss_periodic_x_kernel(const double* __restrict__ in, double* __restrict__ out, int npix, const double* __restrict__ w, int radius) {
    __shared__ double line[1024 + 2 * GP_RMAX];
    const int row = blockIdx.y, x0 = blockIdx.x * 1024;
    const double* src = in + (size_t)row * npix;
    for (int i = threadIdx.x; i < 1024 + 2 * radius; i += 256) {
        int x = (x0 - radius + i) % npix;                 // periodic; the line may reach past a map narrower than 1024 pixels
        if (x < 0) x += npix;
        line[i] = src[x];
    }
    __syncthreads();
    const int o0 = threadIdx.x * 4;                        // outputs o0 .. o0 + 3 read line[o0 + t .. o0 + t + 3], t = 0 .. 2R
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    double v0 = line[o0], v1 = line[o0 + 1], v2 = line[o0 + 2];
    for (int t = 0; t <= 2 * radius; ++t) {
        const double v3 = line[o0 + t + 3 < 1024 + 2 * radius ? o0 + t + 3 : 0];
        const double wt = w[t];
        a0 = fma(wt, v0, a0);
        a1 = fma(wt, v1, a1);
        a2 = fma(wt, v2, a2);
        a3 = fma(wt, v3, a3);
        v0 = v1; v1 = v2; v2 = v3;
    }
    const int x = x0 + o0;
    double* dst = out + (size_t)row * npix;
    if (x + 3 < npix) {
        dst[x] = a0; dst[x + 1] = a1; dst[x + 2] = a2; dst[x + 3] = a3;
    } else {
        if (x < npix) dst[x] = a0;
        if (x + 1 < npix) dst[x + 1] = a1;
        if (x + 2 < npix) dst[x + 2] = a2;
    }
}
// along the columns: a workgroup produces a tile of 64 rows x 32 columns, thread = (column, group of 8 rows)
__global__ void __launch_bounds__(256)
gauss_periodic_y_kernel(const double* __restrict__ in, double* __restrict__ out, int npix, const double* __restrict__ w, int radius) {
    extern __shared__ double tile[];                      // (64 + 2 radius) rows x 32 columns
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int x = blockIdx.x * 32 + c, y0 = blockIdx.y * 64;
    const int xs = x < npix ? x : npix - 1;
    for (int i = rg; i < 64 + 2 * radius; i += 8) {
        int y = (y0 - radius + i) % npix;
        if (y < 0) y += npix;
        tile[i * 32 + c] = in[(size_t)y * npix + xs];
    }
    __syncthreads();
    const int o0 = rg * 8;
    double a[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = 0.0; v[j] = tile[(o0 + j) * 32 + c]; }
    for (int t = 0; t <= 2 * radius; ++t) {
        const double wt = w[t];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fma(wt, v[j], a[j]);
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = v[j + 1];
        const int nxt = o0 + t + 8;
        v[7] = tile[(nxt < 64 + 2 * radius ? nxt : 0) * 32 + c];
    }
    if (x < npix) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (y0 + o0 + j < npix) out[(size_t)(y0 + o0 + j) * npix + x] = a[j];
    }
}

#define AST_FWD(call)                  \
    do {                               \
        int rc_ = (call);              \
        if (rc_ != AST_OK) return rc_; \
    } while (0)

}  // namespace

struct ast_lens_plan {
    int nc = 0;                   // the map: nc x nc pixels of side bsz / nc
    int ncf = 0;                  // the transform: (2 ncf)^2.  ncf = nc, or - EMBEDDED - the next power of two >= nc
    double bsz = 0.0;
    // embedded plans (nc is not a size the hand-written passes cover): the convolution the reference computes on its
    // (2 nc)^2 periodic grid only ever pairs pixel offsets |d| < nc (kappa is zero outside its corner), so it is the
    // LINEAR convolution with K(d) = +-f((|d| + 1/2) dx), cut at r > nc dx - which any periodic grid of >= 2 nc - 1 points
    // computes as well.  kappa is copied into the corner of a zeroed ncf x ncf array, the kernels are sampled on the
    // (2 ncf)^2 grid with the MAP's dx and cut-off, and the nc x nc corner of the result is copied out.
    size_t pitch = 0;             // row pitch (complex elements) of spec / prod / kspec: lens_plan_pitch
    double* kin = nullptr;        // ncf x ncf staging input (zero outside the nc x nc corner, for the plan's lifetime)
    double* kout[2] = {nullptr, nullptr};
    ast_fft_plan* r2c = nullptr;
    ast_fft_plan* c2r = nullptr;
    double* pad = nullptr;        // (2nc)^2 real: kernel images (plan set-up), c2r output
    double* pad_in = nullptr;     // (2nc)^2 real: padded kappa; zero outside the top-left corner, for the plan's lifetime
    double2* prod2 = nullptr;     // second product spectrum (kappa_to_alphas forms both in one pass)
    double2* spec = nullptr;      // kappa spectrum
    double2* prod = nullptr;      // product spectrum (c2r input, overwritten by rocFFT)
    double2* kspec[3] = {nullptr, nullptr, nullptr};   // alpha1, alpha2, phi kernel spectra
    bool kready[3] = {false, false, false};
    // `cols`: the padded transform as row transforms (rocFFT, batched 1-D) of the nc rows that are not zero plus the
    // two-pass column transforms of lens_fft.hip (which skip the zero half and leave the spectra in their permuted row
    // order - every spectrum of this plan goes the same way).  Otherwise: rocFFT's 2-D plans r2c / c2r.
    bool cols = false;
    bool split_cols = false;               // AST_LENS_SPLIT_COLS: the kappa spectrum goes through memory (A / B measurements)
    bool rows = false;                     // with cols: the nc non-zero rows by lens_fft.hip's row kernels too (nc = 128 .. 4096):
                                           // kappa is read unpadded, the inverse stores the scaled corner
    ast_fft_plan* rows_fwd = nullptr;      // nc rows of 2nc reals -> nc rows of nc + 1 complex
    ast_fft_plan* rows_fwd_all = nullptr;  // all 2nc rows (kernel images)
    ast_fft_plan* rows_inv = nullptr;      // nc rows of nc + 1 complex -> nc rows of 2nc reals
};

struct ast_smooth_plan {
    int npix = 0;
    ast_fft_plan* r2c = nullptr;
    ast_fft_plan* c2r = nullptr;
    double2* spec = nullptr;
    double* tmp = nullptr;
    double* w_d = nullptr;
    int w_cap = 0;
    std::vector<double> w_h;
    double w_sigma = -1.0;        // the weights in w_d belong to this sigma and kind (0: none yet): the upload - a pageable,
    int w_kind = 0;               // stream-ordered copy that blocks the host behind everything queued - happens once per sigma
};

extern "C" int ast_kappa_stack(const void* const* planes, const double* wnum, const double* wden, int nplanes,
                               size_t count, int dtype, void* out, int aligned16, void* stream) {
    AST_CHECK_ARG(planes && out && nplanes >= 1);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG((wnum == nullptr) == (wden == nullptr));
    if (count == 0) return AST_OK;
    unsigned g = ast::stream_grid(count, 256);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("kappa_stack", s);
    // 16-byte accesses when every plane (and the output) is 16-byte aligned and the count divides; the pointers live
    // on the device, so alignment is the caller's promise via AST_KAPPA_ALIGNED below - checked here for `out` only
    const bool wide = aligned16 && count % (16 / (dtype == AST_F32 ? 4 : 8)) == 0 && ((uintptr_t)out & 15) == 0;
    if (dtype == AST_F32) {
        if (wide) kappa_stack_kernel<float, 4><<<g, 256, 0, s>>>((const float* const*)planes, wnum, wden, nplanes, count, (float*)out);
        else kappa_stack_kernel<float, 1><<<g, 256, 0, s>>>((const float* const*)planes, wnum, wden, nplanes, count, (float*)out);
    } else {
        static const int depth = getenv("AST_KSTACK_DEPTH") ? atoi(getenv("AST_KSTACK_DEPTH")) : KSTACK_DEPTH;
        static const int grid_env = getenv("AST_KSTACK_GRID") ? atoi(getenv("AST_KSTACK_GRID")) : 0;
        if (grid_env > 0 && (unsigned)grid_env < g) g = (unsigned)grid_env;
        if (wide && depth == 8) kappa_stack_kernel<double, 2, 8><<<g, 256, 0, s>>>((const double* const*)planes, wnum, wden, nplanes, count, (double*)out);
        else if (wide && depth == 16) kappa_stack_kernel<double, 2, 16><<<g, 256, 0, s>>>((const double* const*)planes, wnum, wden, nplanes, count, (double*)out);
        else if (wide) kappa_stack_kernel<double, 2><<<g, 256, 0, s>>>((const double* const*)planes, wnum, wden, nplanes, count, (double*)out);
        else kappa_stack_kernel<double, 1><<<g, 256, 0, s>>>((const double* const*)planes, wnum, wden, nplanes, count, (double*)out);
    }
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_lens_plan_destroy(ast_lens_plan* p) {
    if (!p) return AST_OK;
    ast_fft_plan_destroy(p->r2c);
    ast_fft_plan_destroy(p->c2r);
    ast_fft_plan_destroy(p->rows_fwd);
    ast_fft_plan_destroy(p->rows_fwd_all);
    ast_fft_plan_destroy(p->rows_inv);
    if (p->pad) (void)hipFree(p->pad);
    if (p->kin) (void)hipFree(p->kin);
    for (auto* k : p->kout) if (k) (void)hipFree(k);
    if (p->pad_in) (void)hipFree(p->pad_in);
    if (p->prod2) (void)hipFree(p->prod2);
    if (p->spec) (void)hipFree(p->spec);
    if (p->prod) (void)hipFree(p->prod);
    for (auto* k : p->kspec) if (k) (void)hipFree(k);
    delete p;
    return AST_OK;
}

__global__ void __launch_bounds__(256)
copy2d_kernel(const double* __restrict__ in, size_t in_pitch, double* __restrict__ out, size_t out_pitch, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t r = i / cols, c = i % cols;
        out[r * out_pitch + c] = in[r * in_pitch + c];
    }
}

// Row pitch of the plan's half spectra.  With the hand-written rows and columns it is a whole number of 128-byte lines: a
// column tile's row piece (16 columns, 256 bytes) then covers two lines of its own.  At the natural pitch nc + 1 the piece
// starts 16 bytes further in every row and shares its first and last line with the neighbouring tiles - which run on other
// XCDs, behind other L2s, so those lines were fetched twice.  (AST_LENS_PITCH_ALIGN: elements, 1 = natural pitch.)
static size_t lens_plan_pitch(bool hand_rows, size_t nh) {
    if (!hand_rows) return nh;                         // (the rocFFT row plans are made for the natural pitch)
    const char* v = getenv("AST_LENS_PITCH_ALIGN");
    const size_t a = v ? (size_t)atoll(v) : 8;
    return a > 1 ? (nh + a - 1) / a * a : nh;
}

extern "C" int ast_lens_plan_create(ast_lens_plan** out, int nc, double bsz) {
    AST_CHECK_ARG(out != nullptr && nc >= 1 && nc <= 16384 && bsz > 0.0);
    auto* p = new ast_lens_plan();
    p->nc = nc;
    p->ncf = nc;
    p->bsz = bsz;
    // a size the hand-written rows and columns do not cover is embedded in the next power of two that they do
    // (AST_LENS_NO_EMBED=1: rocFFT plans of the exact size instead, as in rounds 1-3)
    if (!(ast_lens_cols_supported(2 * (size_t)nc) && ast_lens_rows_supported((size_t)nc)) && nc <= 8192 && !getenv("AST_LENS_NO_EMBED") &&
        !getenv("AST_LENS_ROCFFT_2D") && !getenv("AST_LENS_ROCFFT_ROWS")) {
        int f = 128;
        while (f < nc) f *= 2;
        // ... unless the map is large AND just above a power of two (nc = 4100 -> 8192: 4 x the transform area and 15 GB of
        // plan buffers instead of 4): then the exact-size rocFFT plans of rounds 1-3 stay (AST_LENS_EMBED_ALWAYS=1 overrides)
        if (nc <= 2048 || 2 * (long long)f <= 3 * (long long)nc || getenv("AST_LENS_EMBED_ALWAYS")) p->ncf = f;
    }
    const bool embedded = p->ncf != nc;
    const size_t n2 = 2 * (size_t)p->ncf, nh = n2 / 2 + 1;
    const size_t lens[2] = {n2, n2};
    int rc = AST_OK;
    p->cols = ast_lens_cols_supported(n2) != 0 && !getenv("AST_LENS_ROCFFT_2D");
    p->rows = p->cols && ast_lens_rows_supported((size_t)p->ncf) != 0 && !getenv("AST_LENS_ROCFFT_ROWS");
    p->split_cols = getenv("AST_LENS_SPLIT_COLS") != nullptr;
    p->pitch = lens_plan_pitch(p->rows, nh);
    if (p->rows) {
        // hand-written rows and columns: no rocFFT plan at all
    } else if (p->cols) {
        const size_t len1[1] = {n2}, one[1] = {1};
        rc = ast_fft_plan_create_general(&p->rows_fwd, AST_FFT_R2C, AST_F64, 1, len1, one, one, (size_t)nc, n2, nh, 1.0, 0);
        if (rc == AST_OK) rc = ast_fft_plan_create_general(&p->rows_fwd_all, AST_FFT_R2C, AST_F64, 1, len1, one, one, n2, n2, nh, 1.0, 0);
        if (rc == AST_OK) rc = ast_fft_plan_create_general(&p->rows_inv, AST_FFT_C2R, AST_F64, 1, len1, one, one, (size_t)nc, nh, n2, 1.0, 0);
    } else {
        rc = ast_fft_plan_create(&p->r2c, AST_FFT_R2C, AST_F64, 2, lens, 1, 1.0, 0);
        if (rc == AST_OK) rc = ast_fft_plan_create(&p->c2r, AST_FFT_C2R, AST_F64, 2, lens, 1, 1.0, 0);
    }
    if (rc != AST_OK) { ast_lens_plan_destroy(p); return rc; }
    hipError_t e = hipMalloc(&p->pad, n2 * n2 * sizeof(double));
    if (e == hipSuccess && !p->rows) e = hipMalloc(&p->pad_in, n2 * n2 * sizeof(double));      // (the row kernels read kappa unpadded)
    if (e == hipSuccess && !p->rows) e = hipMemset(p->pad_in, 0, n2 * n2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&p->spec, n2 * p->pitch * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc(&p->prod, n2 * p->pitch * sizeof(double2));
    if (e == hipSuccess && embedded) {
        const size_t sq = (size_t)p->ncf * p->ncf * sizeof(double);
        e = hipMalloc(&p->kin, sq);
        if (e == hipSuccess) e = hipMemset(p->kin, 0, sq);
        if (e == hipSuccess) e = hipMalloc(&p->kout[0], sq);
        if (e == hipSuccess) e = hipMalloc(&p->kout[1], sq);
    }
    if (e != hipSuccess) {
        ast::set_error("ast_lens_plan_create: hipMalloc -> %s", hipGetErrorString(e));
        ast_lens_plan_destroy(p);
        return AST_ERR_HIP;
    }
    *out = p;
    return AST_OK;
}

// kernel spectrum `which`, built on first use and cached in the plan
static int lens_kernel_spectrum(ast_lens_plan* p, int which, hipStream_t s) {
    if (p->kready[which]) return AST_OK;
    const size_t n2 = 2 * (size_t)p->ncf, nh = n2 / 2 + 1;
    if (!p->kspec[which]) AST_CHECK_HIP(hipMalloc(&p->kspec[which], n2 * p->pitch * sizeof(double2)));
    const double dsx = p->bsz / (double)p->nc;
    // sampled with the MAP's pixel size and cut at the map's side nc dx (= Dcell * Ncc / 2 of lensing_funcs.c:58) - for an
    // embedded plan on the larger grid
    iso_kernel_build<<<ast::stream_grid(n2 * n2, 256), 256, 0, s>>>((int)n2, dsx, which, p->pad, dsx * (double)p->nc);
    AST_CHECK_LAUNCH();
    if (p->rows) {
        AST_FWD(ast_lens_rows_forward_full(p->pad, (size_t)p->ncf, n2, p->kspec[which], p->pitch, s));
        AST_FWD(ast_lens_cols_forward(p->kspec[which], n2, p->pitch, nh, n2, s));
    } else if (p->cols) {
        AST_FWD(ast_fft_exec(p->rows_fwd_all, p->pad, p->kspec[which], s));
        AST_FWD(ast_lens_cols_forward(p->kspec[which], n2, nh, nh, n2, s));
    } else {
        AST_FWD(ast_fft_exec(p->r2c, p->pad, p->kspec[which], s));
    }
    p->kready[which] = true;
    return AST_OK;
}

// rows of the padded kappa -> p->spec (nc rows of nc + 1 complex; the column transform is the caller's)
static int lens_rows_forward(ast_lens_plan* p, const double* kappa, hipStream_t s) {
    if (p->rows) return ast_lens_rows_forward(kappa, (size_t)p->ncf, p->spec, p->pitch, s);
    {
        AST_PROF("lens.zero_pad", s);
        pad_corner_kernel<<<ast::stream_grid((size_t)p->nc * p->nc, 256), 256, 0, s>>>(kappa, p->nc, p->pad_in);
    }
    AST_CHECK_LAUNCH();
    return ast_fft_exec(p->rows_fwd, p->pad_in, p->spec, s);                // rows 0 .. nc - 1; the rest is never read
}

// the row part after the columns: out = corner of the inverse row transforms of prod, scaled
static int lens_rows_inverse(ast_lens_plan* p, double2* prod, double* out, hipStream_t s) {
    const size_t n2 = 2 * (size_t)p->ncf;
    if (p->rows) {                                                          // corner_matrix and out / (nx ny) * dx dy in the store
        const double dsx = p->bsz / (double)p->nc;
        return ast_lens_rows_inverse(prod, p->pitch, (size_t)p->ncf, dsx * dsx / (double)(n2 * n2), out, s);
    }
    AST_FWD(ast_fft_exec(p->rows_inv, prod, p->pad, s));                    // nc rows of 2nc reals
    AST_PROF("lens.crop_scale", s);
    crop_scale_kernel<<<ast::stream_grid((size_t)p->nc * p->nc, 256), 256, 0, s>>>(p->pad, p->nc, p->bsz / (double)p->nc, out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// cols path: outs[m] = corner of irfft2(rfft2(padded kappa) * kspec[which[m]]), m < nmul <= 2.  Columns: one forward
// pass, one pass that finishes the forward transform, multiplies and starts the inverse(s), one inverse pass per output
// (ast_lens_cols_convolve); AST_LENS_SPLIT_COLS=1 keeps the spectrum in memory instead (forward, then one inverse per
// kernel with the product fused into its first pass: three array passes more for two outputs).
static int lens_convolve_cols(ast_lens_plan* p, const double* kappa_in, const int* which, double* const* outs_in, int nmul, hipStream_t s) {
    const size_t n2 = 2 * (size_t)p->ncf, nh = n2 / 2 + 1;
    const bool embedded = p->ncf != p->nc;
    const double* kappa = kappa_in;
    double* outs[2] = {outs_in[0], nmul == 2 ? outs_in[1] : nullptr};
    if (embedded) {                                   // map -> corner of the zeroed ncf x ncf array; results into staging arrays
        AST_PROF("lens.embed", s);
        copy2d_kernel<<<ast::stream_grid((size_t)p->nc * p->nc, 256), 256, 0, s>>>(kappa_in, (size_t)p->nc, p->kin, (size_t)p->ncf, p->nc, p->nc);
        AST_CHECK_LAUNCH();
        kappa = p->kin;
        outs[0] = p->kout[0];
        outs[1] = p->kout[1];
    }
    struct Crop {                                     // (runs on every return path below)
        ast_lens_plan* p; double* const* dst; int nmul; hipStream_t s; bool on;
        int run() const {
            if (!on) return AST_OK;
            AST_PROF("lens.embed", s);
            for (int m = 0; m < nmul; ++m)
                copy2d_kernel<<<ast::stream_grid((size_t)p->nc * p->nc, 256), 256, 0, s>>>(p->kout[m], (size_t)p->ncf, dst[m], (size_t)p->nc, p->nc, p->nc);
            AST_CHECK_LAUNCH();
            return AST_OK;
        }
    } crop{p, outs_in, nmul, s, embedded};
    AST_FWD(lens_rows_forward(p, kappa, s));
    if (p->split_cols) {
        AST_FWD(ast_lens_cols_forward(p->spec, n2, p->pitch, nh, (size_t)p->ncf, s));
        for (int m = 0; m < nmul; ++m) {
            AST_FWD(ast_lens_cols_inverse(p->spec, p->kspec[which[m]], p->prod, n2, p->pitch, nh, (size_t)p->ncf, s));
            AST_FWD(lens_rows_inverse(p, p->prod, outs[m], s));
        }
        return crop.run();
    }
    if (nmul == 2 && !p->prod2) AST_CHECK_HIP(hipMalloc(&p->prod2, n2 * p->pitch * sizeof(double2)));
    const void* muls[2] = {p->kspec[which[0]], p->kspec[which[nmul - 1]]};
    void* prods[2] = {p->prod, nmul == 2 ? (void*)p->prod2 : (void*)p->prod};
    AST_FWD(ast_lens_cols_convolve(p->spec, n2, p->pitch, nh, (size_t)p->ncf, muls, prods, nmul, (size_t)p->ncf, s));
    for (int m = 0; m < nmul; ++m) AST_FWD(lens_rows_inverse(p, (double2*)prods[m], outs[m], s));
    return crop.run();
}

static int lens_forward(ast_lens_plan* p, const double* kappa, hipStream_t s) {       // rocFFT 2-D route
    {
        AST_PROF("lens.zero_pad", s);
        pad_corner_kernel<<<ast::stream_grid((size_t)p->nc * p->nc, 256), 256, 0, s>>>(kappa, p->nc, p->pad_in);
    }
    AST_CHECK_LAUNCH();
    return ast_fft_exec(p->r2c, p->pad_in, p->spec, s);       // out of place: the real-to-complex transform leaves its input alone
}

static int lens_crop(ast_lens_plan* p, double2* prod, double* out, hipStream_t s) {     // prod is overwritten (rocFFT C2R)
    AST_FWD(ast_fft_exec(p->c2r, prod, p->pad, s));
    AST_PROF("lens.crop_scale", s);
    crop_scale_kernel<<<ast::stream_grid((size_t)p->nc * p->nc, 256), 256, 0, s>>>(p->pad, p->nc, p->bsz / (double)p->nc, out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

static int lens_convolve(ast_lens_plan* p, int which, double* out, hipStream_t s) {
    const size_t n2 = 2 * (size_t)p->ncf, nh = n2 / 2 + 1;
    {
        AST_PROF("lens.cmul", s);
        cmul_kernel<<<ast::stream_grid(n2 * nh, 256), 256, 0, s>>>(p->spec, p->kspec[which], p->prod, n2 * nh);
    }
    AST_CHECK_LAUNCH();
    return lens_crop(p, p->prod, out, s);
}

extern "C" int ast_kappa_to_alphas(ast_lens_plan* p, const double* kappa, double* alpha1, double* alpha2, void* stream) {
    AST_CHECK_ARG(p && kappa && alpha1 && alpha2);
    hipStream_t s = ast::as_stream(stream);
    AST_FWD(lens_kernel_spectrum(p, 0, s));
    AST_FWD(lens_kernel_spectrum(p, 1, s));
    if (p->cols) {
        const int which[2] = {0, 1};
        double* const outs[2] = {alpha1, alpha2};
        return lens_convolve_cols(p, kappa, which, outs, 2, s);
    }
    AST_FWD(lens_forward(p, kappa, s));
    const size_t n2 = 2 * (size_t)p->ncf, nh = n2 / 2 + 1;
    if (!p->prod2) AST_CHECK_HIP(hipMalloc(&p->prod2, n2 * nh * sizeof(double2)));
    {
        AST_PROF("lens.cmul", s);
        cmul2_kernel<<<ast::stream_grid(n2 * nh, 256), 256, 0, s>>>(p->spec, p->kspec[0], p->kspec[1], p->prod, p->prod2, n2 * nh);
    }
    AST_CHECK_LAUNCH();
    AST_FWD(lens_crop(p, p->prod, alpha1, s));
    AST_FWD(lens_crop(p, p->prod2, alpha2, s));
    return AST_OK;
}

extern "C" int ast_kappa_to_phi(ast_lens_plan* p, const double* kappa, double* phi, void* stream) {
    AST_CHECK_ARG(p && kappa && phi);
    hipStream_t s = ast::as_stream(stream);
    AST_FWD(lens_kernel_spectrum(p, 2, s));
    if (p->cols) {
        const int which[1] = {2};
        double* const outs[1] = {phi};
        return lens_convolve_cols(p, kappa, which, outs, 1, s);
    }
    AST_FWD(lens_forward(p, kappa, s));
    AST_FWD(lens_convolve(p, 2, phi, s));
    return AST_OK;
}

// --- libglsg.so-compatible host entry points (lensing_funcs.h:5,7) ---
namespace {
std::mutex g_host_mutex;
ast_lens_plan* g_host_plan = nullptr;

int host_lens(double* kappa0, int nc, double bsz, double* o1, double* o2, bool phi) {
    std::lock_guard<std::mutex> lock(g_host_mutex);
    if (!g_host_plan || g_host_plan->nc != nc || g_host_plan->bsz != bsz) {
        ast_lens_plan_destroy(g_host_plan);
        g_host_plan = nullptr;
        AST_FWD(ast_lens_plan_create(&g_host_plan, nc, bsz));
    }
    const size_t bytes = (size_t)nc * nc * sizeof(double);
    double *k_d = nullptr, *a_d = nullptr, *b_d = nullptr;
    hipError_t e = hipMalloc(&k_d, bytes);
    if (e == hipSuccess) e = hipMalloc(&a_d, bytes);
    if (e == hipSuccess && !phi) e = hipMalloc(&b_d, bytes);
    int rc = AST_OK;
    if (e != hipSuccess) {
        ast::set_error("kappa0_to_%s: hipMalloc -> %s", phi ? "phi" : "alphas", hipGetErrorString(e));
        rc = AST_ERR_HIP;
    }
    if (rc == AST_OK && hipMemcpy(k_d, kappa0, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = AST_ERR_HIP;
    if (rc == AST_OK) rc = phi ? ast_kappa_to_phi(g_host_plan, k_d, a_d, nullptr)
                               : ast_kappa_to_alphas(g_host_plan, k_d, a_d, b_d, nullptr);
    if (rc == AST_OK && hipDeviceSynchronize() != hipSuccess) rc = AST_ERR_HIP;
    if (rc == AST_OK && hipMemcpy(o1, a_d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = AST_ERR_HIP;
    if (rc == AST_OK && !phi && hipMemcpy(o2, b_d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = AST_ERR_HIP;
    if (k_d) (void)hipFree(k_d);
    if (a_d) (void)hipFree(a_d);
    if (b_d) (void)hipFree(b_d);
    return rc;
}
}  // namespace

extern "C" void kappa0_to_alphas(double* kappa0, int Nc, double bsz, double* alpha1, double* alpha2) {
    // void like the original; a failure is fatal rather than silent garbage
    if (host_lens(kappa0, Nc, bsz, alpha1, alpha2, false) != AST_OK) {
        fprintf(stderr, "libastrild_hip: kappa0_to_alphas failed: %s\n", ast_last_error());
        abort();
    }
}

extern "C" void kappa0_to_phi(double* kappa0, int Nc, double bsz, double* phi) {
    if (host_lens(kappa0, Nc, bsz, phi, nullptr, true) != AST_OK) {
        fprintf(stderr, "libastrild_hip: kappa0_to_phi failed: %s\n", ast_last_error());
        abort();
    }
}

// ------------------------------------------------------------- smoothing
extern "C" int ast_smooth_plan_destroy(ast_smooth_plan* p) {
    if (!p) return AST_OK;
    ast_fft_plan_destroy(p->r2c);
    ast_fft_plan_destroy(p->c2r);
    if (p->spec) (void)hipFree(p->spec);
    if (p->tmp) (void)hipFree(p->tmp);
    if (p->w_d) (void)hipFree(p->w_d);
    delete p;
    return AST_OK;
}

extern "C" int ast_smooth_plan_create(ast_smooth_plan** out, int npix) {
    AST_CHECK_ARG(out != nullptr && npix >= 2 && npix <= 32768);
    auto* p = new ast_smooth_plan();
    p->npix = npix;
    const size_t lens[2] = {(size_t)npix, (size_t)npix};
    const size_t nh = npix / 2 + 1;
    int rc = ast_fft_plan_create(&p->r2c, AST_FFT_R2C, AST_F64, 2, lens, 1, 1.0, 0);
    if (rc == AST_OK) rc = ast_fft_plan_create(&p->c2r, AST_FFT_C2R, AST_F64, 2, lens, 1, 1.0, 0);
    if (rc != AST_OK) { ast_smooth_plan_destroy(p); return rc; }
    p->w_cap = 2 * 4 * npix + 1;
    hipError_t e = hipMalloc(&p->spec, (size_t)npix * nh * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc(&p->tmp, (size_t)npix * npix * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&p->w_d, (size_t)p->w_cap * sizeof(double));
    if (e != hipSuccess) {
        ast::set_error("ast_smooth_plan_create: hipMalloc -> %s", hipGetErrorString(e));
        ast_smooth_plan_destroy(p);
        return AST_ERR_HIP;
    }
    *out = p;
    return AST_OK;
}

extern "C" int ast_gaussian_smooth(ast_smooth_plan* p, double* img, double sigma_px, int mode, void* stream) {
    AST_CHECK_ARG(p && img && sigma_px > 0.0 && mode >= 0 && mode <= 2);
    hipStream_t s = ast::as_stream(stream);
    const int npix = p->npix;
    if (mode == 0 && sigma_px >= 2.5 && (int)std::ceil(8.5 * sigma_px) <= GP_RMAX && npix >= 2 &&
        2 * (int)std::ceil(8.5 * sigma_px) + 1 <= p->w_cap && !getenv("AST_SMOOTH_FFT")) {
        // the same periodic convolution in real space (see gauss_periodic_x_kernel)
        const int radius = (int)std::ceil(8.5 * sigma_px);
        if (p->w_sigma != sigma_px || p->w_kind != 1) {
            p->w_h.assign(2 * radius + 1, 0.0);
            const double norm = 1.0 / (sigma_px * std::sqrt(2.0 * M_PI));
            for (int k = -radius; k <= radius; ++k) p->w_h[k + radius] = norm * std::exp(-0.5 * (double)(k * k) / (sigma_px * sigma_px));
            AST_CHECK_HIP(hipMemcpyAsync(p->w_d, p->w_h.data(), p->w_h.size() * sizeof(double), hipMemcpyHostToDevice, s));
            p->w_sigma = sigma_px;
            p->w_kind = 1;
        }
        AST_PROF("smooth.periodic_passes", s);
        gauss_periodic_x_kernel<<<dim3((unsigned)((npix + 1023) / 1024), (unsigned)npix), 256, 0, s>>>(img, p->tmp, npix, p->w_d, radius);
        const size_t ylds = (size_t)(64 + 2 * radius) * 32 * sizeof(double);
        static ast::PerDeviceOnce y_once;
        if (y_once.need()) {
            AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gauss_periodic_y_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)((64 + 2 * GP_RMAX) * 32 * sizeof(double))));
            y_once.mark();
        }
        gauss_periodic_y_kernel<<<dim3((unsigned)((npix + 31) / 32), (unsigned)((npix + 63) / 64)), 256, ylds, s>>>(p->tmp, img, npix, p->w_d, radius);
        AST_CHECK_LAUNCH();
        return AST_OK;
    }
    if (mode == 0) {
        const size_t nh = npix / 2 + 1;
        AST_FWD(ast_fft_exec(p->r2c, img, p->spec, s));
        {
            AST_PROF("smooth.gauss_filter", s);
            gauss_fft_filter_kernel<<<ast::stream_grid((size_t)npix * nh, 256), 256, 0, s>>>(p->spec, npix, sigma_px);
        }
        AST_CHECK_LAUNCH();
        return ast_fft_exec(p->c2r, p->spec, img, s);
    }
    // scipy.ndimage._filters._gaussian_kernel1d: radius = int(truncate*sigma + 0.5), truncate = 4
    const int radius = (int)(4.0 * sigma_px + 0.5);
    AST_CHECK_ARG(2 * radius + 1 <= p->w_cap);
    p->w_kind = 0;                // the real-space ("gaussian") weights below overwrite w_d
    p->w_h.assign(2 * radius + 1, 0.0);
    double sum = 0.0;
    const double sigma2 = sigma_px * sigma_px;
    for (int k = -radius; k <= radius; ++k) {
        p->w_h[k + radius] = std::exp(-0.5 / sigma2 * (double)(k * k));
        sum += p->w_h[k + radius];
    }
    for (auto& v : p->w_h) v /= sum;
    AST_CHECK_HIP(hipMemcpyAsync(p->w_d, p->w_h.data(), p->w_h.size() * sizeof(double), hipMemcpyHostToDevice, s));
    unsigned g = ast::stream_grid((size_t)npix * npix, 256);
    gauss_real_pass_kernel<<<g, 256, 0, s>>>(img, p->tmp, npix, 0, p->w_d, radius, mode == 2);
    gauss_real_pass_kernel<<<g, 256, 0, s>>>(p->tmp, img, npix, 1, p->w_d, radius, mode == 2);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
