// f-3: flat-sky window filters of rays/utils/filters.py that are
// pure element-wise / stencil work on the map: the DGD3 / DGD1 dipole windows
// (filters.py:305-400: a sum of Gaussians differentiated with np.gradient, times the image)
// and the Hann apodization (filters.py:150-178).  fp64 like the reference.
#include "ast_common.h"
#include <cmath>
#include <vector>

namespace {

// gaussian_field(dist, sigma) = exp(-dist^2 / (2 sigma^2)) / (2 pi sigma^2)   (filters.py:403-413)
__device__ inline double gaussian_field(double dist, double sigma) {
    return exp(-(dist * dist) / (2.0 * (sigma * sigma))) / (2.0 * M_PI * (sigma * sigma));
}

// dist[i][j] = hypot(x_j, y_i), x = linspace(1, n, n) - n/2 - 0.5; order 3: the DGD3 combination
// G(s/2) - G(s) + G(2s); order 1: G(s/2) alone (the reference's "DGD1")
__global__ void __launch_bounds__(256)
dgd_base_kernel(double* __restrict__ out, int npix, double sigma, int order) {
    const size_t total = (size_t)npix * npix;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / npix), j = (int)(idx % npix);
        const double x = (double)(j + 1) - (double)npix / 2.0 - 0.5;
        const double y = (double)(i + 1) - (double)npix / 2.0 - 0.5;
        const double dist = sqrt(x * x + y * y);
        out[idx] = order == 3 ? (gaussian_field(dist, sigma * 0.5) - gaussian_field(dist, sigma)) +
                                    gaussian_field(dist, sigma * 2.0)
                              : gaussian_field(dist, sigma * 0.5);
    }
}

// np.gradient(f, h, axis=axis, edge_order=2) for uniform spacing (numpy/lib/_function_base_impl.py):
// interior (f[i+1] - f[i-1]) / (2h); edges a f0 + b f1 + c f2 with a = -1.5/h, b = 2/h, c = -0.5/h
__global__ void __launch_bounds__(256)
gradient_kernel(const double* __restrict__ in, double* __restrict__ out, int npix, int axis, double h) {
    const size_t total = (size_t)npix * npix;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t step = axis == 0 ? (size_t)npix : 1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / npix), j = (int)(idx % npix);
        const int k = axis == 0 ? i : j;
        double v;
        if (k == 0) {
            v = (-1.5 / h) * in[idx] + (2.0 / h) * in[idx + step] + (-0.5 / h) * in[idx + 2 * step];
        } else if (k == npix - 1) {
            v = (0.5 / h) * in[idx - 2 * step] + (-2.0 / h) * in[idx - step] + (1.5 / h) * in[idx];
        } else {
            v = (in[idx + step] - in[idx - step]) / (2.0 * h);
        }
        out[idx] = v;
    }
}

// SkyUtils.convert_deflection_to_shear (sky_utils.py:342-362): np.gradient(.., h) with numpy's default edge_order = 1 -
// interior (f[i+1] - f[i-1]) / (2h), edges (f[1] - f[0]) / h and (f[n-1] - f[n-2]) / h - of both deflection components
// along both axes, then the reference's expressions term by term:
//   al11 = 1 - d0 a1, al12 = -d1 a1, al21 = -d0 a2, al22 = 1 - d1 a2, shear1 = 0.5 (al11 - al22), shear2 = 0.5 (al21 + al12)
// (no contraction: the library is built with -ffp-contract=off, so the result equals numpy's bit for bit)
__device__ inline double grad1(const double* __restrict__ f, size_t idx, size_t step, int k, int n, double h) {
    if (n == 1) return 0.0;
    if (k == 0) return (f[idx + step] - f[idx]) / h;
    if (k == n - 1) return (f[idx] - f[idx - step]) / h;
    return (f[idx + step] - f[idx - step]) / (2.0 * h);
}
__global__ void __launch_bounds__(256)
deflection_to_shear_kernel(const double* __restrict__ a1, const double* __restrict__ a2, int npix, double h,
                           double* __restrict__ g1, double* __restrict__ g2) {
    const size_t total = (size_t)npix * npix;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / npix), j = (int)(idx % npix);
        const double al11 = 1.0 - grad1(a1, idx, (size_t)npix, i, npix, h);
        const double al12 = -grad1(a1, idx, 1, j, npix, h);
        const double al21 = -grad1(a2, idx, (size_t)npix, i, npix, h);
        const double al22 = 1.0 - grad1(a2, idx, 1, j, npix, h);
        g1[idx] = 0.5 * (al11 - al22);
        g2[idx] = 0.5 * (al21 + al12);
    }
}

__global__ void __launch_bounds__(256)
multiply_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] * b[i];
}

// img * outer(hann(n), hann(n)), hann(n)[k] = 0.5 - 0.5 cos(2 pi k / (n - 1))   (scipy.signal.hann, sym=True)
__global__ void __launch_bounds__(256)
hann_kernel(const double* __restrict__ img, double* __restrict__ out, int npix) {
    const size_t total = (size_t)npix * npix;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const double f = npix > 1 ? 2.0 * M_PI / (double)(npix - 1) : 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / npix), j = (int)(idx % npix);
        const double wi = npix > 1 ? 0.5 - 0.5 * cos(f * (double)i) : 1.0;
        const double wj = npix > 1 ? 0.5 - 0.5 * cos(f * (double)j) : 1.0;
        out[idx] = img[idx] * (wi * wj);
    }
}

// scipy.ndimage boundary extension: 0 = "reflect" (d c b a | a b c d | d c b a), 1 = "nearest"
__device__ inline int extend_idx(int i, int n, int mode) {
    if (mode == 1) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    const int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - 1 - m;
}

// NI_Correlate1D (scipy/ndimage/src/ni_filters.c) along `axis`: sym = +1 symmetric, -1 antisymmetric,
// 0 general; w has 2 radius + 1 taps
__global__ void __launch_bounds__(256)
correlate1d_kernel(const double* __restrict__ in, double* __restrict__ out, int npix, int axis,
                   const double* __restrict__ w, int radius, int sym, int mode) {
    const size_t total = (size_t)npix * npix;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / npix), j = (int)(idx % npix);
        const int c = axis == 0 ? i : j;
        auto at = [&](int k) -> double {
            const int r = extend_idx(k, npix, mode);
            return axis == 0 ? in[(size_t)r * npix + j] : in[(size_t)i * npix + r];
        };
        double acc;
        if (sym > 0) {
            acc = at(c) * w[radius];
            for (int k = -radius; k < 0; ++k) acc += (at(c + k) + at(c - k)) * w[k + radius];
        } else if (sym < 0) {
            acc = at(c) * w[radius];
            for (int k = -radius; k < 0; ++k) acc += (at(c + k) - at(c - k)) * w[k + radius];
        } else {
            acc = at(c - radius) * w[0];
            for (int k = -radius + 1; k <= radius; ++k) acc += at(c + k) * w[k + radius];
        }
        out[idx] = acc;
    }
}

// scipy.ndimage.convolve(img, window) (mode "reflect", origin 0): the window is flipped and, for an
// even extent, the origin moves by -1 (scipy/ndimage/_filters.py _correlate_or_convolve), i.e.
// out[i, j] = sum_{a, b} window[kh-1-a, kw-1-b] * img[i + a - ca, j + b - cb], ca = kh/2 - (kh even).
// Taps are added in the row-major order NI_Correlate uses.  16 x 16 output tile per workgroup with
// its input footprint staged in LDS when it fits.
constexpr int CT = 16;
__global__ void __launch_bounds__(CT * CT)
convolve2d_kernel(const double* __restrict__ img, const double* __restrict__ window, double* __restrict__ out,
                  int npix, int kh, int kw, int use_lds) {
    extern __shared__ double tile[];
    const int ca = kh / 2 - ((kh & 1) ? 0 : 1);
    const int cb = kw / 2 - ((kw & 1) ? 0 : 1);
    const int ti = threadIdx.x / CT, tj = threadIdx.x % CT;
    const int i0 = blockIdx.y * CT, j0 = blockIdx.x * CT;
    const int th = CT + kh - 1, tw = CT + kw - 1;
    if (use_lds) {
        for (int t = threadIdx.x; t < th * tw; t += CT * CT) {
            const int a = t / tw, b = t % tw;
            tile[t] = img[(size_t)extend_idx(i0 + a - ca, npix, 0) * npix + extend_idx(j0 + b - cb, npix, 0)];
        }
        __syncthreads();
    }
    const int i = i0 + ti, j = j0 + tj;
    if (i >= npix || j >= npix) return;
    double acc = 0.0;
    for (int a = 0; a < kh; ++a) {
        const double* wrow = window + (size_t)(kh - 1 - a) * kw;
        if (use_lds) {
            const double* trow = tile + (size_t)(ti + a) * tw + tj;
            for (int b = 0; b < kw; ++b) acc += trow[b] * wrow[kw - 1 - b];
        } else {
            const size_t r = (size_t)extend_idx(i + a - ca, npix, 0) * npix;
            for (int b = 0; b < kw; ++b) acc += img[r + extend_idx(j + b - cb, npix, 0)] * wrow[kw - 1 - b];
        }
    }
    out[(size_t)i * npix + j] = acc;
}

// ring mean of Filters.aperture_photometry: fixed-order two-stage reduction (no atomics)
constexpr int RING_BLOCKS = 1024;
__device__ inline bool in_ring(int i, int j, int npix, double alpha_pix) {
    const double x = (double)(j + 1) - (double)npix / 2.0 - 0.5;
    const double y = (double)(i + 1) - (double)npix / 2.0 - 0.5;
    const double d = sqrt(x * x + y * y);
    return alpha_pix < d && d < alpha_pix * sqrt(2.0);
}
__global__ void __launch_bounds__(256)
ring_partial_kernel(const double* __restrict__ img, int npix, double alpha_pix, double* __restrict__ psum,
                    double* __restrict__ pcnt) {
    __shared__ double ssum[256], scnt[256];
    const size_t total = (size_t)npix * npix;
    double s = 0.0, c = 0.0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
        if (in_ring((int)(idx / npix), (int)(idx % npix), npix, alpha_pix)) { s += img[idx]; c += 1.0; }
    ssum[threadIdx.x] = s;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { ssum[threadIdx.x] += ssum[threadIdx.x + k]; scnt[threadIdx.x] += scnt[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { psum[blockIdx.x] = ssum[0]; pcnt[blockIdx.x] = scnt[0]; }
}
__global__ void __launch_bounds__(256)
ring_final_kernel(double* __restrict__ psum, double* __restrict__ pcnt, int nblocks) {
    __shared__ double ssum[256], scnt[256];
    double s = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { s += psum[b]; c += pcnt[b]; }
    ssum[threadIdx.x] = s;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { ssum[threadIdx.x] += ssum[threadIdx.x + k]; scnt[threadIdx.x] += scnt[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) psum[0] = ssum[0] / scnt[0];      // np.mean of an empty ring is nan there too
}
__global__ void __launch_bounds__(256)
subtract_scalar_kernel(const double* __restrict__ img, double* __restrict__ out, size_t n, const double* __restrict__ v) {
    const double m = v[0];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = img[i] - m;
}

// scipy.ndimage._filters._gaussian_kernel1d(sigma, order, radius)[::-1]
std::vector<double> gaussian_kernel1d(double sigma, int order, int radius) {
    const double sigma2 = sigma * sigma;
    const int n = 2 * radius + 1;
    std::vector<double> phi(n), w(n);
    double sum = 0.0;
    for (int k = -radius; k <= radius; ++k) { phi[k + radius] = std::exp(-0.5 / sigma2 * (double)(k * k)); sum += phi[k + radius]; }
    for (auto& v : phi) v /= sum;
    if (order == 0) return phi;
    // q <- (D + P) q, order times: D = diag(1..order, +1), P = diag(-1/sigma2, -1)
    std::vector<double> q(order + 1, 0.0), t(order + 1);
    q[0] = 1.0;
    for (int it = 0; it < order; ++it) {
        for (int r = 0; r <= order; ++r) {
            double v = 0.0;
            if (r + 1 <= order) v += (double)(r + 1) * q[r + 1];
            if (r >= 1) v += (1.0 / -sigma2) * q[r - 1];
            t[r] = v;
        }
        q = t;
    }
    for (int k = -radius; k <= radius; ++k) {
        double v = 0.0;
        for (int e = 0; e <= order; ++e) v += std::pow((double)k, (double)e) * q[e];
        w[k + radius] = v * phi[k + radius];
    }
    // gaussian_filter1d hands correlate1d the reversed kernel
    std::vector<double> rev(w.rbegin(), w.rend());
    return rev;
}

// scipy.ndimage.zoom(img, nout / nin, order=1, grid_mode=True) for nout <= nin: output pixel o samples the input at
// (o + 1/2) nin / nout - 1/2 (pixel centres, NI_ZoomShift with grid_mode) by bilinear interpolation; shrinking never
// reaches past the border, so no boundary mode is involved (the far neighbour of the last pixel carries weight zero).
__global__ void __launch_bounds__(256)
zoom_linear_kernel(const double* __restrict__ in, int nin, double* __restrict__ out, int nout) {
    const size_t total = (size_t)nout * nout;
    const double zoom = (double)nin / (double)nout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int oi = (int)(idx / nout), oj = (int)(idx % nout);
        const double ci = ((double)oi + 0.5) * zoom - 0.5, cj = ((double)oj + 0.5) * zoom - 0.5;
        const int i0 = (int)floor(ci), j0 = (int)floor(cj);
        const double ti = ci - (double)i0, tj = cj - (double)j0;
        const int i1 = min(i0 + 1, nin - 1), j1 = min(j0 + 1, nin - 1);
        const double wi[2] = {1.0 - ti, ti}, wj[2] = {1.0 - tj, tj};
        const double a = in[(size_t)i0 * nin + j0], b = in[(size_t)i0 * nin + j1];
        const double c = in[(size_t)i1 * nin + j0], d = in[(size_t)i1 * nin + j1];
        // the order of ni_interpolation.c's loop over the 2 x 2 support: rows outer, columns inner
        out[idx] = ((a * (wi[0] * wj[0]) + b * (wi[0] * wj[1])) + c * (wi[1] * wj[0])) + d * (wi[1] * wj[1]);
    }
}

}  // namespace

// out = img * d^order/d(axis)^order [ G(sigma/2) - G(sigma) + G(2 sigma) ]  (order 3, DGD3)  or
//       img * d/d(axis) G(sigma/2)                                         (order 1, DGD1);
// derivatives by repeated np.gradient(..., h, edge_order=2).  work_d: 2 * npix^2 doubles.
extern "C" int ast_dgd_filter(const double* img, double* out, double* work, int npix, double sigma_pix, double h,
                              int axis, int order, void* stream) {
    AST_CHECK_ARG(img && out && work && npix >= 3);
    AST_CHECK_ARG(sigma_pix > 0.0 && h > 0.0 && (axis == 0 || axis == 1) && (order == 1 || order == 3));
    hipStream_t s = ast::as_stream(stream);
    const size_t total = (size_t)npix * npix;
    const unsigned g = ast::stream_grid(total, 256);
    double* a = work;
    double* b = work + total;
    AST_PROF("dgd_filter", s);
    dgd_base_kernel<<<g, 256, 0, s>>>(a, npix, sigma_pix, order);
    for (int k = 0; k < order; ++k) {
        gradient_kernel<<<g, 256, 0, s>>>(a, b, npix, axis, h);
        double* t = a; a = b; b = t;
    }
    multiply_kernel<<<g, 256, 0, s>>>(a, img, out, total);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// The resampling half of skimage.transform.resize(img, (nout, nout), anti_aliasing=True) as SkyArray.resize uses it
// (sky_array.py:475-496; the Gaussian prefilter half is ast_gaussian_smooth mode 1 with sigma = (nin / nout - 1) / 2).
extern "C" int ast_deflection_to_shear(const double* alpha1, const double* alpha2, int npix, double h, double* gamma1,
                                       double* gamma2, void* stream) {
    AST_CHECK_ARG(alpha1 && alpha2 && gamma1 && gamma2 && npix >= 2 && h > 0.0);
    AST_CHECK_ARG(gamma1 != alpha1 && gamma1 != alpha2 && gamma2 != alpha1 && gamma2 != alpha2 && gamma1 != gamma2);
    deflection_to_shear_kernel<<<ast::stream_grid((size_t)npix * npix, 256), 256, 0, ast::as_stream(stream)>>>(alpha1, alpha2, npix, h,
                                                                                                           gamma1, gamma2);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_zoom_linear(const double* img, int nin, double* out, int nout, void* stream) {
    AST_CHECK_ARG(img && out && img != out && nout >= 1 && nout <= nin);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("zoom_linear", s);
    zoom_linear_kernel<<<ast::stream_grid((size_t)nout * nout, 256), 256, 0, s>>>(img, nin, out, nout);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_hann_apodize(const double* img, double* out, int npix, void* stream) {
    AST_CHECK_ARG(img && out && npix >= 1);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("hann_apodize", s);
    hann_kernel<<<ast::stream_grid((size_t)npix * npix, 256), 256, 0, s>>>(img, out, npix);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// scipy.ndimage.gaussian_filter(img, sigma, order=(order0, order1), mode=..., truncate=4): one
// correlate1d per axis (axis 0 first), kernel from _gaussian_kernel1d.  Used by
// Filters.gaussian_third_derivative_convolution (filters.py:260-304; mode "nearest").
// work_d: npix^2 + 2 (2 radius + 1) doubles, radius = int(4 sigma + 0.5).
extern "C" int ast_gaussian_filter_order(const double* img, double* out, double* work, size_t work_doubles, int npix,
                                         double sigma, int order0, int order1, int mode, void* stream) {
    AST_CHECK_ARG(img && out && work && npix >= 1 && sigma > 0.0);
    AST_CHECK_ARG(order0 >= 0 && order0 <= 8 && order1 >= 0 && order1 <= 8 && (mode == 0 || mode == 1));
    hipStream_t s = ast::as_stream(stream);
    const int radius = (int)(4.0 * sigma + 0.5);
    const size_t total = (size_t)npix * npix;
    const size_t taps = (size_t)2 * radius + 1;
    AST_CHECK_ARG(work_doubles >= total + 2 * taps);
    double* tmp = work;
    double* w_d = work + total;
    const int orders[2] = {order0, order1};
    int sym[2];
    std::vector<double> both;
    for (int a = 0; a < 2; ++a) {
        std::vector<double> w = gaussian_kernel1d(sigma, orders[a], radius);
        // the symmetry test of NI_Correlate1D (tolerance DBL_EPSILON)
        bool is_sym = true, is_anti = true;
        for (int k = 1; k <= radius; ++k) {
            if (std::fabs(w[radius + k] - w[radius - k]) > 2.220446049250313e-16) is_sym = false;
            if (std::fabs(w[radius + k] + w[radius - k]) > 2.220446049250313e-16) is_anti = false;
        }
        sym[a] = is_sym ? 1 : (is_anti ? -1 : 0);
        both.insert(both.end(), w.begin(), w.end());
    }
    AST_CHECK_HIP(hipMemcpyAsync(w_d, both.data(), both.size() * sizeof(double), hipMemcpyHostToDevice, s));
    AST_CHECK_HIP(hipStreamSynchronize(s));      // `both` is a host temporary
    const unsigned g = ast::stream_grid(total, 256);
    AST_PROF("gaussian_filter_order", s);
    correlate1d_kernel<<<g, 256, 0, s>>>(img, tmp, npix, 0, w_d, radius, sym[0], mode);
    correlate1d_kernel<<<g, 256, 0, s>>>(tmp, out, npix, 1, w_d + taps, radius, sym[1], mode);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// scipy.ndimage.convolve(img, window) with the defaults (mode "reflect", origin 0); used by
// Filters.gaussian_compensated (filters.py:415-459).  window_d: kh x kw doubles on the device.
extern "C" int ast_convolve2d(const double* img, const double* window, double* out, int npix, int kh, int kw,
                              void* stream) {
    AST_CHECK_ARG(img && window && out && img != out && npix >= 1 && kh >= 1 && kw >= 1);
    hipStream_t s = ast::as_stream(stream);
    const size_t lds = (size_t)(CT + kh - 1) * (CT + kw - 1) * sizeof(double);
    const int use_lds = lds <= 64 * 1024;
    dim3 grid((npix + CT - 1) / CT, (npix + CT - 1) / CT);
    AST_PROF("convolve2d", s);
    convolve2d_kernel<<<grid, CT * CT, use_lds ? lds : 0, s>>>(img, window, out, npix, kh, kw, use_lds);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// Filters.aperture_photometry (filters.py:40-73): out = img - mean(img[ring]), ring =
// alpha_pix < pixel distance to the map centre < alpha_pix sqrt(2).  work_d: 2 * 1024 doubles.
extern "C" int ast_aperture_photometry(const double* img, double* out, double* work, int npix, double alpha_pix,
                                       void* stream) {
    AST_CHECK_ARG(img && out && work && npix >= 1 && alpha_pix >= 0.0);
    hipStream_t s = ast::as_stream(stream);
    const size_t total = (size_t)npix * npix;
    AST_PROF("aperture_photometry", s);
    ring_partial_kernel<<<RING_BLOCKS, 256, 0, s>>>(img, npix, alpha_pix, work, work + RING_BLOCKS);
    ring_final_kernel<<<1, 256, 0, s>>>(work, work + RING_BLOCKS, RING_BLOCKS);
    subtract_scalar_kernel<<<ast::stream_grid(total, 256), 256, 0, s>>>(img, out, total, work);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
