// f-3 (second half): flat-sky angular power spectrum and equilateral bispectrum of a square map.
//
// Reference call sites: power_spectra/angular_power_spectrum.py:38-53
// (lenstools ConvergenceMap.powerSpectrum) and bispectra/bispectrum_2d.py:33-50
// (ConvergenceMap.bispectrum(configuration="equilateral")).  lenstools is un-vendored and unpinned;
// its published algorithm is restated (PARITY UNPINNED):
//   ft = rfft2(map), pixels (i, j), j <= N/2;  lx = min(i, N - i) * 2 pi / theta,  ly = j * 2 pi / theta;
//   a pixel falls into bin k when  edges[k] < |l| <= edges[k+1];
//   P(k) = mean over the bin's pixels of |ft|^2  *  (theta / N^2)^2      (every half-plane pixel counts once);
//   B(k) = mean over closed triangles l1 + l2 + l3 = 0 with all three sides in bin k of ft ft ft
//          * theta^4 / N^6, evaluated with the FFT estimator: ring-filter the spectrum, transform
//          back, sum the cube; the triangle count comes from the same with ft = 1.
#include "ast_common.h"

namespace {

constexpr int MAX_BINS = 1024;

// edges[k] < l <= edges[k+1]  ->  k, or -1
__device__ inline int ring_bin(double l, const double* __restrict__ edges, int nb) {
    if (!(l > edges[0]) || l > edges[nb]) return -1;
    int lo = 0, hi = nb;                 // invariant: edges[lo] < l <= edges[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (l > edges[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ inline double pixel_l(int i, int j, int n, double dl) {
    const double lx = (double)min(i, n - i) * dl, ly = (double)j * dl;
    return sqrt(lx * lx + ly * ly);
}

__global__ void __launch_bounds__(256)
flat_power_bin_kernel(const double2* __restrict__ a, const double2* __restrict__ b, int n, double dl,
                      const double* __restrict__ edges, int nb, double* __restrict__ psum,
                      unsigned long long* __restrict__ hits) {
    __shared__ double se[MAX_BINS + 1], sp[MAX_BINS];
    __shared__ unsigned int sh[MAX_BINS];
    for (int i = threadIdx.x; i <= nb; i += 256) se[i] = edges[i];
    for (int i = threadIdx.x; i < nb; i += 256) { sp[i] = 0.0; sh[i] = 0u; }
    __syncthreads();
    const int nz = n / 2 + 1;
    const size_t total = (size_t)n * nz, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int i = (int)(p / nz), j = (int)(p % nz);
        const int k = ring_bin(pixel_l(i, j, n, dl), se, nb);
        if (k < 0) continue;
        const double2 x = a[p], y = b ? b[p] : x;
        atomicAdd(&sp[k], x.x * y.x + x.y * y.y);          // Re(a conj(b))
        atomicAdd(&sh[k], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256) {
        if (sh[i]) {
            atomicAdd(&psum[i], sp[i]);
            atomicAdd(&hits[i], (unsigned long long)sh[i]);
        }
    }
}

// out = in * 1[lo < |l| <= hi]  (in = NULL: the bare indicator) on the half plane of an n x n map
__global__ void __launch_bounds__(256)
ring_filter_kernel(const double2* __restrict__ in, double2* __restrict__ out, int n, double dl, double lo, double hi) {
    const int nz = n / 2 + 1;
    const size_t total = (size_t)n * nz, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const double l = pixel_l((int)(p / nz), (int)(p % nz), n, dl);
        const bool inside = l > lo && l <= hi;
        double2 v = in ? in[p] : make_double2(1.0, 0.0);
        if (!inside) v = make_double2(0.0, 0.0);
        out[p] = v;
    }
}

}  // namespace

extern "C" int ast_flat_power_bin(const void* ft1, const void* ft2, int npix, double angle_rad, const double* edges_d,
                                  int nbins, double* psum_d, unsigned long long* hits_d, void* stream) {
    AST_CHECK_ARG(ft1 && edges_d && psum_d && hits_d && npix >= 2 && angle_rad > 0.0);
    AST_CHECK_ARG(nbins >= 1 && nbins <= MAX_BINS);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("flat_power_bin", s);
    const size_t total = (size_t)npix * (npix / 2 + 1);
    flat_power_bin_kernel<<<ast::stream_grid(total, 256), 256, 0, s>>>((const double2*)ft1, (const double2*)ft2, npix,
                                                                      2.0 * M_PI / angle_rad, edges_d, nbins, psum_d, hits_d);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_ring_filter_2d(const void* in, void* out, int npix, double angle_rad, double l_lo, double l_hi,
                                  void* stream) {
    AST_CHECK_ARG(out && npix >= 2 && angle_rad > 0.0 && l_hi > l_lo);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("ring_filter_2d", s);
    const size_t total = (size_t)npix * (npix / 2 + 1);
    ring_filter_kernel<<<ast::stream_grid(total, 256), 256, 0, s>>>((const double2*)in, (double2*)out, npix,
                                                                   2.0 * M_PI / angle_rad, l_lo, l_hi);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
