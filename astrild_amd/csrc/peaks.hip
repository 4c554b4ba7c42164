// a-8: SkyArray.wl_peak_counts (rays/skys/sky_array.py:435-472) - local maxima of a map and the
// order statistics behind its np.percentile bounds.
//
// lenstools' ConvergenceMap.locatePeaks (un-vendored; restated): a pixel of the INTERIOR of the map
// (the outermost rows / columns are never peaks) is a peak when its value is strictly larger than
// all 8 neighbours and lies in [thresholds[0], thresholds[-1]).  Peak heights and flat pixel indices
// are appended to caller-supplied arrays (order of appearance is not defined: the host sorts by index,
// which is lenstools' row-major scan order).
#include "ast_common.h"
#include <cstring>

namespace {

__device__ inline unsigned long long d2key(double d) {       // order-preserving double -> uint64
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
inline double key2d_host(unsigned long long k) {
    unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}

template <typename T>
__global__ void __launch_bounds__(256)
peak_find_kernel(const T* __restrict__ img, int npix, double lo, double hi, size_t cap, T* __restrict__ values,
                 long long* __restrict__ index, unsigned long long* __restrict__ count) {
    // 16 x 16 pixel tile per workgroup, the 18 x 18 footprint staged in LDS
    __shared__ T tile[18][19];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 18 * 18; i += 256) {
        const int ly = i / 18, lx = i % 18;
        const int gy = min(max(y0 + ly - 1, 0), npix - 1), gx = min(max(x0 + lx - 1, 0), npix - 1);
        tile[ly][lx] = img[(size_t)gy * npix + gx];
    }
    __syncthreads();
    const int x = x0 + tx, y = y0 + ty;
    if (x < 1 || y < 1 || x >= npix - 1 || y >= npix - 1) return;
    const T c = tile[ty + 1][tx + 1];
    bool peak = true;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            if (dy != 1 || dx != 1) peak = peak && c > tile[ty + dy][tx + dx];
    if (!peak || !((double)c >= lo && (double)c < hi)) return;
    const unsigned long long k = atomicAdd(count, 1ull);
    if (k < cap) {
        values[k] = c;
        index[k] = (long long)y * npix + x;
    }
}

// histogram of one digit (bits shift .. shift + width - 1, width <= 11) of the sortable keys of the elements
// whose higher bits equal `prefix`
template <typename T>
__global__ void __launch_bounds__(256)
key_digit_hist_kernel(const T* __restrict__ buf, size_t n, unsigned long long prefix, int shift, int width, int top,
                      unsigned long long* __restrict__ hist) {
    __shared__ unsigned int lh[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long k = d2key((double)buf[i]);
        if (top || (k >> (shift + width)) == prefix) atomicAdd(&lh[(k >> shift) & ((1ull << width) - 1ull)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

}  // namespace

extern "C" int ast_peak_find(const void* img, int dtype, int npix, double lo, double hi, size_t cap, void* values,
                             long long* index, unsigned long long* count, void* stream) {
    AST_CHECK_ARG(img && values && index && count && npix >= 3);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    hipStream_t s = ast::as_stream(stream);
    AST_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(unsigned long long), s));
    const dim3 grid((npix + 15) / 16, (npix + 15) / 16);
    AST_PROF("peak_find", s);
    if (dtype == AST_F32)
        peak_find_kernel<float><<<grid, 256, 0, s>>>((const float*)img, npix, lo, hi, cap, (float*)values, index, count);
    else
        peak_find_kernel<double><<<grid, 256, 0, s>>>((const double*)img, npix, lo, hi, cap, (double*)values, index, count);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// out_host[j] = the ks_host[j]-th smallest element (0-based) of buf_d, exactly, by a most-significant-digit radix
// select on order-preserving 64-bit keys: six passes of an 11-bit digit histogram per k (2048 bins; the last digit
// has 9 bits).  Synchronous: the histograms come back to the host between passes.  scratch_d: 2048 uint64.
extern "C" int ast_order_statistics(const void* buf, int dtype, size_t count, const size_t* ks_host, int nk,
                                    double* out_host, unsigned long long* scratch, void* stream) {
    AST_CHECK_ARG(buf && ks_host && out_host && scratch && count > 0 && nk > 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    hipStream_t s = ast::as_stream(stream);
    const unsigned g = ast::stream_grid(count, 256);
    static unsigned long long hist[2048];
    for (int j = 0; j < nk; ++j) {
        AST_CHECK_ARG(ks_host[j] < count);
        size_t k = ks_host[j];
        unsigned long long prefix = 0;
        // digits from the top: bits 63..53, 52..42, 41..31, 30..20, 19..9 (11 bits each), 8..0 (9 bits)
        const int shifts[6] = {53, 42, 31, 20, 9, 0}, widths[6] = {11, 11, 11, 11, 11, 9};
        for (int pass = 0; pass < 6; ++pass) {
            AST_CHECK_HIP(hipMemsetAsync(scratch, 0, sizeof(hist), s));
            if (dtype == AST_F32)
                key_digit_hist_kernel<float><<<g, 256, 0, s>>>((const float*)buf, count, prefix, shifts[pass], widths[pass],
                                                               pass == 0, scratch);
            else
                key_digit_hist_kernel<double><<<g, 256, 0, s>>>((const double*)buf, count, prefix, shifts[pass], widths[pass],
                                                                pass == 0, scratch);
            AST_CHECK_LAUNCH();
            AST_CHECK_HIP(hipMemcpyAsync(hist, scratch, sizeof(hist), hipMemcpyDeviceToHost, s));
            AST_CHECK_HIP(hipStreamSynchronize(s));
            int bin = 0;
            for (; bin < 2048; ++bin) {
                if (k < hist[bin]) break;
                k -= hist[bin];
            }
            AST_CHECK_ARG(bin < 2048);
            prefix = (prefix << widths[pass]) | (unsigned long long)bin;
        }
        out_host[j] = key2d_host(prefix);
    }
    return AST_OK;
}
