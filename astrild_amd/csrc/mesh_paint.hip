// a-1 / a-2: particle -> grid mass assignment.
//
//   ast_ngp_assign : numpy fancy-assignment semantics (last write wins),
//                    power_spectra/power_spectrum_3d.py:142-148.
//   ast_paint      : pmesh-convention NGP/CIC/TSC scatter-add with hardware
//                    float atomics straight into HBM
//                    (particles/hutils/stats_subfind.py:130-131).
//
// Index and sub-cell arithmetic is float64 for both grid dtypes: with float32
// positions the product x * (n/L) would otherwise lose ~1e-4 of a cell at
// n = 1024, which shows up at the 1e-4 level in low-k power of cold lattices.
#include "ast_common.h"
#include "paint_window.h"

namespace {

using ast::Window;

template <typename T, int W>
__global__ void __launch_bounds__(256)
paint_direct_kernel(const T* __restrict__ pos, const T* __restrict__ mass, size_t np, int n,
                    double inv_dx, double shift, double scale, int x_start, int nx_alloc, T* __restrict__ grid,
                    unsigned long long* dropped) {
    constexpr int LO = Window<W>::LO;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long ndrop = 0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        double fx, fy, fz;
        const int bx = ast::locate<W>(__fma_rn((double)pos[3 * p + 0], inv_dx, shift), n, fx);
        const int by = ast::locate<W>(__fma_rn((double)pos[3 * p + 1], inv_dx, shift), n, fy);
        const int bz = ast::locate<W>(__fma_rn((double)pos[3 * p + 2], inv_dx, shift), n, fz);
        const T m = (T)((mass ? (double)mass[p] : 1.0) * scale);
        T wx[W], wy[W], wz[W];
        Window<W>::weights(fx, wx);
        Window<W>::weights(fy, wy);
        Window<W>::weights(fz, wz);
        int jy[W], jz[W];
#pragma unroll
        for (int a = 0; a < W; ++a) {
            jy[a] = ast::wrap1(by - LO + a, n);
            jz[a] = ast::wrap1(bz - LO + a, n);
        }
#pragma unroll
        for (int a = 0; a < W; ++a) {
            int px = ast::wrap1(bx - LO + a, n) - x_start;
            if (px < 0) px += n;
            if (px >= nx_alloc) {
                ++ndrop;
                continue;
            }
            const T ma = m * wx[a];
#pragma unroll
            for (int b = 0; b < W; ++b) {
                const T mab = ma * wy[b];
                T* row = grid + ((size_t)px * n + jy[b]) * n;
#pragma unroll
                for (int c = 0; c < W; ++c) atomicAdd(row + jz[c], mab * wz[c]);
            }
        }
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

// --- NGP assign: pass 1 records the highest particle index per cell, pass 2
// lets exactly that particle write its value (== numpy's sequential result).
template <typename T>
__device__ inline bool ngp_cell(const T* x, const T* y, const T* z, size_t p, int n, size_t& cell) {
    // (npar * coord).astype(int): float64 product, truncation toward zero
    long long ix = (long long)((double)n * (double)x[p]);
    long long iy = (long long)((double)n * (double)y[p]);
    long long iz = (long long)((double)n * (double)z[p]);
    if (ix < 0 || iy < 0 || iz < 0 || ix >= n || iy >= n || iz >= n) return false;
    cell = ((size_t)ix * n + iy) * n + iz;
    return true;
}

template <typename T>
__global__ void ngp_owner_kernel(const T* x, const T* y, const T* z, size_t np, int n,
                                 uint32_t* owner, unsigned long long* dropped) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long ndrop = 0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        size_t cell;
        if (ngp_cell(x, y, z, p, n, cell)) atomicMax(&owner[cell], (uint32_t)(p + 1));
        else ++ndrop;
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

template <typename T>
__global__ void ngp_write_kernel(const T* x, const T* y, const T* z, const T* v, size_t np, int n,
                                 const uint32_t* owner, T* grid) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        size_t cell;
        if (ngp_cell(x, y, z, p, n, cell) && owner[cell] == (uint32_t)(p + 1)) grid[cell] = v[p];
    }
}

template <typename T>
int launch_paint(int window, const T* pos, const T* mass, size_t np, int n, double boxsize,
                 double scale, int x_start, int nx_alloc, T* grid, unsigned long long* dropped,
                 double shift_cells, hipStream_t s) {
    const double inv_dx = (double)n / boxsize;
    unsigned g = ast::stream_grid(np, 256);
    AST_PROF("paint_direct", s);
    switch (window) {
        case AST_WIN_NGP:
            paint_direct_kernel<T, 1><<<g, 256, 0, s>>>(pos, mass, np, n, inv_dx, shift_cells, scale, x_start, nx_alloc, grid, dropped);
            break;
        case AST_WIN_CIC:
            paint_direct_kernel<T, 2><<<g, 256, 0, s>>>(pos, mass, np, n, inv_dx, shift_cells, scale, x_start, nx_alloc, grid, dropped);
            break;
        default:
            paint_direct_kernel<T, 3><<<g, 256, 0, s>>>(pos, mass, np, n, inv_dx, shift_cells, scale, x_start, nx_alloc, grid, dropped);
            break;
    }
    AST_CHECK_LAUNCH();
    return AST_OK;
}

}  // namespace

extern "C" int ast_paint(int window, int dtype, const void* pos, const void* mass, size_t np, int nmesh,
                         double boxsize, double scale, int x_start, int nx_alloc, void* grid,
                         unsigned long long* dropped, double shift_cells, void* stream) {
    AST_CHECK_ARG(window == AST_WIN_NGP || window == AST_WIN_CIC || window == AST_WIN_TSC);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nmesh > 0 && boxsize > 0.0);
    AST_CHECK_ARG(x_start >= 0 && x_start < nmesh && nx_alloc > 0 && nx_alloc <= nmesh);
    AST_CHECK_ARG(grid != nullptr);
    if (np == 0) return AST_OK;
    AST_CHECK_ARG(pos != nullptr);
    hipStream_t s = ast::as_stream(stream);
    if (dtype == AST_F32)
        return launch_paint<float>(window, (const float*)pos, (const float*)mass, np, nmesh, boxsize, scale,
                                   x_start, nx_alloc, (float*)grid, dropped, shift_cells, s);
    return launch_paint<double>(window, (const double*)pos, (const double*)mass, np, nmesh, boxsize, scale,
                                x_start, nx_alloc, (double*)grid, dropped, shift_cells, s);
}

extern "C" int ast_ngp_assign(const void* x, const void* y, const void* z, const void* values, int dtype,
                              size_t np, int npar, void* grid, uint32_t* owner,
                              unsigned long long* dropped, void* stream) {
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(npar > 0 && grid != nullptr && owner != nullptr);
    AST_CHECK_ARG(np < 0xffffffffull);
    hipStream_t s = ast::as_stream(stream);
    const size_t ng = (size_t)npar * npar * npar;
    const size_t esz = dtype == AST_F32 ? 4 : 8;
    AST_CHECK_HIP(hipMemsetAsync(grid, 0, ng * esz, s));
    AST_CHECK_HIP(hipMemsetAsync(owner, 0, ng * sizeof(uint32_t), s));
    if (np == 0) return AST_OK;
    AST_CHECK_ARG(x && y && z && values);
    unsigned g = ast::stream_grid(np, 256);
    AST_PROF("ngp_assign", s);
    if (dtype == AST_F32) {
        ngp_owner_kernel<float><<<g, 256, 0, s>>>((const float*)x, (const float*)y, (const float*)z, np, npar, owner, dropped);
        ngp_write_kernel<float><<<g, 256, 0, s>>>((const float*)x, (const float*)y, (const float*)z, (const float*)values, np, npar, owner, (float*)grid);
    } else {
        ngp_owner_kernel<double><<<g, 256, 0, s>>>((const double*)x, (const double*)y, (const double*)z, np, npar, owner, dropped);
        ngp_write_kernel<double><<<g, 256, 0, s>>>((const double*)x, (const double*)y, (const double*)z, (const double*)values, np, npar, owner, (double*)grid);
    }
    AST_CHECK_LAUNCH();
    return AST_OK;
}
