// f-2: analytic NFW halo signals painted onto a sky map.
//
// Replaces SkyUtils.NFW_deflection_angle_map / NFW_temperature_perturbation_map /
// add_patch_to_map / analytic_Halo_signal_to_SkyArray
// (rays/skys/sky_utils.py:79-282): the reference builds each halo's stamp with numpy
// (complex arithmetic for the x > 1 branch) and adds it into the map, halo after halo,
// optionally across joblib processes.  Here one launch evaluates every in-bounds
// stamp pixel of every halo and adds it with an fp64 atomic; pixels clipped by the map
// edge are never computed.  Baxter et al. 2015 (1412.7521) Sec. 3.2, Eqs. 6-8.
#include "ast_common.h"
#include <cmath>

namespace {

constexpr double kGcm2 = 4.785e-20;           // G/c^2 [Mpc/M_sun]  (sky_utils.py:19)
constexpr double kCLight = 299792.458;        // km/s              (sky_utils.py:17)

// Eq. 7, real-valued on both branches:  x < 1: atanh;  x > 1: sqrt(1-x^2) = i sqrt(x^2-1) and
// atanh(i y) = i atan(y), so the imaginary units cancel.  x = 0 and x = 1 give nan/inf in the
// reference and are zeroed by its nan_to_num.
__device__ inline double nfw_f(double x) {
    if (!(x > 0.0) || x == 1.0) return 0.0;
    if (x < 1.0) {
        const double y = sqrt((1.0 - x) / (1.0 + x));
        return (log(0.5 * x) + 2.0 / sqrt(1.0 - x * x) * atanh(y)) / x;
    }
    const double y = sqrt((x - 1.0) / (x + 1.0));
    return (log(0.5 * x) + 2.0 / sqrt(x * x - 1.0) * atan(y)) / x;
}

struct HaloSoA {
    const double* r200_deg;
    const double* m200;
    const double* c_nfw;
    const double* dist;        // angular diameter distance [Mpc]
    const double* vel_x;       // transverse velocities [km/s] (dT only)
    const double* vel_y;
    const int* stamp_npix;     // stamp edge length [pixels]
    const int* cen_x;          // stamp centre on the map: column (theta1_pix) ...
    const int* cen_y;          // ... and row (theta2_pix)
};

__global__ void __launch_bounds__(256)
nfw_paint_kernel(HaloSoA h, double extent, int dir_mask, int suppress, double suppression_r, int signal,
                 double* __restrict__ map, int npix) {
    const int halo = blockIdx.y;
    const int S = h.stamp_npix[halo];
    const int rad = S / 2;                                  // int(len(simg) / 2), sky_utils.py:159
    const int x0 = h.cen_x[halo] - rad, y0 = h.cen_y[halo] - rad;
    const int j_lo = x0 < 0 ? -x0 : 0, j_hi = min(S, npix - x0);
    const int i_lo = y0 < 0 ? -y0 : 0, i_hi = min(S, npix - y0);
    if (j_hi <= j_lo || i_hi <= i_lo) return;
    const int w = j_hi - j_lo;
    const long long total = (long long)w * (i_hi - i_lo);

    const double theta = h.r200_deg[halo], m200 = h.m200[halo], c = h.c_nfw[halo], dist = h.dist[halo];
    const double r200 = tan(theta * M_PI / 180.0) * dist;                               // [Mpc]
    const double stop = 2.0 * r200 * extent;
    const double step = S > 1 ? stop / (double)(S - 1) : 0.0;                            // np.linspace
    const double amp = m200 * c * c / (log(1.0 + c) - c / (1.0 + c)) / 4.0 / M_PI;       // Eq. 8
    const double cst = 16.0 * M_PI * kGcm2 * amp / c / r200;                             // Eq. 6
    const double rs = r200 / c;
    const double sup_r = suppression_r * r200;
    const double vx = (signal == 1 && h.vel_x) ? h.vel_x[halo] : 0.0;
    const double vy = (signal == 1 && h.vel_y) ? h.vel_y[halo] : 0.0;

    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = i_lo + (int)(t / w), j = j_lo + (int)(t % w);
        const double ex = (j == S - 1 ? stop : (double)j * step) - r200 * extent;        // thetax = edges[j]
        const double ey = (i == S - 1 ? stop : (double)i * step) - r200 * extent;        // thetay = edges[i]
        const double rr = sqrt(ex * ex + ey * ey);
        const double f = nfw_f(rr / rs);
        double sup = 1.0;
        if (suppress) { const double q = rr / sup_r; sup = exp(-(q * q * q)); }
        double out = 0.0;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (!(dir_mask & (1 << d))) continue;
            double a = rr > 0.0 ? cst * ((d == 0 ? ex : ey) / rr) * f : 0.0;
            if (!isfinite(a)) a = 0.0;                                                   // nan_to_num
            a *= sup;
            if (fabs(a) > 100.0) a = 0.0;                                                // "remove unphysical results"
            out += signal == 1 ? -a * (d == 0 ? vx : vy) / kCLight : a;
        }
        if (out != 0.0) atomicAdd(&map[(size_t)(y0 + i) * npix + (x0 + j)], out);
    }
}

__global__ void __launch_bounds__(256)
add_patch_kernel(double* __restrict__ limg, int nl, const double* __restrict__ simg, int ns, int cx, int cy) {
    const int rad = ns / 2;
    const int x0 = cx - rad, y0 = cy - rad;
    const long long total = (long long)ns * ns;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t / ns), j = (int)(t % ns);
        const int y = y0 + i, x = x0 + j;
        if (x < 0 || y < 0 || x >= nl || y >= nl) continue;
        limg[(size_t)y * nl + x] += simg[t];            // one stamp per launch: no two threads share a pixel
    }
}

}  // namespace

extern "C" int ast_nfw_paint(const double* r200_deg, const double* m200, const double* c_nfw, const double* dist,
                             const double* vel_x, const double* vel_y, const int* stamp_npix, const int* cen_x,
                             const int* cen_y, size_t nhalo, double extent, int dir_mask, int suppress,
                             double suppression_r, int signal, double* map, int npix, void* stream) {
    AST_CHECK_ARG(map != nullptr && npix > 0);
    AST_CHECK_ARG(signal == 0 || signal == 1);
    AST_CHECK_ARG(dir_mask >= 1 && dir_mask <= 3);
    AST_CHECK_ARG(signal == 1 || dir_mask != 3);     // alpha: "Only 0 and 1 are valid direction indications"
    AST_CHECK_ARG(extent > 0.0);
    if (nhalo == 0) return AST_OK;
    AST_CHECK_ARG(r200_deg && m200 && c_nfw && dist && stamp_npix && cen_x && cen_y);
    AST_CHECK_ARG(signal == 0 || (vel_x && vel_y));
    AST_CHECK_ARG(nhalo <= 65535);
    HaloSoA h{r200_deg, m200, c_nfw, dist, vel_x, vel_y, stamp_npix, cen_x, cen_y};
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("nfw_paint", s);
    const size_t per = ((size_t)npix * npix + 255) / 256;
    dim3 grid((unsigned)(per > 256 ? 256 : per), (unsigned)nhalo);
    nfw_paint_kernel<<<grid, 256, 0, s>>>(h, extent, dir_mask, suppress, suppression_r, signal, map, npix);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_add_patch(double* limg, int nl, const double* simg, int ns, int cen_x, int cen_y, void* stream) {
    AST_CHECK_ARG(limg && simg && nl > 0 && ns > 0);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("add_patch", s);
    add_patch_kernel<<<ast::stream_grid((size_t)ns * ns, 256), 256, 0, s>>>(limg, nl, simg, ns, cen_x, cen_y);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
