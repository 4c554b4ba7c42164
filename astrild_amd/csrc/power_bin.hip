// a-5: FFTPower(mode="1d") k-shell binning of the half spectrum, plus the
// shell filter / triple product of the FFT bispectrum estimator (a-10) and the
// slab pack/unpack of the distributed transpose.
//
// Binning layout: one wave walks one (i0, i1) row of the spectrum with its 64
// lanes along the contiguous half axis, so loads are coalesced; lanes add into
// the workgroup's LDS shell table, which is flushed to HBM once per workgroup.
// The data-independent sums (sum w|k|, sum w) come from a separate geometry
// kernel so the data pass carries one LDS atomic per mode instead of three.
#include "ast_common.h"

namespace {

__device__ inline int freq(int i, int n) { return i > n / 2 ? i - n : i; }

__device__ inline int isqrt_i(long long v) {
    // |m|^2 <= 3 * 4096^2 < 2^26: the float sqrt is within 1 of the integer root
    int r = (int)__fsqrt_rn((float)v);
    if ((long long)r * r > v) --r;
    if ((long long)(r + 1) * (r + 1) <= v) ++r;
    return r;
}

template <typename C> struct cplx_traits;
template <> struct cplx_traits<float2> { using real = float; };
template <> struct cplx_traits<double2> { using real = double; };

constexpr int MAX_SHELLS = 4096;  // nmesh <= 8192

// Data pass: psum only.  One wave per (i0, i1) row, lanes along the contiguous
// half axis (coalesced 8/16-byte loads); every lane adds its weighted power
// straight into the workgroup's LDS shell table (ds_add_f64), which is flushed
// to HBM once per workgroup.  Along a row the shell index moves by at most one
// per mode, so same-address conflicts inside a wave stay short.
template <typename C>
__global__ void __launch_bounds__(256)
power_bin_kernel(const C* __restrict__ s1, const C* __restrict__ s2, int n, double pnorm, double kf_rule,
                 int i0_start, int i0_count, int i1_start, int i1_count, double* psum) {
    extern __shared__ double lp[];
    const int nb = n / 2 - 1;
    for (int i = threadIdx.x; i < nb + 2; i += blockDim.x) lp[i] = 0.0;
    __syncthreads();

    const int nz = n / 2 + 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const long long nrows = (long long)i0_count * i1_count;
    for (long long row = (long long)blockIdx.x * waves_per_block + wave; row < nrows;
         row += (long long)gridDim.x * waves_per_block) {
        const int a = (int)(row / i1_count), b = (int)(row % i1_count);
        const int m0 = freq(i0_start + a, n), m1 = freq(i1_start + b, n);
        const long long base = (long long)m0 * m0 + (long long)m1 * m1;
        const C* r1 = s1 + (size_t)row * nz;
        const C* r2 = s2 ? s2 + (size_t)row * nz : nullptr;
        for (int iz = lane; iz < nz; iz += 64) {
            const long long m2 = base + (long long)iz * iz;
            int sh = isqrt_i(m2);                                // shell = sh - 1; sh == 0 is the DC mode
            if (kf_rule != 0.0 && (long long)sh * sh == m2 && sh > 0) sh = ast::float64_edge_norm(sh, m0, m1, iz, kf_rule);
            if (sh < 1 || sh > nb) continue;
            const C x = r1[iz];
            const C y = r2 ? r2[iz] : x;
            const double w = (iz > 0 && iz < n / 2) ? 2.0 : 1.0;
            atomicAdd(&lp[sh], w * ((double)x.x * (double)y.x + (double)x.y * (double)y.y));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        const double v = lp[i + 1];
        if (v != 0.0) atomicAdd(&psum[i], v * pnorm);
    }
}

// Geometry pass: sum w |k| and sum w per shell depend only on the lattice block,
// not on the data; callers cache them per (nmesh, L, block).
__global__ void __launch_bounds__(256)
shell_geometry_kernel(int n, double kf, double kf_rule, int i0_start, int i0_count, int i1_start, int i1_count,
                      double* ksum, unsigned long long* nmodes) {
    extern __shared__ double lk[];
    const int nb = n / 2 - 1;
    unsigned long long* lm = reinterpret_cast<unsigned long long*>(lk + nb + 2);
    for (int i = threadIdx.x; i < nb + 2; i += blockDim.x) { lk[i] = 0.0; lm[i] = 0ull; }
    __syncthreads();
    const int nz = n / 2 + 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const long long nrows = (long long)i0_count * i1_count;
    for (long long row = (long long)blockIdx.x * waves_per_block + wave; row < nrows;
         row += (long long)gridDim.x * waves_per_block) {
        const int a = (int)(row / i1_count), b = (int)(row % i1_count);
        const int m0 = freq(i0_start + a, n), m1 = freq(i1_start + b, n);
        const long long base = (long long)m0 * m0 + (long long)m1 * m1;
        for (int iz = lane; iz < nz; iz += 64) {
            const long long m2 = base + (long long)iz * iz;
            int sh = isqrt_i(m2);
            if (kf_rule != 0.0 && (long long)sh * sh == m2 && sh > 0) sh = ast::float64_edge_norm(sh, m0, m1, iz, kf_rule);
            if (sh < 1 || sh > nb) continue;
            const unsigned long long w = (iz > 0 && iz < n / 2) ? 2ull : 1ull;
            atomicAdd(&lk[sh], (double)w * sqrt((double)m2));
            atomicAdd(&lm[sh], w);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        const unsigned long long m = lm[i + 1];
        if (m) {
            atomicAdd(&ksum[i], lk[i + 1] * kf);
            atomicAdd(&nmodes[i], m);
        }
    }
}

template <typename C>
__global__ void __launch_bounds__(256)
shell_filter_kernel(const C* __restrict__ in, C* __restrict__ out, int n, long long lo2, long long hi2,
                    int i0_start, int i0_count, int i1_start, int i1_count) {
    const int nz = n / 2 + 1;
    const size_t total = (size_t)i0_count * i1_count * nz;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int iz = (int)(i % nz);
        const size_t row = i / nz;
        const int m0 = freq(i0_start + (int)(row / i1_count), n), m1 = freq(i1_start + (int)(row % i1_count), n);
        const long long m2 = (long long)m0 * m0 + (long long)m1 * m1 + (long long)iz * iz;
        const bool inside = m2 >= lo2 && m2 < hi2;
        C v;
        if (in) v = in[i]; else { v.x = 1; v.y = 0; }
        if (!inside) { v.x = 0; v.y = 0; }
        out[i] = v;
    }
}


// The shell indicator on the FULL lattice as a real array, and the real part of a half spectrum unfolded onto the full
// lattice: together with a forward float64 transform they give the bispectrum estimator's I_s(x) = sum_{k in s} e^{ikx}
// without an inverse transform - the indicator is real and even (k in s <=> -k in s), so its inverse transform equals
// its forward one and is real and even itself: I_s(x) = Re FFT[1_s](x), and for x_z > N/2 the value at -x.
__global__ void __launch_bounds__(256)
shell_mask_real_kernel(double* __restrict__ out, int n, long long lo2, long long hi2) {
    const size_t total = (size_t)n * n * n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int m2i = freq((int)(i % n), n), m1 = freq((int)((i / n) % n), n), m0 = freq((int)(i / ((size_t)n * n)), n);
        const long long m2 = (long long)m0 * m0 + (long long)m1 * m1 + (long long)m2i * m2i;
        out[i] = (m2 >= lo2 && m2 < hi2) ? 1.0 : 0.0;
    }
}
__global__ void __launch_bounds__(256)
half_real_to_full_kernel(const double2* __restrict__ spec, double* __restrict__ out, int n) {
    const int nz = n / 2 + 1;
    const size_t total = (size_t)n * n * n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int z = (int)(i % n), y = (int)((i / n) % n), x = (int)(i / ((size_t)n * n));
        if (z >= nz) { z = n - z; y = y ? n - y : 0; x = x ? n - x : 0; }
        out[i] = spec[((size_t)x * n + y) * nz + z].x;
    }
}

// f-1: interlacing + window compensation of a catalogue-painted mesh (nbodykit CatalogMesh with interlaced=True /
// compensated=True, the parameters astrild writes at power_spectrum_3d.py:197-212; nbodykit is un-vendored, its
// published formulas restated).  c1 is the spectrum of the plain paint, c2 of the paint shifted by half a cell
// (ast_paint shift_cells = 0.5); w_i = 2 pi m_i / N is the circular frequency of axis i:
//   interlaced:   c1 <- (c1 + c2 * exp(i (wx + wy + wz) / 2)) / 2        (Sefusatti et al. 2016: odd images cancel)
//   compensated:  c1 <- c1 / prod_i W(w_i),
//     interlaced:      W = sinc(w/2)^p                     (CompensateCIC p = 2, CompensateTSC p = 3)
//     not interlaced:  W = sqrt(1 - 2/3 s) (CIC),  sqrt(1 - s + 2/15 s^2) (TSC),  s = sin^2(w/2)
//                                                          (Compensate*Shotnoise: aliased shot noise, Jing 2005)
// All factors in double for both dtypes.
template <typename C>
__global__ void __launch_bounds__(256)
interlace_compensate_kernel(C* __restrict__ c1, const C* __restrict__ c2, int n, int window, int compensate,
                            int i0_start, int i0_count, int i1_start, int i1_count) {
    using R = typename cplx_traits<C>::real;
    const int nz = n / 2 + 1;
    const size_t total = (size_t)i0_count * i1_count * nz;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const double dw = 2.0 * M_PI / (double)n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int iz = (int)(i % nz);
        const size_t row = i / nz;
        const double w[3] = {dw * freq(i0_start + (int)(row / i1_count), n), dw * freq(i1_start + (int)(row % i1_count), n),
                             dw * iz};
        double re = (double)c1[i].x, im = (double)c1[i].y;
        if (c2) {
            double sn, cs;
            sincos(0.5 * (w[0] + w[1] + w[2]), &sn, &cs);
            const double re2 = (double)c2[i].x, im2 = (double)c2[i].y;
            re = 0.5 * re + 0.5 * (re2 * cs - im2 * sn);
            im = 0.5 * im + 0.5 * (re2 * sn + im2 * cs);
        }
        if (compensate) {
            double f = 1.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double h = 0.5 * w[a], sh = sin(h);
                if (c2) {
                    const double sc = h == 0.0 ? 1.0 : sh / h;
                    f *= window == AST_WIN_TSC ? sc * sc * sc : sc * sc;
                } else {
                    const double s2 = sh * sh;
                    f *= window == AST_WIN_TSC ? sqrt(1.0 - s2 + 2.0 / 15.0 * s2 * s2) : sqrt(1.0 - 2.0 / 3.0 * s2);
                }
            }
            re /= f;
            im /= f;
        }
        C o;
        o.x = (R)re;
        o.y = (R)im;
        c1[i] = o;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
triple_sum_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, size_t n, double* out) {
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc += (double)a[i] * (double)b[i] * (double)c[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

// All triangle sums of a set of shell fields in ONE pass over the fields: a workgroup stages a chunk of TRI_CHUNK
// cells of every field in LDS, then thread (t, part) adds the products of triangle t over its part of the chunk
// (lanes of a wave = different triangles, same cell: the LDS row pitch is odd, distinct fields hit distinct banks,
// equal fields broadcast).  Every field is read from HBM once instead of once per triangle it appears in (75
// triangles over 31 shells at 512^3: 16.6 GB instead of 121 GB).  Products and sums in double, fixed order:
// partial[block][thread], then triple_reduce_kernel adds blocks and parts in index order - deterministic.
#ifndef TRI_ABLATE
#define TRI_ABLATE 0          // perf experiments: 1 = no products (loads + LDS staging only), 2 = no global loads
#endif
#ifndef TRI_F32_PRODUCTS
#define TRI_F32_PRODUCTS 1
#endif
constexpr int TRI_CHUNK = 256;
constexpr int TRI_THREADS = 256;
constexpr int TRI_BLOCKS = 2048;          // persistent workgroups (8 per CU), grid-stride over the chunks
// PIPE (nfields <= TRI_PIPE_FIELDS): the NEXT chunk's values are loaded into registers before the current chunk's products
// are formed, so the global loads run beside the LDS-bound product loop instead of before it (measured at 512^3, 31 fields,
// 75 triangles: loads + staging alone 2.9 ms, products alone 3.1 ms, one after the other per workgroup 4.5 ms).
constexpr int TRI_PIPE_FIELDS = 32;
template <typename T, bool PIPE>
__global__ void __launch_bounds__(TRI_THREADS)
triple_sums_kernel(const T* const* __restrict__ fields, int nfields, const int* __restrict__ tri, int ntri, int parts,
                   size_t n, double* __restrict__ partial) {
    extern __shared__ unsigned char tri_lds[];
    T* v = reinterpret_cast<T*>(tri_lds);                       // [nfields][TRI_CHUNK + 1]
    constexpr int P = TRI_CHUNK + 1;
    const int t = threadIdx.x % ntri, part = threadIdx.x / ntri;
    const bool worker = part < parts;
    int ia = 0, ib = 0, ic = 0;
    if (worker) {
        ia = tri[3 * t] * P; ib = tri[3 * t + 1] * P; ic = tri[3 * t + 2] * P;
        // a repeated field first (the product does not care about the order): (a, b, b) -> (b, b, a), (a, b, a) -> (a, a, b)
        if (ib == ic) { const int a = ia; ia = ib; ic = a; }
        else if (ia == ic) { const int b = ib; ib = ia; ic = b; }
    }
    // every triangle of this wave repeats a field (equilateral, isosceles and squeezed bins all do): two LDS reads per term
    // instead of three
    const bool wave_pairs = __all((int)(!worker || ia == ib)) != 0;
    const int per = (TRI_CHUNK + parts - 1) / parts;
    const int c_lo = part * per, c_hi = min(c_lo + per, TRI_CHUNK);
    double acc = 0.0;
    const size_t nchunks = (n + TRI_CHUNK - 1) / TRI_CHUNK;
    typedef const __attribute__((address_space(1))) T* gptr_t;   // the pointers come from memory: say that they are global, or the loads are flat
    auto products = [&]() {
        if (worker && !(TRI_ABLATE & 1)) {            // two-level sum: a chunk's terms first (keeps the round-off of the
            double sub = 0.0;                         // long running sum at sqrt(chunks), not sqrt(cells))
            if (wave_pairs && sizeof(T) == 4 && TRI_F32_PRODUCTS) {           // (uniform over the wave)
#pragma unroll 4
                for (int c = c_lo; c < c_hi; ++c) {
                    const float x = (float)v[ia + c];
                    sub += (double)(x * x * (float)v[ic + c]);
                }
            } else
#pragma unroll 4
            for (int c = c_lo; c < c_hi; ++c) {
#if TRI_F32_PRODUCTS
                // fp32 fields: the two products in the fields' own precision (each rounds at 6e-8, unbiased - the fields carry
                // the fp32 transform's 1e-7 already), ONE conversion and the running sum in double: 4 vector instructions
                // per term instead of 6 (the product loop is bound by vector issue and LDS reads in equal parts)
                if (sizeof(T) == 4) { sub += (double)((float)v[ia + c] * (float)v[ib + c] * (float)v[ic + c]); continue; }
#endif
                sub += (double)v[ia + c] * (double)v[ib + c] * (double)v[ic + c];
            }
            acc += sub;
        }
    };
    if (PIPE) {
        T reg[TRI_PIPE_FIELDS];
        auto fetch = [&](size_t ch) {                 // unconditional loads: cells past the end re-read the last one and are zeroed
            const size_t cell = ch * TRI_CHUNK + threadIdx.x;
            const size_t lc = cell < n ? cell : n - 1;
            const T keep = cell < n ? (T)1 : (T)0;
#pragma unroll
            for (int f = 0; f < TRI_PIPE_FIELDS; ++f)
                reg[f] = ((TRI_ABLATE & 2) ? (T)(lc & 7) : ((gptr_t)fields[min(f, nfields - 1)])[lc]) * keep;
        };
        if ((size_t)blockIdx.x < nchunks) fetch(blockIdx.x);
        for (size_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
#pragma unroll
            for (int f = 0; f < TRI_PIPE_FIELDS; ++f)
                if (f < nfields) v[f * P + threadIdx.x] = reg[f];
            __syncthreads();
            if (ch + gridDim.x < nchunks) fetch(ch + gridDim.x);      // in flight across the product loop
            products();
            __syncthreads();
        }
    } else {
        for (size_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
            // unconditional loads in groups of 8 (a predicated load is a branch plus a full wait each: 31 serial memory
            // latencies per chunk); cells past the end re-read the last one and are zeroed
            const size_t cell = ch * TRI_CHUNK + threadIdx.x;
            const size_t lc = cell < n ? cell : n - 1;
            const T keep = cell < n ? (T)1 : (T)0;
            for (int f0 = 0; f0 < nfields; f0 += 8) {
                T tmp[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    tmp[j] = (TRI_ABLATE & 2) ? (T)(lc & 7) : ((gptr_t)fields[min(f0 + j, nfields - 1)])[lc];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (f0 + j < nfields) v[(f0 + j) * P + threadIdx.x] = tmp[j] * keep;
            }
            __syncthreads();
            products();
            __syncthreads();
        }
    }
    partial[(size_t)blockIdx.x * TRI_THREADS + threadIdx.x] = worker ? acc : 0.0;
}

__global__ void __launch_bounds__(256)
triple_reduce_kernel(const double* __restrict__ partial, int nblocks, int ntri, int parts, double* __restrict__ out) {
    const int t = blockIdx.x;                                   // one workgroup per triangle
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks * parts; i += 256) {
        const int blk = i / parts, part = i % parts;
        acc += partial[(size_t)blk * TRI_THREADS + part * ntri + t];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[t] = (w[0] + w[1]) + (w[2] + w[3]);
}

// (n0, n1, n2) -> parts x (n0, n1/parts, n2)
template <typename C, bool UNPACK>
__global__ void __launch_bounds__(256)
slab_pack_kernel(const C* __restrict__ in, C* __restrict__ out, size_t n0, size_t n1, size_t n2, int parts) {
    const size_t c1 = n1 / parts;
    const size_t total = n0 * n1 * n2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t k = i % n2;
        const size_t j = (i / n2) % n1;
        const size_t a = i / (n2 * n1);
        const size_t part = j / c1, jl = j % c1;
        const size_t packed = ((part * n0 + a) * c1 + jl) * n2 + k;
        if (UNPACK) out[i] = in[packed]; else out[packed] = in[i];
    }
}

}  // namespace

extern "C" int ast_power_bin_1d(const void* spec1, const void* spec2, int dtype, int nmesh, double boxsize,
                                int i0_start, int i0_count, int i1_start, int i1_count,
                                double* ksum, double* psum, long long* nmodes, int binning, void* stream) {
    AST_CHECK_ARG((spec1 && psum) || (!spec1 && !spec2 && !psum));
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    AST_CHECK_ARG((ksum == nullptr) == (nmodes == nullptr));
    AST_CHECK_ARG(psum || ksum);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nmesh >= 4 && nmesh % 2 == 0 && nmesh / 2 - 1 <= MAX_SHELLS && boxsize > 0.0);
    AST_CHECK_ARG(i0_start >= 0 && i0_count >= 0 && i0_start + i0_count <= nmesh);
    AST_CHECK_ARG(i1_start >= 0 && i1_count >= 0 && i1_start + i1_count <= nmesh);
    const long long nrows = (long long)i0_count * i1_count;
    if (nrows == 0) return AST_OK;
    const int nb = nmesh / 2 - 1;
    const double kf = 2.0 * M_PI / boxsize;
    const double kf_rule = binning == AST_BIN_FLOAT64 ? kf : 0.0;
    const double pnorm = boxsize * boxsize * boxsize;
    const long long need = (nrows + 3) / 4;
    const unsigned g = (unsigned)(need > 4096 ? 4096 : need);
    hipStream_t s = ast::as_stream(stream);
    if (psum) {
        AST_PROF("power_bin", s);
        const size_t lds = (size_t)(nb + 2) * sizeof(double);
        if (dtype == AST_F32)
            power_bin_kernel<float2><<<g, 256, lds, s>>>((const float2*)spec1, (const float2*)spec2, nmesh, pnorm, kf_rule,
                                                          i0_start, i0_count, i1_start, i1_count, psum);
        else
            power_bin_kernel<double2><<<g, 256, lds, s>>>((const double2*)spec1, (const double2*)spec2, nmesh, pnorm, kf_rule,
                                                           i0_start, i0_count, i1_start, i1_count, psum);
    }
    if (ksum) {
        AST_PROF("shell_geometry", s);
        const size_t lds = (size_t)(nb + 2) * (sizeof(double) + sizeof(unsigned long long));
        shell_geometry_kernel<<<g, 256, lds, s>>>(nmesh, kf, kf_rule, i0_start, i0_count, i1_start, i1_count, ksum,
                                                   reinterpret_cast<unsigned long long*>(nmodes));
    }
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_shell_filter(const void* in, void* out, int dtype, int nmesh, int m_lo, int m_hi, int i0_start,
                                int i0_count, int i1_start, int i1_count, void* stream) {
    AST_CHECK_ARG(out != nullptr);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nmesh >= 4 && nmesh % 2 == 0 && m_lo >= 0 && m_hi > m_lo);
    const long long lo2 = (long long)m_lo * m_lo, hi2 = (long long)m_hi * m_hi;
    AST_CHECK_ARG(i0_start >= 0 && i0_count >= 0 && i0_start + i0_count <= nmesh);
    AST_CHECK_ARG(i1_start >= 0 && i1_count >= 0 && i1_start + i1_count <= nmesh);
    const size_t total = (size_t)i0_count * i1_count * (nmesh / 2 + 1);
    if (total == 0) return AST_OK;
    unsigned g = ast::stream_grid(total, 256);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("shell_filter", s);
    if (dtype == AST_F32)
        shell_filter_kernel<float2><<<g, 256, 0, s>>>((const float2*)in, (float2*)out, nmesh, lo2, hi2, i0_start, i0_count, i1_start, i1_count);
    else
        shell_filter_kernel<double2><<<g, 256, 0, s>>>((const double2*)in, (double2*)out, nmesh, lo2, hi2, i0_start, i0_count, i1_start, i1_count);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_shell_mask_real(double* out, int nmesh, int m_lo, int m_hi, void* stream) {
    AST_CHECK_ARG(out != nullptr && nmesh >= 4 && nmesh % 2 == 0 && m_lo >= 0 && m_hi > m_lo);
    shell_mask_real_kernel<<<ast::stream_grid((size_t)nmesh * nmesh * nmesh, 256), 256, 0, ast::as_stream(stream)>>>(
        out, nmesh, (long long)m_lo * m_lo, (long long)m_hi * m_hi);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_half_real_to_full(const void* spec, double* out, int nmesh, void* stream) {
    AST_CHECK_ARG(spec != nullptr && out != nullptr && (const void*)out != spec && nmesh >= 4 && nmesh % 2 == 0);
    half_real_to_full_kernel<<<ast::stream_grid((size_t)nmesh * nmesh * nmesh, 256), 256, 0, ast::as_stream(stream)>>>(
        (const double2*)spec, out, nmesh);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_interlace_compensate(void* c1, const void* c2, int dtype, int nmesh, int window, int compensate,
                                        int i0_start, int i0_count, int i1_start, int i1_count, void* stream) {
    AST_CHECK_ARG(c1 != nullptr && (c2 != nullptr || compensate));
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(window == AST_WIN_CIC || window == AST_WIN_TSC);
    AST_CHECK_ARG(nmesh >= 4 && nmesh % 2 == 0);
    AST_CHECK_ARG(i0_start >= 0 && i0_count >= 0 && i0_start + i0_count <= nmesh);
    AST_CHECK_ARG(i1_start >= 0 && i1_count >= 0 && i1_start + i1_count <= nmesh);
    const size_t total = (size_t)i0_count * i1_count * (nmesh / 2 + 1);
    if (total == 0) return AST_OK;
    unsigned g = ast::stream_grid(total, 256);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("interlace_compensate", s);
    if (dtype == AST_F32)
        interlace_compensate_kernel<float2><<<g, 256, 0, s>>>((float2*)c1, (const float2*)c2, nmesh, window, compensate != 0,
                                                             i0_start, i0_count, i1_start, i1_count);
    else
        interlace_compensate_kernel<double2><<<g, 256, 0, s>>>((double2*)c1, (const double2*)c2, nmesh, window, compensate != 0,
                                                              i0_start, i0_count, i1_start, i1_count);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_triple_product_sum(const void* a, const void* b, const void* c, int dtype, size_t count,
                                      double* out, void* stream) {
    AST_CHECK_ARG(a && b && c && out);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    if (count == 0) return AST_OK;
    unsigned g = ast::stream_grid(count, 256);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("triple_product_sum", s);
    if (dtype == AST_F32)
        triple_sum_kernel<float><<<g, 256, 0, s>>>((const float*)a, (const float*)b, (const float*)c, count, out);
    else
        triple_sum_kernel<double><<<g, 256, 0, s>>>((const double*)a, (const double*)b, (const double*)c, count, out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" size_t ast_triple_product_sums_scratch_bytes(void) { return (size_t)TRI_BLOCKS * TRI_THREADS * sizeof(double); }

extern "C" int ast_triple_product_sums(const void* const* fields, int nfields, int dtype, size_t count, const int* tri,
                                       int ntri, void* scratch, double* out, void* stream) {
    AST_CHECK_ARG(fields && tri && scratch && out);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nfields >= 1 && ntri >= 1 && ntri <= TRI_THREADS);
    const size_t esz = dtype == AST_F32 ? 4 : 8;
    const size_t lds = (size_t)nfields * (TRI_CHUNK + 1) * esz;
    AST_CHECK_ARG(lds <= 160 * 1024);
    hipStream_t s = ast::as_stream(stream);
    const int parts = TRI_THREADS / ntri;
    const size_t nchunks = (count + TRI_CHUNK - 1) / TRI_CHUNK;
    // one resident wave of workgroups: 256 CUs x as many as the LDS lets a CU hold (no tail of a second round)
    size_t resident = (160 * 1024) / (lds ? lds : 1);
    resident = 256 * (resident < 1 ? 1 : resident > 8 ? 8 : resident);
    const int blocks = (int)(nchunks < resident ? (nchunks ? nchunks : 1) : resident);
    AST_PROF("triple_product_sums", s);
    if (dtype == AST_F32) {
        static ast::PerDeviceOnce attr_once;
        if (attr_once.need()) {
            AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&triple_sums_kernel<float, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&triple_sums_kernel<float, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_once.mark();
        }
        // the pipelined variant always issues TRI_PIPE_FIELDS loads per thread and chunk (slots past nfields re-read the last
        // field): it pays for many fields (the bispectrum's 31 shells), not for the three fields of a single triangle
        if (nfields > TRI_PIPE_FIELDS / 2 && nfields <= TRI_PIPE_FIELDS && !getenv("AST_TRI_NO_PIPE"))
            triple_sums_kernel<float, true><<<blocks, TRI_THREADS, lds, s>>>((const float* const*)fields, nfields, tri, ntri, parts, count,
                                                                             (double*)scratch);
        else
            triple_sums_kernel<float, false><<<blocks, TRI_THREADS, lds, s>>>((const float* const*)fields, nfields, tri, ntri, parts, count,
                                                                              (double*)scratch);
    } else {
        static ast::PerDeviceOnce attr_once;
        if (attr_once.need()) {
            AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&triple_sums_kernel<double, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_once.mark();
        }
        triple_sums_kernel<double, false><<<blocks, TRI_THREADS, lds, s>>>((const double* const*)fields, nfields, tri, ntri, parts, count,
                                                                           (double*)scratch);
    }
    AST_CHECK_LAUNCH();
    triple_reduce_kernel<<<ntri, 256, 0, s>>>((const double*)scratch, blocks, ntri, parts, out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

template <bool UNPACK>
static int slab_pack_impl(const void* in, void* out, int dtype, size_t n0, size_t n1, size_t n2, int parts, void* stream) {
    AST_CHECK_ARG(in && out && in != out);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(parts >= 1 && n1 % (size_t)parts == 0);
    const size_t total = n0 * n1 * n2;
    if (total == 0) return AST_OK;
    unsigned g = ast::stream_grid(total, 256);
    hipStream_t s = ast::as_stream(stream);
    if (dtype == AST_F32)
        slab_pack_kernel<float2, UNPACK><<<g, 256, 0, s>>>((const float2*)in, (float2*)out, n0, n1, n2, parts);
    else
        slab_pack_kernel<double2, UNPACK><<<g, 256, 0, s>>>((const double2*)in, (double2*)out, n0, n1, n2, parts);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_slab_pack(const void* in, void* out, int dtype, size_t n0, size_t n1, size_t n2, int parts, void* stream) {
    return slab_pack_impl<false>(in, out, dtype, n0, n1, n2, parts, stream);
}
extern "C" int ast_slab_unpack(const void* in, void* out, int dtype, size_t n0, size_t n1, size_t n2, int parts, void* stream) {
    return slab_pack_impl<true>(in, out, dtype, n0, n1, n2, parts, stream);
}
