// Error state, fill / divide / add / accumulate streaming kernels, min-max and
// histogram reductions, synthetic particle generator.
#include "ast_common.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace ast {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
struct ProfRec { const char* name; hipEvent_t a, b; };
std::mutex g_prof_mutex;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t s) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    ProfRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].b, stream);
}

}  // namespace ast

extern "C" int ast_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(ast::g_prof_mutex);
    for (auto& r : ast::g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    ast::g_prof.clear();
    ast::g_prof_on = on != 0;
    return AST_OK;
}

extern "C" int ast_profile_report(char* buf, size_t cap) {
    AST_CHECK_ARG(buf != nullptr && cap > 0);
    std::lock_guard<std::mutex> lock(ast::g_prof_mutex);
    struct Agg { const char* name; int calls; double ms; };
    std::vector<Agg> agg;
    for (auto& r : ast::g_prof) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        bool found = false;
        for (auto& a : agg) if (strcmp(a.name, r.name) == 0) { a.calls++; a.ms += ms; found = true; break; }
        if (!found) agg.push_back({r.name, 1, ms});
    }
    size_t off = 0;
    buf[0] = 0;
    for (auto& a : agg) {
        int n = snprintf(buf + off, cap - off, "%s,%d,%.6f\n", a.name, a.calls, a.ms);
        if (n < 0 || (size_t)n >= cap - off) break;
        off += (size_t)n;
    }
    return AST_OK;
}

extern "C" int ast_version(void) { return 100; }
extern "C" const char* ast_last_error(void) { return ast::g_err; }

namespace {

template <typename T>
__global__ void fill_kernel(T* buf, size_t n, T v) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) buf[i] = v;
}

template <typename T>
__global__ void divide_kernel(T* buf, size_t n, T d) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) buf[i] = buf[i] / d;
}

template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

// 64-lane butterfly reductions
__device__ inline double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// order-preserving map double <-> uint64 so min/max can use integer atomics
__device__ inline unsigned long long d2key(double d) {
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double key2d(unsigned long long k) {
    unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// V elements per load (16 bytes when the buffer is aligned), four loads in flight per thread, extrema reduced over the
// wave and then over the workgroup: ONE pair of atomics per workgroup, and few workgroups (1024 threads, two per CU).
// The same-address atomics serialise at the memory side: one pair per WAVE (8192 waves) took 0.2 ms for a 134 MB map
// that streams in 0.03, one pair per 256-thread workgroup (2048 of them) 0.057 ms.
constexpr int MINMAX_BLOCK = 1024, MINMAX_GRID = 512;
template <typename T, int V>
__global__ void __launch_bounds__(MINMAX_BLOCK) minmax_kernel(const T* __restrict__ buf, size_t n, unsigned long long* keys) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    __shared__ double slo[MINMAX_BLOCK / 64], shi[MINMAX_BLOCK / 64];
    const size_t nvec = n / V;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double l[4] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX}, h[4] = {-DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
    const vec_t* vb = reinterpret_cast<const vec_t*>(buf);
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        vec_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vb[i + j * stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < V; ++k) { l[j] = fmin(l[j], (double)v[j][k]); h[j] = fmax(h[j], (double)v[j][k]); }
        }
    }
    double lo = DBL_MAX, hi = -DBL_MAX;
    for (; i < nvec; i += stride) {
        const vec_t v = vb[i];
#pragma unroll
        for (int k = 0; k < V; ++k) { lo = fmin(lo, (double)v[k]); hi = fmax(hi, (double)v[k]); }
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * V) {           // the elements past the last whole vector
        const double v = (double)buf[nvec * V + threadIdx.x];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    lo = fmin(fmin(lo, l[0]), fmin(fmin(l[1], l[2]), l[3]));
    hi = fmax(fmax(hi, h[0]), fmax(fmax(h[1], h[2]), h[3]));
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x < 64) {
        lo = threadIdx.x < MINMAX_BLOCK / 64 ? slo[threadIdx.x] : DBL_MAX;
        hi = threadIdx.x < MINMAX_BLOCK / 64 ? shi[threadIdx.x] : -DBL_MAX;
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (threadIdx.x == 0) {
            atomicMin(&keys[0], d2key(lo));
            atomicMax(&keys[1], d2key(hi));
        }
    }
}

__global__ void minmax_init(unsigned long long* keys) {
    keys[0] = ~0ull;
    keys[1] = 0ull;
}
__global__ void minmax_finish(unsigned long long* keys) {
    double lo = key2d(keys[0]), hi = key2d(keys[1]);
    reinterpret_cast<double*>(keys)[0] = lo;
    reinterpret_cast<double*>(keys)[1] = hi;
}

// Fixed-order sum in double: AST_SUM_PARTS block partials (each block a fixed contiguous range,
// lanes a fixed stride, tree in a fixed order), then one block adds the partials in index order.
template <typename T>
__global__ void __launch_bounds__(256) sum_stage1_kernel(const T* __restrict__ buf, size_t n, double* __restrict__ part) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    double acc = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) acc += (double)buf[i];
    __shared__ double sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void sum_stage2_kernel(const double* __restrict__ part, int nparts, double* out) {
    double t = 0.0;
    for (int i = 0; i < nparts; ++i) t += part[i];
    *out = t;
}

// np.histogram with uniform bins (numpy/lib/_histograms_impl.py fast path):
//   f = (v - lo) * (nbins / (hi - lo));  idx = (int) f;  idx == nbins -> nbins-1;
//   then numpy corrects against the float64 edges: v < edge[idx] -> idx-1,
//   v >= edge[idx+1] && idx != nbins-1 -> idx+1.   edges = linspace(lo, hi, nbins+1).
// range_d (or null): {lo, hi} on the device (ast_minmax's output) - the range=None case of np.histogram without a
// host round trip; a degenerate range is widened by +-0.5 like numpy does.
// V elements per load (16 bytes when the buffer is aligned), four loads in flight; the workgroup's LDS histogram is kept
// in `copies` replicas (lane mod copies picks one) so that the lanes of a wave that hit the same popular bin do not
// queue on one LDS word; 1024-thread workgroups, at most 512 of them: nbins global atomics each at the end.
constexpr int HIST_BLOCK = 1024, HIST_GRID = 512, HIST_MAXB = 4096;
template <typename T, int V>
__global__ void __launch_bounds__(HIST_BLOCK)
hist_kernel(const T* __restrict__ buf, size_t n, double lo, double hi, int nbins, int copies, unsigned long long* counts,
            const double* __restrict__ range_d = nullptr) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    __shared__ unsigned int lh[HIST_MAXB];
    for (int i = threadIdx.x; i < nbins * copies; i += HIST_BLOCK) lh[i] = 0;
    if (range_d) {
        lo = range_d[0];
        hi = range_d[1];
        if (lo == hi) { lo -= 0.5; hi += 0.5; }
    }
    __syncthreads();
    const double norm = (double)nbins / (hi - lo);
    const double step = (hi - lo) / (double)nbins;
    unsigned int* mine = lh + (threadIdx.x & (copies - 1)) * nbins;
    auto put = [&](double v) {
        if (!(v >= lo && v <= hi)) return;
        int idx = (int)((v - lo) * norm);
        if (idx == nbins) idx = nbins - 1;
        // numpy's linspace: edge[i] = lo + i*step (last forced to hi)
        const double e0 = lo + idx * step;
        const double e1 = (idx + 1 == nbins) ? hi : lo + (idx + 1) * step;
        if (v < e0) idx -= 1;
        else if (v >= e1 && idx != nbins - 1) idx += 1;
        atomicAdd(&mine[idx], 1u);
    };
    const size_t nvec = n / V;
    const size_t stride = (size_t)gridDim.x * HIST_BLOCK;
    size_t i = (size_t)blockIdx.x * HIST_BLOCK + threadIdx.x;
    const vec_t* vb = reinterpret_cast<const vec_t*>(buf);
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        vec_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vb[i + j * stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < V; ++k) put((double)v[j][k]);
        }
    }
    for (; i < nvec; i += stride) {
        const vec_t v = vb[i];
#pragma unroll
        for (int k = 0; k < V; ++k) put((double)v[k]);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * V) put((double)buf[nvec * V + threadIdx.x]);   // past the last whole vector
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += HIST_BLOCK) {
        unsigned long long total = 0;
        for (int c = 0; c < copies; ++c) total += lh[c * nbins + b];
        if (total) atomicAdd(&counts[b], total);
    }
}

template <typename T>
void launch_hist(const T* buf, size_t count, double lo, double hi, int nbins, unsigned long long* counts, const double* range_d,
                 hipStream_t s) {
    int copies = 1;
    while (copies < 32 && 2 * copies * nbins <= HIST_MAXB) copies *= 2;
    constexpr int W = 16 / (int)sizeof(T);
    const bool wide = ((uintptr_t)buf & 15) == 0;
    const size_t per = wide ? W : 1;
    const size_t need = ((count + per - 1) / per + HIST_BLOCK - 1) / HIST_BLOCK;
    const unsigned g = (unsigned)(need > (size_t)HIST_GRID ? (size_t)HIST_GRID : need);
    if (wide) hist_kernel<T, W><<<g, HIST_BLOCK, 0, s>>>(buf, count, lo, hi, nbins, copies, counts, range_d);
    else hist_kernel<T, 1><<<g, HIST_BLOCK, 0, s>>>(buf, count, lo, hi, nbins, copies, counts, range_d);
}

// ---- counter-based normal generator for the synthetic particle set ----
__device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// A pseudo-random PERMUTATION of [0, count): a four-round Feistel network on 2 * half bits (the smallest even width that
// covers count), round function = the splitmix64 finaliser keyed by (key, round), walked along its cycle until the value
// falls below count (a bijection of [0, 2^bits) restricted this way is a bijection of [0, count); < 4 steps on average).
// "shuffled = a fixed permutation from seed + 1".
__device__ inline uint64_t feistel_perm(uint64_t t, uint64_t count, int half, uint64_t key) {
    const uint64_t mask = (1ull << half) - 1ull;
    uint64_t v = t;
    do {
        uint64_t l = v >> half, r = v & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint64_t f = mix64(r ^ mix64(key + (uint64_t)round)) & mask;
            const uint64_t nl = r;
            r = l ^ f;
            l = nl;
        }
        v = (l << half) | r;
    } while (v >= count);
    return v;
}

constexpr uint64_t SYNTH_SHUFFLE_PRNG = ~0ull;      // shuffle_stride value that selects the Feistel permutation

template <typename T>
__global__ void synth_kernel(T* pos, size_t first, size_t count, int npside, double boxsize,
                             double sigma, uint64_t seed, uint64_t shuffle_stride) {
    const double h = boxsize / npside;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    int half = 1;
    while ((1ull << (2 * half)) < count) ++half;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        // shuffled order permutes the particles INSIDE [first, first + count), so a rank's
        // range keeps holding its own lattice planes
        uint64_t id = first + (shuffle_stride == SYNTH_SHUFFLE_PRNG ? feistel_perm(t, count, half, seed + 1)
                               : shuffle_stride ? (uint64_t)(((unsigned __int128)t * shuffle_stride) % count) : t);
        uint64_t k = id % npside, j = (id / npside) % npside, i = id / ((uint64_t)npside * npside);
        double q[3] = {(i + 0.5) * h, (j + 0.5) * h, (k + 0.5) * h};
        // two Box-Muller pairs give three normals
        uint64_t a = mix64(seed ^ mix64(id * 4 + 0)), b = mix64(seed ^ mix64(id * 4 + 1));
        uint64_t c = mix64(seed ^ mix64(id * 4 + 2)), d = mix64(seed ^ mix64(id * 4 + 3));
        double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0);   // (0,1)
        double u2 = (b >> 11) * (1.0 / 9007199254740992.0);
        double u3 = ((c >> 11) + 1.0) * (1.0 / 9007199254740993.0);
        double u4 = (d >> 11) * (1.0 / 9007199254740992.0);
        double r1 = sqrt(-2.0 * log(u1)), r2 = sqrt(-2.0 * log(u3));
        double s1, c1, s2, c2;
        sincos(6.283185307179586 * u2, &s1, &c1);
        sincos(6.283185307179586 * u4, &s2, &c2);
        double xi[3] = {r1 * c1, r1 * s1, r2 * c2};
        for (int dim = 0; dim < 3; ++dim) {
            double x = q[dim] + sigma * xi[dim];
            x -= floor(x / boxsize) * boxsize;
            T xt = (T)x;
            if (xt >= (T)boxsize) xt = (T)0;   // the cast may round up to L
            pos[3 * t + dim] = xt;
        }
    }
}

// ---- clustered synthetic set: a lattice collapsing onto attractors ----
// Evolved snapshots are what PowerSpectrum3D and SubFind.power_spectrum really paint (stats_subfind.py:125-131): most
// particles in a smooth background, a heavy tail in a few dense knots.  Lattice site q (k fastest, like the lattice set) is
// pulled towards every attractor h within 3 R_h:  x = q - sum_h A exp(-r^2 / (2 R_h^2)) (q - c_h)  (minimum image) + the
// half-cell Gaussian jitter of the lattice set.  Inside ~R_h / 2 the lattice is compressed by 1 - A in radius: with A = 0.95
// and R_h up to 48 cells a knot holds several 10^5 particles in a few cells - tile occupancies of 100 x the mean - while
// neighbours in memory stay neighbours in space outside the knots (file order of a real snapshot).  Attractor h: centre
// uniform in the box, R_h = (6 + 42 u^3) L / npside with u uniform - many small, few large - all from hashes of (seed, h).
constexpr int SYNTH_MAX_ATTRACTORS = 1024;
template <typename T>
__global__ void __launch_bounds__(256)
synth_clustered_kernel(T* pos, size_t count, int npside, double boxsize, double sigma, uint64_t seed, int nattr, double amp,
                       uint64_t shuffle) {
    __shared__ double ax[SYNTH_MAX_ATTRACTORS], ay[SYNTH_MAX_ATTRACTORS], az[SYNTH_MAX_ATTRACTORS], ar[SYNTH_MAX_ATTRACTORS];
    const double h = boxsize / npside;
    for (int a = threadIdx.x; a < nattr; a += 256) {
        const uint64_t k = seed * 0x9e3779b97f4a7c15ull + 77;
        ax[a] = (mix64(k ^ mix64(4 * (uint64_t)a + 0)) >> 11) * (1.0 / 9007199254740992.0) * boxsize;
        ay[a] = (mix64(k ^ mix64(4 * (uint64_t)a + 1)) >> 11) * (1.0 / 9007199254740992.0) * boxsize;
        az[a] = (mix64(k ^ mix64(4 * (uint64_t)a + 2)) >> 11) * (1.0 / 9007199254740992.0) * boxsize;
        const double u = (mix64(k ^ mix64(4 * (uint64_t)a + 3)) >> 11) * (1.0 / 9007199254740992.0);
        ar[a] = (6.0 + 42.0 * u * u * u) * h;
    }
    __syncthreads();
    int half = 1;
    while ((1ull << (2 * half)) < count) ++half;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const uint64_t id = shuffle ? feistel_perm(t, count, half, seed + 1) : t;
        const uint64_t kk = id % npside, jj = (id / npside) % npside, ii = id / ((uint64_t)npside * npside);
        const double q[3] = {(ii + 0.5) * h, (jj + 0.5) * h, (kk + 0.5) * h};
        double d[3] = {0.0, 0.0, 0.0};
        for (int a = 0; a < nattr; ++a) {
            double e[3] = {q[0] - ax[a], q[1] - ay[a], q[2] - az[a]};
#pragma unroll
            for (int c = 0; c < 3; ++c) e[c] -= boxsize * rint(e[c] / boxsize);
            const double r2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], rr = ar[a] * ar[a];
            if (r2 < 9.0 * rr) {
                const double w = amp * exp(-0.5 * r2 / rr);
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] -= w * e[c];
            }
        }
        uint64_t a0 = mix64(seed ^ mix64(id * 4 + 0)), b0 = mix64(seed ^ mix64(id * 4 + 1));
        uint64_t c0 = mix64(seed ^ mix64(id * 4 + 2)), d0 = mix64(seed ^ mix64(id * 4 + 3));
        const double u1 = ((a0 >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (b0 >> 11) * (1.0 / 9007199254740992.0);
        const double u3 = ((c0 >> 11) + 1.0) * (1.0 / 9007199254740993.0), u4 = (d0 >> 11) * (1.0 / 9007199254740992.0);
        const double r1 = sqrt(-2.0 * log(u1)), r2g = sqrt(-2.0 * log(u3));
        double s1, c1, s2, c2;
        sincos(6.283185307179586 * u2, &s1, &c1);
        sincos(6.283185307179586 * u4, &s2, &c2);
        const double xi[3] = {r1 * c1, r1 * s1, r2g * c2};
        for (int dim = 0; dim < 3; ++dim) {
            double x = q[dim] + d[dim] + sigma * xi[dim];
            x -= floor(x / boxsize) * boxsize;
            T xt = (T)x;
            if (xt >= (T)boxsize) xt = (T)0;
            pos[3 * t + dim] = xt;
        }
    }
}

}  // namespace

extern "C" int ast_synth_clustered_particles(void* pos, int dtype, size_t count, int npside, double boxsize, double sigma,
                                             uint64_t seed, int nattractors, double amplitude, int shuffle, void* stream) {
    AST_CHECK_ARG(pos != nullptr || count == 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(npside > 0 && boxsize > 0 && count == (size_t)npside * npside * npside);
    AST_CHECK_ARG(nattractors >= 0 && nattractors <= SYNTH_MAX_ATTRACTORS && amplitude >= 0.0 && amplitude < 1.0);
    if (count == 0) return AST_OK;
    const unsigned g = ast::stream_grid(count, 256);
    if (dtype == AST_F32)
        synth_clustered_kernel<float><<<g, 256, 0, ast::as_stream(stream)>>>((float*)pos, count, npside, boxsize, sigma, seed, nattractors, amplitude, shuffle ? 1 : 0);
    else
        synth_clustered_kernel<double><<<g, 256, 0, ast::as_stream(stream)>>>((double*)pos, count, npside, boxsize, sigma, seed, nattractors, amplitude, shuffle ? 1 : 0);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_fill(void* buf, int dtype, size_t count, double value, void* stream) {
    AST_CHECK_ARG(buf != nullptr || count == 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    if (count == 0) return AST_OK;
    unsigned g = ast::stream_grid(count, 256);
    if (dtype == AST_F32)
        fill_kernel<float><<<g, 256, 0, ast::as_stream(stream)>>>((float*)buf, count, (float)value);
    else
        fill_kernel<double><<<g, 256, 0, ast::as_stream(stream)>>>((double*)buf, count, value);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_divide(void* buf, int dtype, size_t count, double divisor, void* stream) {
    AST_CHECK_ARG(buf != nullptr || count == 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    if (count == 0) return AST_OK;
    unsigned g = ast::stream_grid(count, 256);
    if (dtype == AST_F32)
        divide_kernel<float><<<g, 256, 0, ast::as_stream(stream)>>>((float*)buf, count, (float)divisor);
    else
        divide_kernel<double><<<g, 256, 0, ast::as_stream(stream)>>>((double*)buf, count, divisor);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_add(const void* a, const void* b, void* out, int dtype, size_t count, void* stream) {
    AST_CHECK_ARG((a && b && out) || count == 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    if (count == 0) return AST_OK;
    unsigned g = ast::stream_grid(count, 256);
    if (dtype == AST_F32)
        add_kernel<float><<<g, 256, 0, ast::as_stream(stream)>>>((const float*)a, (const float*)b, (float*)out, count);
    else
        add_kernel<double><<<g, 256, 0, ast::as_stream(stream)>>>((const double*)a, (const double*)b, (double*)out, count);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// The on-box streaming ceiling the roofline fractions are also quoted against ("also measure an on-box
// streaming-copy kernel"): 16 bytes per lane and access, four accesses in flight per lane, consecutive workgroups on
// consecutive 16-KB pieces.  mode 0 = copy (read + write), 1 = read only (the sum keeps the loads alive; one store per
// workgroup that never happens: the guard value is not reachable), 2 = write only.
typedef float vfloat4 __attribute__((ext_vector_type(4)));
template <bool NTL, bool NTS, int U>
__global__ void __launch_bounds__(256) stream_copy_kernel(vfloat4* __restrict__ dst, const vfloat4* __restrict__ src, size_t n16, int mode) {
    const size_t per_block = 256 * U;
    vfloat4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t base = (size_t)blockIdx.x * per_block; base < n16; base += (size_t)gridDim.x * per_block) {
        vfloat4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t i = base + (size_t)j * 256 + threadIdx.x;
            v[j] = (mode != 2 && i < n16) ? (NTL ? __builtin_nontemporal_load(src + i) : src[i]) : vfloat4{1.f, 2.f, 3.f, 4.f};
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t i = base + (size_t)j * 256 + threadIdx.x;
            if (mode == 1) acc += v[j];
            else if (i < n16) { if (NTS) __builtin_nontemporal_store(v[j], dst + i); else dst[i] = v[j]; }
        }
    }
    if (mode == 1 && acc.x + acc.y + acc.z + acc.w == -1.2345e38f) dst[0] = acc;
}

// mode bits 0-1: 0 copy, 1 read, 2 write.  Tuning bits (scripts/micro/copy_rate.py; honoured when bit 8 is set): 4 plain
// loads, 5 plain stores, 6 one piece per workgroup instead of a grid-stride loop, 7 eight accesses per lane instead of four.
extern "C" int ast_stream_copy(void* dst, const void* src, size_t bytes, int mode, void* stream) {
    // defaults = the fastest variant per operation on the MI355X (scripts/micro/copy_rate.py, 4 GiB): copy 5.78 TB/s with plain
    // loads and stores, read 7.18 with nontemporal loads, write 6.18 with plain stores - one 16-KB piece per workgroup in all
    // three (a grid-stride loop over 8192 workgroups: 5.2 / 7.0 / 5.7).  Bit 8 set: bits 4-7 choose the variant explicitly.
    const int op = mode & 3, tune = (mode & 256) ? (mode >> 4) & 15 : (op == 1 ? 4 : 7);
    AST_CHECK_ARG(dst != nullptr && (src != nullptr || op == 2));
    AST_CHECK_ARG(mode >= 0 && mode < 512 && op <= 2 && (mode & 12) == 0 && ((mode & 256) || (mode & 240) == 0));
    AST_CHECK_ARG(bytes % 16 == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0);
    if (bytes == 0) return AST_OK;
    const int U = (tune & 8) ? 8 : 4;
    const size_t n16 = bytes / 16, blocks = (n16 + 256 * U - 1) / (256 * U);
    AST_CHECK_ARG(blocks < (1ull << 31));
    const unsigned g = (tune & 4) ? (unsigned)blocks : (unsigned)(blocks < 256u * 32u ? blocks : 256u * 32u);
    vfloat4* d = (vfloat4*)dst;
    const vfloat4* sp = (const vfloat4*)src;
    hipStream_t st = ast::as_stream(stream);
#define AST_COPY_CASE(NTL, NTS, UU) stream_copy_kernel<NTL, NTS, UU><<<g, 256, 0, st>>>(d, sp, n16, op)
    switch (tune & 11) {
        case 0: AST_COPY_CASE(true, true, 4); break;
        case 1: AST_COPY_CASE(false, true, 4); break;
        case 2: AST_COPY_CASE(true, false, 4); break;
        case 3: AST_COPY_CASE(false, false, 4); break;
        case 8: AST_COPY_CASE(true, true, 8); break;
        case 9: AST_COPY_CASE(false, true, 8); break;
        case 10: AST_COPY_CASE(true, false, 8); break;
        default: AST_COPY_CASE(false, false, 8); break;
    }
#undef AST_COPY_CASE
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_accumulate(void* dst, const void* src, int dtype, size_t count, void* stream) {
    return ast_add(dst, src, dst, dtype, count, stream);
}

extern "C" int ast_minmax(const void* buf, int dtype, size_t count, double* out, void* stream) {
    AST_CHECK_ARG(buf && out && count > 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    hipStream_t s = ast::as_stream(stream);
    auto* keys = reinterpret_cast<unsigned long long*>(out);
    minmax_init<<<1, 1, 0, s>>>(keys);
    const bool wide = ((uintptr_t)buf & 15) == 0;
    const size_t per = wide ? 16 / (dtype == AST_F32 ? 4 : 8) : 1;
    const size_t need = ((count + per - 1) / per + MINMAX_BLOCK - 1) / MINMAX_BLOCK;
    const unsigned g = (unsigned)(need > (size_t)MINMAX_GRID ? (size_t)MINMAX_GRID : need);
    if (dtype == AST_F32) {
        if (wide) minmax_kernel<float, 4><<<g, MINMAX_BLOCK, 0, s>>>((const float*)buf, count, keys);
        else minmax_kernel<float, 1><<<g, MINMAX_BLOCK, 0, s>>>((const float*)buf, count, keys);
    } else {
        if (wide) minmax_kernel<double, 2><<<g, MINMAX_BLOCK, 0, s>>>((const double*)buf, count, keys);
        else minmax_kernel<double, 1><<<g, MINMAX_BLOCK, 0, s>>>((const double*)buf, count, keys);
    }
    minmax_finish<<<1, 1, 0, s>>>(keys);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_sum(const void* buf, int dtype, size_t count, double* out, void* stream) {
    AST_CHECK_ARG(buf && out && count > 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    hipStream_t s = ast::as_stream(stream);
    if (dtype == AST_F32)
        sum_stage1_kernel<float><<<AST_SUM_PARTS, 256, 0, s>>>((const float*)buf, count, out + 1);
    else
        sum_stage1_kernel<double><<<AST_SUM_PARTS, 256, 0, s>>>((const double*)buf, count, out + 1);
    sum_stage2_kernel<<<1, 1, 0, s>>>(out + 1, AST_SUM_PARTS, out);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_histogram(const void* buf, int dtype, size_t count, double lo, double hi, int nbins,
                             long long* counts, void* stream) {
    AST_CHECK_ARG(buf && counts);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nbins > 0 && nbins <= 4096);
    AST_CHECK_ARG(hi > lo);
    if (count == 0) return AST_OK;
    auto* c = reinterpret_cast<unsigned long long*>(counts);
    if (dtype == AST_F32) launch_hist((const float*)buf, count, lo, hi, nbins, c, nullptr, ast::as_stream(stream));
    else launch_hist((const double*)buf, count, lo, hi, nbins, c, nullptr, ast::as_stream(stream));
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// np.histogram(buf, bins=nbins) with range=None in one stream-ordered call: min/max into range_d (2 doubles, as
// ast_minmax), then the counts against that range read from the device.  The caller fetches counts and range together.
extern "C" int ast_histogram_auto(const void* buf, int dtype, size_t count, int nbins, long long* counts, double* range_d,
                                  void* stream) {
    AST_CHECK_ARG(buf && counts && range_d && count > 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nbins > 0 && nbins <= 4096);
    const int rc = ast_minmax(buf, dtype, count, range_d, stream);
    if (rc != AST_OK) return rc;
    auto* c = reinterpret_cast<unsigned long long*>(counts);
    if (dtype == AST_F32) launch_hist((const float*)buf, count, 0.0, 1.0, nbins, c, range_d, ast::as_stream(stream));
    else launch_hist((const double*)buf, count, 0.0, 1.0, nbins, c, range_d, ast::as_stream(stream));
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_synth_lattice_particles(void* pos, int dtype, size_t first, size_t count, int npside,
                                           double boxsize, double sigma, uint64_t seed,
                                           uint64_t shuffle_stride, void* stream) {
    AST_CHECK_ARG(pos != nullptr || count == 0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(npside > 0 && boxsize > 0);
    AST_CHECK_ARG(first + count <= (size_t)npside * npside * npside);
    if (count == 0) return AST_OK;
    unsigned g = ast::stream_grid(count, 256);
    if (dtype == AST_F32)
        synth_kernel<float><<<g, 256, 0, ast::as_stream(stream)>>>((float*)pos, first, count, npside, boxsize, sigma, seed, shuffle_stride);
    else
        synth_kernel<double><<<g, 256, 0, ast::as_stream(stream)>>>((double*)pos, first, count, npside, boxsize, sigma, seed, shuffle_stride);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// ---- particle routing for slab-decomposed paints ----
// Destination of a particle = the slab that owns its base plane (the cell floor(s) for CIC, the nearest grid
// point floor(s + 1/2) for NGP / TSC, wrapped into the box), slabs of nloc = nmesh / nparts planes.
namespace {
constexpr int ROUTE_MAX_PARTS = 64;

template <typename T>
__device__ inline int route_dest(T x, double inv_dx, double half, int n, int nloc) {
    const double fl = floor((double)x * inv_dx + half);
    const double dn = (double)n;
    double r = fl - floor(fl / dn) * dn;
    r = r >= dn ? r - dn : r;
    r = r < 0.0 ? r + dn : r;
    return (int)r / nloc;
}

// MODE 0: counts[part] += particles; MODE 1: scatter to out at cursor[part] (reserved per block through LDS counts)
template <typename T, int MODE>
__global__ void __launch_bounds__(256)
route_kernel(const T* __restrict__ pos, const T* __restrict__ mass, size_t np, double inv_dx, double half, int n, int nloc,
             int nparts, unsigned long long* __restrict__ counts, T* __restrict__ out_pos, T* __restrict__ out_mass) {
    __shared__ unsigned int lcount[ROUTE_MAX_PARTS];
    __shared__ unsigned long long lbase[ROUTE_MAX_PARTS];
    const size_t per_block = 256 * 8;
    for (size_t b0 = (size_t)blockIdx.x * per_block; b0 < np; b0 += (size_t)gridDim.x * per_block) {
        if (threadIdx.x < ROUTE_MAX_PARTS) lcount[threadIdx.x] = 0;
        __syncthreads();
        int dest[8];
        unsigned int slot[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t p = b0 + (size_t)u * 256 + threadIdx.x;
            dest[u] = -1;
            if (p < np) {
                dest[u] = route_dest(pos[3 * p], inv_dx, half, n, nloc);
                slot[u] = atomicAdd(&lcount[dest[u]], 1u);
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nparts && lcount[threadIdx.x])
            lbase[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)lcount[threadIdx.x]);
        if (MODE == 1) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (dest[u] < 0) continue;
                const size_t p = b0 + (size_t)u * 256 + threadIdx.x;
                const size_t q = (size_t)(lbase[dest[u]] + slot[u]);
                out_pos[3 * q + 0] = pos[3 * p + 0];
                out_pos[3 * q + 1] = pos[3 * p + 1];
                out_pos[3 * q + 2] = pos[3 * p + 2];
                if (mass) out_mass[q] = mass[p];
            }
        }
        __syncthreads();
    }
}
}  // namespace

// counts_d[part] (uint64, caller zero-fills) += number of particles whose base plane belongs to slab `part`.
extern "C" int ast_route_count(const void* pos, int dtype, size_t np, int nmesh, double boxsize, int window, int nparts,
                               unsigned long long* counts, void* stream) {
    AST_CHECK_ARG(counts && (pos || np == 0) && nmesh > 0 && boxsize > 0.0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nparts >= 1 && nparts <= ROUTE_MAX_PARTS && nmesh % nparts == 0);
    if (np == 0) return AST_OK;
    const double half = window == AST_WIN_CIC ? 0.0 : 0.5;
    const unsigned g = ast::stream_grid((np + 7) / 8, 256);
    hipStream_t s = ast::as_stream(stream);
    if (dtype == AST_F32)
        route_kernel<float, 0><<<g, 256, 0, s>>>((const float*)pos, nullptr, np, nmesh / boxsize, half, nmesh, nmesh / nparts, nparts, counts, nullptr, nullptr);
    else
        route_kernel<double, 0><<<g, 256, 0, s>>>((const double*)pos, nullptr, np, nmesh / boxsize, half, nmesh, nmesh / nparts, nparts, counts, nullptr, nullptr);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// Particles grouped by destination slab: slab `part` occupies out[cursor_d[part] ...) where cursor_d holds the
// exclusive prefix sums of ast_route_count on entry (it is advanced; order inside a slab's range is not defined).
extern "C" int ast_route_scatter(const void* pos, const void* mass, int dtype, size_t np, int nmesh, double boxsize,
                                 int window, int nparts, unsigned long long* cursor, void* out_pos, void* out_mass,
                                 void* stream) {
    AST_CHECK_ARG(cursor && (pos || np == 0) && (out_pos || np == 0) && nmesh > 0 && boxsize > 0.0);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG((mass == nullptr) == (out_mass == nullptr));
    AST_CHECK_ARG(nparts >= 1 && nparts <= ROUTE_MAX_PARTS && nmesh % nparts == 0);
    if (np == 0) return AST_OK;
    const double half = window == AST_WIN_CIC ? 0.0 : 0.5;
    const unsigned g = ast::stream_grid((np + 7) / 8, 256);
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("route_scatter", s);
    if (dtype == AST_F32)
        route_kernel<float, 1><<<g, 256, 0, s>>>((const float*)pos, (const float*)mass, np, nmesh / boxsize, half, nmesh, nmesh / nparts, nparts, cursor, (float*)out_pos, (float*)out_mass);
    else
        route_kernel<double, 1><<<g, 256, 0, s>>>((const double*)pos, (const double*)mass, np, nmesh / boxsize, half, nmesh, nmesh / nparts, nparts, cursor, (double*)out_pos, (double*)out_mass);
    AST_CHECK_LAUNCH();
    return AST_OK;
}
