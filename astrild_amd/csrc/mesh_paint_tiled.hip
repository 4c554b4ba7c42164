// a-2, LDS-tiled: CIC / TSC scatter-add through LDS-resident grid tiles.
//
// Random global float atomics run ~17x below the coalesced atomic rate on
// MI355X (MI355X_MICROARCH.md "Global float atomics"), so the deposit is
// restructured so that all 8 / 27 updates of a particle land in LDS:
//
//   A  run scan   one thread per particle (coalesced position loads): tile id
//                 of its base cell; inside each wave, consecutive particles
//                 with the same tile form a run {tile, first particle, length}.
//                 Only run heads allocate a slot and bump the tile's run count.
//                 No particle data is moved or copied.
//   B  scan       exclusive prefix sum of the per-tile run counts.
//   C  bucket     runs are written tile-major.
//   D  deposit    one workgroup per tile: zero an LDS tile (+ window halo),
//                 each wave walks runs of this tile (a run is a contiguous
//                 piece of the particle array, so the loads stay coalesced),
//                 ds_add_f32/f64 into LDS, one flush of the tile to HBM.
//
// Spatially coherent input (lattice / Morton / slab ordered snapshots) gives
// long runs and the run list is tiny; fully shuffled input degrades to one run
// per particle — still correct, and no slower than a key-value sort would be.
#include "ast_common.h"

namespace {

constexpr int TX = 8, TY = 8, TZ = 32;   // owned cells per tile

template <int W> struct Win;
template <> struct Win<2> {
    static constexpr int LO = 0;
    __device__ static inline void eval(double s, long long& i0, double* w) {
        double fl = floor(s);
        double f = s - fl;
        i0 = (long long)fl;
        double a = 1.0 - f, b = f, sum = a + b;
        w[0] = a / sum;
        w[1] = b / sum;
    }
};
template <> struct Win<3> {
    static constexpr int LO = 1;
    __device__ static inline void eval(double s, long long& i0, double* w) {
        double ic = floor(s + 0.5);
        double d = s - ic;
        i0 = (long long)ic - 1;
        double hm = 0.5 - d, hp = 0.5 + d;
        double a = 0.5 * (hm * hm), b = 0.75 - d * d, c = 0.5 * (hp * hp);
        double sum = (a + b) + c;
        w[0] = a / sum;
        w[1] = b / sum;
        w[2] = c / sum;
    }
};

__device__ inline int wrapi(long long i, int n) {
    long long r = i % n;
    return (int)(r < 0 ? r + n : r);
}

struct TileGeom {
    int n, x_start, nx_alloc;
    int ntx, nty, ntz;
    double inv_dx;
};

// base cell (window centre for TSC, lower corner for CIC) -> tile id, or
// 0xffffffff when the base plane is outside the buffer.
template <typename T, int W>
__device__ inline uint32_t tile_of(const T* pos, size_t p, const TileGeom& g) {
    long long i0;
    double w[W];
    Win<W>::eval((double)pos[3 * p + 0] * g.inv_dx, i0, w);
    int bx = wrapi(i0 + Win<W>::LO, g.n) - g.x_start;
    if (bx < 0) bx += g.n;
    if (bx >= g.nx_alloc) return 0xffffffffu;
    Win<W>::eval((double)pos[3 * p + 1] * g.inv_dx, i0, w);
    int by = wrapi(i0 + Win<W>::LO, g.n);
    Win<W>::eval((double)pos[3 * p + 2] * g.inv_dx, i0, w);
    int bz = wrapi(i0 + Win<W>::LO, g.n);
    return (uint32_t)(((bx / TX) * g.nty + by / TY) * g.ntz + bz / TZ);
}

// run entry: tile (25 bits) | length (7 bits, 1..64 stored as len-1... 6 bits + spare) | first particle (32 bits)
__device__ inline uint64_t pack_run(uint32_t tile, uint32_t len, uint32_t start) {
    return ((uint64_t)tile << 39) | ((uint64_t)(len - 1) << 32) | start;
}

template <typename T, int W>
__global__ void __launch_bounds__(256)
run_scan_kernel(const T* __restrict__ pos, size_t np, TileGeom g, uint64_t* __restrict__ runs,
                unsigned long long* __restrict__ nruns, uint32_t* __restrict__ tile_count,
                unsigned long long* dropped) {
    const int lane = threadIdx.x & 63;
    const size_t nchunks = (np + 255) / 256;
    for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const size_t p = chunk * 256 + threadIdx.x;
        const bool valid = p < np;
        uint32_t key = valid ? tile_of<T, W>(pos, p, g) : 0xffffffffu;
        const bool live = key != 0xffffffffu;
        const uint32_t prev = __shfl_up(key, 1, 64);
        const bool head = live && (lane == 0 || prev != key);
        // a run ends where the next head starts or at the first dead lane
        const unsigned long long hmask = __ballot(head);
        const unsigned long long dmask = __ballot(!live);
        if (valid && !live && dropped) atomicAdd(dropped, 1ull);
        if (hmask == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(nruns, (unsigned long long)__popcll(hmask));
        base = __shfl(base, 0, 64);
        if (head) {
            const unsigned long long stop = (hmask | dmask) & ~((2ull << lane) - 1ull);  // bits above lane
            const int end = stop ? __ffsll((long long)stop) - 1 : 64;
            const unsigned long long below = hmask & ((1ull << lane) - 1ull);
            runs[base + __popcll(below)] = pack_run(key, (uint32_t)(end - lane), (uint32_t)p);
            atomicAdd(&tile_count[key], 1u);
        }
    }
}

// ---- exclusive scan of tile_count (3 kernels, 1024 items per block) ----
__global__ void __launch_bounds__(256)
scan_blocks_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ block_sums, uint32_t n) {
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < wave; ++i) woff += wsum[i];
    uint32_t ex = woff + inc - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + inc;
}

__global__ void __launch_bounds__(256)
scan_sums_kernel(uint32_t* block_sums, uint32_t nblocks, uint32_t* total_out) {
    // single workgroup, serial over 256-wide strips (nblocks <= 32768)
    __shared__ uint32_t carry, wsum[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = carry;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        if (i < nblocks) block_sums[i] = woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(256)
scan_add_kernel(uint32_t* out, const uint32_t* __restrict__ block_sums, uint32_t n) {
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) out[base + i] += add;
}

__global__ void __launch_bounds__(256)
bucket_runs_kernel(const uint64_t* __restrict__ runs, const unsigned long long* __restrict__ nruns,
                   const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_fill,
                   uint64_t* __restrict__ sorted) {
    const unsigned long long nr = *nruns;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += stride) {
        const uint64_t r = runs[i];
        const uint32_t tile = (uint32_t)(r >> 39);
        const uint32_t slot = tile_off[tile] + atomicAdd(&tile_fill[tile], 1u);
        sorted[slot] = r;
    }
}

template <typename T, int W>
__global__ void __launch_bounds__(256)
tile_deposit_kernel(const T* __restrict__ pos, const T* __restrict__ mass, TileGeom g, double scale,
                    const uint64_t* __restrict__ sorted, const uint32_t* __restrict__ tile_off,
                    const uint32_t* __restrict__ tile_count, T* __restrict__ grid,
                    unsigned long long* dropped) {
    constexpr int LX = TX + W - 1, LY = TY + W - 1, LZ = TZ + W - 1;
    constexpr int LO = Win<W>::LO;
    __shared__ T tile[LX * LY * LZ];
    const uint32_t t = blockIdx.x;
    const uint32_t nr = tile_count[t];
    if (nr == 0) return;                       // uniform for the workgroup
    const uint32_t r0 = tile_off[t];
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) tile[i] = (T)0;
    __syncthreads();

    const int tz = t % g.ntz, ty = (t / g.ntz) % g.nty, tx = t / (g.ntz * g.nty);
    const int ox = tx * TX, oy = ty * TY, oz = tz * TZ;   // owned origin (buffer plane / global y, z)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t r = wave; r < nr; r += 4) {
        const uint64_t run = sorted[r0 + r];
        const uint32_t start = (uint32_t)run;
        const int len = (int)((run >> 32) & 0x7f) + 1;
        if (lane < len) {
            const size_t p = (size_t)start + lane;
            long long ix0, iy0, iz0;
            double wx[W], wy[W], wz[W];
            Win<W>::eval((double)pos[3 * p + 0] * g.inv_dx, ix0, wx);
            Win<W>::eval((double)pos[3 * p + 1] * g.inv_dx, iy0, wy);
            Win<W>::eval((double)pos[3 * p + 2] * g.inv_dx, iz0, wz);
            int bx = wrapi(ix0 + LO, g.n) - g.x_start;
            if (bx < 0) bx += g.n;
            const int lx = bx - ox;                               // 0..TX-1 by construction of the run
            const int ly = wrapi(iy0 + LO, g.n) - oy;
            const int lz = wrapi(iz0 + LO, g.n) - oz;
            const double m = (mass ? (double)mass[p] : 1.0) * scale;
#pragma unroll
            for (int a = 0; a < W; ++a) {
                const double ma = m * wx[a];
#pragma unroll
                for (int b = 0; b < W; ++b) {
                    const double mab = ma * wy[b];
                    T* row = &tile[((lx + a) * LY + (ly + b)) * LZ + lz];
#pragma unroll
                    for (int c = 0; c < W; ++c) atomicAdd(row + c, (T)(mab * wz[c]));
                }
            }
        }
    }
    __syncthreads();

    // flush: LDS cell (a, b, c) is buffer plane ox + a - LO, global (oy + b - LO, oz + c - LO)
    unsigned long long ndrop = 0;
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) {
        const T v = tile[i];
        if (v == (T)0) continue;
        const int c = i % LZ, b = (i / LZ) % LY, a = i / (LZ * LY);
        int px = ox + a - LO;
        if (g.nx_alloc == g.n) px = wrapi(px, g.n);
        else if (px < 0 || px >= g.nx_alloc) { ++ndrop; continue; }
        const int gy = wrapi(oy + b - LO, g.n), gz = wrapi(oz + c - LO, g.n);
        atomicAdd(&grid[((size_t)px * g.n + gy) * g.n + gz], v);
    }
    if (dropped && ndrop) atomicAdd(dropped, ndrop);
}

struct Workspace {
    unsigned long long* nruns;      // [0] run counter, [1] scan total (as u32)
    uint32_t* tile_count;
    uint32_t* tile_off;
    uint32_t* tile_fill;
    uint32_t* block_sums;
    uint64_t* runs;
    uint64_t* sorted;
    size_t bytes;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

Workspace carve(void* base, size_t np, uint32_t ntiles) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = (char*)base + off; off += align256(bytes); return p; };
    w.nruns = (unsigned long long*)take(64);
    w.tile_count = (uint32_t*)take((size_t)ntiles * 4);
    w.tile_fill = (uint32_t*)take((size_t)ntiles * 4);
    w.tile_off = (uint32_t*)take((size_t)ntiles * 4);
    w.block_sums = (uint32_t*)take((size_t)((ntiles + 1023) / 1024 + 1) * 4);
    w.runs = (uint64_t*)take(np * 8);
    w.sorted = (uint64_t*)take(np * 8);
    w.bytes = off;
    return w;
}

bool tiled_geometry(int nmesh, int nx_alloc, TileGeom& g, uint32_t& ntiles) {
    if (nmesh % TY || nmesh % TZ) return false;
    g.n = nmesh;
    g.nx_alloc = nx_alloc;
    g.ntx = (nx_alloc + TX - 1) / TX;
    g.nty = nmesh / TY;
    g.ntz = nmesh / TZ;
    const unsigned long long nt = (unsigned long long)g.ntx * g.nty * g.ntz;
    if (nt >= (1ull << 25)) return false;
    ntiles = (uint32_t)nt;
    return true;
}

template <typename T, int W>
int run_tiled(const T* pos, const T* mass, size_t np, TileGeom g, uint32_t ntiles, double scale, T* grid,
              void* workspace, unsigned long long* dropped, hipStream_t s) {
    Workspace w = carve(workspace, np, ntiles);
    // counters, tile_count, tile_fill are contiguous at the front of the workspace
    const size_t zero_bytes = (size_t)((char*)w.tile_off - (char*)w.nruns);
    AST_CHECK_HIP(hipMemsetAsync(w.nruns, 0, zero_bytes, s));
    const size_t nchunks = (np + 255) / 256;
    unsigned ga = (unsigned)(nchunks > 8192 ? 8192 : nchunks);
    run_scan_kernel<T, W><<<ga, 256, 0, s>>>(pos, np, g, w.runs, w.nruns, w.tile_count, dropped);
    const uint32_t nblk = (ntiles + 1023) / 1024;
    scan_blocks_kernel<<<nblk, 256, 0, s>>>(w.tile_count, w.tile_off, w.block_sums, ntiles);
    scan_sums_kernel<<<1, 256, 0, s>>>(w.block_sums, nblk, (uint32_t*)(w.nruns + 1));
    scan_add_kernel<<<nblk, 256, 0, s>>>(w.tile_off, w.block_sums, ntiles);
    bucket_runs_kernel<<<2048, 256, 0, s>>>(w.runs, w.nruns, w.tile_off, w.tile_fill, w.sorted);
    tile_deposit_kernel<T, W><<<ntiles, 256, 0, s>>>(pos, mass, g, scale, w.sorted, w.tile_off, w.tile_count, grid, dropped);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

}  // namespace

extern "C" size_t ast_paint_tiled_workspace_bytes(size_t np, int nmesh, int nx_alloc) {
    TileGeom g;
    uint32_t ntiles = 0;
    if (nmesh <= 0 || nx_alloc <= 0 || !tiled_geometry(nmesh, nx_alloc, g, ntiles)) return 0;
    return carve(nullptr, np, ntiles).bytes;
}

extern "C" int ast_paint_tiled(int window, int dtype, const void* pos, const void* mass, size_t np, int nmesh,
                               double boxsize, double scale, int x_start, int nx_alloc, void* grid,
                               void* workspace, size_t workspace_bytes, unsigned long long* dropped,
                               void* stream) {
    AST_CHECK_ARG(window == AST_WIN_CIC || window == AST_WIN_TSC);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(nmesh > 0 && boxsize > 0.0);
    AST_CHECK_ARG(x_start >= 0 && x_start < nmesh && nx_alloc > 0 && nx_alloc <= nmesh);
    AST_CHECK_ARG(grid != nullptr);
    AST_CHECK_ARG(np < 0xffffffffull);
    if (np == 0) return AST_OK;
    AST_CHECK_ARG(pos != nullptr && workspace != nullptr);
    TileGeom g;
    uint32_t ntiles = 0;
    if (!tiled_geometry(nmesh, nx_alloc, g, ntiles)) {
        ast::set_error("ast_paint_tiled: nmesh must be a multiple of %d with fewer than 2^25 tiles", TZ);
        return AST_ERR_ARG;
    }
    g.x_start = x_start;
    g.inv_dx = (double)nmesh / boxsize;
    if (workspace_bytes < carve(nullptr, np, ntiles).bytes) {
        ast::set_error("ast_paint_tiled: workspace too small (%zu < %zu bytes)", workspace_bytes,
                       carve(nullptr, np, ntiles).bytes);
        return AST_ERR_WORKSPACE;
    }
    hipStream_t s = ast::as_stream(stream);
    if (dtype == AST_F32) {
        if (window == AST_WIN_CIC)
            return run_tiled<float, 2>((const float*)pos, (const float*)mass, np, g, ntiles, scale, (float*)grid, workspace, dropped, s);
        return run_tiled<float, 3>((const float*)pos, (const float*)mass, np, g, ntiles, scale, (float*)grid, workspace, dropped, s);
    }
    if (window == AST_WIN_CIC)
        return run_tiled<double, 2>((const double*)pos, (const double*)mass, np, g, ntiles, scale, (double*)grid, workspace, dropped, s);
    return run_tiled<double, 3>((const double*)pos, (const double*)mass, np, g, ntiles, scale, (double*)grid, workspace, dropped, s);
}
