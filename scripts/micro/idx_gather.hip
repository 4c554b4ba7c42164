// Microbenchmark: index load -> dependent 12-byte gather (runs of 32 consecutive particles), as the
// tiled deposit does, at different occupancies and loads in flight per thread (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int U>
__global__ void __launch_bounds__(256) k(const float* __restrict__ pos, const uint32_t* __restrict__ index, size_t np,
                                         float* out) {
    extern __shared__ float pad[];
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t s = (size_t)blockIdx.x * 256 * U + threadIdx.x; s < np; s += stride) {
        uint32_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = index[s + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t p = idx[u];
            acc += pos[3 * p] + pos[3 * p + 1] + pos[3 * p + 2];
        }
    }
    if (acc == 1234.5f) { out[0] = acc; pad[threadIdx.x] = acc; }
}

// each workgroup walks its own contiguous chunk of the index (like one tile column of the deposit)
template <int U>
__global__ void __launch_bounds__(256) kchunk(const float* __restrict__ pos, const uint32_t* __restrict__ index, size_t np,
                                              size_t chunk, size_t skew, float* out) {
    extern __shared__ float pad[];
    float acc = 0.f;
    const size_t c0 = (size_t)blockIdx.x * chunk;
    const size_t start = skew ? ((size_t)blockIdx.x * 2654435761u >> 7) % (chunk / (256 * U)) * (256 * U) : 0;
    for (size_t o = 0; o < chunk; o += 256 * U) {
        size_t s = start + o;
        if (s >= chunk) s -= chunk;
        s += c0 + threadIdx.x;
        uint32_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = index[s + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t p = idx[u];
            acc += pos[3 * p] + pos[3 * p + 1] + pos[3 * p + 2];
        }
    }
    if (acc == 1234.5f) { out[0] = acc; pad[threadIdx.x] = acc; }
}

// the same walk with the deposit kernel's loop skeleton: per-iteration scalar load of a tile count,
// positions prefetched one batch ahead and indices two ahead
template <int U, bool SCALAR, bool PIPE>
__global__ void __launch_bounds__(256) kwalk(const float* __restrict__ pos, const uint32_t* __restrict__ index,
                                             const uint32_t* __restrict__ counts, size_t chunk, float* out) {
    extern __shared__ float pad[];
    float acc = 0.f;
    const size_t c0 = (size_t)blockIdx.x * chunk;
    const int nit = (int)(chunk / (256 * U));
    uint32_t ia[U], ib[U];
    float pa[3 * U], pb[3 * U];
    auto ldi = [&](int it, uint32_t (&idx)[U]) {
        const size_t s = c0 + (size_t)min(it, nit - 1) * 256 * U + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = index[s + u * 256];
    };
    auto ldp = [&](const uint32_t (&idx)[U], float (&p)[3 * U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t q = (size_t)idx[u] * 3; p[3 * u] = pos[q]; p[3 * u + 1] = pos[q + 1]; p[3 * u + 2] = pos[q + 2]; }
    };
    auto use = [&](const float (&p)[3 * U]) {
#pragma unroll
        for (int u = 0; u < 3 * U; ++u) acc += p[u];
    };
    if (PIPE) {
        ldi(0, ib); ldi(1, ia); ldp(ib, pa);
        for (int it = 0; it < nit; it += 2) {
            if (SCALAR) { const uint32_t c = counts[blockIdx.x * 64 + (it & 63)]; if (c == 0x12345u) break; }
            ldp(ia, pb); ldi(it + 2, ib); use(pa);
            if (SCALAR) { const uint32_t c = counts[blockIdx.x * 64 + ((it + 1) & 63)]; if (c == 0x12345u) break; }
            ldp(ib, pa); ldi(it + 3, ia); use(pb);
        }
    } else {
        for (int it = 0; it < nit; ++it) {
            if (SCALAR) { const uint32_t c = counts[blockIdx.x * 64 + (it & 63)]; if (c == 0x12345u) break; }
            ldi(it, ia); ldp(ia, pa); use(pa);
        }
    }
    if (acc == 1234.5f) { out[0] = acc; pad[threadIdx.x] = acc; }
}

template <int U, bool SCALAR, bool PIPE>
void runwalk(const float* pos, const uint32_t* index, const uint32_t* counts, size_t np, float* out, const char* what) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    const int nwg = 16384;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        kwalk<U, SCALAR, PIPE><<<nwg, 256, 24 * 1024>>>(pos, index, counts, np / nwg, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    printf("U=%d walk %-40s %7.3f ms  %6.2f TB/s\n", U, what, ms, np * 16.0 / ms / 1e9);
}

template <int U>
void runchunk(const float* pos, const uint32_t* index, size_t np, float* out, int nwg, size_t skew, const char* what) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        kchunk<U><<<nwg, 256, 24 * 1024>>>(pos, index, np, np / nwg, skew, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    printf("U=%d chunk walk, %6d WGs %-22s %7.3f ms  %6.2f TB/s\n", U, nwg, what, ms, np * 16.0 / ms / 1e9);
}

__global__ void fill_index(uint32_t* index, size_t np, unsigned long long nruns, unsigned long long mult) {
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < np; s += (size_t)gridDim.x * 256) {
        const unsigned long long run = s / 32;
        index[s] = (uint32_t)(((run * mult) % nruns) * 32 + s % 32);
    }
}

// tile-major order of a 1024^3 lattice stored z-fastest (8 x 8 x 32 tiles, columns of 32 tiles): the
// deposit's real index for lattice-ordered input
__global__ void fill_index_lattice(uint32_t* index, size_t np) {
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < np; s += (size_t)gridDim.x * 256) {
        const unsigned j = s % 2048, tz = (s / 2048) % 32, col = s / 65536, ty = col % 128, tx = col / 128;
        const unsigned ix = 8 * tx + j / 256, iy = 8 * ty + (j / 32) % 8, iz = 32 * tz + j % 32;
        index[s] = (ix * 1024u + iy) * 1024u + iz;
    }
}

template <int U>
void run(const float* pos, const uint32_t* index, size_t np, float* out, int lds_bytes, const char* what) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        k<U><<<256 * 16, 256, lds_bytes>>>(pos, index, np, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    printf("U=%d %-28s %7.3f ms  %6.2f TB/s (16 B / particle)\n", U, what, ms, np * 16.0 / ms / 1e9);
}

int main() {
    const size_t np = (size_t)1 << 30;
    float *pos, *out;
    uint32_t* index;
    hipMalloc(&pos, np * 12);
    hipMalloc(&index, np * 4);
    hipMalloc(&out, 4);
    hipMemset(pos, 0, np * 12);
    fill_index<<<4096, 256>>>(index, np, np / 32, 2654435761ull | 1ull);
    hipDeviceSynchronize();
    run<1>(pos, index, np, out, 0, "8 WG/CU");
    run<4>(pos, index, np, out, 0, "8 WG/CU");
    run<8>(pos, index, np, out, 0, "8 WG/CU");
    run<4>(pos, index, np, out, 36 * 1024, "4 WG/CU (LDS-limited)");
    run<4>(pos, index, np, out, 70 * 1024, "2 WG/CU (LDS-limited)");
    run<8>(pos, index, np, out, 70 * 1024, "2 WG/CU (LDS-limited)");
    run<4>(pos, index, np, out, 150 * 1024, "1 WG/CU (LDS-limited)");
    runchunk<4>(pos, index, np, out, 16384, 0, "lockstep");
    runchunk<4>(pos, index, np, out, 16384, 1, "skewed start");
    runchunk<4>(pos, index, np, out, 1536, 0, "lockstep");
    runchunk<4>(pos, index, np, out, 1536, 1, "skewed start");
    fill_index_lattice<<<4096, 256>>>(index, np);
    hipDeviceSynchronize();
    runchunk<4>(pos, index, np, out, 16384, 0, "LATTICE lockstep");
    runchunk<4>(pos, index, np, out, 16384, 1, "LATTICE skewed");
    run<4>(pos, index, np, out, 0, "LATTICE grid-stride 8 WG/CU");
    uint32_t* counts;
    hipMalloc(&counts, 16384 * 64 * 4);
    hipMemset(counts, 0, 16384 * 64 * 4);
    runwalk<4, false, false>(pos, index, counts, np, out, "plain");
    runwalk<4, true, false>(pos, index, counts, np, out, "plain + scalar load/iter");
    runwalk<4, false, true>(pos, index, counts, np, out, "pipelined");
    runwalk<4, true, true>(pos, index, counts, np, out, "pipelined + scalar load/iter");
    return 0;
}
