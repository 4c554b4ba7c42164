// Microbenchmark: the column walk's LDS accumulation pattern (CIC: 8 adds per particle into a 9 x 9 x 33 tile; a half wave
// = 32 particles consecutive in z, each jittered into one of the neighbouring (x, y) rows) by cell format (gfx950):
//   V0  64-bit fixed point, ds_add_u64 (what the walk does)
//   V1  32-bit cells, ds_add_u32 (no carry handling: the ceiling of a lo/hi split)
//   V2  32-bit lo plane with RETURNING adds + carry test, rare ds_add_u32 on a hi plane (a correct lo/hi split)
//   V3  V0 with the z pitch padded 33 -> 48 cells (rows half the banks apart)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int LX = 9, LY = 9;

template <int V, int LZ>
__global__ void __launch_bounds__(256) k(unsigned long long* out, int iters, unsigned seed) {
    __shared__ unsigned long long tile64[V == 0 || V == 3 ? LX * LY * LZ : 1];
    __shared__ unsigned lo[V == 1 || V == 2 ? LX * LY * LZ : 1], hi[V == 2 ? LX * LY * LZ : 1];
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) {
        if (V == 0 || V == 3) tile64[i] = 0;
        if (V == 1 || V == 2) lo[i] = 0x80000000u;
        if (V == 2) hi[i] = 0;
    }
    __syncthreads();
    unsigned s = seed ^ (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 31;
    unsigned long long chk = 0;
    for (int it = 0; it < iters; ++it) {
        // base cell: the window's row (it-dependent) +- jitter in x and y (32 % each way out of the row), z = lane +- jitter
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        int lx = 3 + (int)(it & 3), ly = 3 + (int)((it >> 2) & 3);
        const unsigned jx = r & 63, jy = (r >> 6) & 63, jz = (r >> 12) & 63;
        lx += jx < 10 ? -1 : jx < 20 ? 1 : 0;
        ly += jy < 10 ? -1 : jy < 20 ? 1 : 0;
        int lz = lane + (jz < 10 ? -1 : jz < 20 ? 1 : 0);
        lz = lz < 0 ? 0 : lz > 31 ? 31 : lz;
        const unsigned term = 1000u + (r & 1023u);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int cell = ((lx + a) * LY + (ly + b)) * LZ + lz + c;
                    if (V == 0 || V == 3) atomicAdd(&tile64[cell], (unsigned long long)term);
                    else if (V == 1) atomicAdd(&lo[cell], term);
                    else {
                        const unsigned old = atomicAdd(&lo[cell], term);
                        if (old + term < old) atomicAdd(&hi[cell], 1u);        // carry: rare
                    }
                }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) chk += (V == 0 || V == 3) ? tile64[i] : (unsigned long long)lo[i] + (V == 2 ? hi[i] : 0u);
    if (chk == 12345) out[blockIdx.x] = chk;
}

template <int V, int LZ> void run(const char* name) {
    unsigned long long* out;
    hipMalloc(&out, 4096 * 8);
    const int blocks = 256 * 6, iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<V, LZ><<<blocks, 256>>>(out, 16, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<V, LZ><<<blocks, 256>>>(out, iters, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double parts = (double)blocks * 256 * iters;
    printf("%-44s %8.3f ms  %7.2f G particles/s (8 adds each)\n", name, ms, parts / ms / 1e6);
    hipFree(out);
}

int main() {
    run<0, 33>("V0 u64 cells, pitch 33");
    run<3, 48>("V3 u64 cells, pitch 48");
    run<3, 34>("V3 u64 cells, pitch 34");
    run<1, 33>("V1 u32 cells, no carry (ceiling)");
    run<2, 33>("V2 u32 lo (returning) + carry -> hi");
    run<1, 34>("V1 u32 cells, pitch 34");
    return 0;
}
