// Microbenchmark: LDS atomic add throughput by type and address pattern (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

template <typename T, int PATTERN>
__global__ void __launch_bounds__(256) k(T* out, int iters, unsigned seed) {
    __shared__ T tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = (T)0;
    __syncthreads();
    unsigned s = seed ^ (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        int a;
        if (PATTERN == 0) a = (threadIdx.x + it * 33) & 4095;            // lane-distinct, consecutive banks
        else if (PATTERN == 1) { s = s * 1664525u + 1013904223u; a = (s >> 10) & 4095; }  // random
        else if (PATTERN == 2) a = ((lane >> 1) + it * 33 + (threadIdx.x >> 6) * 64) & 4095;  // pairs share an address
        else a = (it * 7) & 4095;                                           // all lanes same address
        atomicAdd(&tile[a], (T)1);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[5];
}

template <typename T, int P>
void run(const char* name) {
    T* out;
    hipMalloc(&out, 4096 * sizeof(T));
    const int blocks = 2048, iters = 2048;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<T, P><<<blocks, 256>>>(out, 16, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<T, P><<<blocks, 256>>>(out, iters, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters;
    printf("%-28s %8.3f ms  %8.2f Gatomics/s  %6.2f lanes/clk/CU (2.4GHz,256CU)\n", name, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
    hipFree(out);
}

int main() {
    run<float, 0>("f32 distinct");
    run<float, 1>("f32 random");
    run<float, 2>("f32 pairs-same-addr");
    run<float, 3>("f32 all-same-addr");
    run<unsigned, 0>("u32 distinct");
    run<unsigned, 1>("u32 random");
    run<unsigned, 2>("u32 pairs-same-addr");
    run<unsigned, 3>("u32 all-same-addr");
    run<double, 0>("f64 distinct");
    run<double, 1>("f64 random");
    run<unsigned long long, 0>("u64 distinct");
    run<unsigned long long, 1>("u64 random");
    return 0;
}
