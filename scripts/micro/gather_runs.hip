// Microbenchmark: bandwidth of gathering 12-byte particles in contiguous runs of R particles whose
// start positions are scattered over a 12.9 GB array (the tiled deposit's access pattern for
// lattice-ordered input: R = tile extent in z), vs a plain stream (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// each wave handles 64 consecutive "slots"; slot s -> particle perm(s / R) * R + s % R
__global__ void __launch_bounds__(256) k(const float* __restrict__ pos, size_t np, int R, unsigned long long nruns,
                                         unsigned long long mult, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < np; s += stride) {
        const unsigned long long run = s / R;
        const unsigned long long prun = (run * mult) % nruns;           // mult coprime to nruns: a permutation
        const size_t p = (size_t)prun * R + s % R;
        acc += pos[3 * p] + pos[3 * p + 1] + pos[3 * p + 2];
    }
    if (acc == 1234.5f) out[0] = acc;
}

int main() {
    const size_t np = (size_t)1 << 30;
    float *pos, *out;
    hipMalloc(&pos, np * 12);
    hipMalloc(&out, 4);
    hipMemset(pos, 0, np * 12);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int Rs[] = {0, 32, 64, 128, 256, 1024, 4096};
    for (int R : Rs) {
        const int r = R ? R : 1 << 20;
        const unsigned long long nruns = np / r;
        const unsigned long long mult = R ? 2654435761ull % nruns | 1ull : 1ull;   // odd; nruns is a power of two
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            k<<<256 * 24, 256>>>(pos, np, r, nruns, mult, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("run length %7d particles (%8d B): %7.3f ms  %6.2f TB/s\n", R ? R : 0, r * 12, ms, np * 12.0 / ms / 1e9);
    }
    return 0;
}
