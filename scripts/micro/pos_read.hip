// Microbenchmark: how fast can 2^30 AoS float3 positions (12.9 GB) be streamed in, by load shape (gfx950)?
//   A  three dword loads per lane, stride 12 B (what a naive pos[3p+c] compiles to when it is not merged)
//   B  one dwordx3 per lane (the paint's grouping kernel today)
//   C  dwordx4 per lane, fully contiguous (no per-particle view: the ceiling)
//   D  dwordx4 per lane + wave-local LDS transpose to one particle per lane (what a kernel would do to get C's rate)
// Each variant with one 4096-particle interval per workgroup (the grouping kernel's grid) and UNROLL loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ pos, size_t np, float* out) {
    __shared__ float lds[4][3 * 256];          // per wave: 256 particles x 3 floats
    float acc = 0.f;
    const size_t p0 = (size_t)blockIdx.x * 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (MODE == 0 || MODE == 1) {
#pragma unroll
        for (int trip = 0; trip < 4; ++trip) {
            float x[4], y[4], z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t p = p0 + trip * 1024 + u * 256 + tid;
                if (MODE == 0) {
                    x[u] = __builtin_nontemporal_load(pos + 3 * p);
                    y[u] = __builtin_nontemporal_load(pos + 3 * p + 1);
                    z[u] = __builtin_nontemporal_load(pos + 3 * p + 2);
                } else {
                    x[u] = pos[3 * p]; y[u] = pos[3 * p + 1]; z[u] = pos[3 * p + 2];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += x[u] * 1.5f + y[u] * 2.5f + z[u];
        }
    } else if (MODE == 2) {
        const f4* q = reinterpret_cast<const f4*>(pos + 3 * p0);       // 4096 particles = 3072 f4
#pragma unroll
        for (int trip = 0; trip < 3; ++trip) {
            f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = q[(trip * 4 + u) * 256 + tid];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x * 1.5f + v[u].y * 2.5f + v[u].z + v[u].w;
        }
    } else {
        // wave w handles particles p0 + w * 1024 .. + 1024 in four rounds of 256: 3 dwordx4 per lane (contiguous 3 KB),
        // through the wave's own LDS slice, back as x, y, z of particle (round, j * 64 + lane), j < 4
        const f4* q = reinterpret_cast<const f4*>(pos + 3 * (p0 + (size_t)wave * 1024));
        float* mine = lds[wave];
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            f4 v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = q[round * 192 + u * 64 + lane];
#pragma unroll
            for (int u = 0; u < 3; ++u) *reinterpret_cast<f4*>(mine + 4 * (u * 64 + lane)) = v[u];
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the wave's own stores have landed
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pp = j * 64 + lane;
                acc += mine[3 * pp] * 1.5f + mine[3 * pp + 1] * 2.5f + mine[3 * pp + 2];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
    }
    if (acc == 1234.5f) out[0] = acc;
}

template <int MODE> void run(const char* name, const float* pos, size_t np, float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        k<MODE><<<(unsigned)(np / 4096), 256>>>(pos, np, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %5.2f TB/s\n", name, best, np * 12.0 / best / 1e9);
}

int main() {
    const size_t np = (size_t)1 << 30;
    float *pos, *out;
    hipMalloc(&pos, np * 12);
    hipMalloc(&out, 4);
    hipMemset(pos, 0, np * 12);
    run<0>("A three nontemporal dword loads per lane", pos, np, out);
    run<1>("B pos[3p+c] (merged to dwordx3)", pos, np, out);
    run<2>("C dwordx4 contiguous", pos, np, out);
    run<3>("D dwordx4 + wave-local LDS transpose", pos, np, out);
    return 0;
}
