// Microbenchmark: issue cost of the VALU instructions the paint kernels are made of (gfx950).
// Each kernel runs UNROLL independent chains of one instruction per thread, enough waves to fill
// every SIMD; reported: cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X) X(mul_f64) X(add_f64) X(fma_f64) X(floor_f64) X(cvt_i32_f64) X(cvt_f64_i32) X(cvt_f64_f32) X(cvt_f32_f64) \
    X(pk_add_f32) X(pk_mul_f32) X(pk_fma_f32) X(mul_f32) X(fma_f32) X(floor_f32) X(cvt_i32_f32) X(min_u32) X(add_u32) X(mad_u64_u32) X(lshl_add_u64)

enum Op {
#define X(n) OP_##n,
    OPS(X)
#undef X
    OP_COUNT
};

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
    constexpr int U = 8;
    double d[U];
    float f[U];
    int i32[U];
    unsigned long long u64[U];
    for (int u = 0; u < U; ++u) { d[u] = seed + u + threadIdx.x; f[u] = (float)d[u]; i32[u] = (int)d[u]; u64[u] = (unsigned long long)d[u]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (OP == OP_mul_f64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u]) : "v"(seed));
            if (OP == OP_add_f64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[u]) : "v"(seed));
            if (OP == OP_fma_f64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[u]) : "v"(seed));
            if (OP == OP_floor_f64) asm volatile("v_floor_f64 %0, %0" : "+v"(d[u]));
            if (OP == OP_cvt_i32_f64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i32[u]) : "v"(d[u]));
            if (OP == OP_cvt_f64_i32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[u]) : "v"(i32[u]));
            if (OP == OP_cvt_f64_f32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[u]) : "v"(f[u]));
            if (OP == OP_cvt_f32_f64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[u]) : "v"(d[u]));
            if (OP == OP_pk_add_f32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[u]) : "v"(seed));
            if (OP == OP_pk_mul_f32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[u]) : "v"(seed));
            if (OP == OP_pk_fma_f32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[u]) : "v"(seed));
            if (OP == OP_mul_f32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[u]) : "v"((float)seed));
            if (OP == OP_fma_f32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[u]) : "v"((float)seed));
            if (OP == OP_floor_f32) asm volatile("v_floor_f32 %0, %0" : "+v"(f[u]));
            if (OP == OP_cvt_i32_f32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i32[u]) : "v"(f[u]));
            if (OP == OP_min_u32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(i32[u]) : "v"(it));
            if (OP == OP_add_u32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i32[u]) : "v"(it));
            if (OP == OP_mad_u64_u32) asm volatile("v_mad_u64_u32 %0, vcc, %1, 12, %0" : "+v"(u64[u]) : "v"(i32[u]) : "vcc");
            if (OP == OP_lshl_add_u64) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(u64[u]));
        }
    }
    double s = 0;
    for (int u = 0; u < U; ++u) s += d[u] + f[u] + i32[u] + (double)u64[u];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char* name) {
    double* out;
    hipMalloc(&out, 8);
    const int blocks = 256 * 8, iters = 4096;          // 8 workgroups = 32 waves per CU = 8 per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 16, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, iters, 1.0000001);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_simd = (double)blocks * 4 * iters * 8 / (256.0 * 4);
    printf("%-16s %8.3f ms   %6.2f cycles / wave-instruction / SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
    hipFree(out);
}

int main() {
#define X(n) run<OP_##n>(#n);
    OPS(X)
#undef X
    return 0;
}
