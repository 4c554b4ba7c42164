// Shared plumbing for libastrild_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/astrild_hip.h"

namespace ast {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Memory-bound streaming kernels: cap the grid at 256 CUs x 8 blocks and
// grid-stride the rest (cdna_hip_programming.md Guideline 11).
inline unsigned stream_grid(size_t work_items, unsigned block) {
    size_t need = (work_items + block - 1) / block;
    if (need < 1) need = 1;
    return (unsigned)(need > 2048 ? 2048 : need);
}

// Optional per-launch HIP-event timing (ast_profile_enable): a scope records an
// event pair on the launch stream; ast_profile_report aggregates by name.
struct ProfScope {
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

}  // namespace ast

namespace ast {
// FFTPower's shell membership.  Mathematically shell = floor(|m|) - 1 on integer lattice vectors m; nbodykit
// decides it in float64: digitize(kx^2 + ky^2 + kz^2, kedges^2) with k_i = k_F * m_i and kedges = arange(k_F, ...,
// k_F) (element i = k_F + i * k_F).  The two agree except for vectors whose norm is EXACTLY an integer r
// (perfect-square |m|^2): those sit on an edge and the rounding of the float expressions decides between shell r - 1
// and r - 2.  This restates that arithmetic for such a vector (no fma: the library is built with
// -ffp-contract=off); kf = 0 selects the integer rule.  Returns floor(|m|) or floor(|m|) - 1.
__device__ inline int float64_edge_norm(int r, int mx, int my, int mz, double kf) {
    const double kx = kf * (double)mx, ky = kf * (double)my, kz = kf * (double)mz;
    const double k2 = (kx * kx + ky * ky) + kz * kz;
    const double e = kf + (double)(r - 1) * kf;
    return k2 < e * e ? r - 1 : r;
}
}  // namespace ast

namespace ast {
// "has this process already raised the dynamic-LDS limit of this kernel on the CURRENT device?" - the attribute is
// per device, so the flag is too (one process per GPU is the rule, but nothing forbids a second device).
struct PerDeviceOnce {
    std::atomic<bool> done[64] = {};
    // need(): the attribute calls have to be made (again: they are idempotent, so two racing first callers both
    // making them is harmless); mark() after they have SUCCEEDED.
    bool need() const {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        return !done[dev].load(std::memory_order_acquire);
    }
    void mark() {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    }
};
}  // namespace ast

#define AST_PROF(name, stream) ast::ProfScope ast_prof_scope_(name, stream)

#define AST_CHECK_ARG(cond)                                              \
    do {                                                                 \
        if (!(cond)) {                                                   \
            ast::set_error("%s: bad argument: %s", __func__, #cond);     \
            return AST_ERR_ARG;                                          \
        }                                                                \
    } while (0)

#define AST_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            ast::set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(e_));  \
            return AST_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

#define AST_CHECK_LAUNCH() AST_CHECK_HIP(hipGetLastError())
