// Shared plumbing for libastrild_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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
