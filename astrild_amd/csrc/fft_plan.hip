// a-4: rocFFT plans behind the C-ABI (3D/2D R2C, C2R, strided 1-D C2C).
// rocFFT wants lengths fastest-axis-first; the C-ABI takes numpy-style shapes.
#include "ast_common.h"
#include <rocfft/rocfft.h>
#include <mutex>
#include <vector>

struct ast_fft_plan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
    bool inplace = false;
};

namespace {
std::once_flag g_setup_once;
void ensure_setup() {
    std::call_once(g_setup_once, [] { rocfft_setup(); });
}

#define AST_CHECK_FFT(expr)                                                  \
    do {                                                                     \
        rocfft_status st_ = (expr);                                          \
        if (st_ != rocfft_status_success) {                                  \
            ast::set_error("%s: %s -> rocfft_status %d", __func__, #expr, (int)st_); \
            return AST_ERR_ROCFFT;                                           \
        }                                                                    \
    } while (0)

int finish_plan(ast_fft_plan* p) {
    AST_CHECK_FFT(rocfft_execution_info_create(&p->info));
    AST_CHECK_FFT(rocfft_plan_get_work_buffer_size(p->plan, &p->work_bytes));
    if (p->work_bytes) {
        AST_CHECK_HIP(hipMalloc(&p->work, p->work_bytes));
        AST_CHECK_FFT(rocfft_execution_info_set_work_buffer(p->info, p->work, p->work_bytes));
    }
    return AST_OK;
}

rocfft_transform_type kind_to_type(int kind) {
    switch (kind) {
        case AST_FFT_R2C: return rocfft_transform_type_real_forward;
        case AST_FFT_C2R: return rocfft_transform_type_real_inverse;
        case AST_FFT_C2C_FWD: return rocfft_transform_type_complex_forward;
        default: return rocfft_transform_type_complex_inverse;
    }
}
}  // namespace

extern "C" int ast_fft_plan_create(ast_fft_plan** out, int kind, int dtype, int rank, const size_t* lengths,
                                   size_t batch, double scale, int inplace) {
    AST_CHECK_ARG(out != nullptr && lengths != nullptr);
    AST_CHECK_ARG(kind >= AST_FFT_R2C && kind <= AST_FFT_C2C_INV);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(rank >= 1 && rank <= 3 && batch >= 1);
    for (int i = 0; i < rank; ++i) AST_CHECK_ARG(lengths[i] >= 1);
    ensure_setup();
    size_t rl[3];
    for (int i = 0; i < rank; ++i) rl[i] = lengths[rank - 1 - i];
    auto* p = new ast_fft_plan();
    p->inplace = inplace != 0;
    rocfft_plan_description desc = nullptr;
    AST_CHECK_FFT(rocfft_plan_description_create(&desc));
    if (scale != 1.0) AST_CHECK_FFT(rocfft_plan_description_set_scale_factor(desc, scale));
    if (inplace && (kind == AST_FFT_R2C || kind == AST_FFT_C2R)) {
        // in-place real transforms use rows padded to 2*(n/2+1) reals
        size_t nh = rl[0] / 2 + 1;
        size_t rs[3] = {1, 2 * nh, 2 * nh * (rank > 1 ? rl[1] : 1)};
        size_t cs[3] = {1, nh, nh * (rank > 1 ? rl[1] : 1)};
        size_t rdist = 2 * nh, cdist = nh;
        for (int i = 1; i < rank; ++i) { rdist *= rl[i]; cdist *= rl[i]; }
        if (kind == AST_FFT_R2C)
            AST_CHECK_FFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real,
                          rocfft_array_type_hermitian_interleaved, nullptr, nullptr, rank, rs, rdist, rank, cs, cdist));
        else
            AST_CHECK_FFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_hermitian_interleaved,
                          rocfft_array_type_real, nullptr, nullptr, rank, cs, cdist, rank, rs, rdist));
    }
    rocfft_status st = rocfft_plan_create(&p->plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                                          kind_to_type(kind),
                                          dtype == AST_F32 ? rocfft_precision_single : rocfft_precision_double,
                                          (size_t)rank, rl, batch, desc);
    rocfft_plan_description_destroy(desc);
    if (st != rocfft_status_success) {
        delete p;
        ast::set_error("ast_fft_plan_create: rocfft_plan_create -> rocfft_status %d", (int)st);
        return AST_ERR_ROCFFT;
    }
    int rc = finish_plan(p);
    if (rc != AST_OK) { ast_fft_plan_destroy(p); return rc; }
    *out = p;
    return AST_OK;
}

extern "C" int ast_fft_plan_create_strided_1d(ast_fft_plan** out, int kind, int dtype, size_t length,
                                              size_t stride, size_t batch, size_t dist, double scale) {
    AST_CHECK_ARG(out != nullptr);
    AST_CHECK_ARG(kind == AST_FFT_C2C_FWD || kind == AST_FFT_C2C_INV);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(length >= 1 && stride >= 1 && batch >= 1 && dist >= 1);
    ensure_setup();
    auto* p = new ast_fft_plan();
    p->inplace = true;
    rocfft_plan_description desc = nullptr;
    AST_CHECK_FFT(rocfft_plan_description_create(&desc));
    if (scale != 1.0) AST_CHECK_FFT(rocfft_plan_description_set_scale_factor(desc, scale));
    size_t st1[1] = {stride};
    AST_CHECK_FFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved,
                  rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, st1, dist, 1, st1, dist));
    size_t len1[1] = {length};
    rocfft_status st = rocfft_plan_create(&p->plan, rocfft_placement_inplace, kind_to_type(kind),
                                          dtype == AST_F32 ? rocfft_precision_single : rocfft_precision_double,
                                          1, len1, batch, desc);
    rocfft_plan_description_destroy(desc);
    if (st != rocfft_status_success) {
        delete p;
        ast::set_error("ast_fft_plan_create_strided_1d: rocfft_plan_create -> rocfft_status %d", (int)st);
        return AST_ERR_ROCFFT;
    }
    int rc = finish_plan(p);
    if (rc != AST_OK) { ast_fft_plan_destroy(p); return rc; }
    *out = p;
    return AST_OK;
}

extern "C" int ast_fft_plan_create_general(ast_fft_plan** out, int kind, int dtype, int rank, const size_t* lengths,
                                           const size_t* in_strides, const size_t* out_strides, size_t batch,
                                           size_t in_dist, size_t out_dist, double scale, int inplace) {
    AST_CHECK_ARG(out != nullptr && lengths != nullptr && in_strides != nullptr && out_strides != nullptr);
    AST_CHECK_ARG(kind >= AST_FFT_R2C && kind <= AST_FFT_C2C_INV);
    AST_CHECK_ARG(dtype == AST_F32 || dtype == AST_F64);
    AST_CHECK_ARG(rank >= 1 && rank <= 3 && batch >= 1);
    ensure_setup();
    size_t rl[3], ris[3], ros[3];
    for (int i = 0; i < rank; ++i) {
        rl[i] = lengths[rank - 1 - i];
        ris[i] = in_strides[rank - 1 - i];
        ros[i] = out_strides[rank - 1 - i];
    }
    auto* p = new ast_fft_plan();
    p->inplace = inplace != 0;
    rocfft_plan_description desc = nullptr;
    AST_CHECK_FFT(rocfft_plan_description_create(&desc));
    if (scale != 1.0) AST_CHECK_FFT(rocfft_plan_description_set_scale_factor(desc, scale));
    rocfft_array_type it = rocfft_array_type_complex_interleaved, ot = rocfft_array_type_complex_interleaved;
    if (kind == AST_FFT_R2C) { it = rocfft_array_type_real; ot = rocfft_array_type_hermitian_interleaved; }
    if (kind == AST_FFT_C2R) { it = rocfft_array_type_hermitian_interleaved; ot = rocfft_array_type_real; }
    AST_CHECK_FFT(rocfft_plan_description_set_data_layout(desc, it, ot, nullptr, nullptr, rank, ris, in_dist,
                                                          rank, ros, out_dist));
    rocfft_status st = rocfft_plan_create(&p->plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                                          kind_to_type(kind),
                                          dtype == AST_F32 ? rocfft_precision_single : rocfft_precision_double,
                                          (size_t)rank, rl, batch, desc);
    rocfft_plan_description_destroy(desc);
    if (st != rocfft_status_success) {
        delete p;
        ast::set_error("ast_fft_plan_create_general: rocfft_plan_create -> rocfft_status %d", (int)st);
        return AST_ERR_ROCFFT;
    }
    int rc = finish_plan(p);
    if (rc != AST_OK) { ast_fft_plan_destroy(p); return rc; }
    *out = p;
    return AST_OK;
}

extern "C" size_t ast_fft_plan_work_bytes(const ast_fft_plan* plan) { return plan ? plan->work_bytes : 0; }

extern "C" int ast_fft_exec(ast_fft_plan* plan, void* in, void* out, void* stream) {
    AST_CHECK_ARG(plan != nullptr && in != nullptr);
    AST_CHECK_ARG(plan->inplace || out != nullptr);
    AST_CHECK_FFT(rocfft_execution_info_set_stream(plan->info, stream));
    void* ib[1] = {in};
    void* ob[1] = {out};
    AST_PROF("rocfft_execute", ast::as_stream(stream));
    AST_CHECK_FFT(rocfft_execute(plan->plan, ib, plan->inplace ? nullptr : ob, plan->info));
    return AST_OK;
}

extern "C" int ast_fft_plan_destroy(ast_fft_plan* plan) {
    if (!plan) return AST_OK;
    if (plan->info) rocfft_execution_info_destroy(plan->info);
    if (plan->plan) rocfft_plan_destroy(plan->plan);
    if (plan->work) (void)hipFree(plan->work);
    delete plan;
    return AST_OK;
}
