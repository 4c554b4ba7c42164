// SURVEY.md §8(b)'s communication entries of the C-ABI: a binding of libastrild_hip.so ALONE (no torch) gets the slab
// transpose and the shell-sum reduction over RCCL / xGMI.  The Python side of this repository reaches RCCL through
// torch.distributed ("nccl" is RCCL on ROCm) and never calls these; they take the same buffers in the same layouts
// (ast_fft_tile_c2c_disc's send buffer, the receive block of ast_fft_tile_disc_block_power).
//
// RCCL is loaded lazily (dlopen of librccl.so.1 at the first ast_comm_* call): the library keeps no link-time dependency
// on it, and a process that already holds an RCCL (torch's) gets that one.
#include "ast_common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

const Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.h) return &g_rccl;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { ast::set_error("ast_comm: cannot load librccl.so.1: %s", dlerror()); return nullptr; }
    Rccl r;
    r.h = h;
#define AST_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { ast::set_error("ast_comm: %s missing in librccl", name); return nullptr; }
    AST_SYM(GetUniqueId, "ncclGetUniqueId")
    AST_SYM(CommInitRank, "ncclCommInitRank")
    AST_SYM(CommDestroy, "ncclCommDestroy")
    AST_SYM(GroupStart, "ncclGroupStart")
    AST_SYM(GroupEnd, "ncclGroupEnd")
    AST_SYM(Send, "ncclSend")
    AST_SYM(Recv, "ncclRecv")
    AST_SYM(AllReduce, "ncclAllReduce")
    AST_SYM(GetErrorString, "ncclGetErrorString")
#undef AST_SYM
    g_rccl = r;
    return &g_rccl;
}
}  // namespace

struct ast_comm { ncclComm_t comm; int nranks, rank; };

#define AST_CHECK_NCCL(R, expr)                                                                   \
    do {                                                                                          \
        ncclResult_t e_ = (expr);                                                                 \
        if (e_ != ncclSuccess) {                                                                  \
            ast::set_error("%s: %s -> %s", __func__, #expr, (R)->GetErrorString(e_));             \
            return AST_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

extern "C" int ast_comm_unique_id(void* id_out, size_t id_bytes) {
    AST_CHECK_ARG(id_out != nullptr && id_bytes >= sizeof(ncclUniqueId));
    const Rccl* R = rccl();
    if (!R) return AST_ERR_HIP;
    AST_CHECK_NCCL(R, R->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
    return AST_OK;
}

extern "C" int ast_comm_init(ast_comm** out, int nranks, int rank, const void* id, size_t id_bytes) {
    AST_CHECK_ARG(out != nullptr && nranks >= 1 && rank >= 0 && rank < nranks && id != nullptr && id_bytes >= sizeof(ncclUniqueId));
    const Rccl* R = rccl();
    if (!R) return AST_ERR_HIP;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    AST_CHECK_NCCL(R, R->CommInitRank(&c, nranks, uid, rank));
    *out = new ast_comm{c, nranks, rank};
    return AST_OK;
}

extern "C" int ast_comm_destroy(ast_comm* c) {
    if (!c) return AST_OK;
    const Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return AST_OK;
}

// The slab transpose as ONE group of point-to-point operations (xGMI is point to point: all links of a GPU carry their
// piece at once): for every peer q != rank, send_count[q] elements of `dtype` (AST_F32 / AST_F64 REAL elements - a
// complex value is two) from send_d + send_offset[q] and recv_count[q] into recv_d + recv_offset[q] (offsets in
// elements).  The rank's own piece never travels (ast_fft_tile_c2c_disc writes it into the receive block itself); a
// non-zero count at q == rank is copied on the stream.  Asynchronous on `stream`.
extern "C" int ast_slab_transpose(ast_comm* c, const void* send, const size_t* send_offset, const size_t* send_count, void* recv,
                                  const size_t* recv_offset, const size_t* recv_count, int dtype, void* stream) {
    AST_CHECK_ARG(c != nullptr && send_offset && send_count && recv_offset && recv_count && (dtype == AST_F32 || dtype == AST_F64));
    const Rccl* R = rccl();
    if (!R) return AST_ERR_HIP;
    hipStream_t s = ast::as_stream(stream);
    const size_t esz = dtype == AST_F32 ? 4 : 8;
    const ncclDataType_t dt = dtype == AST_F32 ? ncclFloat32 : ncclFloat64;
    AST_CHECK_NCCL(R, R->GroupStart());
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank) continue;
        if (send_count[q]) AST_CHECK_NCCL(R, R->Send((const char*)send + send_offset[q] * esz, send_count[q], dt, q, c->comm, s));
        if (recv_count[q]) AST_CHECK_NCCL(R, R->Recv((char*)recv + recv_offset[q] * esz, recv_count[q], dt, q, c->comm, s));
    }
    AST_CHECK_NCCL(R, R->GroupEnd());
    const int me = c->rank;
    if (send_count[me] && recv_count[me]) {
        AST_CHECK_ARG(send_count[me] == recv_count[me] && send != nullptr && recv != nullptr);
        AST_CHECK_HIP(hipMemcpyAsync((char*)recv + recv_offset[me] * esz, (const char*)send + send_offset[me] * esz,
                                     send_count[me] * esz, hipMemcpyDeviceToDevice, s));
    }
    return AST_OK;
}

// In-place sum over the ranks of `count` float64 values (the shell sums: N/2 - 1 of them; the low-k modes).
extern "C" int ast_comm_allreduce_sum(ast_comm* c, double* buf, size_t count, void* stream) {
    AST_CHECK_ARG(c != nullptr && (buf != nullptr || count == 0));
    const Rccl* R = rccl();
    if (!R) return AST_ERR_HIP;
    if (count == 0) return AST_OK;
    AST_CHECK_NCCL(R, R->AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, ast::as_stream(stream)));
    return AST_OK;
}
