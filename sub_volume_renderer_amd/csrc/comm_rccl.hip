// Multi-GPU gather of per-rank frame regions over RCCL (xGMI), behind the C ABI (include/svr.h).
//
// The reference has no multi-GPU path at all.  Here the frame is sharded by pixels, one
// process per GPU, and the only exchange is "every rank's rendered region -> root": one grouped
// ncclSend / ncclRecv per rank and plane (the primitive RCCL's own ncclGather is built from).  Root has a direct
// xGMI link to every peer, so this is one hop per peer; no ring algorithm is involved or wanted.
//
// RCCL is bound at run time (dlopen / dlsym): the process usually already holds a copy (PyTorch ships one and
// torch.distributed's "nccl" backend IS that RCCL), and two different RCCL builds must not be mixed in one
// process; libsvr_hip.so itself therefore has no link-time dependency on librccl and loads on machines without it.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>      // types and prototypes only: every call goes through a dlsym'd pointer of the prototype's own type

#include "svr_internal.h"

static_assert(SVR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "svr.h carries ncclUniqueId as SVR_COMM_ID_BYTES opaque bytes");
static_assert(sizeof(ncclUniqueId) == SVR_COMM_ID_BYTES, "ncclUniqueId layout");

namespace {

struct Rccl {
    void* handle = nullptr;
    // pointer types are taken from rccl.h's own declarations, so a changed signature fails to compile here instead of
    // drifting silently behind dlsym
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclSend)           Send = nullptr;
    decltype(&ncclRecv)           Recv = nullptr;
    decltype(&ncclGroupStart)     GroupStart = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process already holds first (PyTorch's), then the system one
        const char* names[] = { "librccl.so.1", "librccl.so" };
        for (const char* n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            const char* why = dlerror();               // one call: dlerror() clears the message it returns
            r.error = std::string("librccl not found: ") + (why ? why : "no loader message");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    });
    return r;
}

int rccl_fail(const char* what, ncclResult_t e) {
    Rccl& r = rccl();
    svr_set_error(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
    return SVR_ERR_HIP;
}

#define SVR_RCCL_TRY(expr)                                        \
    do {                                                          \
        const ncclResult_t e_ = (expr);                           \
        if (e_ != ncclSuccess) return rccl_fail(#expr, e_);                 \
    } while (0)

}  // namespace

extern "C" {

int svr_comm_unique_id(char out_id[SVR_COMM_ID_BYTES]) {
    if (!out_id) { svr_set_error("svr_comm_unique_id: null argument"); return SVR_ERR_INVALID; }
    Rccl& r = rccl();
    if (!r.error.empty()) { svr_set_error("svr_comm_unique_id: " + r.error); return SVR_ERR_HIP; }
    ncclUniqueId id;
    SVR_RCCL_TRY(r.GetUniqueId(&id));
    memcpy(out_id, id.internal, sizeof(id.internal));
    return SVR_OK;
}

int svr_comm_init(svr_ctx* c, const char id_bytes[SVR_COMM_ID_BYTES], int rank, int nranks) {
    if (!c || !id_bytes) { svr_set_error("svr_comm_init: null argument"); return SVR_ERR_INVALID; }
    if (nranks < 1 || rank < 0 || rank >= nranks) { svr_set_error("svr_comm_init: rank out of range"); return SVR_ERR_INVALID; }
    if (c->comm) { svr_set_error("svr_comm_init: the context already has a communicator"); return SVR_ERR_INVALID; }
    Rccl& r = rccl();
    if (!r.error.empty()) { svr_set_error("svr_comm_init: " + r.error); return SVR_ERR_HIP; }
    int prev = -1;
    SVR_HIP_TRY(hipGetDevice(&prev));
    SVR_HIP_TRY(hipSetDevice(c->device));            // the communicator binds to the context's GPU
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, sizeof(id.internal));
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r.CommInitRank(&comm, nranks, id, rank);
    (void)hipSetDevice(prev);
    if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
    c->comm = comm; c->comm_rank = rank; c->comm_size = nranks;
    return SVR_OK;
}

int svr_comm_destroy(svr_ctx* c) {
    if (!c) { svr_set_error("svr_comm_destroy: null ctx"); return SVR_ERR_INVALID; }
    if (!c->comm) return SVR_OK;
    Rccl& r = rccl();
    ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    c->comm = nullptr;
    SVR_RCCL_TRY(r.CommDestroy(comm));
    return SVR_OK;
}

int svr_gather_tiles(svr_ctx* c, int nplanes, const void* const* local, void* const* gathered,
                     const size_t* bytes_per_rank, int root, void* stream) {
    if (!c || !local || !bytes_per_rank) { svr_set_error("svr_gather_tiles: null argument"); return SVR_ERR_INVALID; }
    if (!c->comm) { svr_set_error("svr_gather_tiles: svr_comm_init has not been called"); return SVR_ERR_INVALID; }
    if (nplanes < 1 || nplanes > 8) { svr_set_error("svr_gather_tiles: 1 to 8 planes per call"); return SVR_ERR_INVALID; }
    if (root < 0 || root >= c->comm_size) { svr_set_error("svr_gather_tiles: root out of range"); return SVR_ERR_INVALID; }
    const bool is_root = c->comm_rank == root;
    if (is_root && !gathered) { svr_set_error("svr_gather_tiles: root needs the gathered buffers"); return SVR_ERR_INVALID; }
    for (int p = 0; p < nplanes; ++p)
        if (!local[p] || (is_root && !gathered[p])) { svr_set_error("svr_gather_tiles: null plane pointer"); return SVR_ERR_INVALID; }
    Rccl& r = rccl();
    ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(c->device);
    // every plane of every peer in ONE group: the transfers progress concurrently, one per xGMI link
    SVR_RCCL_TRY(r.GroupStart());
    ncclResult_t e = ncclSuccess;
    for (int p = 0; p < nplanes && e == ncclSuccess; ++p) {
        const size_t n = bytes_per_rank[p];
        if (n == 0) continue;
        if (is_root) {
            char* base = static_cast<char*>(gathered[p]);
            for (int k = 0; k < c->comm_size && e == ncclSuccess; ++k)
                if (k != root) e = r.Recv(base + (size_t)k * n, n, ncclInt8, k, comm, s);
        } else {
            e = r.Send(local[p], n, ncclInt8, root, comm, s);
        }
    }
    const ncclResult_t ge = r.GroupEnd();
    if (e != ncclSuccess) return rccl_fail("ncclSend/ncclRecv", e);
    if (ge != ncclSuccess) return rccl_fail("ncclGroupEnd", ge);
    if (is_root)                                     // root's own region: a device copy on the same stream
        for (int p = 0; p < nplanes; ++p)
            if (bytes_per_rank[p])
                SVR_HIP_TRY(hipMemcpyAsync(static_cast<char*>(gathered[p]) + (size_t)root * bytes_per_rank[p], local[p],
                                           bytes_per_rank[p], hipMemcpyDeviceToDevice, s));
    return SVR_OK;
}

}  // extern "C"
