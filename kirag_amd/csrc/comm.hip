// kr_comm / kr_shard_allgather_topk: the one exchange step of the row-sharded search for hosts that do not bring their own
// collective library: RCCL (xGMI) all-gather of every rank's [nq, k] result lists + the device merge, enqueued on the caller's stream.
//
// Replaces the reference's gather-to-rank-0 (`utils/utils.py:145-155`) for the search results; the Python mirror (`kirag_amd/parallel.py`) can use
// either this (comm="rccl") or `torch.distributed` (default).  RCCL is resolved at run time (dlopen of librccl.so.1: the copy a torch process has
// already loaded is reused), so the library has no link-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"

namespace kr {
namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string err;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("KIRAG_AMD_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            r.err = dlerror();
        }
        if (!r.h) return;
        auto sym = [&](const char* s) {
            void* p = dlsym(r.h, s);
            if (!p) { r.err = std::string("missing symbol ") + s; }
            return p;
        };
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(sym("ncclAllGather"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(sym("ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(sym("ncclGroupEnd"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.group_start || !r.group_end || !r.error_string) {
            dlclose(r.h); r.h = nullptr;
        }
    });
    return r;
}

int need_rccl() {
    Rccl& r = rccl();
    if (!r.h) return fail(KR_ESTATE, "RCCL is not loadable (%s): set KIRAG_AMD_RCCL_LIB or use the host's own collective + kr_topk_merge_device", r.err.c_str());
    return 0;
}

#define KR_NCCL(expr)                                                                                                   \
    do {                                                                                                                \
        ncclResult_t _r = (expr);                                                                                       \
        if (_r != ncclSuccess) return kr::fail(KR_EHIP, "%s failed: %s (%s:%d)", #expr, rccl().error_string(_r), __FILE__, __LINE__); \
    } while (0)

}  // namespace
}  // namespace kr

struct kr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    char* ws = nullptr;          // [world][nq*k] int64 rows, then [world][nq*k] fp32 scores
    size_t ws_bytes = 0;
};

using namespace kr;

extern "C" {

int kr_comm_unique_id(void* id128) {
    if (!id128) return fail(KR_EINVAL, "null id buffer");
    KR_TRY(need_rccl());
    static_assert(sizeof(ncclUniqueId) == KR_COMM_ID_BYTES, "KR_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId id;
    KR_NCCL(rccl().get_unique_id(&id));
    __builtin_memcpy(id128, &id, sizeof id);
    return 0;
}

int kr_comm_create(const void* id128, int rank, int world, int device, kr_comm** out) {
    if (!out) return fail(KR_EINVAL, "null out pointer");
    *out = nullptr;
    if (!id128 || world <= 0 || rank < 0 || rank >= world) return fail(KR_EINVAL, "bad communicator arguments (rank %d of %d)", rank, world);
    KR_TRY(need_rccl());
    KR_TRY(select_device(device));
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof id);
    kr_comm* c = new kr_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = rccl().comm_init_rank(&c->comm, world, id, rank);       // blocks until all `world` ranks have called it
    if (r != ncclSuccess) {
        delete c;
        return fail(KR_EHIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, rccl().error_string(r));
    }
    *out = c;
    return 0;
}

int kr_comm_destroy(kr_comm* c) {
    if (!c) return 0;
    int rc = 0;
    if (select_device(c->device) == 0) {
        (void)hipDeviceSynchronize();
        if (c->ws) (void)hipFree(c->ws);
    }
    if (c->comm && rccl().h) {
        ncclResult_t r = rccl().comm_destroy(c->comm);
        if (r != ncclSuccess) rc = fail(KR_EHIP, "ncclCommDestroy failed: %s", rccl().error_string(r));
    }
    delete c;
    return rc;
}

int kr_comm_rank(const kr_comm* c) { return c ? c->rank : -1; }
int kr_comm_world(const kr_comm* c) { return c ? c->world : -1; }

int kr_shard_allgather_topk(kr_comm* c, const float* scores_local, const int64_t* rows_local, int nq, int k, float* out_scores, int64_t* out_rows,
                            void* stream) {
    if (!c || !scores_local || !rows_local || !out_scores || !out_rows || nq < 0 || k <= 0) return fail(KR_EINVAL, "bad all-gather arguments");
    if ((int64_t)c->world * k > 8192)
        return fail(KR_EINVAL, "world * k = %lld exceeds the device merge's 8192 entries per query (gather with the host's collective and use kr_topk_merge)",
                    (long long)c->world * k);
    if (nq == 0) return 0;
    KR_TRY(select_device(c->device));
    if (!is_device_pointer(scores_local) || !is_device_pointer(rows_local)) return fail(KR_EINVAL, "the local lists must be device pointers");
    const size_t per = (size_t)nq * k;
    const size_t need = per * c->world * 12;
    if (need > c->ws_bytes) {
        if (c->ws) { KR_HIP(hipDeviceSynchronize()); KR_HIP(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
        KR_HIP(hipMalloc(reinterpret_cast<void**>(&c->ws), need));
        c->ws_bytes = need;
    }
    int64_t* all_rows = reinterpret_cast<int64_t*>(c->ws);
    float* all_sc = reinterpret_cast<float*>(c->ws + per * c->world * 8);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    KR_NCCL(rccl().group_start());                                            // one fused launch for both lists
    ncclResult_t r1 = rccl().all_gather(rows_local, all_rows, per, ncclInt64, c->comm, st);
    ncclResult_t r2 = rccl().all_gather(scores_local, all_sc, per, ncclFloat32, c->comm, st);
    KR_NCCL(rccl().group_end());
    KR_NCCL(r1); KR_NCCL(r2);
    return kr_topk_merge_device(all_sc, (int64_t)per, all_rows, (int64_t)per, c->world, nq, k, out_scores, out_rows, c->device, stream);
}

}  // extern "C"
