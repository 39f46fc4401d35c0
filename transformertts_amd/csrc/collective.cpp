// ttsmi_allreduce_sum_f32 and the communicator helpers: a thin RCCL wrapper for bindings that have no collective library
// of their own (the Python host side uses torch.distributed, whose "nccl" backend IS RCCL).
// RCCL is resolved at first use with dlopen - the copy already loaded in the process if there is one (torch bundles
// its own librccl: two instances in one process must not happen), else the system one - so libttsmi has no link-time
// dependency on it and a single-GPU user never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ttsmi.h"

void ttsmi_set_error(const char* fmt, ...);

namespace {
struct UniqueId { char bytes[128]; };                    // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, UniqueId, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*error_string_fn)(int);

struct Rccl {
    void* handle = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    all_reduce_fn all_reduce = nullptr;
    error_string_fn error_string = nullptr;
    bool ok = false;
    char why[256] = "symbols missing";                   // dlerror() captured ONCE, at the failing dlopen (a second call returns NULL)
    Rccl() {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {                    // already in the process (torch's copy)?
            handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (handle) break;
        }
        for (int i = 0; !handle && i < 2; ++i) {
            handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (!handle) {
                const char* e = dlerror();
                if (e) { strncpy(why, e, sizeof(why) - 1); why[sizeof(why) - 1] = 0; }
            }
        }
        if (!handle) return;
        get_unique_id = (get_unique_id_fn)dlsym(handle, "ncclGetUniqueId");
        comm_init_rank = (comm_init_rank_fn)dlsym(handle, "ncclCommInitRank");
        comm_destroy = (comm_destroy_fn)dlsym(handle, "ncclCommDestroy");
        all_reduce = (all_reduce_fn)dlsym(handle, "ncclAllReduce");
        error_string = (error_string_fn)dlsym(handle, "ncclGetErrorString");
        ok = get_unique_id && comm_init_rank && comm_destroy && all_reduce;
    }
};
Rccl& rccl() {
    static Rccl r;                                       // initialised once (C++11), like the TTSMI_* knobs
    return r;
}
int fail(const char* who, int rc) {
    Rccl& r = rccl();
    ttsmi_set_error("%s: RCCL error %d (%s)", who, rc, r.error_string ? r.error_string(rc) : "?");
    return TTSMI_ERR_LAUNCH;
}
int need(const char* who) {
    if (rccl().ok) return TTSMI_OK;
    ttsmi_set_error("%s: librccl.so.1 could not be loaded (%s)", who, rccl().why);
    return TTSMI_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" {

int ttsmi_comm_unique_id(void* id128) {
    if (!id128) { ttsmi_set_error("comm_unique_id: null pointer"); return TTSMI_ERR_INVALID_ARG; }
    if (int rc = need("comm_unique_id")) return rc;
    UniqueId id;
    if (int rc = rccl().get_unique_id(&id)) return fail("comm_unique_id", rc);
    memcpy(id128, id.bytes, sizeof(id.bytes));
    return TTSMI_OK;
}

int ttsmi_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        ttsmi_set_error("comm_init_rank: bad argument (nranks %d, rank %d)", nranks, rank);
        return TTSMI_ERR_INVALID_ARG;
    }
    if (int rc = need("comm_init_rank")) return rc;
    UniqueId id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    if (int rc = rccl().comm_init_rank(comm, nranks, id, rank)) return fail("comm_init_rank", rc);
    return TTSMI_OK;
}

int ttsmi_comm_destroy(void* comm) {
    if (!comm) return TTSMI_OK;
    if (int rc = need("comm_destroy")) return rc;
    if (int rc = rccl().comm_destroy(comm)) return fail("comm_destroy", rc);
    return TTSMI_OK;
}

int ttsmi_allreduce_sum_f32(void* comm, float* buf, int64_t n, ttsmi_stream_t stream) {
    if (!comm || !buf || n < 0) { ttsmi_set_error("allreduce_sum_f32: bad argument"); return TTSMI_ERR_INVALID_ARG; }
    if (n == 0) return TTSMI_OK;
    if (int rc = need("allreduce_sum_f32")) return rc;
    // ncclFloat32 = 7, ncclSum = 0 (rccl.h); in place
    if (int rc = rccl().all_reduce(buf, buf, (size_t)n, 7, 0, comm, (hipStream_t)stream)) return fail("allreduce_sum_f32", rc);
    return TTSMI_OK;
}

}  // extern "C"
