// Collectives of the data-parallel step over RCCL (xGMI), behind the C ABI: the gradient all-reduce (one per step, in pieces that
// overlap the convolution backward), the all-gather / reduce-scatter of the Q1 latent exchange (vae_model/decoder.py:109-110 mixes
// rows of the GLOBAL batch).  The reference is single-GPU (utils/parameters.py:163-164): nothing there to mirror.
//
// RCCL is bound at run time (dlopen): a single-GPU user of libvaecap needs no RCCL at all, and a process that already carries an RCCL
// (PyTorch-ROCm bundles one) must not end up with two -- the already-loaded library is preferred (RTLD_NOLOAD by soname), then the
// ROCm install; VC_RCCL_LIB overrides.  Every RCCL failure becomes a non-zero return code with ncclGetErrorString in vc_last_error.
#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "vaecap.h"

// The part of the NCCL API (rccl.h) this file binds, declared HERE: libvaecap builds on a ROCm install without the RCCL development
// headers (the library is bound at run time, see above).  Values as published in rccl.h / nccl.h and stable across their releases:
// ncclUniqueId is 128 opaque bytes, ncclSuccess = 0, ncclInProgress = 7, ncclSum = 0, ncclFloat32 = 7.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
}

namespace vc {

enum { VC_ECOMM = 10003 };

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    char path[256] = {0};
};

static Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("VC_RCCL_LIB");
        const char* loaded[] = {"librccl.so.1", "librccl.so"};
        const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (env && *env) {
            r.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
            snprintf(r.path, sizeof(r.path), "%s", env);
        }
        for (int i = 0; !r.handle && i < 2; ++i) {
            r.handle = dlopen(loaded[i], RTLD_NOW | RTLD_NOLOAD);   // the RCCL this process already carries (e.g. PyTorch's)
            if (r.handle) snprintf(r.path, sizeof(r.path), "%s (already loaded)", loaded[i]);
        }
        for (int i = 0; !r.handle && i < 3; ++i) {
            r.handle = dlopen(fresh[i], RTLD_NOW | RTLD_LOCAL);
            if (r.handle) snprintf(r.path, sizeof(r.path), "%s", fresh[i]);
        }
        if (!r.handle) return;
#define VC_SYM(field, name) *(void**)(&r.field) = dlsym(r.handle, name)
        VC_SYM(GetUniqueId, "ncclGetUniqueId");
        VC_SYM(CommInitRank, "ncclCommInitRank");
        VC_SYM(CommDestroy, "ncclCommDestroy");
        VC_SYM(CommAbort, "ncclCommAbort");
        VC_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
        VC_SYM(AllReduce, "ncclAllReduce");
        VC_SYM(AllGather, "ncclAllGather");
        VC_SYM(ReduceScatter, "ncclReduceScatter");
        VC_SYM(GetErrorString, "ncclGetErrorString");
        VC_SYM(GetVersion, "ncclGetVersion");
#undef VC_SYM
    });
    const bool ok = r.handle && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommAbort && r.AllReduce && r.AllGather && r.ReduceScatter &&
                    r.GetErrorString;
    return ok ? &r : nullptr;
}

static int rccl_fail(const Rccl* r, ncclResult_t e, const char* fn, const char* what) {
    snprintf(last_error_buf(), 512, "%s: %s failed: %s (RCCL result %d)", fn, what, r->GetErrorString(e), (int)e);
    return VC_ECOMM;
}

// The handle the ABI gives out.  Destroyed / aborted communicators keep their (small) wrapper, marked dead: a late call returns an error
// code instead of touching freed RCCL state.
struct Comm {
    uint32_t magic;
    ncclComm_t comm;
    int world, rank, device;
    bool alive;
};
constexpr uint32_t COMM_MAGIC = 0x76636363u;   // "vccc"

static int comm_check(void* h, const char* fn, Comm** out, Rccl** r) {
    Comm* c = (Comm*)h;
    if (!c || c->magic != COMM_MAGIC) return fail(VC_EINVAL, "%s: not a communicator handle (vc_comm_init_rank)", fn);
    if (!c->alive) return fail(VC_ECOMM, "%s: the communicator was destroyed or aborted", fn);
    *r = rccl();
    if (!*r) return fail(VC_ECOMM, "%s: RCCL is not loadable", fn);
    if ((*r)->CommGetAsyncError) {   // an error another rank raised (or a network failure) surfaces on the next call
        ncclResult_t ae = ncclSuccess;
        const ncclResult_t e = (*r)->CommGetAsyncError(c->comm, &ae);
        if (e != ncclSuccess) return rccl_fail(*r, e, fn, "ncclCommGetAsyncError");
        if (ae != ncclSuccess && ae != ncclInProgress) return rccl_fail(*r, ae, fn, "an earlier collective (asynchronous error)");
    }
    *out = c;
    return 0;
}

}  // namespace vc

extern "C" int vc_comm_available(void) { return vc::rccl() ? 1 : 0; }

extern "C" int vc_comm_unique_id(void* id128) {
    using namespace vc;
    VC_CHECK_ARG(id128, "null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    Rccl* r = rccl();
    if (!r) return fail(VC_ECOMM, "%s: RCCL is not loadable (librccl.so.1; set VC_RCCL_LIB)", __func__);
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail(r, e, __func__, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int vc_comm_init_rank(int world, int rank, const void* id128, int device, void** comm_out) {
    using namespace vc;
    VC_CHECK_ARG(comm_out && id128 && world >= 1 && rank >= 0 && rank < world, "bad argument");
    *comm_out = nullptr;
    Rccl* r = rccl();
    if (!r) return fail(VC_ECOMM, "%s: RCCL is not loadable (librccl.so.1; set VC_RCCL_LIB)", __func__);
    const hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return fail((int)he, "%s: hipSetDevice failed", __func__);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, world, id, rank);
    if (e != ncclSuccess) return rccl_fail(r, e, __func__, "ncclCommInitRank");
    Comm* h = new Comm{COMM_MAGIC, c, world, rank, device, true};
    *comm_out = h;
    return 0;
}

extern "C" int vc_comm_info(void* comm, int* world, int* rank, int* rccl_version) {
    using namespace vc;
    Comm* c; Rccl* r;
    const int rc = comm_check(comm, __func__, &c, &r);
    if (rc) return rc;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) { *rccl_version = 0; if (r->GetVersion) r->GetVersion(rccl_version); }
    return 0;
}

extern "C" int vc_comm_destroy(void* comm) {
    using namespace vc;
    Comm* c = (Comm*)comm;
    VC_CHECK_ARG(c && c->magic == COMM_MAGIC, "not a communicator handle");
    if (!c->alive) return 0;
    c->alive = false;
    Rccl* r = rccl();
    if (!r) return fail(VC_ECOMM, "%s: RCCL is not loadable", __func__);
    const ncclResult_t e = r->CommDestroy(c->comm);
    return e == ncclSuccess ? 0 : rccl_fail(r, e, __func__, "ncclCommDestroy");
}

extern "C" int vc_comm_abort(void* comm) {
    using namespace vc;
    Comm* c = (Comm*)comm;
    VC_CHECK_ARG(c && c->magic == COMM_MAGIC, "not a communicator handle");
    if (!c->alive) return 0;
    c->alive = false;
    Rccl* r = rccl();
    if (!r) return fail(VC_ECOMM, "%s: RCCL is not loadable", __func__);
    const ncclResult_t e = r->CommAbort(c->comm);
    return e == ncclSuccess ? 0 : rccl_fail(r, e, __func__, "ncclCommAbort");
}

extern "C" int vc_allreduce_sum_f32(void* comm, void* stream, float* buf, size_t n) {
    using namespace vc;
    Comm* c; Rccl* r;
    const int rc = comm_check(comm, __func__, &c, &r);
    if (rc) return rc;
    VC_CHECK_ARG(buf || n == 0, "null pointer");
    if (n == 0) return 0;
    const ncclResult_t e = r->AllReduce(buf, buf, n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    return e == ncclSuccess ? 0 : rccl_fail(r, e, __func__, "ncclAllReduce");
}

extern "C" int vc_allgather_f32(void* comm, void* stream, const float* in, float* out, size_t n_per_rank) {
    using namespace vc;
    Comm* c; Rccl* r;
    const int rc = comm_check(comm, __func__, &c, &r);
    if (rc) return rc;
    VC_CHECK_ARG((in && out) || n_per_rank == 0, "null pointer");
    if (n_per_rank == 0) return 0;
    const ncclResult_t e = r->AllGather(in, out, n_per_rank, ncclFloat32, c->comm, (hipStream_t)stream);
    return e == ncclSuccess ? 0 : rccl_fail(r, e, __func__, "ncclAllGather");
}

extern "C" int vc_reducescatter_sum_f32(void* comm, void* stream, const float* in, float* out, size_t n_per_rank) {
    using namespace vc;
    Comm* c; Rccl* r;
    const int rc = comm_check(comm, __func__, &c, &r);
    if (rc) return rc;
    VC_CHECK_ARG((in && out) || n_per_rank == 0, "null pointer");
    if (n_per_rank == 0) return 0;
    const ncclResult_t e = r->ReduceScatter(in, out, n_per_rank, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    return e == ncclSuccess ? 0 : rccl_fail(r, e, __func__, "ncclReduceScatter");
}
