// RCCL collectives of the path behind the C ABI: the SUM all-reduce of the flat f32 gradient buffer and the
// variable-length all-gather of a CLaSS round's rows, on an in-process communicator (one process per GPU; the unique id is
// exchanged by the launcher).  The reference has no multi-device code: this is new.
//
// librccl is bound at RUN TIME (dlopen): inside a PyTorch process the copy torch.distributed has already loaded is reused
// (two RCCL instances in one process would each spin up their own proxy threads and IPC state); a host without RCCL can
// still load libcpg_hip.so - the cpg_comm_* entry points then fail with a message instead of the library failing to load.
#include <dlfcn.h>
#include <mutex>
#include "cpg_internal.h"

namespace {

typedef struct { char internal[128]; } rcclUniqueId;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40-43)
typedef void* rcclComm;
enum { RCCL_SUM = 0, RCCL_INT8 = 0, RCCL_FLOAT32 = 7 };   // ncclRedOp_t / ncclDataType_t values (rccl.h:448,459,466)

struct Rccl {
    int (*GetUniqueId)(rcclUniqueId*);
    int (*CommInitRank)(rcclComm*, int, rcclUniqueId, int);
    int (*CommDestroy)(rcclComm);
    int (*CommCount)(const rcclComm, int*);
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm, hipStream_t);
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char* (*GetErrorString)(int);
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // the copy the process already has (torch's)
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
#define CPG_SYM(field, name) *(void**)(&r.field) = dlsym(h, name)
        CPG_SYM(GetUniqueId, "ncclGetUniqueId");
        CPG_SYM(CommInitRank, "ncclCommInitRank");
        CPG_SYM(CommDestroy, "ncclCommDestroy");
        CPG_SYM(CommCount, "ncclCommCount");
        CPG_SYM(AllReduce, "ncclAllReduce");
        CPG_SYM(Broadcast, "ncclBroadcast");
        CPG_SYM(GroupStart, "ncclGroupStart");
        CPG_SYM(GroupEnd, "ncclGroupEnd");
        CPG_SYM(GetErrorString, "ncclGetErrorString");
#undef CPG_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.GroupStart && r.GroupEnd &&
               r.GetErrorString;
    });
    return r;
}

int fail(const char* what, int rc) {
    cpg_set_error("%s: %s", what, rccl().ok ? rccl().GetErrorString(rc) : "librccl.so is not loadable");
    return rc ? -100 - rc : -100;
}

}  // namespace

#define CPG_RCCL(call, what)                  \
    do {                                      \
        if (!rccl().ok) return fail(what, 0); \
        const int rc__ = (call);              \
        if (rc__ != 0) return fail(what, rc__); \
    } while (0)

// 1 when librccl could be bound
CPG_EXPORT int cpg_comm_available(void) { return rccl().ok ? 1 : 0; }

// rank 0 calls this and hands the 128 bytes to every rank (any out-of-band channel)
CPG_EXPORT int cpg_comm_unique_id(void* id128) {
    CPG_CHECK_ARG(id128);
    CPG_RCCL(rccl().GetUniqueId((rcclUniqueId*)id128), "ncclGetUniqueId");
    return 0;
}

// collective over all ranks (each on its own device: hipSetDevice first); *comm receives the communicator handle
CPG_EXPORT int cpg_comm_init(const void* id128, int rank, int world, void** comm) {
    CPG_CHECK_ARG(id128 && comm && world >= 1 && rank >= 0 && rank < world);
    rcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    CPG_RCCL(rccl().CommInitRank((rcclComm*)comm, world, id, rank), "ncclCommInitRank");
    return 0;
}

// ranks the communicator itself reports (ncclCommCount): what a first multi-GPU run prints next to the launcher's WORLD_SIZE
CPG_EXPORT int cpg_comm_count(void* comm) {
    if (!rccl().ok || !rccl().CommCount || !comm) return -1;
    int n = -1;
    const int rc = rccl().CommCount((rcclComm)comm, &n);
    return rc == 0 ? n : -1;
}

CPG_EXPORT int cpg_comm_destroy(void* comm) {
    CPG_CHECK_ARG(comm);
    CPG_RCCL(rccl().CommDestroy((rcclComm)comm), "ncclCommDestroy");
    return 0;
}

// buf[0..n) := sum over ranks, in place, asynchronous on `stream` (the gradient exchange of train_vae's step; the 1/world
// factor is folded into cpg_adam_step's gscale)
CPG_EXPORT int cpg_allreduce_f32(void* comm, float* buf, size_t n, void* stream) {
    CPG_CHECK_ARG(comm && buf && n > 0);
    CPG_RCCL(rccl().AllReduce(buf, buf, n, RCCL_FLOAT32, RCCL_SUM, (rcclComm)comm, (hipStream_t)stream), "ncclAllReduce");
    return 0;
}

// Variable-length all-gather of bytes: rank r contributes counts[r] bytes (counts: HOST array of `world` entries, identical on
// every rank - exchange it first, e.g. with cpg_allgatherv itself on 8-byte entries); recv receives the contributions back to
// back in rank order.  send may be null when counts[rank] == 0.  One grouped set of broadcasts: every rank is the root of one.
CPG_EXPORT int cpg_allgatherv(void* comm, const void* send, const size_t* counts, int rank, int world, void* recv, void* stream) {
    CPG_CHECK_ARG(comm && counts && recv && world >= 1 && rank >= 0 && rank < world && (send || counts[rank] == 0));
    CPG_RCCL(rccl().GroupStart(), "ncclGroupStart");
    size_t off = 0;
    for (int r = 0; r < world; ++r) {
        if (counts[r] > 0) {
            char* dst = (char*)recv + off;
            const int rc = rccl().Broadcast(r == rank ? send : (const void*)dst, dst, counts[r], RCCL_INT8, r, (rcclComm)comm,
                                            (hipStream_t)stream);
            if (rc != 0) {
                rccl().GroupEnd();
                return fail("ncclBroadcast", rc);
            }
        }
        off += counts[r];
    }
    CPG_RCCL(rccl().GroupEnd(), "ncclGroupEnd");
    return 0;
}
