// comm.hip -- see comm.h.  The exchange of the data-parallel step: one all-reduce(SUM) over the gradient tensors,
// fp32 on the wire, ring over xGMI; the reference (lrcn.jl:369-394) is single-device, this has no counterpart there.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

namespace {

struct Api {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    char why[256] = "";
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) {
            snprintf(a.why, sizeof(a.why), "cannot open librccl.so.1: %s", dlerror());
            return;
        }
#define SYM(field, name)                                                                \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name));                   \
    if (!a.field) {                                                                      \
        snprintf(a.why, sizeof(a.why), "librccl has no symbol %s", name);                \
        return;                                                                          \
    }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllReduce, "ncclAllReduce")
        SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        a.ok = true;
    });
    return a;
}

}  // namespace

struct LrcnComm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
};

int comm_available(char *err, size_t errn) {
    Api &a = api();
    if (!a.ok) {
        snprintf(err, errn, "%s", a.why);
        return -1;
    }
    return 0;
}

int comm_unique_id(void *out128, char *err, size_t errn) {
    Api &a = api();
    if (!a.ok) {
        snprintf(err, errn, "%s", a.why);
        return -1;
    }
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = a.GetUniqueId(&id);
    if (r != ncclSuccess) {
        snprintf(err, errn, "ncclGetUniqueId: %s", a.GetErrorString(r));
        return -1;
    }
    memcpy(out128, &id, sizeof(id));
    return 0;
}

LrcnComm *comm_create(int world, int rank, const void *unique_id128, char *err, size_t errn) {
    Api &a = api();
    if (!a.ok) {
        snprintf(err, errn, "%s", a.why);
        return nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    LrcnComm *c = new LrcnComm();
    c->world = world;
    c->rank = rank;
    const ncclResult_t r = a.CommInitRank(&c->comm, world, id, rank);  // collective: every rank calls it with the same id
    if (r != ncclSuccess) {
        snprintf(err, errn, "ncclCommInitRank(world %d, rank %d): %s", world, rank, a.GetErrorString(r));
        delete c;
        return nullptr;
    }
    return c;
}

void comm_destroy(LrcnComm *c) {
    if (!c) return;
    if (c->comm) (void)api().CommDestroy(c->comm);
    delete c;
}

int comm_world(const LrcnComm *c) { return c ? c->world : 1; }

int comm_allreduce_f32(LrcnComm *c, float *buf, size_t count, hipStream_t stream, char *err, size_t errn) {
    if (!c || count == 0) return 0;
    Api &a = api();
    const ncclResult_t r = a.AllReduce(buf, buf, count, ncclFloat, ncclSum, c->comm, stream);
    if (r != ncclSuccess) {
        snprintf(err, errn, "ncclAllReduce(%zu floats): %s", count, a.GetErrorString(r));
        return -1;
    }
    return 0;
}

int comm_group_begin(LrcnComm *c) { return c ? (api().GroupStart() == ncclSuccess ? 0 : -1) : 0; }
int comm_group_end(LrcnComm *c) { return c ? (api().GroupEnd() == ncclSuccess ? 0 : -1) : 0; }
