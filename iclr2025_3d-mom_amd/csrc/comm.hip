// Collectives of a multi-GPU step as plain C calls over librccl (one process per GPU, xGMI): mom_comm_* of include/mom4d.h.
//
// Why not torch.distributed: every call into it costs the rank's host 40-50 us (Python -> c10d -> ProcessGroupNCCL: work objects,
// events, a watchdog entry), a step has three (camera-batch shard) or six (tile-row shard) of them, and the host paces a rank
// (profiles/r05_probes/dist_one_rank*.json: one rank through RCCL ran at 0.93 / 0.80 of the unsharded step with NO byte on the wire).
// A ctypes call into ncclAllReduce on a raw stream handle costs about 2 us; ordering against the compute streams goes through the
// marks of csrc/stream_order.hip.
//
// librccl is resolved at RUN time (dlopen), not linked: the library must load on a machine without RCCL (a single-GPU user, the CPU
// build check), and in a process that has imported torch the SONAME librccl.so.1 is already mapped from torch/lib -- dlopen returns
// THAT copy, so the process never holds two RCCLs.  The communicator is created from a 128-byte id that rank 0 makes
// (mom_comm_unique_id) and the host mirror hands to the other ranks through whatever rendezvous it already has (the torch store).
#include "mom_common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

// the six enums and the id of rccl.h this file needs, restated (their values are ABI of NCCL 2.x / RCCL: nccl.h has kept them fixed)
typedef struct { char internal[128]; } IdBytes;
typedef void* Comm;
enum { kNcclSuccess = 0 };
enum { kNcclSum = 0, kNcclMax = 2 };
enum { kNcclInt32 = 2, kNcclFloat32 = 7 };

struct Api {
    void* handle = nullptr;
    int (*GetUniqueId)(IdBytes*) = nullptr;
    int (*CommInitRank)(Comm*, int, IdBytes, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Api g_api;
std::once_flag g_once;
char g_err[256] = "";

void set_err(const char* what, int rc)
{
    const char* s = (g_api.GetErrorString && rc > 0) ? g_api.GetErrorString(rc) : "";
    snprintf(g_err, sizeof(g_err), "%s%s%s (rc %d)", what, s[0] ? ": " : "", s, rc);
}

void load_api()
{
    const char* names[] = {getenv("MOM_RCCL_LIB"), "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if (!n || !n[0]) continue;
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.handle) break;
    }
    if (!g_api.handle) { snprintf(g_err, sizeof(g_err), "librccl not found (%s)", dlerror()); return; }
#define SYM(field, name) *(void**)(&g_api.field) = dlsym(g_api.handle, name); if (!g_api.field) { snprintf(g_err, sizeof(g_err), "%s missing in librccl", name); return; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(CommAbort, "ncclCommAbort")
    SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather") SYM(ReduceScatter, "ncclReduceScatter")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_api.ok = true;
}

bool api() { std::call_once(g_once, load_api); return g_api.ok; }

int dtype_of(int t) { return t == MOM_COMM_F32 ? kNcclFloat32 : (t == MOM_COMM_I32 ? kNcclInt32 : -1); }
int op_of(int o) { return o == MOM_COMM_SUM ? kNcclSum : (o == MOM_COMM_MAX ? kNcclMax : -1); }

}  // namespace

struct MomComm {
    Comm comm;
    int world, rank;
};

extern "C" {

int mom_comm_available(void) { return api() ? 1 : 0; }
const char* mom_comm_last_error(void) { return g_err; }

int mom_comm_unique_id(void* id128)
{
    if (!id128) return MOM_EINVAL;
    if (!api()) return MOM_EUNAVAILABLE;
    IdBytes id;
    const int rc = g_api.GetUniqueId(&id);
    if (rc != kNcclSuccess) { set_err("ncclGetUniqueId", rc); return MOM_ELAUNCH; }
    memcpy(id128, id.internal, sizeof(id.internal));
    return MOM_OK;
}

int mom_comm_create(MomComm** out, const void* id128, int world, int rank)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return MOM_EINVAL;
    if (!api()) return MOM_EUNAVAILABLE;
    IdBytes id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm c = nullptr;
    const int rc = g_api.CommInitRank(&c, world, id, rank);        // on the CURRENT device (hipSetDevice before the call)
    if (rc != kNcclSuccess || !c) { set_err("ncclCommInitRank", rc); return MOM_ELAUNCH; }
    *out = new MomComm{c, world, rank};
    return MOM_OK;
}

int mom_comm_destroy(MomComm* c)
{
    if (!c) return MOM_OK;
    const int rc = g_api.ok ? g_api.CommDestroy(c->comm) : kNcclSuccess;
    delete c;
    if (rc != kNcclSuccess) { set_err("ncclCommDestroy", rc); return MOM_ELAUNCH; }
    return MOM_OK;
}

int mom_comm_abort(MomComm* c)
{
    if (!c) return MOM_OK;
    const int rc = g_api.ok ? g_api.CommAbort(c->comm) : kNcclSuccess;
    delete c;
    return rc == kNcclSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_comm_world(const MomComm* c) { return c ? c->world : 0; }
int mom_comm_rank(const MomComm* c) { return c ? c->rank : -1; }

int mom_comm_group_start(void)
{
    if (!api()) return MOM_EUNAVAILABLE;
    const int rc = g_api.GroupStart();
    if (rc != kNcclSuccess) { set_err("ncclGroupStart", rc); return MOM_ELAUNCH; }
    return MOM_OK;
}

int mom_comm_group_end(void)
{
    if (!api()) return MOM_EUNAVAILABLE;
    const int rc = g_api.GroupEnd();
    if (rc != kNcclSuccess) { set_err("ncclGroupEnd", rc); return MOM_ELAUNCH; }
    return MOM_OK;
}

int mom_comm_all_reduce(MomComm* c, void* buf, size_t count, int dtype, int op, mom_stream_t stream)
{
    if (!c || dtype_of(dtype) < 0 || op_of(op) < 0 || (!buf && count)) return MOM_EINVAL;
    if (!count) return MOM_OK;
    const int rc = g_api.AllReduce(buf, buf, count, dtype_of(dtype), op_of(op), c->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) { set_err("ncclAllReduce", rc); return MOM_ELAUNCH; }
    return MOM_OK;
}

int mom_comm_all_gather(MomComm* c, void* buf, size_t count_per_rank, int dtype, mom_stream_t stream)
{
    if (!c || dtype_of(dtype) < 0 || (!buf && count_per_rank)) return MOM_EINVAL;
    if (!count_per_rank) return MOM_OK;
    // in place: this rank's slab already sits at buf + rank * count_per_rank (rccl.h: "In-place operations will happen if
    // sendbuff == recvbuff + rank * sendcount")
    const char* own = (const char*)buf + (size_t)c->rank * count_per_rank * 4;
    const int rc = g_api.AllGather(own, buf, count_per_rank, dtype_of(dtype), c->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) { set_err("ncclAllGather", rc); return MOM_ELAUNCH; }
    return MOM_OK;
}

int mom_comm_reduce_scatter(MomComm* c, void* buf, size_t count_per_rank, int dtype, int op, mom_stream_t stream)
{
    if (!c || dtype_of(dtype) < 0 || op_of(op) < 0 || (!buf && count_per_rank)) return MOM_EINVAL;
    if (!count_per_rank) return MOM_OK;
    // in place: the reduced block of this rank lands at buf + rank * count_per_rank ("recvbuff == sendbuff + rank * recvcount")
    char* own = (char*)buf + (size_t)c->rank * count_per_rank * 4;
    const int rc = g_api.ReduceScatter(buf, own, count_per_rank, dtype_of(dtype), op_of(op), c->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) { set_err("ncclReduceScatter", rc); return MOM_ELAUNCH; }
    return MOM_OK;
}

}  // extern "C"
