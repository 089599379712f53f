/* capi_shard.cpp - the multi-GPU side of the C-ABI (BASELINE.json north_star): one process (or thread) per
 * GPU, whole sequences sharded over the ranks, NO data-path collective; the one exchange is the ORB vocabulary, which rank 0
 * loads and broadcasts over RCCL (xGMI inside a node).  The reference has no counterpart: its only parallelism is the three
 * extractor threads of src/Frame.cc:124-134 - this is what a C++ host of the batched-sequence mode calls instead of the Python
 * torch.distributed path bench.py uses.
 *
 * RCCL is loaded at run time (dlopen librccl.so.1): libdrfe.so itself stays free of the dependency, a single-GPU host never
 * touches it, and a missing library is a DRFE_ERR_STATE with a message, not a load failure. */
#include "drfe_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl& rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.lib) break;
        }
        if (!R.lib) { R.err = std::string("RCCL not found: ") + dlerror(); return; }
#define SYM(field, name)                                                                      \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, name));                        \
    if (!R.field) { R.err = std::string("RCCL lacks ") + name; return; }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(Broadcast, "ncclBroadcast")
        SYM(AllReduce, "ncclAllReduce")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    });
    return R;
}

static_assert(sizeof(ncclUniqueId) == DRFE_SHARD_ID_BYTES, "drfe.h carries ncclUniqueId as DRFE_SHARD_ID_BYTES opaque bytes");

} // namespace

struct drfe_shard {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int nranks = 1, rank = 0, device = 0;
    void* d_buf = nullptr; size_t cap = 0;
    std::string err;
};

static thread_local std::string g_shardErr;

extern "C" {

const char* drfe_shard_last_error(const drfe_shard* s) { return s ? s->err.c_str() : g_shardErr.c_str(); }

int drfe_shard_unique_id(uint8_t* id)
{
    if (!id) return DRFE_ERR_INVALID;
    Rccl& R = rccl();
    if (!R.err.empty()) { g_shardErr = R.err; return DRFE_ERR_STATE; }
    ncclUniqueId u;
    const ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) { g_shardErr = std::string("ncclGetUniqueId: ") + R.GetErrorString(r); return DRFE_ERR_HIP; }
    std::memcpy(id, &u, sizeof(u));
    return DRFE_OK;
}

int drfe_shard_create(const uint8_t* id, int nranks, int rank, int device, drfe_shard** out)
{
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return DRFE_ERR_INVALID;
    *out = nullptr;
    Rccl& R = rccl();
    if (!R.err.empty()) { g_shardErr = R.err; return DRFE_ERR_STATE; }
    if (hipSetDevice(device) != hipSuccess) { g_shardErr = "drfe_shard_create: hipSetDevice failed"; return DRFE_ERR_HIP; }
    drfe_shard* s = new (std::nothrow) drfe_shard();
    if (!s) return DRFE_ERR_INVALID;
    s->nranks = nranks; s->rank = rank; s->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t r = R.CommInitRank(&s->comm, nranks, u, rank);
    if (r != ncclSuccess) { g_shardErr = std::string("ncclCommInitRank: ") + R.GetErrorString(r); delete s; return DRFE_ERR_HIP; }
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { g_shardErr = "drfe_shard_create: stream"; R.CommDestroy(s->comm); delete s; return DRFE_ERR_HIP; }
    *out = s;
    return DRFE_OK;
}

void drfe_shard_destroy(drfe_shard* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->comm) (void)rccl().CommDestroy(s->comm);
    if (s->d_buf) (void)hipFree(s->d_buf);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static int shard_scratch(drfe_shard* s, size_t bytes)
{
    if (s->cap >= bytes) return DRFE_OK;
    if (s->d_buf) (void)hipFree(s->d_buf);
    s->d_buf = nullptr; s->cap = 0;
    if (hipMalloc(&s->d_buf, bytes) != hipSuccess) { s->err = "drfe_shard: device staging buffer"; return DRFE_ERR_HIP; }
    s->cap = bytes;
    return DRFE_OK;
}

/* a HOST buffer of `bytes` bytes from rank `root` to every rank: staged through device memory, one ncclBroadcast (the
 * vocabulary's ~58 MB node table at initialisation) */
int drfe_shard_broadcast(drfe_shard* s, void* buf, size_t bytes, int root)
{
    if (!s || (!buf && bytes) || root < 0 || root >= s->nranks) { if (s) s->err = "drfe_shard_broadcast: invalid argument"; return DRFE_ERR_INVALID; }
    if (bytes == 0) return DRFE_OK;
    Rccl& R = rccl();
    if (hipSetDevice(s->device) != hipSuccess) { s->err = "drfe_shard_broadcast: hipSetDevice"; return DRFE_ERR_HIP; }
    int rc = shard_scratch(s, bytes);
    if (rc != DRFE_OK) return rc;
    if (s->rank == root && hipMemcpyAsync(s->d_buf, buf, bytes, hipMemcpyHostToDevice, s->stream) != hipSuccess) { s->err = "drfe_shard_broadcast: upload"; return DRFE_ERR_HIP; }
    const ncclResult_t r = R.Broadcast(s->d_buf, s->d_buf, bytes, ncclUint8, root, s->comm, s->stream);
    if (r != ncclSuccess) { s->err = std::string("ncclBroadcast: ") + R.GetErrorString(r); return DRFE_ERR_HIP; }
    if (s->rank != root && hipMemcpyAsync(buf, s->d_buf, bytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess) { s->err = "drfe_shard_broadcast: download"; return DRFE_ERR_HIP; }
    if (hipStreamSynchronize(s->stream) != hipSuccess) { s->err = "drfe_shard_broadcast: synchronize"; return DRFE_ERR_HIP; }
    return DRFE_OK;
}

/* end-of-run reporting: element-wise MAX over the ranks of n_max doubles (times) and SUM of n_sum 64-bit counters (frames), in
 * place on every rank.  Not on the data path. */
int drfe_shard_reduce_report(drfe_shard* s, double* max_inout, int n_max, long long* sum_inout, int n_sum)
{
    if (!s || n_max < 0 || n_sum < 0 || (n_max && !max_inout) || (n_sum && !sum_inout)) { if (s) s->err = "drfe_shard_reduce_report: invalid argument"; return DRFE_ERR_INVALID; }
    Rccl& R = rccl();
    if (hipSetDevice(s->device) != hipSuccess) { s->err = "drfe_shard_reduce_report: hipSetDevice"; return DRFE_ERR_HIP; }
    const size_t bytes = (size_t)(n_max + n_sum) * 8;
    if (bytes == 0) return DRFE_OK;
    int rc = shard_scratch(s, bytes);
    if (rc != DRFE_OK) return rc;
    double* dm = static_cast<double*>(s->d_buf);
    long long* ds = reinterpret_cast<long long*>(dm + n_max);
    hipError_t e = hipSuccess;
    if (n_max) e = hipMemcpyAsync(dm, max_inout, (size_t)n_max * 8, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess && n_sum) e = hipMemcpyAsync(ds, sum_inout, (size_t)n_sum * 8, hipMemcpyHostToDevice, s->stream);
    if (e != hipSuccess) { s->err = "drfe_shard_reduce_report: upload"; return DRFE_ERR_HIP; }
    ncclResult_t r = ncclSuccess;
    if (n_max) r = R.AllReduce(dm, dm, (size_t)n_max, ncclDouble, ncclMax, s->comm, s->stream);
    if (r == ncclSuccess && n_sum) r = R.AllReduce(ds, ds, (size_t)n_sum, ncclInt64, ncclSum, s->comm, s->stream);
    if (r != ncclSuccess) { s->err = std::string("ncclAllReduce: ") + R.GetErrorString(r); return DRFE_ERR_HIP; }
    if (n_max) e = hipMemcpyAsync(max_inout, dm, (size_t)n_max * 8, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && n_sum) e = hipMemcpyAsync(sum_inout, ds, (size_t)n_sum * 8, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) { s->err = "drfe_shard_reduce_report: download"; return DRFE_ERR_HIP; }
    return DRFE_OK;
}

/* which sequences rank `rank` of `nranks` processes when `n_sequences` are dealt round robin (sequence i -> rank i % nranks):
 * writes up to cap indices, returns their number.  Host arithmetic, no RCCL. */
int drfe_shard_sequences_of_rank(int n_sequences, int nranks, int rank, int* out, int cap)
{
    if (n_sequences < 0 || nranks < 1 || rank < 0 || rank >= nranks || (cap > 0 && !out)) return DRFE_ERR_INVALID;
    int n = 0;
    for (int i = rank; i < n_sequences; i += nranks) { if (n < cap) out[n] = i; n++; }
    return n;
}

} /* extern "C" */
