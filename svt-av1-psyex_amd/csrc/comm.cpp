// comm.cpp -- the multi-GPU exchange of the open-loop ME results behind the C-ABI (include/svt_hip_comm.h): RCCL all-gather over xGMI
// of the per-b64 results of every rank's row band, so that a C host -- the reference's motion-estimation process collects the results
// of all superblocks of a picture before it posts the picture (Codec/me_process.c:174-313) -- needs nothing but this library.
// librccl is opened at run time (dlopen): single-GPU users of libsvthip.so do not need it.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <string.h>
#include <new>
#include <rccl/rccl.h>
#include "svt_hip_internal.h"
#include "../../include/svt_hip_comm.h"

namespace {

struct Rccl {
    void *lib;
    decltype(&ncclGetUniqueId)    GetUniqueId;
    decltype(&ncclCommInitRank)   CommInitRank;
    decltype(&ncclCommDestroy)    CommDestroy;
    decltype(&ncclCommCount)      CommCount;
    decltype(&ncclAllGather)      AllGather;
    decltype(&ncclBroadcast)      Broadcast;
    decltype(&ncclGroupStart)     GroupStart;
    decltype(&ncclGroupEnd)       GroupEnd;
    decltype(&ncclGetErrorString) GetErrorString;
};

std::mutex g_rccl_mu;
Rccl       g_rccl;
bool       g_rccl_ready = false;

int load_rccl(SvtHipContext *ctx) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_ready) return SVT_HIP_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_DEVICE, "librccl.so not found: %s", dlerror());
    g_rccl.lib = lib;
#define SYM(field, name)                                                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));                                   \
    if (!g_rccl.field) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_DEVICE, "librccl.so lacks %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount") SYM(AllGather, "ncclAllGather")
    SYM(Broadcast, "ncclBroadcast") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl_ready = true;
    return SVT_HIP_OK;
}

#define RCCL_CHECK(ctx, call)                                                                                                          \
    do {                                                                                                                               \
        ncclResult_t r_ = (call);                                                                                                      \
        if (r_ != ncclSuccess) return svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "%s failed: %s", #call, g_rccl.GetErrorString(r_));        \
    } while (0)

} // namespace

struct SvtHipComm {
    SvtHipContext *ctx;
    ncclComm_t     comm;
    int            rank, world;
    hipStream_t    stream;                       // the exchange runs beside the compute of the context stream
    hipEvent_t     ready[SVT_HIP_COMM_SLOTS];    // context stream: the results of slot k are written
    hipEvent_t     landed[SVT_HIP_COMM_SLOTS];   // exchange stream: slot k's exchange has completed
    bool           used[SVT_HIP_COMM_SLOTS];
};

static_assert(SVT_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as opaque bytes");

extern "C" {

int svt_hip_comm_unique_id(SvtHipContext *ctx, uint8_t id[SVT_HIP_COMM_ID_BYTES]) {
    if (!ctx || !id) return SVT_HIP_ERR_BAD_PARAM;
    if (int rc = load_rccl(ctx)) return rc;
    ncclUniqueId u;
    RCCL_CHECK(ctx, g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SVT_HIP_OK;
}

int svt_hip_comm_create(SvtHipContext *ctx, const uint8_t id[SVT_HIP_COMM_ID_BYTES], int rank, int world, SvtHipComm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return SVT_HIP_ERR_BAD_PARAM;
    *out = nullptr;
    if (int rc = load_rccl(ctx)) return rc;
    SVT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SvtHipComm *c = new (std::nothrow) SvtHipComm();
    if (!c) return SVT_HIP_ERR_NO_MEMORY;
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < SVT_HIP_COMM_SLOTS && ok; k++)
        ok = hipEventCreateWithFlags(&c->ready[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->landed[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        svt_hip_comm_destroy(c);
        return svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "comm: stream / event creation failed");
    }
    *out = c;
    return SVT_HIP_OK;
}

void svt_hip_comm_destroy(SvtHipComm *c) {
    if (!c) return;
    hipSetDevice(c->ctx->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    for (int k = 0; k < SVT_HIP_COMM_SLOTS; k++) {
        if (c->ready[k]) hipEventDestroy(c->ready[k]);
        if (c->landed[k]) hipEventDestroy(c->landed[k]);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// Common head of an exchange: the exchange stream picks up behind what the context stream has enqueued so far (the ME launch that
// writes `send`); the compute enqueued afterwards on the context stream runs beside the exchange.
static int exchange_begin(SvtHipComm *c, int slot) {
    if (!c || slot < 0 || slot >= SVT_HIP_COMM_SLOTS) return SVT_HIP_ERR_BAD_PARAM;
    SVT_HIP_CHECK(c->ctx, hipSetDevice(c->ctx->device));
    SVT_HIP_CHECK(c->ctx, hipEventRecord(c->ready[slot], c->ctx->stream));
    SVT_HIP_CHECK(c->ctx, hipStreamWaitEvent(c->stream, c->ready[slot], 0));
    return SVT_HIP_OK;
}
static int exchange_end(SvtHipComm *c, int slot) {
    SVT_HIP_CHECK(c->ctx, hipEventRecord(c->landed[slot], c->stream));
    c->used[slot] = true;
    return SVT_HIP_OK;
}

int svt_hip_me_results_all_gather(SvtHipComm *c, int slot, const void *send_dev, void *recv_dev, size_t bytes_per_rank) {
    if (!c || slot < 0 || slot >= SVT_HIP_COMM_SLOTS) return SVT_HIP_ERR_BAD_PARAM;
    if (!send_dev || !recv_dev) return svt_hip_fail(c->ctx, SVT_HIP_ERR_BAD_PARAM, "all-gather: null buffer"); // before any stream is touched
    if (int rc = exchange_begin(c, slot)) return rc;
    if (bytes_per_rank) RCCL_CHECK(c->ctx, g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, c->stream));
    return exchange_end(c, slot);
}

int svt_hip_me_results_all_gather_v(SvtHipComm *c, int slot, const void *send_dev, void *recv_dev, const size_t *offsets, const size_t *bytes) {
    if (!c || slot < 0 || slot >= SVT_HIP_COMM_SLOTS) return SVT_HIP_ERR_BAD_PARAM;
    if (!send_dev || !recv_dev || !offsets || !bytes) return svt_hip_fail(c->ctx, SVT_HIP_ERR_BAD_PARAM, "all-gather-v: null argument"); // before any stream is touched
    if (int rc = exchange_begin(c, slot)) return rc;
    // one broadcast per rank, fused into one group: rank r's `bytes[r]` live bytes land at recv + offsets[r] on every rank
    RCCL_CHECK(c->ctx, g_rccl.GroupStart());
    for (int r = 0; r < c->world; r++)
        if (bytes[r]) {
            ncclResult_t e = g_rccl.Broadcast(r == c->rank ? send_dev : static_cast<const uint8_t *>(recv_dev) + offsets[r], static_cast<uint8_t *>(recv_dev) + offsets[r],
                                              bytes[r], ncclUint8, r, c->comm, c->stream);
            if (e != ncclSuccess) {
                g_rccl.GroupEnd();
                return svt_hip_fail(c->ctx, SVT_HIP_ERR_LAUNCH, "ncclBroadcast(root %d) failed: %s", r, g_rccl.GetErrorString(e));
            }
        }
    RCCL_CHECK(c->ctx, g_rccl.GroupEnd());
    return exchange_end(c, slot);
}

int svt_hip_comm_stream_wait(SvtHipComm *c, int slot) {
    if (!c || slot < 0 || slot >= SVT_HIP_COMM_SLOTS) return SVT_HIP_ERR_BAD_PARAM;
    if (c->used[slot]) SVT_HIP_CHECK(c->ctx, hipStreamWaitEvent(c->ctx->stream, c->landed[slot], 0));
    return SVT_HIP_OK;
}

int svt_hip_comm_sync(SvtHipComm *c) {
    if (!c) return SVT_HIP_ERR_BAD_PARAM;
    SVT_HIP_CHECK(c->ctx, hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}

// the number of ranks RCCL itself reports for the communicator (ncclCommCount): what a caller records next to a multi-GPU measurement
int svt_hip_comm_count(SvtHipComm *c, int *ranks) {
    if (!c || !ranks) return SVT_HIP_ERR_BAD_PARAM;
    RCCL_CHECK(c->ctx, g_rccl.CommCount(c->comm, ranks));
    return SVT_HIP_OK;
}

int svt_hip_comm_rank(const SvtHipComm *c) { return c ? c->rank : -1; }
int svt_hip_comm_world(const SvtHipComm *c) { return c ? c->world : 0; }

} // extern "C"
