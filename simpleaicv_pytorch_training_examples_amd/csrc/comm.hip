// Gradient all-reduce side of the data-parallel step, on RCCL directly (no torch types, no Python in the loop).
//
// What it replaces: the reducer of nn.parallel.DistributedDataParallel that the reference wraps its model in
// (reference tools/train_classification_model.py:217-227, tools/scripts.py:183-226): contiguous ranges of the flat
// fp32 gradient arena ("buckets") are averaged over the ranks as soon as their last gradient has been produced.
//
// Inside a hipGraph capture (engine.StepGraph) the collectives go to a communication stream and overlap backward:
//   compute stream      ... wgrad kernels of bucket b ...  record(ev_in) ............ wait(ev_out) optimizer
//   communication stream                                   wait(ev_in) ncclAllReduce(bucket b) ... record(ev_out)
// In an eagerly launched step they run on the producing stream itself, in order (see on_side_stream() below for the
// measurement behind that choice).
//
// One communicator per process (= per GPU); every call only enqueues.
// RCCL is resolved at run time with dlopen/dlsym: the copy PyTorch already loaded (soname librccl.so.1) when the
// host is the Python mirror, the ROCm one for a C/C++ host -- the library has no link-time RCCL dependency, and a
// process must never hold two RCCL copies.  xGMI is point-to-point (7 links per GPU): the bucket size is the caller's
// knob (engine.DistributedDataParallel: 48 MiB buckets, a small last bucket so the tail of backward is short).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "saicv_internal.h"
#include "../../include/saicv_hip.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_why[256] = "symbols missing";      // why loading failed (dlerror() may be read only once)

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);          // the copy the process already holds, if any
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* e = dlerror();              // one call: it clears the error it returns
        snprintf(g_rccl_why, sizeof(g_rccl_why), "%s", e ? e : "dlopen failed");
        return;
    }
    g_rccl.handle = h;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(dlsym(h, "ncclBroadcast"));
    g_rccl.ReduceScatter = reinterpret_cast<decltype(g_rccl.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce &&
                g_rccl.Broadcast && g_rccl.ReduceScatter && g_rccl.AllGather && g_rccl.GetErrorString;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) {
        saicv::set_error("saicv_comm: RCCL (librccl.so.1) could not be loaded: %s", g_rccl_why);
        return nullptr;
    }
    return &g_rccl;
}

#define COMM_HIP(call, what)                                                              \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            saicv::set_error("saicv_comm %s: %s", what, hipGetErrorString(e_));           \
            return -2;                                                                    \
        }                                                                                 \
    } while (0)

#define COMM_RCCL(r, call, what)                                                          \
    do {                                                                                  \
        ncclResult_t n_ = (call);                                                         \
        if (n_ != ncclSuccess) {                                                          \
            saicv::set_error("saicv_comm %s: RCCL: %s", what, (r)->GetErrorString(n_));   \
            return -3;                                                                    \
        }                                                                                 \
    } while (0)

}  // namespace

struct saicv_comm {
    ncclComm_t nccl = nullptr;
    hipStream_t side = nullptr;       // communication stream
    hipEvent_t ev_in = nullptr;       // producer stream -> communication stream
    hipEvent_t ev_out = nullptr;      // communication stream -> consumer stream
    int world = 0, rank = 0;
    bool overlap_eager = false;       // SAICV_COMM_MODE=events: communication stream also outside graph capture
    unsigned long long buckets = 0;   // buckets enqueued since creation
    unsigned long long bytes = 0;
};

namespace {

// Where a collective runs.  Inside a hipGraph capture: on the communication stream, ordered by events -- they become graph
// edges and the all-reduce overlaps the rest of backward.  In an eagerly launched step: on the PRODUCER stream itself.
// Measured on MI355X (profiles/r02_ddp_eager_path.md): as soon as the communication stream has to be ordered after a point
// of a busy compute stream -- event wait, stream wait-value, or a helper thread that waits on the host and then launches
// -- every kernel of the step runs 20-40 us longer (ResNet-50: 23.6 -> 31-38 ms per step in a world of one), whereas a
// ring all-reduce of the whole 102 MB gradient costs well under a millisecond on xGMI.  Serialising it is the cheap side.
bool on_side_stream(const saicv_comm* c, void* stream) {
    if (c->overlap_eager) return true;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cap != hipStreamCaptureStatusNone;
}

}  // namespace

// ranks of the largest LIVE communicator of this process (common.h): igemm.hip sizes the weight-gradient round with it.
// Recomputed when a communicator is destroyed (it only ever grew, so a single-GPU model trained later in the same
// process kept the 85 % split).
int g_saicv_comm_world = 1;
static int g_live_worlds[64];
static int g_n_live = 0;
static void comm_world_changed() {
    int w = 1;
    for (int i = 0; i < g_n_live; ++i) w = g_live_worlds[i] > w ? g_live_worlds[i] : w;
    g_saicv_comm_world = w;
}

extern "C" {

int saicv_comm_available(void) {
    return rccl() ? 0 : -1;
}

int saicv_comm_unique_id(void* id128) {
    const Rccl* r = rccl();
    if (!r) return -1;
    if (!id128) { saicv::set_error("saicv_comm_unique_id: null buffer"); return -1; }
    ncclUniqueId id;
    COMM_RCCL(r, r->GetUniqueId(&id), "unique_id");
    ::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int saicv_comm_create(const void* id128, int world, int rank, saicv_comm** out) {
    if (!id128 || !out) { saicv::set_error("saicv_comm_create: null argument"); return -1; }
    if (world < 1 || rank < 0 || rank >= world) {
        saicv::set_error("saicv_comm_create: rank %d outside world of %d", rank, world);
        return -1;
    }
    const Rccl* r = rccl();
    if (!r) return -1;
    saicv_comm* c = new saicv_comm();
    c->world = world;
    c->rank = rank;
    ncclUniqueId id;
    ::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t n = r->CommInitRank(&c->nccl, world, id, rank);       // collective over the ranks; uses the current device
    if (n != ncclSuccess) {
        saicv::set_error("saicv_comm_create: ncclCommInitRank: %s", r->GetErrorString(n));
        delete c;
        return -3;
    }
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);                         // hi = numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
        saicv::set_error("saicv_comm_create: stream / event creation failed: %s", hipGetErrorString(hipGetLastError()));
        saicv_comm_destroy(c);
        return -2;
    }
    const char* mode = getenv("SAICV_COMM_MODE");
    c->overlap_eager = mode && strcmp(mode, "events") == 0;
    if (g_n_live < 64) g_live_worlds[g_n_live++] = world;             // collectives now share this GPU with the step's kernels
    comm_world_changed();
    *out = c;
    return 0;
}

int saicv_comm_allreduce_bucket(saicv_comm* c, float* grads, size_t n, int average, void* producer_stream) {
    if (!c || !c->nccl) { saicv::set_error("saicv_comm_allreduce_bucket: no communicator"); return -1; }
    if (!grads && n) { saicv::set_error("saicv_comm_allreduce_bucket: null bucket"); return -1; }
    if (n == 0) return 0;
    const Rccl* r = rccl();
    if (!r) return -1;
    if (on_side_stream(c, producer_stream)) {
        COMM_HIP(hipEventRecord(c->ev_in, static_cast<hipStream_t>(producer_stream)), "allreduce_bucket (record)");
        COMM_HIP(hipStreamWaitEvent(c->side, c->ev_in, 0), "allreduce_bucket (wait)");
        COMM_RCCL(r, r->AllReduce(grads, grads, n, ncclFloat32, average ? ncclAvg : ncclSum, c->nccl, c->side),
                  "allreduce_bucket");
    } else {
        COMM_RCCL(r, r->AllReduce(grads, grads, n, ncclFloat32, average ? ncclAvg : ncclSum, c->nccl,
                                  static_cast<hipStream_t>(producer_stream)), "allreduce_bucket");
    }
    c->buckets += 1;
    c->bytes += n * sizeof(float);
    return 0;
}

// stream a collective of this communicator runs on, ordered after producer_stream's work so far (see on_side_stream)
static int comm_stream_for(saicv_comm* c, void* producer_stream, hipStream_t* out, const char* what) {
    if (on_side_stream(c, producer_stream)) {
        COMM_HIP(hipEventRecord(c->ev_in, static_cast<hipStream_t>(producer_stream)), what);
        COMM_HIP(hipStreamWaitEvent(c->side, c->ev_in, 0), what);
        *out = c->side;
    } else {
        *out = static_cast<hipStream_t>(producer_stream);
    }
    return 0;
}

int saicv_comm_reduce_scatter(saicv_comm* c, const float* grads, float* shard, size_t n_per_rank, int average, void* producer_stream) {
    if (!c || !c->nccl) { saicv::set_error("saicv_comm_reduce_scatter: no communicator"); return -1; }
    if (n_per_rank == 0) return 0;
    if (!grads || !shard) { saicv::set_error("saicv_comm_reduce_scatter: null buffer"); return -1; }
    const Rccl* r = rccl();
    if (!r) return -1;
    hipStream_t s = nullptr;
    if (int e = comm_stream_for(c, producer_stream, &s, "reduce_scatter (order)")) return e;
    COMM_RCCL(r, r->ReduceScatter(grads, shard, n_per_rank, ncclFloat32, average ? ncclAvg : ncclSum, c->nccl, s), "reduce_scatter");
    c->buckets += 1;
    c->bytes += n_per_rank * (size_t)c->world * sizeof(float);
    return 0;
}

int saicv_comm_all_gather(saicv_comm* c, const float* shard, float* buf, size_t n_per_rank, void* producer_stream) {
    if (!c || !c->nccl) { saicv::set_error("saicv_comm_all_gather: no communicator"); return -1; }
    if (n_per_rank == 0) return 0;
    if (!shard || !buf) { saicv::set_error("saicv_comm_all_gather: null buffer"); return -1; }
    const Rccl* r = rccl();
    if (!r) return -1;
    hipStream_t s = nullptr;
    if (int e = comm_stream_for(c, producer_stream, &s, "all_gather (order)")) return e;
    COMM_RCCL(r, r->AllGather(shard, buf, n_per_rank, ncclFloat32, c->nccl, s), "all_gather");
    c->buckets += 1;
    c->bytes += n_per_rank * (size_t)c->world * sizeof(float);
    return 0;
}

int saicv_comm_broadcast(saicv_comm* c, void* buf, size_t bytes, int root, void* stream) {
    if (!c || !c->nccl) { saicv::set_error("saicv_comm_broadcast: no communicator"); return -1; }
    if (root < 0 || root >= c->world) { saicv::set_error("saicv_comm_broadcast: root %d outside world of %d", root, c->world); return -1; }
    if (bytes == 0) return 0;
    if (!buf) { saicv::set_error("saicv_comm_broadcast: null buffer"); return -1; }
    const Rccl* r = rccl();
    if (!r) return -1;
    // every collective of this communicator runs on ITS stream, in issue order; the caller's stream is ordered before
    // (buf is ready) and after (buf is overwritten) by events
    hipStream_t user = static_cast<hipStream_t>(stream);
    if (!on_side_stream(c, stream)) {
        COMM_RCCL(r, r->Broadcast(buf, buf, bytes, ncclUint8, root, c->nccl, user), "broadcast");
        return 0;
    }
    COMM_HIP(hipEventRecord(c->ev_in, user), "broadcast (record)");
    COMM_HIP(hipStreamWaitEvent(c->side, c->ev_in, 0), "broadcast (wait)");
    COMM_RCCL(r, r->Broadcast(buf, buf, bytes, ncclUint8, root, c->nccl, c->side), "broadcast");
    COMM_HIP(hipEventRecord(c->ev_out, c->side), "broadcast (record out)");
    COMM_HIP(hipStreamWaitEvent(user, c->ev_out, 0), "broadcast (wait out)");
    return 0;
}

int saicv_comm_join(saicv_comm* c, void* consumer_stream) {
    if (!c) { saicv::set_error("saicv_comm_join: no communicator"); return -1; }
    if (!on_side_stream(c, consumer_stream)) return 0;        // the collectives ran on the consumer's own stream
    COMM_HIP(hipEventRecord(c->ev_out, c->side), "join (record)");
    COMM_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), c->ev_out, 0), "join (wait)");
    return 0;
}

int saicv_comm_stats(const saicv_comm* c, int* world, int* rank, unsigned long long* buckets, unsigned long long* bytes) {
    if (!c) { saicv::set_error("saicv_comm_stats: no communicator"); return -1; }
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (buckets) *buckets = c->buckets;
    if (bytes) *bytes = c->bytes;
    return 0;
}

int saicv_comm_destroy(saicv_comm* c) {
    if (!c) return 0;
    if (c->side) hipStreamSynchronize(c->side);
    if (c->nccl && g_rccl.ok) g_rccl.CommDestroy(c->nccl);
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_out) hipEventDestroy(c->ev_out);
    if (c->side) hipStreamDestroy(c->side);
    for (int i = 0; i < g_n_live; ++i)
        if (g_live_worlds[i] == c->world) { g_live_worlds[i] = g_live_worlds[--g_n_live]; break; }
    comm_world_changed();
    delete c;
    return 0;
}

}  // extern "C"
