// cluster.hip -- the id-range shards of one node behind ONE handle, for a single-process caller (the Go shim):
// kdb_cluster_create / kdb_sharded_search_batch / kdb_sharded_flat_scan_batch.
//
// The reference is single process and has no counterpart; north_star defines the path: the index shards by vector-id
// range across the GPUs of one node, a query batch visits every shard, and ONE all-gather of the per-shard top-k over
// xGMI (RCCL) precedes the merge.  Per call:
//   1. the queries reach device 0 by one H2D copy and the other devices by an RCCL broadcast (xGMI, not n x PCIe);
//   2. every shard runs kdb_search_batch_dev / kdb_flat_scan_batch_dev on its own device and stream, writing the packed
//      block ids[B][k] | raw distances[B][k] | count[B] straight into its slot of that device's send buffer
//      (shards_per_device consecutive slots);
//   3. ONE ncclAllGather (grouped over the devices of this process): every device ends up with all G blocks;
//   4. the call's ROOT device merges G*k candidates per query (merge_topk_kernel: total order (key, global id), global id =
//      id_base[g] + local id) and the answers go back in one D2H copy.
// Calls of several threads overlap (round 3).  A cluster keeps KDB_CLANES = 2 lanes: per device a stream, query / send /
// receive buffers and allow-list slices of its own; a call takes the next lane (and holds it to its end), enqueues under one
// short lock and then waits -- outside that lock -- for ONE event on its root device's stream: the copy of the answers is
// the last operation of the call, and everything else of the call precedes it through stream order and events.  Queries,
// sliced allow lists and answers pass through a page-locked buffer of the lane (copies from / to the caller's pageable
// memory would hold the enqueuing thread -- and with it the enqueue lock -- until the device gets there).  So while
// call i is in its all-gather / merge / D2H, call i+1 (other lane) is already walking the shards (every kdb_index keeps two
// scratch sets for exactly this).  The collectives of BOTH lanes go through ONE communicator and ONE collective stream per
// device, in the order the calls were enqueued -- the same order on every device, which is what RCCL needs to be
// deadlock-free; events tie a lane's streams to it (queries ready -> broadcast -> walks -> all-gather -> merge).
// Why all-gather and not a gather to the root: north_star names it; on point-to-point xGMI every peer has its own link, so
// delivering a 0.7 MB block to seven peers takes the time of delivering it to one; and it makes every device a possible root:
// lane 0 merges on device 0, lane 1 on device 1 (when there is one), so consecutive calls do not queue behind one device's
// merge and PCIe link.
// int8 shards exchange and merge their distances as the reference's float64 (hnsw_index.go:2429-2454 computes and orders
// doubles): block = dist64[B][k] | ids[B][k] | count[B]; KDB_SEARCH_DIST_F64 hands the doubles to the caller, without it they
// are rounded to float on the way out -- the ORDER is the float64 order either way.
// RCCL is resolved at run time (dlopen librccl.so.1): the library carries no link-time dependency on it, and a process that
// never creates a cluster never loads it.  With one device the collective calls still run (one rank), so the 1-GPU box of
// the test rig exercises the same code as an 8-GPU node.
// The one-process-per-GPU deployment (torchrun; bench.py --gpus N) is kektordb_amd/shard.py: same packed block, same merge
// kernel, torch.distributed's RCCL communicator instead of ncclCommInitAll.
#include "kdb_internal.h"
#include <atomic>
#include <dlfcn.h>
#include <string.h>
#include <vector>

namespace {

typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t; // ncclSuccess = 0
enum { KDB_NCCL_INT32 = 2 };
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.so, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.so, "ncclCommAbort");
        r.CommCount = (decltype(r.CommCount))dlsym(r.so, "ncclCommCount");
        r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
        r.Broadcast = (decltype(r.Broadcast))dlsym(r.so, "ncclBroadcast");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.so, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.so, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.Broadcast && r.GroupStart && r.GroupEnd;
    });
    return r;
}

#define KDB_NCCL(call)                                                                                      \
    do {                                                                                                    \
        ncclResult_t _r = (call);                                                                           \
        if (_r != 0) {                                                                                      \
            kdb_set_error("%s failed: %s", #call, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?"); \
            return KDB_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)

constexpr int KDB_CLANES = 2;

struct LaneDev { // one lane's resources on one device
    hipStream_t stream = nullptr;
    hipEvent_t ev_ready = nullptr;  // this lane's inputs of the next collective are ready (recorded on `stream`)
    hipEvent_t ev_coll = nullptr;   // the collective has delivered (recorded on the device's collective stream)
    float *d_q = nullptr;           // [B][dim] queries
    size_t q_bytes = 0;
    uint32_t *d_send = nullptr;     // [spd][L] this device's packed blocks
    uint32_t *d_recv = nullptr;     // [n_dev][spd][L] everybody's
    size_t send_bytes = 0, recv_bytes = 0;
    std::vector<uint64_t *> d_allow; // per local shard: local allow bitset (or empty)
    std::vector<size_t> allow_bytes;
    uint64_t *d_allow_g = nullptr;   // the caller's GLOBAL allow list, uploaded once per device and call; sliced by a kernel
    size_t allow_g_bytes = 0;
    uint32_t *d_bases = nullptr;    // [G] (root devices only)
    uint32_t *d_out = nullptr;      // merged ids | dist | count (root devices only)
    size_t out_bytes = 0;
};

struct DevSlot {
    int device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t coll = nullptr;     // every collective of this device, of either lane, in enqueue order
    LaneDev lane[KDB_CLANES];
};

struct Lane {
    std::mutex mu;                  // held by the call that uses the lane, from its first enqueue to its last wait
    uint32_t root = 0;              // index into devs: where this lane's calls merge
    hipEvent_t done = nullptr;      // on the root's lane stream: the answers have reached the lane's page-locked buffer
    unsigned char *h_pin = nullptr; // page-locked (portable): queries | sliced allow lists | merged answers of the call in
    size_t h_pin_bytes = 0;         // flight -- every copy of a call is truly asynchronous, nothing waits under `enq`
    size_t h_out = 0;               // offset of the answers (dist | ids | count) of the call in flight
};

} // namespace

struct kdb_cluster {
    std::vector<kdb_index *> shards; // device-major order
    std::vector<uint32_t> id_base;
    std::vector<DevSlot> devs;
    uint32_t spd = 1;                // shards per device
    uint32_t dim = 0, metric = 0, precision = 0;
    Lane lanes[KDB_CLANES];
    std::mutex enq;                  // enqueue order = collective order on every device
    uint64_t seq = 0;                // under enq_pick
    std::mutex enq_pick;
    // A failure INSIDE an RCCL group leaves the communicators with a collective queued on some devices and not on others: the
    // next collective would wait for ever.  The handle is then POISONED: every communicator is aborted (ncclCommAbort ends
    // whatever is queued), and every later call returns KDB_ERR_STATE at once -- destroy the cluster and create a new one.
    std::atomic<bool> poisoned{false};
    std::string poison_why;          // written once, before `poisoned` is set
    std::atomic<uint32_t> inject{0}; // test hook (kdb_cluster_debug_fail_next): fail inside the next 1 = broadcast, 2 = all-gather group
};

extern "C" void kdb_cluster_destroy(kdb_cluster *c) {
    if (!c) return;
    for (DevSlot &d : c->devs) {
        (void)hipSetDevice(d.device);
        (void)hipDeviceSynchronize();
        if (d.comm && rccl().ok) (void)rccl().CommDestroy(d.comm);
        for (LaneDev &l : d.lane) {
            for (void *p : {(void *)l.d_q, (void *)l.d_send, (void *)l.d_recv, (void *)l.d_bases, (void *)l.d_out, (void *)l.d_allow_g})
                if (p) (void)hipFree(p);
            for (uint64_t *p : l.d_allow)
                if (p) (void)hipFree(p);
            if (l.ev_ready) (void)hipEventDestroy(l.ev_ready);
            if (l.ev_coll) (void)hipEventDestroy(l.ev_coll);
            if (l.stream) (void)hipStreamDestroy(l.stream);
        }
        if (d.coll) (void)hipStreamDestroy(d.coll);
    }
    for (Lane &l : c->lanes) {
        if (!c->devs.empty()) (void)hipSetDevice(c->devs[l.root].device);
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.h_pin) (void)hipHostFree(l.h_pin);
    }
    delete c;
}

extern "C" int kdb_cluster_create(kdb_index *const *shards, const uint32_t *id_base, uint32_t n_shards, kdb_cluster **out) {
    if (!shards || !id_base || !out || n_shards == 0) {
        kdb_set_error("cluster_create: null argument or no shard");
        return KDB_ERR_INVALID;
    }
    *out = nullptr;
    for (uint32_t g = 0; g < n_shards; g++) {
        if (!shards[g]) {
            kdb_set_error("cluster_create: shard %u is null", g);
            return KDB_ERR_INVALID;
        }
        const kdb_index_desc &a = shards[0]->desc, &b = shards[g]->desc;
        if (a.dim != b.dim || a.metric != b.metric || a.precision != b.precision) {
            kdb_set_error("cluster_create: shard %u differs from shard 0 in dim / metric / precision", g);
            return KDB_ERR_INVALID;
        }
        if (g && id_base[g] < id_base[g - 1]) {
            kdb_set_error("cluster_create: id bases must ascend (shard g owns the ids behind id_base[g])");
            return KDB_ERR_INVALID;
        }
    }
    // shards of one device must be consecutive and every device must hold the same number (equal-sized all-gather blocks)
    std::vector<int> dev_of;
    std::vector<uint32_t> per_dev;
    for (uint32_t g = 0; g < n_shards; g++) {
        const int d = shards[g]->device;
        if (dev_of.empty() || dev_of.back() != d) {
            for (int seen : dev_of)
                if (seen == d) {
                    kdb_set_error("cluster_create: the shards of device %d are not consecutive", d);
                    return KDB_ERR_INVALID;
                }
            dev_of.push_back(d);
            per_dev.push_back(0);
        }
        per_dev.back()++;
    }
    for (uint32_t n : per_dev)
        if (n != per_dev[0]) {
            kdb_set_error("cluster_create: every device must hold the same number of shards");
            return KDB_ERR_INVALID;
        }
    if (!rccl().ok) {
        kdb_set_error("cluster_create: RCCL (librccl.so.1) could not be loaded");
        return KDB_ERR_UNSUPPORTED;
    }
    kdb_cluster *c = new (std::nothrow) kdb_cluster();
    if (!c) return KDB_ERR_OOM;
    c->shards.assign(shards, shards + n_shards);
    c->id_base.assign(id_base, id_base + n_shards);
    c->spd = per_dev[0];
    c->dim = shards[0]->desc.dim;
    c->metric = shards[0]->desc.metric;
    c->precision = shards[0]->desc.precision;
    c->devs.resize(dev_of.size());
    auto fail = [&](int code) {
        kdb_cluster_destroy(c);
        return code;
    };
    std::vector<ncclComm_t> comms(dev_of.size());
    {
        ncclResult_t r = rccl().CommInitAll(comms.data(), (int)dev_of.size(), dev_of.data());
        if (r != 0) {
            kdb_set_error("ncclCommInitAll over %zu device(s) failed: %s", dev_of.size(), rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
            return fail(KDB_ERR_HIP);
        }
    }
    for (size_t i = 0; i < dev_of.size(); i++) { // every communicator has an owner before anything else can fail
        c->devs[i].device = dev_of[i];
        c->devs[i].comm = comms[i];
    }
    for (size_t i = 0; i < dev_of.size(); i++) {
        DevSlot &d = c->devs[i];
        bool ok = hipSetDevice(d.device) == hipSuccess && hipStreamCreateWithFlags(&d.coll, hipStreamNonBlocking) == hipSuccess;
        for (LaneDev &l : d.lane) {
            l.d_allow.assign(c->spd, nullptr);
            l.allow_bytes.assign(c->spd, 0);
            ok = ok && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&l.ev_ready, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&l.ev_coll, hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) {
            kdb_set_error("cluster_create: stream / event creation failed on device %d", d.device);
            return fail(KDB_ERR_HIP);
        }
    }
    for (int li = 0; li < KDB_CLANES; li++) { // lane li merges on device li (mod the number of devices)
        Lane &l = c->lanes[li];
        l.root = (uint32_t)li % (uint32_t)c->devs.size();
        LaneDev &r = c->devs[l.root].lane[li];
        if (hipSetDevice(c->devs[l.root].device) != hipSuccess || hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess ||
            hipMalloc(&r.d_bases, (size_t)n_shards * 4) != hipSuccess ||
            hipMemcpy(r.d_bases, id_base, (size_t)n_shards * 4, hipMemcpyHostToDevice) != hipSuccess) {
            kdb_set_error("cluster_create: id base upload failed");
            return fail(KDB_ERR_HIP);
        }
    }
    *out = c;
    return KDB_OK;
}

// A lane's device buffer grows: only that lane's stream and the device's collective stream ever touch it, and the lane is
// held by the calling thread -- so those two streams are all that has to drain (no device-wide wait; not under `enq`).
static int ensure_bytes(void **p, size_t *have, size_t want, hipStream_t lane_stream, hipStream_t coll) {
    if (*have >= want) return KDB_OK;
    if (*p) {
        KDB_HIP(hipStreamSynchronize(lane_stream));
        KDB_HIP(hipStreamSynchronize(coll));
        KDB_HIP(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    KDB_HIP(hipMalloc(p, want + want / 4));
    *have = want + want / 4;
    return KDB_OK;
}

// bits [base+1, base+count] of a dense GLOBAL bitset -> the shard's local bitset (local id i <-> global id base+i).
// One thread per 64-bit word of the slice; the global list was uploaded ONCE for the device, whatever the number of shards.
__global__ void slice_allow_kernel(const uint64_t *__restrict__ g, size_t g_words, uint32_t base, uint32_t count, uint64_t *__restrict__ out) {
    const size_t n_out = ((size_t)count >> 6) + 1;
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_out) return;
    const uint64_t gb = (uint64_t)base + 64ull * w; // local bits 64w .. 64w+63 = global bits base + 64w ..
    const size_t gw = (size_t)(gb >> 6);
    const unsigned sh = (unsigned)(gb & 63u);
    uint64_t v = gw < g_words ? g[gw] >> sh : 0ull;
    if (sh && gw + 1 < g_words) v |= g[gw + 1] << (64u - sh);
    if (w == 0) v &= ~1ull; // local id 0 does not exist
    if (w == n_out - 1) {   // ids above count
        const uint32_t last = count & 63u;
        v &= last == 63u ? ~0ull : ((2ull << last) - 1ull);
    }
    out[w] = v;
}

// Abort every communicator and refuse further calls.  Called with `enq` held, at most once.
static void poison(kdb_cluster *c, const char *why) {
    if (c->poisoned.load()) return;
    c->poison_why = why ? why : "?";
    for (DevSlot &d : c->devs) {
        (void)hipSetDevice(d.device);
        if (d.comm) { // ncclCommAbort ends the collectives that are queued or running without their peers
            if (rccl().CommAbort) (void)rccl().CommAbort(d.comm);
            d.comm = nullptr;
        }
    }
    c->poisoned.store(true);
}

// an RCCL group that is closed on every path; *inconsistent = some devices may hold a collective the others lack
template <typename F>
static int rccl_group(bool *inconsistent, F body) {
    KDB_NCCL(rccl().GroupStart());
    const int rc = body();
    const ncclResult_t e = rccl().GroupEnd();
    if (rc || e != 0) *inconsistent = true;
    if (rc) return rc;
    KDB_NCCL(e);
    return KDB_OK;
}

struct CallPlan { // sizes of one call (shared by the preparation and the enqueue step)
    uint32_t G, nd, spd;
    bool i8, out64;
    size_t bk, L, o_ids, o_dist, o_cnt, qbytes, allow_g_bytes, out_span, h_allow, h_out;
};
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Everything of a call that may ALLOCATE or wait for a stream: done with the lane held but WITHOUT the enqueue lock, so a
// batch larger than any before, or a 12.5 MB allow list, holds up nobody else's enqueue.
static int sharded_prepare(kdb_cluster *c, int li, const CallPlan &pl, const float *queries, const uint64_t *allow_bits) {
    Lane &lane = c->lanes[li];
    int rc;
    for (uint32_t i = 0; i < pl.nd; i++) {
        DevSlot &ds = c->devs[i];
        LaneDev &d = ds.lane[li];
        KDB_HIP(hipSetDevice(ds.device));
        if ((rc = ensure_bytes((void **)&d.d_q, &d.q_bytes, pl.qbytes, d.stream, ds.coll))) return rc;
        if ((rc = ensure_bytes((void **)&d.d_send, &d.send_bytes, (size_t)pl.spd * pl.L * 4, d.stream, ds.coll))) return rc;
        if ((rc = ensure_bytes((void **)&d.d_recv, &d.recv_bytes, (size_t)pl.G * pl.L * 4, d.stream, ds.coll))) return rc;
        if (allow_bits) {
            if ((rc = ensure_bytes((void **)&d.d_allow_g, &d.allow_g_bytes, pl.allow_g_bytes, d.stream, ds.coll))) return rc;
            for (uint32_t ls = 0; ls < pl.spd; ls++) {
                const size_t ab = (((size_t)c->shards[i * pl.spd + ls]->count >> 6) + 1) * 8;
                if ((rc = ensure_bytes((void **)&d.d_allow[ls], &d.allow_bytes[ls], ab, d.stream, ds.coll))) return rc;
            }
        }
    }
    DevSlot &rdev = c->devs[lane.root];
    LaneDev &root = rdev.lane[li];
    KDB_HIP(hipSetDevice(rdev.device));
    if ((rc = ensure_bytes((void **)&root.d_out, &root.out_bytes, pl.out_span + 16, root.stream, rdev.coll))) return rc;
    // the call's page-locked buffer: queries | the global allow list | answers.  The lane is ours (its last call has waited
    // for `done`), so nothing in flight uses the old one when it has to grow.
    const size_t pin_need = pl.h_out + al256(pl.out_span);
    if (lane.h_pin_bytes < pin_need) {
        if (lane.h_pin) (void)hipHostFree(lane.h_pin);
        lane.h_pin = nullptr;
        lane.h_pin_bytes = 0;
        KDB_HIP(hipHostMalloc((void **)&lane.h_pin, pin_need + pin_need / 4, hipHostMallocPortable));
        lane.h_pin_bytes = pin_need + pin_need / 4;
    }
    lane.h_out = pl.h_out;
    memcpy(lane.h_pin, queries, pl.qbytes);
    if (allow_bits) memcpy(lane.h_pin + pl.h_allow, allow_bits, pl.allow_g_bytes);
    return KDB_OK;
}

// everything of one call that is queued on the devices (under `enq`); the caller then waits for lane.done.
// *inconsistent: the failure happened inside an RCCL group -- the communicators must be aborted
static int sharded_enqueue(kdb_cluster *c, int li, const CallPlan &pl, bool flat, uint32_t B, uint32_t k, uint32_t ef, bool have_allow,
                           size_t allow_words, uint32_t flags, bool *inconsistent) {
    Lane &lane = c->lanes[li];
    const uint32_t G = pl.G, nd = pl.nd, spd = pl.spd;
    const size_t L = pl.L, bk = pl.bk;
    int rc;
    DevSlot &rdev = c->devs[lane.root];
    LaneDev &root = rdev.lane[li];
    const uint32_t inject = c->inject.exchange(0u);
    // 1. queries: H2D to the root, RCCL broadcast to the others (collective stream; the lanes' streams wait for it)
    KDB_HIP(hipSetDevice(rdev.device));
    KDB_HIP(hipMemcpyAsync(root.d_q, lane.h_pin, pl.qbytes, hipMemcpyHostToDevice, root.stream));
    if (nd > 1 || inject == 1u) {
        KDB_HIP(hipEventRecord(root.ev_ready, root.stream));
        KDB_HIP(hipStreamWaitEvent(rdev.coll, root.ev_ready, 0));
        rc = rccl_group(inconsistent, [&]() -> int {
            for (uint32_t i = 0; i < nd; i++) {
                KDB_HIP(hipSetDevice(c->devs[i].device));
                LaneDev &d = c->devs[i].lane[li];
                KDB_NCCL(rccl().Broadcast(d.d_q, d.d_q, pl.qbytes / 4, KDB_NCCL_INT32, (int)lane.root, c->devs[i].comm, c->devs[i].coll)); // in place at the root
                if (inject == 1u) {
                    kdb_set_error("injected failure inside the broadcast group (kdb_cluster_debug_fail_next)");
                    return KDB_ERR_HIP;
                }
            }
            return KDB_OK;
        });
        if (rc) return rc;
        for (uint32_t i = 0; i < nd; i++) {
            KDB_HIP(hipSetDevice(c->devs[i].device));
            LaneDev &d = c->devs[i].lane[li];
            KDB_HIP(hipEventRecord(d.ev_coll, c->devs[i].coll));
            KDB_HIP(hipStreamWaitEvent(d.stream, d.ev_coll, 0));
        }
    }
    // 2. every shard searches on its device, into its slot of the send buffer.  No host wait and no host work in this loop:
    //    a global allow list goes to every device ONCE (from the lane's page-locked buffer, each device over its own PCIe
    //    link) and one small kernel per shard cuts the shard's slice out of it
    const uint32_t sflags = pl.i8 ? (flags | KDB_SEARCH_DIST_F64) : flags;
    for (uint32_t g = 0; g < G; g++) {
        LaneDev &d = c->devs[g / spd].lane[li];
        const uint32_t ls = g % spd;
        kdb_index *idx = c->shards[g];
        KDB_HIP(hipSetDevice(c->devs[g / spd].device));
        const uint64_t *d_allow = nullptr;
        if (have_allow) {
            if (ls == 0) KDB_HIP(hipMemcpyAsync(d.d_allow_g, lane.h_pin + pl.h_allow, pl.allow_g_bytes, hipMemcpyHostToDevice, d.stream));
            const size_t n_out = ((size_t)idx->count >> 6) + 1;
            hipLaunchKernelGGL(slice_allow_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, d.stream, d.d_allow_g, allow_words, c->id_base[g],
                               idx->count, d.d_allow[ls]);
            KDB_HIP(hipGetLastError());
            d_allow = d.d_allow[ls];
        }
        uint32_t *blk = d.d_send + (size_t)ls * L;
        rc = flat ? kdb_flat_scan_batch_dev(idx, d.d_q, B, k, d_allow, sflags, blk + pl.o_ids, reinterpret_cast<float *>(blk + pl.o_dist), blk + pl.o_cnt, d.stream)
                  : kdb_search_batch_dev(idx, d.d_q, B, k, ef, d_allow, sflags, blk + pl.o_ids, reinterpret_cast<float *>(blk + pl.o_dist), blk + pl.o_cnt, d.stream);
        if (rc) return rc;
    }
    // 3. the one exchange step: all-gather of the packed blocks over xGMI (collective stream, behind the walks)
    for (uint32_t i = 0; i < nd; i++) {
        KDB_HIP(hipSetDevice(c->devs[i].device));
        LaneDev &d = c->devs[i].lane[li];
        KDB_HIP(hipEventRecord(d.ev_ready, d.stream));
        KDB_HIP(hipStreamWaitEvent(c->devs[i].coll, d.ev_ready, 0));
    }
    rc = rccl_group(inconsistent, [&]() -> int {
        for (uint32_t i = 0; i < nd; i++) {
            if (inject == 2u && i + 1 == nd) { // every device but the last has its all-gather queued: the worst case
                kdb_set_error("injected failure inside the all-gather group (kdb_cluster_debug_fail_next)");
                return KDB_ERR_HIP;
            }
            KDB_HIP(hipSetDevice(c->devs[i].device));
            LaneDev &d = c->devs[i].lane[li];
            KDB_NCCL(rccl().AllGather(d.d_send, d.d_recv, (size_t)spd * L, KDB_NCCL_INT32, c->devs[i].comm, c->devs[i].coll));
        }
        return KDB_OK;
    });
    if (rc) return rc;
    // 4. merge on the root, answers home
    KDB_HIP(hipSetDevice(rdev.device));
    KDB_HIP(hipEventRecord(root.ev_coll, rdev.coll));
    KDB_HIP(hipStreamWaitEvent(root.stream, root.ev_coll, 0));
    unsigned char *ob = reinterpret_cast<unsigned char *>(root.d_out);
    void *m_dist = ob; // (8-byte distances first)
    uint32_t *m_ids = reinterpret_cast<uint32_t *>(ob + bk * 8);
    uint32_t *m_cnt = m_ids + bk;
    if (pl.i8) {
        rc = kdb_launch_merge_topk_f64(G, B, k, root.d_recv + pl.o_ids, reinterpret_cast<const double *>(root.d_recv + pl.o_dist), root.d_recv + pl.o_cnt,
                                       L, L / 2, L, root.d_bases, m_ids, m_dist, pl.out64 ? 1 : 0, m_cnt, root.stream);
    } else {
        const int negate = c->metric == KDB_METRIC_COSINE && c->precision == KDB_PREC_F32;
        rc = kdb_launch_merge_topk(negate, G, B, k, root.d_recv + pl.o_ids, reinterpret_cast<const float *>(root.d_recv + pl.o_dist),
                                   root.d_recv + pl.o_cnt, L, L, root.d_bases, m_ids, reinterpret_cast<float *>(m_dist), m_cnt, root.stream);
    }
    if (rc) return rc;
    KDB_HIP(hipMemcpyAsync(lane.h_pin + lane.h_out, ob, pl.out_span, hipMemcpyDeviceToHost, root.stream)); // dist | ids | count, one copy
    KDB_HIP(hipEventRecord(lane.done, root.stream));
    return KDB_OK;
}

static int sharded_call(kdb_cluster *c, bool flat, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                        const uint64_t *allow_bits, size_t allow_words, uint32_t flags, uint32_t *out_ids, float *out_dist,
                        uint32_t *out_count) {
    if (!c) {
        kdb_set_error("null cluster handle");
        return KDB_ERR_INVALID;
    }
    if (c->poisoned.load()) {
        kdb_set_error("cluster poisoned by an earlier failure inside an RCCL group (%s): its communicators were aborted -- destroy it and create a new one",
                      c->poison_why.c_str());
        return KDB_ERR_STATE;
    }
    if (B == 0) return KDB_OK;
    if (!queries || !out_ids || !out_dist || !out_count || k == 0) {
        kdb_set_error("sharded search: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    if ((flags & KDB_SEARCH_DIST_F64) && c->precision != KDB_PREC_I8) {
        kdb_set_error("sharded search: KDB_SEARCH_DIST_F64 applies to int8 shards (the other precisions compute float32 distances)");
        return KDB_ERR_INVALID;
    }
    if (allow_bits && allow_words == 0) allow_bits = nullptr; // (a list without words names nothing the library could slice: treated as absent)
    CallPlan pl;
    pl.G = (uint32_t)c->shards.size();
    pl.nd = (uint32_t)c->devs.size();
    pl.spd = c->spd;
    pl.i8 = c->precision == KDB_PREC_I8; // distances travel as float64
    pl.out64 = (flags & KDB_SEARCH_DIST_F64) != 0;
    pl.bk = (size_t)B * k;
    // packed block in 32-bit words: float shards ids | dist | count; int8 shards dist64 | ids | count (8-byte aligned blocks)
    pl.L = pl.i8 ? ((3 * pl.bk + B + 1) & ~(size_t)1) : 2 * pl.bk + B;
    pl.o_ids = pl.i8 ? 2 * pl.bk : 0;
    pl.o_dist = pl.i8 ? 0 : pl.bk;
    pl.o_cnt = pl.i8 ? 3 * pl.bk : 2 * pl.bk;
    pl.qbytes = (size_t)B * c->dim * 4;
    pl.allow_g_bytes = allow_bits ? allow_words * 8 : 0;
    pl.out_span = pl.bk * 8 + pl.bk * 4 + (size_t)B * 4;
    pl.h_allow = al256(pl.qbytes);
    pl.h_out = pl.h_allow + al256(pl.allow_g_bytes);
    int li;
    {
        std::lock_guard<std::mutex> pk(c->enq_pick);
        li = (int)(c->seq++ % KDB_CLANES);
    }
    Lane &lane = c->lanes[li];
    std::lock_guard<std::mutex> hold(lane.mu); // the lane's buffers belong to this call until its answers are home
    int rc = sharded_prepare(c, li, pl, queries, allow_bits);
    if (rc) return rc; // nothing was queued
    bool inconsistent = false;
    {
        std::lock_guard<std::mutex> lk(c->enq);
        if (c->poisoned.load()) {
            kdb_set_error("cluster poisoned by an earlier failure inside an RCCL group (%s)", c->poison_why.c_str());
            return KDB_ERR_STATE;
        }
        rc = sharded_enqueue(c, li, pl, flat, B, k, ef, allow_bits != nullptr, allow_words, flags, &inconsistent);
        if (rc != KDB_OK && inconsistent) {
            const std::string why = kdb_last_error(); // (poison() runs HIP / RCCL calls of its own)
            poison(c, why.c_str());
            kdb_set_error("%s", why.c_str());
        }
    }
    if (rc != KDB_OK) { // whatever was queued still uses the lane's buffers: drain before the lane is handed on (after an
        // abort the collectives that wait for a peer have been ended, so this returns)
        for (DevSlot &d : c->devs) {
            (void)hipSetDevice(d.device);
            (void)hipStreamSynchronize(d.lane[li].stream);
            (void)hipStreamSynchronize(d.coll);
        }
        return rc;
    }
    // ONE wait, on the root's stream: the D2H copies are the call's last operations and every other piece of it precedes
    // them (walks -> all-gather -> merge).  The other devices' collective kernels of this call may still be delivering
    // blocks into THEIR receive buffers; the lane's next call orders itself behind them through the collective stream.
    KDB_HIP(hipSetDevice(c->devs[lane.root].device));
    KDB_HIP(hipEventSynchronize(lane.done));
    const size_t bk = pl.bk;
    const unsigned char *h = lane.h_pin + lane.h_out;
    memcpy(out_dist, h, bk * ((flags & KDB_SEARCH_DIST_F64) ? 8 : 4));
    memcpy(out_ids, h + bk * 8, bk * 4);
    memcpy(out_count, h + bk * 8 + bk * 4, (size_t)B * 4);
    return KDB_OK;
}

extern "C" int kdb_sharded_search_batch(kdb_cluster *c, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                                        const uint64_t *allow_bits, uint64_t allow_words, uint32_t flags, uint32_t *out_ids,
                                        float *out_dist, uint32_t *out_count) {
    return sharded_call(c, false, queries, B, k, ef, allow_bits, (size_t)allow_words, flags, out_ids, out_dist, out_count);
}

extern "C" int kdb_sharded_flat_scan_batch(kdb_cluster *c, const float *queries, uint32_t B, uint32_t k,
                                           const uint64_t *allow_bits, uint64_t allow_words, uint32_t flags, uint32_t *out_ids,
                                           float *out_dist, uint32_t *out_count) {
    return sharded_call(c, true, queries, B, k, 0, allow_bits, (size_t)allow_words, flags, out_ids, out_dist, out_count);
}

extern "C" int kdb_cluster_info(const kdb_cluster *c, uint32_t *n_shards, uint32_t *n_devices, uint32_t *shards_per_device) {
    if (!c) return KDB_ERR_INVALID;
    if (n_shards) *n_shards = (uint32_t)c->shards.size();
    if (n_devices) *n_devices = (uint32_t)c->devs.size();
    if (shards_per_device) *shards_per_device = c->spd;
    return KDB_OK;
}

// What the communicator itself reports: the number of ranks ncclCommInitAll gave it (= devices of the cluster; a bench line
// prints it as rccl_ranks_seen), and whether the handle has been poisoned.
extern "C" int kdb_cluster_comm_info(const kdb_cluster *c, uint32_t *ranks_in_communicator, uint32_t *poisoned) {
    if (!c) return KDB_ERR_INVALID;
    if (poisoned) *poisoned = c->poisoned.load() ? 1u : 0u;
    if (ranks_in_communicator) {
        *ranks_in_communicator = 0;
        if (!c->devs.empty() && c->devs[0].comm && rccl().CommCount) {
            int n = 0;
            if (rccl().CommCount(c->devs[0].comm, &n) == 0) *ranks_in_communicator = (uint32_t)n;
        }
    }
    return KDB_OK;
}

// Test hook: the next sharded call fails INSIDE the RCCL group of stage 1 (query broadcast) or 2 (all-gather) -- with the
// collective queued on some devices and not on the last -- so that the poisoning path can be exercised without broken hardware.
extern "C" int kdb_cluster_debug_fail_next(kdb_cluster *c, uint32_t stage) {
    if (!c || stage > 2u) return KDB_ERR_INVALID;
    c->inject.store(stage);
    return KDB_OK;
}
