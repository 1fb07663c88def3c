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
//   4. device 0 merges G*k candidates per query (merge_topk_kernel: total order (key, global id), global id = id_base[g] +
//      local id) and the answers go back in one D2H copy.
// RCCL is resolved at run time (dlopen librccl.so.1): the library carries no link-time dependency on it, and a process that
// never creates a cluster never loads it.  With one device the collective calls still run (one rank), so the 1-GPU box of
// the test rig exercises the same code as an 8-GPU node.
// The one-process-per-GPU deployment (torchrun; bench.py --gpus N) is kektordb_amd/shard.py: same packed block, same merge
// kernel, torch.distributed's RCCL communicator instead of ncclCommInitAll.
#include "kdb_internal.h"
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

struct DevSlot {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    ncclComm_t comm = nullptr;
    float *d_q = nullptr;          // [B][dim] queries
    size_t q_bytes = 0;
    uint32_t *d_send = nullptr;    // [spd][L] this device's packed blocks
    uint32_t *d_recv = nullptr;    // [n_dev][spd][L] everybody's
    size_t send_words = 0, recv_words = 0;
    std::vector<uint64_t *> d_allow; // per local shard: local allow bitset (or empty)
    std::vector<size_t> allow_words;
};

} // namespace

struct kdb_cluster {
    std::vector<kdb_index *> shards; // device-major order
    std::vector<uint32_t> id_base;
    std::vector<DevSlot> devs;
    uint32_t spd = 1;                // shards per device
    uint32_t dim = 0, metric = 0, precision = 0;
    uint32_t *d_bases = nullptr;     // [G] on device 0
    uint32_t *d_out = nullptr;       // merged ids | dist | count on device 0
    size_t out_words = 0;
    std::mutex mu;
};

extern "C" void kdb_cluster_destroy(kdb_cluster *c) {
    if (!c) return;
    for (DevSlot &d : c->devs) {
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm && rccl().ok) (void)rccl().CommDestroy(d.comm);
        for (void *p : {(void *)d.d_q, (void *)d.d_send, (void *)d.d_recv})
            if (p) (void)hipFree(p);
        for (uint64_t *p : d.d_allow)
            if (p) (void)hipFree(p);
        if (d.ev) (void)hipEventDestroy(d.ev);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    if (!c->devs.empty()) (void)hipSetDevice(c->devs[0].device);
    if (c->d_bases) (void)hipFree(c->d_bases);
    if (c->d_out) (void)hipFree(c->d_out);
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
    for (size_t i = 0; i < dev_of.size(); i++) {
        DevSlot &d = c->devs[i];
        d.device = dev_of[i];
        d.comm = comms[i];
        d.d_allow.assign(c->spd, nullptr);
        d.allow_words.assign(c->spd, 0);
        if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) {
            kdb_set_error("cluster_create: stream / event creation failed on device %d", d.device);
            return fail(KDB_ERR_HIP);
        }
    }
    if (hipSetDevice(c->devs[0].device) != hipSuccess || hipMalloc(&c->d_bases, (size_t)n_shards * 4) != hipSuccess ||
        hipMemcpy(c->d_bases, id_base, (size_t)n_shards * 4, hipMemcpyHostToDevice) != hipSuccess) {
        kdb_set_error("cluster_create: id base upload failed");
        return fail(KDB_ERR_HIP);
    }
    *out = c;
    return KDB_OK;
}

static int ensure_bytes(void **p, size_t *have, size_t want) {
    if (*have >= want) return KDB_OK;
    if (*p) {
        KDB_HIP(hipDeviceSynchronize());
        KDB_HIP(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    KDB_HIP(hipMalloc(p, want + want / 4));
    *have = want + want / 4;
    return KDB_OK;
}

// bits [base+1, base+count] of a dense GLOBAL bitset -> the shard's local bitset (local id i <-> global id base+i)
static void slice_allow(const uint64_t *g, size_t g_words, uint32_t base, uint32_t count, std::vector<uint64_t> &out) {
    out.assign(((size_t)count >> 6) + 1, 0ull);
    for (size_t w = 0; w < out.size(); w++) {
        // local bits 64w .. 64w+63 = global bits base + 64w ..
        const uint64_t gb = (uint64_t)base + 64ull * w;
        const size_t gw = (size_t)(gb >> 6);
        const unsigned sh = (unsigned)(gb & 63u);
        uint64_t v = gw < g_words ? g[gw] >> sh : 0ull;
        if (sh && gw + 1 < g_words) v |= g[gw + 1] << (64u - sh);
        out[w] = v;
    }
    out[0] &= ~1ull; // local id 0 does not exist
    const uint32_t last = count & 63u; // ids above count
    out.back() &= last == 63u ? ~0ull : ((2ull << last) - 1ull);
}

static int sharded_call(kdb_cluster *c, bool flat, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                        const uint64_t *allow_bits, size_t allow_words, uint32_t flags, uint32_t *out_ids, float *out_dist,
                        uint32_t *out_count) {
    if (!c) {
        kdb_set_error("null cluster handle");
        return KDB_ERR_INVALID;
    }
    if (B == 0) return KDB_OK;
    if (!queries || !out_ids || !out_dist || !out_count || k == 0) {
        kdb_set_error("sharded search: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    if (flags & KDB_SEARCH_DIST_F64) { // the exchange block and the merge carry float distances
        kdb_set_error("sharded search: KDB_SEARCH_DIST_F64 is a single-index option");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    const uint32_t G = (uint32_t)c->shards.size(), nd = (uint32_t)c->devs.size(), spd = c->spd;
    const size_t L = 2ull * B * k + B; // packed block, 32-bit words
    const size_t qbytes = (size_t)B * c->dim * 4;
    int rc;
    for (uint32_t i = 0; i < nd; i++) {
        DevSlot &d = c->devs[i];
        KDB_HIP(hipSetDevice(d.device));
        if ((rc = ensure_bytes((void **)&d.d_q, &d.q_bytes, qbytes))) return rc;
        size_t sb = d.send_words * 4, rb = d.recv_words * 4;
        if ((rc = ensure_bytes((void **)&d.d_send, &sb, (size_t)spd * L * 4))) return rc;
        if ((rc = ensure_bytes((void **)&d.d_recv, &rb, (size_t)G * L * 4))) return rc;
        d.send_words = sb / 4;
        d.recv_words = rb / 4;
    }
    // 1. queries: H2D to device 0, RCCL broadcast to the others
    {
        DevSlot &d0 = c->devs[0];
        KDB_HIP(hipSetDevice(d0.device));
        KDB_HIP(hipMemcpyAsync(d0.d_q, queries, qbytes, hipMemcpyHostToDevice, d0.stream));
        if (nd > 1) {
            KDB_NCCL(rccl().GroupStart());
            for (uint32_t i = 0; i < nd; i++) {
                DevSlot &d = c->devs[i];
                KDB_HIP(hipSetDevice(d.device));
                KDB_NCCL(rccl().Broadcast(d.d_q, d.d_q, qbytes / 4, KDB_NCCL_INT32, 0, d.comm, d.stream)); // in place at the root
            }
            KDB_NCCL(rccl().GroupEnd());
        }
    }
    // 2. every shard searches on its device, into its slot of the send buffer
    std::vector<uint64_t> host_bits;
    for (uint32_t g = 0; g < G; g++) {
        DevSlot &d = c->devs[g / spd];
        const uint32_t li = g % spd;
        kdb_index *idx = c->shards[g];
        KDB_HIP(hipSetDevice(d.device));
        const uint64_t *d_allow = nullptr;
        if (allow_bits) {
            slice_allow(allow_bits, allow_words, c->id_base[g], idx->count, host_bits);
            size_t have = d.allow_words[li] * 8;
            if ((rc = ensure_bytes((void **)&d.d_allow[li], &have, host_bits.size() * 8))) return rc;
            d.allow_words[li] = have / 8;
            // (pageable source: the copy has consumed host_bits when the call returns)
            KDB_HIP(hipMemcpyAsync(d.d_allow[li], host_bits.data(), host_bits.size() * 8, hipMemcpyHostToDevice, d.stream));
            KDB_HIP(hipStreamSynchronize(d.stream));
            d_allow = d.d_allow[li];
        }
        uint32_t *blk = d.d_send + (size_t)li * L;
        uint32_t *b_ids = blk;
        float *b_dist = reinterpret_cast<float *>(blk + (size_t)B * k);
        uint32_t *b_cnt = blk + 2 * (size_t)B * k;
        rc = flat ? kdb_flat_scan_batch_dev(idx, d.d_q, B, k, d_allow, flags, b_ids, b_dist, b_cnt, d.stream)
                  : kdb_search_batch_dev(idx, d.d_q, B, k, ef, d_allow, flags, b_ids, b_dist, b_cnt, d.stream);
        if (rc) return rc;
    }
    // 3. the one exchange step: all-gather of the packed blocks over xGMI
    KDB_NCCL(rccl().GroupStart());
    for (uint32_t i = 0; i < nd; i++) {
        DevSlot &d = c->devs[i];
        KDB_HIP(hipSetDevice(d.device));
        KDB_NCCL(rccl().AllGather(d.d_send, d.d_recv, (size_t)spd * L, KDB_NCCL_INT32, d.comm, d.stream));
    }
    KDB_NCCL(rccl().GroupEnd());
    // 4. merge on device 0, answers home
    DevSlot &d0 = c->devs[0];
    KDB_HIP(hipSetDevice(d0.device));
    {
        size_t ob = c->out_words * 4;
        if ((rc = ensure_bytes((void **)&c->d_out, &ob, L * 4))) return rc;
        c->out_words = ob / 4;
    }
    uint32_t *m_ids = c->d_out;
    float *m_dist = reinterpret_cast<float *>(c->d_out + (size_t)B * k);
    uint32_t *m_cnt = c->d_out + 2 * (size_t)B * k;
    const int negate = c->metric == KDB_METRIC_COSINE && c->precision == KDB_PREC_F32;
    const size_t bk = (size_t)B * k;
    rc = kdb_launch_merge_topk(negate, G, B, k, d0.d_recv, reinterpret_cast<const float *>(d0.d_recv + bk), d0.d_recv + 2 * bk, L, L,
                               c->d_bases, m_ids, m_dist, m_cnt, d0.stream);
    if (rc) return rc;
    KDB_HIP(hipMemcpyAsync(out_ids, m_ids, bk * 4, hipMemcpyDeviceToHost, d0.stream));
    KDB_HIP(hipMemcpyAsync(out_dist, m_dist, bk * 4, hipMemcpyDeviceToHost, d0.stream));
    KDB_HIP(hipMemcpyAsync(out_count, m_cnt, (size_t)B * 4, hipMemcpyDeviceToHost, d0.stream));
    for (uint32_t i = 0; i < nd; i++) { // every device's part of the call is over before the caller's buffers are reused
        KDB_HIP(hipSetDevice(c->devs[i].device));
        KDB_HIP(hipStreamSynchronize(c->devs[i].stream));
    }
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
