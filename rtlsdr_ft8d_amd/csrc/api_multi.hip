// api_multi.hip -- SURVEY.md section 8(e) for a plain C caller: contiguous frame shards over several contexts / GPUs of
// one node on persistent host threads, and the device-resident RCCL gather of the spot list.
#include "ft8gpu_ctx.h"
#include "shard_pool.h"

#include <algorithm>
#include <dlfcn.h>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <vector>

// SURVEY.md section 8(e) for a C caller: contiguous shards, one host thread and one context per GPU; the
// "gather" is each shard writing its records at its frame offset of the caller's host arrays.
static int run_shards(ft8gpu_ctx *const *ctxs, int ndev, const float *const *iq_of, const int *first, const int *count,
                      bool iq_on_device, struct decoder_results *decodes, int32_t *n_results) {
    std::vector<int> rc((size_t)ndev, 0);
    std::vector<std::string> why((size_t)ndev);
    auto work = [&](int g) {
        struct decoder_results *d = decodes + (size_t)first[g] * kMaxMessages;
        int32_t *n = n_results + first[g];
        rc[g] = iq_on_device ? decode_dev_to_host(ctxs[g], iq_of[g], count[g], d, n)
                             : ft8gpu_decode_batch(ctxs[g], iq_of[g], count[g], d, n, FT8GPU_HOST_PTRS);
        if (rc[g]) why[g] = ft8_err_buffer();                       // the error text is thread-local: hand it to the caller's thread
    };
    // shards 1.. on the persistent workers, shard 0 on the calling thread
    ShardPool &pool = ShardPool::instance();
    ShardPool::Latch latch;
    for (int g = 1; g < ndev; ++g) {
        if (count[g] <= 0) continue;
        if (!pool.post([&work, g] { work(g); }, &latch)) work(g);     // no worker to be had: this shard runs here
    }
    if (count[0] > 0) work(0);
    latch.wait();
    for (int g = 0; g < ndev; ++g)
        if (rc[g]) return ft8_fail("shard %d of %d (frames [%d, %d)): %s", g, ndev, first[g], first[g] + count[g], why[g].c_str());
    return 0;
}

extern "C" {

int ft8gpu_decode_batch_multi(ft8gpu_ctx *const *ctxs, int ndev, const float *iq, int nframes,
                              struct decoder_results *decodes, int32_t *n_results) {
    if (!ctxs || ndev < 1) return ft8_fail("ft8gpu_decode_batch_multi: no contexts");
    if (nframes < 0) return ft8_fail("nframes < 0");
    if (nframes == 0) return 0;
    if (!iq || !decodes || !n_results) return ft8_fail("NULL array argument");
    std::vector<const float *> iq_of((size_t)ndev);
    std::vector<int> first((size_t)ndev), count((size_t)ndev);
    for (int g = 0; g < ndev; ++g) {
        if (!ctxs[g]) return ft8_fail("ctxs[%d] is NULL", g);
        for (int h = 0; h < g; ++h) if (ctxs[h] == ctxs[g]) return ft8_fail("ctxs[%d] and ctxs[%d] are the same context", h, g);
        first[g] = (int)((long long)nframes * g / ndev);
        count[g] = (int)((long long)nframes * (g + 1) / ndev) - first[g];
        iq_of[g] = iq + (size_t)first[g] * 2 * kNSamples;
    }
    return run_shards(ctxs, ndev, iq_of.data(), first.data(), count.data(), false, decodes, n_results);
}

int ft8gpu_decode_batch_multi_dev(ft8gpu_ctx *const *ctxs, int ndev, const float *const *iq_dev, const int *nframes_dev,
                                  struct decoder_results *decodes, int32_t *n_results) {
    if (!ctxs || ndev < 1) return ft8_fail("ft8gpu_decode_batch_multi_dev: no contexts");
    if (!iq_dev || !nframes_dev || !decodes || !n_results) return ft8_fail("NULL array argument");
    std::vector<int> first((size_t)ndev), count((size_t)ndev);
    long long total = 0;
    for (int g = 0; g < ndev; ++g) {
        if (!ctxs[g]) return ft8_fail("ctxs[%d] is NULL", g);
        for (int h = 0; h < g; ++h) if (ctxs[h] == ctxs[g]) return ft8_fail("ctxs[%d] and ctxs[%d] are the same context", h, g);
        if (nframes_dev[g] < 0) return ft8_fail("nframes_dev[%d] < 0", g);
        if (nframes_dev[g] > 0 && !iq_dev[g]) return ft8_fail("iq_dev[%d] is NULL", g);
        first[g] = (int)total;
        count[g] = nframes_dev[g];
        total += nframes_dev[g];
        if (total > 0x7FFFFFFF) return ft8_fail("too many frames");
    }
    return run_shards(ctxs, ndev, iq_dev, first.data(), count.data(), true, decodes, n_results);
}

}  // extern "C"

// ---- device-resident gather of the spot list over RCCL (north_star: "a trivial RCCL gather
// over xGMI for the spot list") for a plain C caller.  ft8gpu_decode_batch_multi[_dev] gather on the HOST, which is
// what the daemon consumes; this entry leaves the whole job's records in HBM of every GPU (e.g. for
// ft8gpu_pskreporter_datagrams or a device-side consumer).  Single-process RCCL: one communicator per GPU
// (ncclCommInitAll), one grouped ncclAllGather per buffer on each context's own stream, so the collective is ordered
// behind the kernels that produce the records and nothing synchronises the host.
// librccl is bound at run time (dlopen), not at link time: libft8gpu.so has no RCCL dependency, a process that
// already mapped an RCCL (PyTorch does) keeps that copy, and a box without RCCL gets a clean error.
namespace {

typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int kNcclUint8 = 1;                      // ncclUint8 of rccl.h (ncclInt8 = 0)

Rccl *rccl() {
    static std::mutex mu;
    static Rccl *r = nullptr;
    std::lock_guard<std::mutex> l(mu);
    if (r && r->lib) return r;
    if (!r) r = new Rccl();
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);            // the copy the process already holds, if any
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { const char *e = dlerror(); r->why = e ? e : "dlopen failed"; return r; }
    r->CommInitAll = (decltype(r->CommInitAll))dlsym(h, "ncclCommInitAll");
    r->CommDestroy = (decltype(r->CommDestroy))dlsym(h, "ncclCommDestroy");
    r->AllGather = (decltype(r->AllGather))dlsym(h, "ncclAllGather");
    r->GroupStart = (decltype(r->GroupStart))dlsym(h, "ncclGroupStart");
    r->GroupEnd = (decltype(r->GroupEnd))dlsym(h, "ncclGroupEnd");
    r->GetErrorString = (decltype(r->GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r->CommInitAll || !r->CommDestroy || !r->AllGather || !r->GroupStart || !r->GroupEnd || !r->GetErrorString) {
        r->why = "librccl lacks a required nccl* symbol";
        return r;
    }
    r->lib = h;
    return r;
}

struct GatherGroup {                                // communicators of one device list, created on first use
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};
std::mutex g_gather_mu;
std::vector<GatherGroup *> g_groups;

}  // namespace

extern "C" int ft8gpu_gather_spots(ft8gpu_ctx *const *ctxs, int ndev, const struct decoder_results *const *decodes_dev,
                                   const int32_t *const *n_results_dev, int frames_per_dev,
                                   struct decoder_results *const *all_decodes_dev, int32_t *const *all_n_results_dev) {
    if (!ctxs || ndev < 1) return ft8_fail("ft8gpu_gather_spots: no contexts");
    if (!decodes_dev || !n_results_dev || !all_decodes_dev || !all_n_results_dev) return ft8_fail("NULL array argument");
    if (frames_per_dev < 0) return ft8_fail("frames_per_dev < 0");
    if (frames_per_dev == 0) return 0;
    std::vector<int> devs((size_t)ndev);
    for (int g = 0; g < ndev; ++g) {
        if (!ctxs[g]) return ft8_fail("ctxs[%d] is NULL", g);
        if (!decodes_dev[g] || !n_results_dev[g] || !all_decodes_dev[g] || !all_n_results_dev[g]) return ft8_fail("NULL buffer for shard %d", g);
        devs[g] = ctxs[g]->device;
        for (int h = 0; h < g; ++h)
            if (devs[h] == devs[g]) return ft8_fail("ft8gpu_gather_spots: ctxs[%d] and ctxs[%d] are on the same GPU %d (RCCL needs one rank per device; "
                                                "several contexts on one GPU gather on the host: ft8gpu_decode_batch_multi_dev)", h, g, devs[g]);
    }
    Rccl *r = rccl();
    if (!r->lib) return ft8_fail("RCCL unavailable: %s", r->why.c_str());
    std::lock_guard<std::mutex> lock(g_gather_mu);
    GatherGroup *grp = nullptr;
    for (GatherGroup *c : g_groups) if (c->devices == devs) grp = c;
    if (!grp) {
        grp = new (std::nothrow) GatherGroup();
        if (!grp) return ft8_fail("out of host memory");
        grp->devices = devs;
        grp->comms.assign((size_t)ndev, nullptr);
        const int rc = r->CommInitAll(grp->comms.data(), ndev, devs.data());
        if (rc != 0) { const char *e = r->GetErrorString(rc); delete grp; return ft8_fail("ncclCommInitAll failed: %s", e ? e : "?"); }
        g_groups.push_back(grp);
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    const size_t rec_bytes = (size_t)frames_per_dev * kMaxMessages * sizeof(struct decoder_results);
    const size_t cnt_bytes = (size_t)frames_per_dev * sizeof(int32_t);
    // RCCL enqueues the grouped collectives on the contexts' streams at ncclGroupEnd(), not at the ncclAllGather calls: every
    // context stays locked from GroupStart to GroupEnd (in address order -- the one lock order any multi-context entry may use),
    // so that a concurrent ft8gpu_set_stream / ft8gpu_destroy cannot replace or destroy a stream RCCL is launching on.
    std::vector<ft8gpu_ctx *> order(ctxs, ctxs + ndev);
    std::sort(order.begin(), order.end(), std::less<ft8gpu_ctx *>());
    std::vector<std::unique_lock<std::mutex>> held;
    held.reserve((size_t)ndev);
    for (ft8gpu_ctx *c : order) held.emplace_back(c->mu);
    int rc = r->GroupStart();
    for (int g = 0; g < ndev && rc == 0; ++g) {
        (void)hipSetDevice(devs[g]);
        rc = r->AllGather(decodes_dev[g], all_decodes_dev[g], rec_bytes, kNcclUint8, grp->comms[g], ctxs[g]->stream);
        if (rc == 0) rc = r->AllGather(n_results_dev[g], all_n_results_dev[g], cnt_bytes, kNcclUint8, grp->comms[g], ctxs[g]->stream);
    }
    const int rc_end = r->GroupEnd();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc == 0) rc = rc_end;
    if (rc != 0) { const char *e = r->GetErrorString(rc); return ft8_fail("RCCL all-gather failed: %s", e ? e : "?"); }
    return 0;       // enqueued on every context's stream; ft8gpu_synchronize(ctxs[g]) or a later entry of that context waits for it
}

extern "C" void ft8gpu_gather_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_gather_mu);
    if (g_groups.empty()) return;                   // no gather was ever performed: nothing to destroy, and librccl is NOT loaded for this
    Rccl *r = rccl();                               // a group exists, so the library is already bound: this only returns the handle
    for (GatherGroup *grp : g_groups) {
        if (r->lib) for (ncclComm_t c : grp->comms) if (c) (void)r->CommDestroy(c);
        delete grp;
    }
    g_groups.clear();
}

extern "C" int ft8gpu_shard_workers(void) { return ShardPool::instance().workers(); }
