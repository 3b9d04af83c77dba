// ft8gpu_api.hip -- context, persistent HBM buffers and the C ABI of include/ft8gpu.h.
// Replaces the process-global FFTW state of the reference (initFFTW/freeFFTW, rtlsdr_ft8d.c:314-347)
// by an explicit, re-entrant context; the reference-named drop-in symbols live in ft8_compat.c.
#include "ft8gpu_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <condition_variable>
#include <deque>
#include <dlfcn.h>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// The reference's quantiser, rtlsdr_ft8d.c:1416 + :1425-1427, as a function of
// y = 1e-12f + mag2*4/(NFFT*NFFT), evaluated with the host's libm exactly as the reference does.
int ref_quant(float y) {
    const float db = 10.0f * log10f(y);
    const int scaled = (int)(2 * db + 240);
    return (scaled < 0) ? 0 : ((scaled > 255) ? 255 : scaled);
}

uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// qthr[k] (k = 1..255) = smallest positive float y with ref_quant(y) >= k.  The device quantiser
// counts thresholds <= y, which reproduces ref_quant bit for bit as long as the host log10f is
// monotone across each threshold (checked below and by sampling).
int build_tables(Ft8Tables *t) {
    for (int i = 0; i < kNfft; i++) t->hann[i] = sinf((M_PI / kNfft) * i);          // rtlsdr_ft8d.c:333
    for (int k = 0; k < kNfft; k++) {
        const double a = 2.0 * M_PI * (double)k / (double)kNfft;
        t->tw[k].x = (float)cos(a);
        t->tw[k].y = (float)(-sin(a));
    }
    memset(t->qthr, 0, sizeof t->qthr);
    const uint32_t lo_bits = f2u(1E-12f), hi_bits = f2u(FLT_MAX);
    t->qthr[0] = 0.0f;
    for (int k = 1; k <= 255; k++) {
        if (ref_quant(u2f(hi_bits)) < k) { t->qthr[k] = INFINITY; continue; }
        if (ref_quant(u2f(lo_bits)) >= k) { t->qthr[k] = u2f(lo_bits); continue; }
        uint32_t lo = lo_bits, hi = hi_bits;              // invariant: q(lo) < k <= q(hi)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (ref_quant(u2f(mid)) >= k) hi = mid; else lo = mid;
        }
        t->qthr[k] = u2f(hi);
        // local monotonicity: a window of neighbouring floats must sit on the right side
        for (uint32_t d = 1; d <= 64; d++) {
            if (ref_quant(u2f(hi + d)) < k || ref_quant(u2f(hi - d)) >= k)
                return fail("host log10f is not monotone around quantiser threshold %d", k);
        }
    }
    for (int k = 256; k < 260; k++) t->qthr[k] = NAN;       // `y >= qthr[256]` must be false for every y, +inf included
    // sampled global check of the threshold form against the direct expression
    uint64_t s = 0x243F6A8885A308D3ull;
    for (int it = 0; it < 200000; it++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const float expo = -12.0f + 18.5f * (float)((s >> 11) & 0xFFFFFF) / 16777216.0f;
        const float y = 1E-12f + powf(10.0f, expo);
        int q = 0;
        for (int k = 1; k <= 255; k++) q += (y >= t->qthr[k]);
        if (q != ref_quant(y)) return fail("quantiser threshold table disagrees with log10f at y=%g", (double)y);
    }
    return 0;
}

}  // namespace

struct ft8gpu_ctx {
    int device = 0;
    int num_cus = 256;
    int max_frames = 0;
    int cap_candidates = 0;
    ft8gpu_params params{ FT8GPU_K_MIN_SCORE, FT8GPU_K_MAX_CANDIDATES, FT8GPU_K_LDPC_ITERS };
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    static constexpr int kTimingSlots = 32, kEvPerSlot = 16, kSideEv0 = 10;
    hipEvent_t ev[kTimingSlots][kEvPerSlot]{};   // ring of per-run stage events (no host sync while timing):
                                                 // 0..9 on the main stream, 10..15 on the side stream
    long runs = 0;                         // pipeline runs recorded since timing was enabled
    int slot_form[kTimingSlots]{};         // which form of the pipeline a slot recorded: 0 one launch per stage, 1 two parts
    hipStream_t side = nullptr;            // carries the serial kernels (heap, spots) of one half-batch
                                           // while the main stream works on the other half
    hipStream_t side2 = nullptr;           // heap replay of part B (beside the one of part A on `side`)
    hipEvent_t dep[4]{};                   // cross-stream dependencies (no timing)
    bool overlap_ok = false;               // main, side and side2 were SEEN to run kernels concurrently (probe_streams)
    int *d_probe = nullptr;                // two ints for that probe
    char overlap_why[160] = "";            // why the overlapped pipeline is off (empty when it is on)
    std::mutex mu;                         // every entry point holds it: concurrent callers of one context serialise
    unsigned debug_flags = 0;              // FT8GPU_DBG_* (test hooks, per context)
    hipStream_t copy = nullptr;            // host-buffer calls: uploads chunk k+1 while chunk k is decoded
    static constexpr int kCopyEvents = 4;
    hipEvent_t copied[kCopyEvents]{};

    Ft8Tables *d_tab = nullptr;
    float *d_iq = nullptr;                 // staging for host-pointer calls
    uint8_t *d_mag = nullptr;
    uint32_t *d_lists = nullptr;
    int32_t *d_list_counts = nullptr;
    ft8gpu_candidate *d_cands = nullptr;
    int32_t *d_counts = nullptr;
    ft8gpu_decode_status *d_status = nullptr;
    struct decoder_results *d_decodes = nullptr;
    int32_t *d_nres = nullptr;
    int16_t *d_scores = nullptr;           // lazily allocated (diagnostic)
    ft8gpu_synth_signal *d_sigs = nullptr;
    size_t sigs_cap = 0;
    void *d_rx_sums = nullptr, *d_rx_p2 = nullptr;     // RX front end scratch
    uint8_t *d_rx_raw = nullptr;
    float *d_rx_iq = nullptr;
    size_t rx_sums_cap = 0, rx_p2_cap = 0, rx_raw_cap = 0, rx_iq_cap = 0;
    uint8_t *d_rep = nullptr;              // host-pointer staging of the report stage
    int32_t *d_rep_len = nullptr;
    uint32_t *d_rep_time = nullptr;
    size_t rep_cap = 0, rep_len_cap = 0, rep_time_cap = 0;
};

namespace {

int alloc_candidate_buffers(ft8gpu_ctx *c, int cap) {
    if (c->d_cands) { (void)hipFree(c->d_cands); c->d_cands = nullptr; }
    if (c->d_status) { (void)hipFree(c->d_status); c->d_status = nullptr; }
    HIP_TRY(hipMalloc(&c->d_cands, (size_t)c->max_frames * cap * sizeof(ft8gpu_candidate)));
    HIP_TRY(hipMalloc(&c->d_status, (size_t)c->max_frames * cap * sizeof(ft8gpu_decode_status)));
    c->cap_candidates = cap;
    return 0;
}

int check_params(const ft8gpu_params *p) {
    if (p->max_candidates < 1 || p->max_candidates > FT8GPU_ABS_MAX_CANDIDATES)
        return fail("max_candidates %d out of range [1, %d]", p->max_candidates, FT8GPU_ABS_MAX_CANDIDATES);
    if (p->ldpc_iters < 1 || p->ldpc_iters > 1000) return fail("ldpc_iters %d out of range", p->ldpc_iters);
    if (p->min_score < -32768 || p->min_score > 32767) return fail("min_score %d out of range", p->min_score);
    return 0;
}

// Every ABI entry that touches a context holds its mutex (two host threads on one context serialise instead
// of racing on the staging buffers and the timing ring) and runs with the context's GPU current, restoring
// the caller's current device on the way out.
struct Entry {
    std::unique_lock<std::mutex> lock;
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit Entry(ft8gpu_ctx *c) : lock(c->mu) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) err = hipSetDevice(c->device); else prev = -1;
    }
    ~Entry() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct StageTimer {
    ft8gpu_ctx *c;
    explicit StageTimer(ft8gpu_ctx *ctx) : c(ctx) {}
    void mark(int i) { if (c->timing) (void)hipEventRecord(c->ev[c->runs % ft8gpu_ctx::kTimingSlots][i], c->stream); }
    void mark_side(int i) { mark_on(c->side, i); }
    void mark_on(hipStream_t s, int i) { if (c->timing) (void)hipEventRecord(c->ev[c->runs % ft8gpu_ctx::kTimingSlots][ft8gpu_ctx::kSideEv0 + i], s); }
    void done(int form) {
        if (!c->timing) return;
        c->slot_form[c->runs % ft8gpu_ctx::kTimingSlots] = form;
        c->runs++;
    }
};

inline int force_ieee(const ft8gpu_ctx *c) { return (c->debug_flags & FT8GPU_DBG_FORCE_IEEE_DIV) ? 1 : 0; }

float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}

// ---- do the context's streams really run kernels side by side? --------------------------------------------------------
// The two-part pipeline below hides the serial kernels (heap replay, spot collection) under the throughput kernels of the
// other part, which only works if the main stream and the two side streams sit on three different hardware queues: HIP
// multiplexes streams onto a few queues, and streams that share one run their kernels one after the other (round 3 lost
// 0.27 ms per step when the context happened to be created after a framework's streams).  Instead of relying on creation
// order, the context MEASURES it: a one-wave kernel on stream A spins until a flag is set or 2 ms have passed, a
// one-thread kernel on stream B sets the flag; A reports whether it saw it.  A side stream that does not co-run with the
// others is replaced by a newly created one (the rejected stream is kept until the search ends so that its queue is
// not handed out again), a few times; if that
// fails too the context runs the plain pipeline (one launch per stage, nothing on side streams) and says so:
// ft8gpu_overlap_active() returns 0 and ft8gpu_last_error() holds the reason.
__global__ void ft8_probe_wait_kernel(int *flag, int *seen, unsigned long long timeout_ticks) {
    const unsigned long long t0 = wall_clock64();
    int ok = 0;
    do {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(16);
    } while (wall_clock64() - t0 < timeout_ticks);
    *seen = ok;
}
__global__ void ft8_probe_set_kernel(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 1: a kernel on `b` ran while a kernel on `a` was running; 0: it did not (within 2 ms); -1: HIP error (g_err set)
int streams_corun(ft8gpu_ctx *c, hipStream_t a, hipStream_t b) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    HIP_TRY(hipStreamSynchronize(a));
    HIP_TRY(hipStreamSynchronize(b));
    HIP_TRY(hipMemsetAsync(c->d_probe, 0, 2 * sizeof(int), a));
    HIP_TRY(hipStreamSynchronize(a));
    hipLaunchKernelGGL(ft8_probe_wait_kernel, dim3(1), dim3(1), 0, a, c->d_probe, c->d_probe + 1, (unsigned long long)khz * 2ull);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ft8_probe_set_kernel, dim3(1), dim3(1), 0, b, c->d_probe);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(a));
    HIP_TRY(hipStreamSynchronize(b));
    int seen = 0;
    HIP_TRY(hipMemcpy(&seen, c->d_probe + 1, sizeof(int), hipMemcpyDeviceToHost));
    return seen ? 1 : 0;
}

// Plain non-blocking streams, default priority.  (Measured in round 4: with the side streams at the highest stream
// priority the pipeline alone runs exactly as fast -- 4.19 against 4.195 ms per 4096 frames -- but beside a one-rank RCCL
// exchange on a framework stream it LOSES 0.8 ms per step (5.09 against 4.23-4.30 ms; 4.37 against 4.24 ms with the
// context created after the process group): queues of different priorities are arbitrated against each other, and the
// hand-offs between the main stream and the collective's stream pay for it.  So no priorities.)
hipError_t create_side_stream(hipStream_t *s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

// (re)establishes c->overlap_ok for the current main stream; replaces side streams that share a queue
int probe_streams(ft8gpu_ctx *c) {
    c->overlap_ok = false;
    c->overlap_why[0] = 0;
    if (!c->d_probe) HIP_TRY(hipMalloc(&c->d_probe, 2 * sizeof(int)));
    std::vector<hipStream_t> rejected;
    auto cleanup = [&] { for (hipStream_t r : rejected) (void)hipStreamDestroy(r); rejected.clear(); };
    int rc = 1;
    for (int attempt = 0; attempt < 6; ++attempt) {
        // which of the side streams fails against the main stream or against its sibling?
        int bad = 0;                                  // 1: side, 2: side2
        if ((rc = streams_corun(c, c->stream, c->side)) < 0) break;
        if (rc == 0) bad = 1;
        if (!bad) { if ((rc = streams_corun(c, c->stream, c->side2)) < 0) break; if (rc == 0) bad = 2; }
        if (!bad) { if ((rc = streams_corun(c, c->side, c->side2)) < 0) break; if (rc == 0) bad = 2; }
        if (!bad) { c->overlap_ok = true; break; }
        hipStream_t fresh = nullptr;
        if (create_side_stream(&fresh) != hipSuccess) { rc = 0; break; }
        hipStream_t &slot = bad == 1 ? c->side : c->side2;
        rejected.push_back(slot);
        slot = fresh;
        rc = 0;
    }
    cleanup();
    if (rc < 0) return -1;
    if (!c->overlap_ok)
        snprintf(c->overlap_why, sizeof c->overlap_why,
                 "the context's side streams do not run beside its main stream (shared hardware queues): plain pipeline, no overlap");
    return 0;
}

// Large batches: the exact heap replay is a serial kernel (a lane or a wave per frame), so the batch is cut into a first
// part A (a quarter) and the rest B, and the replays run on side streams under the throughput kernels of the other part:
//   main : wf(A) sync(A) wf(B) sync(B) ..wait heap(A).. decode(A) ..wait heap(B).. decode(B) spots(A+B)
//   side :            heap(A)
//   side2:                        heap(B)
// heap(A) hides under the waterfall and sync kernels of B, heap(B) under decode(A); what stays exposed is the spot
// collection (27 us) and one extra LDPC-kernel tail.  The heap replay is a dependent chain whose length grows
// with the candidate cap (about 0.1 ms at 120, 0.35 ms at 480) and hardly with the number of frames, so at large caps
// part A is made big enough for decode(A) to cover heap(B).  (Round 2 ran ONE waterfall launch up front, which left
// heap(A) only sync(B) to hide under.  Measured alternatives, profiles/r02_ab_kernels.json and r03_ab_pipeline.json:
// equal halves cost 0.1-0.16 ms more; a K-part pipeline with the waterfall of part k+1 beside the LDPC kernel of part k
// is SLOWER -- the waterfall's large workgroups are not co-scheduled beside the LDPC kernel's small ones.)
int run_pipeline_overlapped(ft8gpu_ctx *c, const float *d_iq, int n, struct decoder_results *d_dec, int32_t *d_nres) {
    StageTimer t(c);
    const ft8gpu_params &p = c->params;
    const int mc = p.max_candidates;
    // size of part A: a quarter of the batch in whole blocks of 64 frames.  Round 2 used 1/16 (its sweep, with the heap
    // replay of part B on the same side stream as that of part A: 1/16 4.99 ms, 2/16 5.03, 4/16 5.03, 8/16 5.22).  With
    // heap(B) on its own stream and the round-3 kernels the order is reversed -- 256 frames 4.23-4.26 ms, 512 4.25,
    // 768 4.25, 1024 4.20-4.24, 1536 4.21-4.23, 2048 4.21 -- decode(A) then covers the whole replay of part B
    // (whose dependent chain grows with the candidate cap, not with the number of frames) and the first LDPC launch fills
    // the machine for longer.
    int n0 = ((n / 4) + 63) & ~63;
    if (n0 < 64) n0 = 64;
    if (n0 > n - 64) n0 = n / 2;
    const int n1 = n - n0;
    const size_t lo = (size_t)n0;                                   // frame offset of part B
    const size_t frame_floats = 2 * (size_t)kNSamples;
    uint8_t *mag1 = c->d_mag + lo * kMagArray;
    uint32_t *lists1 = c->d_lists + lo * kSublistsPerFrame * kSublistCap;
    int32_t *lc1 = c->d_list_counts + lo * kSublistsPerFrame;
    ft8gpu_candidate *cands1 = c->d_cands + lo * mc;
    int32_t *counts1 = c->d_counts + lo;
    ft8gpu_decode_status *st1 = c->d_status + lo * mc;
    hipEvent_t *E = c->dep;          // 0: sync(A) done  1: sync(B) done  2: heap(A)  3: heap(B)

    // main stream events: 0 wf(A) 1 sync(A) 2 wf(B) 9 sync(B) 3 | 4 decode(A) 5 decode(B) 6 spots 7 | 8 end
    // side stream events: 0 heap(A) 1        side2: 2 heap(B) 3
    t.mark(0);
    HIP_TRY(launch_waterfall(d_iq, c->d_mag, c->d_tab, n0, c->num_cus, c->debug_flags, c->stream));
    t.mark(1);
    HIP_TRY(launch_sync(c->d_mag, c->d_lists, c->d_list_counts, nullptr, n0, p.min_score, c->stream));
    HIP_TRY(hipEventRecord(E[0], c->stream));
    t.mark(2);
    HIP_TRY(launch_waterfall(d_iq + lo * frame_floats, mag1, c->d_tab, n1, c->num_cus, c->debug_flags, c->stream));
    t.mark(9);
    HIP_TRY(launch_sync(mag1, lists1, lc1, nullptr, n1, p.min_score, c->stream));
    HIP_TRY(hipEventRecord(E[1], c->stream));
    t.mark(3);
    // side streams: heap(A), heap(B)
    HIP_TRY(hipStreamWaitEvent(c->side, E[0], 0));
    t.mark_side(0);
    // The replay of part B runs beside the VALU-bound LDPC kernel of part A: from about 3000 frames on that kernel runs
    // longer than the 0.55 ms of the one-lane-per-frame replay, which costs a tenth of the issue slots beside it.  The
    // replay of part A runs beside the LDS-bound waterfall kernel, which has VALU slots to spare and no LDS bandwidth:
    // there the wave-per-frame form (VALU lane moves, short chain) is the better neighbour.  Measured in one session:
    // both wave-per-frame 4.157 ms, both lane-per-frame 4.116, this split 4.096.
    const bool hide = n >= 3072;
    HIP_TRY(launch_heap(c->d_lists, c->d_list_counts, c->d_cands, c->d_counts, n0, mc, c->debug_flags, c->side, false));
    t.mark_side(1);
    HIP_TRY(hipEventRecord(E[2], c->side));
    HIP_TRY(hipStreamWaitEvent(c->side2, E[1], 0));
    t.mark_on(c->side2, 2);
    HIP_TRY(launch_heap(lists1, lc1, cands1, counts1, n1, mc, c->debug_flags, c->side2, hide));
    t.mark_on(c->side2, 3);
    HIP_TRY(hipEventRecord(E[3], c->side2));
    // main stream: decode(A), decode(B), spots
    HIP_TRY(hipStreamWaitEvent(c->stream, E[2], 0));
    t.mark(4);
    HIP_TRY(launch_decode(c->d_mag, c->d_cands, c->d_counts, c->d_status, n0, mc, p.ldpc_iters, false, force_ieee(c), c->stream));
    t.mark(5);
    HIP_TRY(hipStreamWaitEvent(c->stream, E[3], 0));
    HIP_TRY(launch_decode(mag1, cands1, counts1, st1, n1, mc, p.ldpc_iters, false, force_ieee(c), c->stream));
    t.mark(6);
    // ONE spot collection for both parts behind the last LDPC launch (the parts' buffers are contiguous).  Until round 4 the
    // spots of part A ran on the side stream beside decode(B); since the kernel takes 27 us for 4096 frames that bought
    // nothing and cost two event hops on the main stream: 4.115 -> 4.077 ms per step in interleaved A/B.
    HIP_TRY(launch_spots(c->d_cands, c->d_counts, c->d_status, n, mc, p.min_score, d_dec, d_nres, c->stream));
    t.mark(7);
    t.mark(8);
    t.done(1);
    return 0;
}

// the pipeline on device pointers; all intermediates in the context's HBM buffers
int run_pipeline(ft8gpu_ctx *c, const float *d_iq, int n, struct decoder_results *d_dec, int32_t *d_nres) {
    if (c->overlap_ok && !(c->debug_flags & FT8GPU_DBG_NO_OVERLAP) && n >= 512) return run_pipeline_overlapped(c, d_iq, n, d_dec, d_nres);
    StageTimer t(c);
    const ft8gpu_params &p = c->params;
    t.mark(0);
    HIP_TRY(launch_waterfall(d_iq, c->d_mag, c->d_tab, n, c->num_cus, c->debug_flags, c->stream));
    t.mark(1);
    HIP_TRY(launch_sync(c->d_mag, c->d_lists, c->d_list_counts, nullptr, n, p.min_score, c->stream));
    t.mark(2);
    HIP_TRY(launch_heap(c->d_lists, c->d_list_counts, c->d_cands, c->d_counts, n, p.max_candidates, c->debug_flags, c->stream));
    t.mark(3);
    HIP_TRY(launch_decode(c->d_mag, c->d_cands, c->d_counts, c->d_status, n, p.max_candidates, p.ldpc_iters, false, force_ieee(c), c->stream));
    t.mark(4);
    HIP_TRY(launch_spots(c->d_cands, c->d_counts, c->d_status, n, p.max_candidates, p.min_score, d_dec, d_nres, c->stream));
    t.mark(5);
    t.done(0);
    return 0;
}

}  // namespace

#define CHECK_COMMON(c, n)                                                              \
    if (!(c)) return fail("ctx is NULL");                                               \
    if ((n) < 0) return fail("nframes < 0");                                            \
    Entry entry_(c);                                                                    \
    HIP_TRY(entry_.err);

extern "C" {

const char *ft8gpu_last_error(void) { return g_err; }

int ft8gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// everything that can fail after the context object exists; the caller destroys it on failure
static int create_body(ft8gpu_ctx *c) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    for (auto &slot : c->ev) for (auto &e : slot) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(create_side_stream(&c->side));
    HIP_TRY(create_side_stream(&c->side2));
    for (auto &e : c->dep) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // (the upload stream of the host-buffer path is created on first use: a context that only sees device pointers
    // keeps its three streams on three hardware queues of their own)
    for (auto &e : c->copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    Ft8Tables *h = (Ft8Tables *)malloc(sizeof(Ft8Tables));
    if (!h) return fail("out of host memory");
    if (build_tables(h)) { free(h); return -1; }
    hipError_t e = hipMalloc(&c->d_tab, sizeof(Ft8Tables));
    if (e == hipSuccess) e = hipMemcpy(c->d_tab, h, sizeof(Ft8Tables), hipMemcpyHostToDevice);
    free(h);
    if (e != hipSuccess) return fail("uploading the constant tables failed: %s", hipGetErrorString(e));
    HIP_TRY(decode_tables_init(c->stream));

    const size_t F = (size_t)c->max_frames;
    HIP_TRY(hipMalloc(&c->d_mag, F * kMagArray));
    HIP_TRY(hipMalloc(&c->d_lists, F * kSublistsPerFrame * kSublistCap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->d_list_counts, F * kSublistsPerFrame * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->d_counts, F * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->d_decodes, F * kMaxMessages * sizeof(struct decoder_results)));
    HIP_TRY(hipMalloc(&c->d_nres, F * sizeof(int32_t)));
    if (alloc_candidate_buffers(c, c->params.max_candidates < 120 ? 120 : c->params.max_candidates)) return -1;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (probe_streams(c)) return -1;
    return 0;
}

int ft8gpu_create(ft8gpu_ctx **out, int device, int max_frames, const ft8gpu_params *params) {
    if (!out) return fail("ft8gpu_create: out is NULL");
    *out = nullptr;
    if (max_frames < 1) return fail("ft8gpu_create: max_frames must be >= 1");
    if (params && check_params(params)) return -1;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("ft8gpu_create: device %d not present (%d visible)", device, ndev);
    int prev = -1;
    (void)hipGetDevice(&prev);
    HIP_TRY(hipSetDevice(device));
    ft8gpu_ctx *c = new (std::nothrow) ft8gpu_ctx();
    if (!c) { if (prev >= 0 && prev != device) (void)hipSetDevice(prev); return fail("out of host memory"); }
    c->device = device;
    c->max_frames = max_frames;
    if (params) c->params = *params;
    const int rc = create_body(c);
    if (rc) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);          // ft8gpu_destroy must not clobber the reason
        ft8gpu_destroy(c);
        memcpy(g_err, keep, sizeof keep);
    } else {
        *out = c;
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);     // the caller's current device is left as it was
    return rc ? -1 : 0;
}

void ft8gpu_destroy(ft8gpu_ctx *c) {
    if (!c) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{ prev };
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *bufs[] = { c->d_tab, c->d_iq, c->d_mag, c->d_lists, c->d_list_counts, c->d_cands, c->d_counts,
                     c->d_status, c->d_decodes, c->d_nres, c->d_scores, c->d_sigs,
                     c->d_rx_sums, c->d_rx_p2, c->d_rx_raw, c->d_rx_iq,
                     c->d_rep, c->d_rep_len, c->d_rep_time, c->d_probe };
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (c->side) (void)hipStreamSynchronize(c->side);
    for (auto &slot : c->ev) for (auto &e : slot) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->dep) if (e) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->side2) { (void)hipStreamSynchronize(c->side2); (void)hipStreamDestroy(c->side2); }
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    for (auto &e : c->copied) if (e) (void)hipEventDestroy(e);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// hip_stream: NULL = the context creates its own (non-blocking) stream; any other value is used as given,
// including hipStreamLegacy ((hipStream_t)1, FT8GPU_STREAM_LEGACY) for the legacy null stream and
// hipStreamPerThread ((hipStream_t)2).
int ft8gpu_set_stream(ft8gpu_ctx *c, void *hip_stream) {
    CHECK_COMMON(c, 0);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->side));
    HIP_TRY(hipStreamSynchronize(c->side2));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    // a borrowed stream may share a hardware queue with a side stream: measure again (and re-roll the side streams)
    if (probe_streams(c)) return -1;
    if (!c->overlap_ok) fail("%s", c->overlap_why);           // not an error: the call succeeds, the reason is on record
    return 0;
}

int ft8gpu_overlap_active(ft8gpu_ctx *c) {
    if (!c) return fail("ctx is NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->overlap_ok) fail("%s", c->overlap_why);
    return c->overlap_ok ? 1 : 0;
}

void *ft8gpu_get_stream(ft8gpu_ctx *c) {
    if (!c) { fail("ctx is NULL"); return nullptr; }
    std::lock_guard<std::mutex> lock(c->mu);
    return (void *)c->stream;
}

int ft8gpu_set_params(ft8gpu_ctx *c, const ft8gpu_params *p) {
    if (!c || !p) return fail("NULL argument");
    if (check_params(p)) return -1;
    CHECK_COMMON(c, 0);
    if (p->max_candidates > c->cap_candidates) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (alloc_candidate_buffers(c, p->max_candidates)) return -1;
    }
    c->params = *p;
    return 0;
}

int ft8gpu_set_debug_flags(ft8gpu_ctx *c, unsigned flags) {
    CHECK_COMMON(c, 0);
    if (flags & ~FT8GPU_DBG_ALL) return fail("ft8gpu_set_debug_flags: unknown bits 0x%x", flags & ~FT8GPU_DBG_ALL);
    if ((flags & FT8GPU_DBG_HEAP_LANE_PER_FRAME) && (flags & FT8GPU_DBG_HEAP_WAVE_PER_FRAME))
        return fail("ft8gpu_set_debug_flags: FT8GPU_DBG_HEAP_LANE_PER_FRAME and FT8GPU_DBG_HEAP_WAVE_PER_FRAME exclude each other");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->debug_flags = flags;
    return 0;
}

int ft8gpu_selftest_bp_math(ft8gpu_ctx *c, uint64_t out[7]) {
    if (!out) return fail("NULL argument");
    CHECK_COMMON(c, 0);
    HIP_TRY(run_bp_math_selftest(out, c->stream));
    return 0;
}

int ft8gpu_enable_timing(ft8gpu_ctx *c, int on) {
    CHECK_COMMON(c, 0);
    c->timing = on != 0;
    c->runs = 0;
    return 0;
}

// mean over the (up to 32 most recent) pipeline runs recorded since ft8gpu_enable_timing(ctx, 1)
int ft8gpu_get_timings(ft8gpu_ctx *c, ft8gpu_timings *out, int32_t *nruns) {
    if (!c || !out) return fail("NULL argument");
    CHECK_COMMON(c, 0);
    if (!c->timing || c->runs == 0) return fail("no timed pipeline run recorded");
    const int n = c->runs < ft8gpu_ctx::kTimingSlots ? (int)c->runs : ft8gpu_ctx::kTimingSlots;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    int launches = 1;
    for (int k = 0; k < n; k++) {
        const int slot = (int)((c->runs - 1 - k) % ft8gpu_ctx::kTimingSlots);
        hipEvent_t *e = c->ev[slot];
        if (c->slot_form[slot] != 0) {
            hipEvent_t *sd = e + ft8gpu_ctx::kSideEv0;
            HIP_TRY(hipEventSynchronize(e[8]));               // (the main stream has waited for both side streams by then)
            acc[0] += elapsed(e[0], e[1]) + elapsed(e[2], e[9]);             // waterfall: both parts
            acc[1] += elapsed(e[1], e[2]) + elapsed(e[9], e[3]);             // sync: both parts
            acc[2] += elapsed(sd[0], sd[1]) + elapsed(sd[2], sd[3]);         // heap: both parts (side stream, overlapped)
            acc[3] += elapsed(e[4], e[5]) + elapsed(e[5], e[6]);             // decode: both launches
            acc[4] += elapsed(e[6], e[7]);                                   // spots: both parts in one launch
            acc[5] += elapsed(e[0], e[8]);
            launches = 2;
        } else {
            HIP_TRY(hipEventSynchronize(e[5]));
            for (int i = 0; i < 5; i++) acc[i] += elapsed(e[i], e[i + 1]);
            acc[5] += elapsed(e[0], e[5]);
        }
    }
    out->waterfall_ms = (float)(acc[0] / n);
    out->sync_ms = (float)(acc[1] / n);
    out->heap_ms = (float)(acc[2] / n);
    out->decode_ms = (float)(acc[3] / n);
    out->spots_ms = (float)(acc[4] / n);
    out->total_ms = (float)(acc[5] / n);
    out->launches_per_stage = launches;
    if (nruns) *nruns = n;
    return 0;
}

int ft8gpu_synchronize(ft8gpu_ctx *c) {
    CHECK_COMMON(c, 0);
    HIP_TRY(hipStreamSynchronize(c->stream));                // the main stream joins the side stream at the end of a run
    return 0;
}

// device memory helpers: they act on the context's GPU (not on whatever device happens to be current)
void *ft8gpu_dev_alloc(ft8gpu_ctx *c, size_t bytes) {
    if (!c) { fail("ctx is NULL"); return nullptr; }
    Entry entry_(c);
    void *p = nullptr;
    if (entry_.err != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { fail("hipMalloc(%zu) on device %d failed", bytes, c->device); return nullptr; }
    return p;
}
void ft8gpu_dev_free(ft8gpu_ctx *c, void *p) {
    if (!c || !p) return;
    Entry entry_(c);
    (void)hipFree(p);
}
// page-locked host memory: the host-buffer entries upload with hipMemcpyAsync, which is a true asynchronous DMA (and
// overlaps the kernels of the previous chunk) only from pinned memory; from pageable memory it is staged through a
// bounce buffer and serialises.  Plain hipHostMalloc / hipHostFree, offered here so that a C caller of the batch
// entries needs no HIP header.
void *ft8gpu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { fail("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void ft8gpu_host_free(void *p) { if (p) (void)hipHostFree(p); }

int ft8gpu_memcpy_h2d(ft8gpu_ctx *c, void *d, const void *s, size_t n) {
    CHECK_COMMON(c, 0);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice));
    return 0;
}
int ft8gpu_memcpy_d2h(ft8gpu_ctx *c, void *d, const void *s, size_t n) {
    CHECK_COMMON(c, 0);
    HIP_TRY(hipStreamSynchronize(c->stream));                // results of the context's own kernels are complete
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost));
    return 0;
}

static constexpr int kHostChunk = 512;      // frames per upload chunk of a host-buffer batch


int ft8gpu_decode_batch(ft8gpu_ctx *c, const float *iq, int nframes, struct decoder_results *decodes,
                        int32_t *n_results, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!iq || !decodes || !n_results) return fail("NULL array argument");
    const size_t frame_floats = 2 * (size_t)kNSamples;
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        if (flags & FT8GPU_DEVICE_PTRS) {
            if (run_pipeline(c, iq + f0 * frame_floats, n, decodes + (size_t)f0 * kMaxMessages, n_results + f0)) return -1;
        } else {
            if (!c->d_iq) HIP_TRY(hipMalloc(&c->d_iq, (size_t)c->max_frames * frame_floats * sizeof(float)));
            // slots of non-CQ messages must keep the caller's bytes (rtlsdr_ft8d.c:1509-1520)
            HIP_TRY(hipMemcpyAsync(c->d_decodes, decodes + (size_t)f0 * kMaxMessages,
                                   (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyHostToDevice, c->stream));
            // the upload is 384 KB per frame and takes longer than the decode: pipeline it in chunks on a
            // copy stream so that the kernels of chunk k run under the upload of chunk k+1
            const int chunk = (n > kHostChunk && !(c->debug_flags & FT8GPU_DBG_NO_OVERLAP)) ? kHostChunk : n;
            int k = 0;
            for (int g0 = 0; g0 < n; g0 += chunk, k++) {
                const int m = (n - g0 < chunk) ? n - g0 : chunk;
                if (chunk < n && !c->copy) HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
                hipStream_t up = (chunk < n) ? c->copy : c->stream;
                HIP_TRY(hipMemcpyAsync(c->d_iq + g0 * frame_floats, iq + (f0 + g0) * frame_floats, m * frame_floats * sizeof(float),
                                       hipMemcpyHostToDevice, up));
                if (up != c->stream) {
                    hipEvent_t e = c->copied[k % ft8gpu_ctx::kCopyEvents];
                    HIP_TRY(hipEventRecord(e, up));
                    HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
                }
                if (run_pipeline(c, c->d_iq + g0 * frame_floats, m, c->d_decodes + (size_t)g0 * kMaxMessages, c->d_nres + g0)) return -1;
            }
            HIP_TRY(hipMemcpyAsync(decodes + (size_t)f0 * kMaxMessages, c->d_decodes,
                                   (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(n_results + f0, c->d_nres, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

// Persistent host workers of the multi-GPU entries: one thread per concurrently running shard beyond the caller's
// own, created on first use and kept for the life of the process (round 2 spawned ndev-1 std::threads per call,
// which is measurable on small batches).  Workers carry no GPU state of their own: every task enters its context
// through the usual Entry guard.  No C++ exception crosses the C ABI: a failed thread creation makes post() return
// false and the caller runs the shard itself.
class ShardPool {
public:
    struct Latch {
        std::mutex m;
        std::condition_variable cv;
        int pending = 0;
        void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [this] { return pending == 0; }); }
    };
    static ShardPool &instance() { static ShardPool *p = new ShardPool(); return *p; }    // never destroyed: no join at exit
    bool post(std::function<void()> fn, Latch *latch) {
        std::unique_lock<std::mutex> l(m_);
        try {
            q_.emplace_back(std::move(fn), latch);
        } catch (...) { return false; }
        // one waiting (or starting) worker per queued job, so that shards never queue up behind each other
        if (idle_ + starting_ < (int)q_.size()) {
            try { std::thread(&ShardPool::loop, this).detach(); ++workers_; ++starting_; }
            catch (...) {
                if (idle_ + starting_ == 0 && workers_ == 0) { q_.pop_back(); return false; }   // nobody would ever run it
            }
        }
        { std::lock_guard<std::mutex> g(latch->m); ++latch->pending; }
        l.unlock();
        cv_.notify_one();
        return true;
    }
    int workers() { std::lock_guard<std::mutex> l(m_); return workers_; }
private:
    void loop() {
        bool first = true;
        for (;;) {
            std::pair<std::function<void()>, Latch *> job;
            {
                std::unique_lock<std::mutex> l(m_);
                if (first) { --starting_; first = false; }
                ++idle_;
                cv_.wait(l, [this] { return !q_.empty(); });
                --idle_;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job.first();
            // The latch lives on the caller's stack (run_shards): the waiter may return, and its frame may die, as soon as
            // it can observe pending == 0 -- which needs the mutex.  So the notification is sent while the mutex is still
            // held; after the unlock this thread never touches the latch again.
            {
                std::lock_guard<std::mutex> g(job.second->m);
                --job.second->pending;
                job.second->cv.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::pair<std::function<void()>, Latch *>> q_;
    int workers_ = 0, idle_ = 0, starting_ = 0;
};

// frames resident on the context's GPU, records to host arrays (used by the multi-GPU entry)
static int decode_dev_to_host(ft8gpu_ctx *c, const float *d_iq, int nframes, struct decoder_results *decodes, int32_t *n_results) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!d_iq || !decodes || !n_results) return fail("NULL array argument");
    const size_t frame_floats = 2 * (size_t)kNSamples;
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        // slots of non-CQ messages must keep the caller's bytes (rtlsdr_ft8d.c:1509-1520)
        HIP_TRY(hipMemcpyAsync(c->d_decodes, decodes + (size_t)f0 * kMaxMessages,
                               (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyHostToDevice, c->stream));
        if (run_pipeline(c, d_iq + f0 * frame_floats, n, c->d_decodes, c->d_nres)) return -1;
        HIP_TRY(hipMemcpyAsync(decodes + (size_t)f0 * kMaxMessages, c->d_decodes,
                               (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(n_results + f0, c->d_nres, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

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
        if (rc[g]) why[g] = g_err;                       // the error text is thread-local: hand it to the caller's thread
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
        if (rc[g]) return fail("shard %d of %d (frames [%d, %d)): %s", g, ndev, first[g], first[g] + count[g], why[g].c_str());
    return 0;
}

int ft8gpu_decode_batch_multi(ft8gpu_ctx *const *ctxs, int ndev, const float *iq, int nframes,
                              struct decoder_results *decodes, int32_t *n_results) {
    if (!ctxs || ndev < 1) return fail("ft8gpu_decode_batch_multi: no contexts");
    if (nframes < 0) return fail("nframes < 0");
    if (nframes == 0) return 0;
    if (!iq || !decodes || !n_results) return fail("NULL array argument");
    std::vector<const float *> iq_of((size_t)ndev);
    std::vector<int> first((size_t)ndev), count((size_t)ndev);
    for (int g = 0; g < ndev; ++g) {
        if (!ctxs[g]) return fail("ctxs[%d] is NULL", g);
        for (int h = 0; h < g; ++h) if (ctxs[h] == ctxs[g]) return fail("ctxs[%d] and ctxs[%d] are the same context", h, g);
        first[g] = (int)((long long)nframes * g / ndev);
        count[g] = (int)((long long)nframes * (g + 1) / ndev) - first[g];
        iq_of[g] = iq + (size_t)first[g] * 2 * kNSamples;
    }
    return run_shards(ctxs, ndev, iq_of.data(), first.data(), count.data(), false, decodes, n_results);
}

int ft8gpu_decode_batch_multi_dev(ft8gpu_ctx *const *ctxs, int ndev, const float *const *iq_dev, const int *nframes_dev,
                                  struct decoder_results *decodes, int32_t *n_results) {
    if (!ctxs || ndev < 1) return fail("ft8gpu_decode_batch_multi_dev: no contexts");
    if (!iq_dev || !nframes_dev || !decodes || !n_results) return fail("NULL array argument");
    std::vector<int> first((size_t)ndev), count((size_t)ndev);
    long long total = 0;
    for (int g = 0; g < ndev; ++g) {
        if (!ctxs[g]) return fail("ctxs[%d] is NULL", g);
        for (int h = 0; h < g; ++h) if (ctxs[h] == ctxs[g]) return fail("ctxs[%d] and ctxs[%d] are the same context", h, g);
        if (nframes_dev[g] < 0) return fail("nframes_dev[%d] < 0", g);
        if (nframes_dev[g] > 0 && !iq_dev[g]) return fail("iq_dev[%d] is NULL", g);
        first[g] = (int)total;
        count[g] = nframes_dev[g];
        total += nframes_dev[g];
        if (total > 0x7FFFFFFF) return fail("too many frames");
    }
    return run_shards(ctxs, ndev, iq_dev, first.data(), count.data(), true, decodes, n_results);
}

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
    if (!ctxs || ndev < 1) return fail("ft8gpu_gather_spots: no contexts");
    if (!decodes_dev || !n_results_dev || !all_decodes_dev || !all_n_results_dev) return fail("NULL array argument");
    if (frames_per_dev < 0) return fail("frames_per_dev < 0");
    if (frames_per_dev == 0) return 0;
    std::vector<int> devs((size_t)ndev);
    for (int g = 0; g < ndev; ++g) {
        if (!ctxs[g]) return fail("ctxs[%d] is NULL", g);
        if (!decodes_dev[g] || !n_results_dev[g] || !all_decodes_dev[g] || !all_n_results_dev[g]) return fail("NULL buffer for shard %d", g);
        devs[g] = ctxs[g]->device;
        for (int h = 0; h < g; ++h)
            if (devs[h] == devs[g]) return fail("ft8gpu_gather_spots: ctxs[%d] and ctxs[%d] are on the same GPU %d (RCCL needs one rank per device; "
                                                "several contexts on one GPU gather on the host: ft8gpu_decode_batch_multi_dev)", h, g, devs[g]);
    }
    Rccl *r = rccl();
    if (!r->lib) return fail("RCCL unavailable: %s", r->why.c_str());
    std::lock_guard<std::mutex> lock(g_gather_mu);
    GatherGroup *grp = nullptr;
    for (GatherGroup *c : g_groups) if (c->devices == devs) grp = c;
    if (!grp) {
        grp = new (std::nothrow) GatherGroup();
        if (!grp) return fail("out of host memory");
        grp->devices = devs;
        grp->comms.assign((size_t)ndev, nullptr);
        const int rc = r->CommInitAll(grp->comms.data(), ndev, devs.data());
        if (rc != 0) { const char *e = r->GetErrorString(rc); delete grp; return fail("ncclCommInitAll failed: %s", e ? e : "?"); }
        g_groups.push_back(grp);
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    const size_t rec_bytes = (size_t)frames_per_dev * kMaxMessages * sizeof(struct decoder_results);
    const size_t cnt_bytes = (size_t)frames_per_dev * sizeof(int32_t);
    int rc = r->GroupStart();
    for (int g = 0; g < ndev && rc == 0; ++g) {
        std::lock_guard<std::mutex> cl(ctxs[g]->mu);
        (void)hipSetDevice(devs[g]);
        rc = r->AllGather(decodes_dev[g], all_decodes_dev[g], rec_bytes, kNcclUint8, grp->comms[g], ctxs[g]->stream);
        if (rc == 0) rc = r->AllGather(n_results_dev[g], all_n_results_dev[g], cnt_bytes, kNcclUint8, grp->comms[g], ctxs[g]->stream);
    }
    const int rc_end = r->GroupEnd();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc == 0) rc = rc_end;
    if (rc != 0) { const char *e = r->GetErrorString(rc); return fail("RCCL all-gather failed: %s", e ? e : "?"); }
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

int ft8gpu_waterfall(ft8gpu_ctx *c, const float *iq, int nframes, uint8_t *mag, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!iq || !mag) return fail("NULL array argument");
    const size_t frame_floats = 2 * (size_t)kNSamples;
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        if (flags & FT8GPU_DEVICE_PTRS) {
            HIP_TRY(launch_waterfall(iq + f0 * frame_floats, mag + (size_t)f0 * kMagArray, c->d_tab, n, c->num_cus, c->debug_flags, c->stream));
        } else {
            if (!c->d_iq) HIP_TRY(hipMalloc(&c->d_iq, (size_t)c->max_frames * frame_floats * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(c->d_iq, iq + f0 * frame_floats, n * frame_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(launch_waterfall(c->d_iq, c->d_mag, c->d_tab, n, c->num_cus, c->debug_flags, c->stream));
            HIP_TRY(hipMemcpyAsync(mag + (size_t)f0 * kMagArray, c->d_mag, (size_t)n * kMagArray, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

int ft8gpu_find_sync(ft8gpu_ctx *c, const uint8_t *mag, int nframes, ft8gpu_candidate *cands, int32_t *counts, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!mag || !cands || !counts) return fail("NULL array argument");
    const int mc = c->params.max_candidates;
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        const bool dev = flags & FT8GPU_DEVICE_PTRS;
        const uint8_t *dm = dev ? mag + (size_t)f0 * kMagArray : c->d_mag;
        ft8gpu_candidate *dc = dev ? cands + (size_t)f0 * mc : c->d_cands;
        int32_t *dn = dev ? counts + f0 : c->d_counts;
        if (!dev) HIP_TRY(hipMemcpyAsync(c->d_mag, mag + (size_t)f0 * kMagArray, (size_t)n * kMagArray, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_sync(dm, c->d_lists, c->d_list_counts, nullptr, n, c->params.min_score, c->stream));
        HIP_TRY(launch_heap(c->d_lists, c->d_list_counts, dc, dn, n, mc, c->debug_flags, c->stream));
        if (!dev) {
            HIP_TRY(hipMemcpyAsync(cands + (size_t)f0 * mc, dc, (size_t)n * mc * sizeof(ft8gpu_candidate), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(counts + f0, dn, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

int ft8gpu_score_map(ft8gpu_ctx *c, const uint8_t *mag, int nframes, int16_t *scores, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!mag || !scores) return fail("NULL array argument");
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        const bool dev = flags & FT8GPU_DEVICE_PTRS;
        if (!dev && !c->d_scores) HIP_TRY(hipMalloc(&c->d_scores, (size_t)c->max_frames * kScoresPerFrame * sizeof(int16_t)));
        const uint8_t *dm = dev ? mag + (size_t)f0 * kMagArray : c->d_mag;
        int16_t *ds = dev ? scores + (size_t)f0 * kScoresPerFrame : c->d_scores;
        if (!dev) HIP_TRY(hipMemcpyAsync(c->d_mag, mag + (size_t)f0 * kMagArray, (size_t)n * kMagArray, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_sync(dm, c->d_lists, c->d_list_counts, ds, n, c->params.min_score, c->stream));
        if (!dev) {
            HIP_TRY(hipMemcpyAsync(scores + (size_t)f0 * kScoresPerFrame, ds, (size_t)n * kScoresPerFrame * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

int ft8gpu_decode_candidates(ft8gpu_ctx *c, const uint8_t *mag, const ft8gpu_candidate *cands, const int32_t *counts,
                             int nframes, ft8gpu_decode_status *status, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!mag || !cands || !counts || !status) return fail("NULL array argument");
    const int mc = c->params.max_candidates;
    // the stage entry reports the exact ldpc_errors; FT8GPU_DBG_PIPELINE_FORM runs the form of the
    // kernel the batch pipeline uses instead (test hook: every field but ldpc_errors must agree)
    const bool count_errors = !(c->debug_flags & FT8GPU_DBG_PIPELINE_FORM);
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        const bool dev = flags & FT8GPU_DEVICE_PTRS;
        const uint8_t *dm = dev ? mag + (size_t)f0 * kMagArray : c->d_mag;
        const ft8gpu_candidate *dc = dev ? cands + (size_t)f0 * mc : c->d_cands;
        const int32_t *dn = dev ? counts + f0 : c->d_counts;
        ft8gpu_decode_status *dst = dev ? status + (size_t)f0 * mc : c->d_status;
        if (!dev) {
            HIP_TRY(hipMemcpyAsync(c->d_mag, mag + (size_t)f0 * kMagArray, (size_t)n * kMagArray, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_cands, cands + (size_t)f0 * mc, (size_t)n * mc * sizeof(ft8gpu_candidate), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_counts, counts + f0, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemsetAsync(c->d_status, 0, (size_t)n * mc * sizeof(ft8gpu_decode_status), c->stream));
        }
        HIP_TRY(launch_decode(dm, dc, dn, dst, n, mc, c->params.ldpc_iters, count_errors, force_ieee(c), c->stream));
        if (!dev) {
            HIP_TRY(hipMemcpyAsync(status + (size_t)f0 * mc, dst, (size_t)n * mc * sizeof(ft8gpu_decode_status), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

int ft8gpu_collect_spots(ft8gpu_ctx *c, const ft8gpu_candidate *cands, const int32_t *counts,
                         const ft8gpu_decode_status *status, int nframes, struct decoder_results *decodes,
                         int32_t *n_results, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!cands || !counts || !status || !decodes || !n_results) return fail("NULL array argument");
    const int mc = c->params.max_candidates;
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        const bool dev = flags & FT8GPU_DEVICE_PTRS;
        if (!dev) {
            HIP_TRY(hipMemcpyAsync(c->d_cands, cands + (size_t)f0 * mc, (size_t)n * mc * sizeof(ft8gpu_candidate), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_counts, counts + f0, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_status, status + (size_t)f0 * mc, (size_t)n * mc * sizeof(ft8gpu_decode_status), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_decodes, decodes + (size_t)f0 * kMaxMessages, (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(launch_spots(c->d_cands, c->d_counts, c->d_status, n, mc, c->params.min_score, c->d_decodes, c->d_nres, c->stream));
            HIP_TRY(hipMemcpyAsync(decodes + (size_t)f0 * kMaxMessages, c->d_decodes, (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(n_results + f0, c->d_nres, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        } else {
            HIP_TRY(launch_spots(cands + (size_t)f0 * mc, counts + f0, status + (size_t)f0 * mc, n, mc, c->params.min_score,
                                 decodes + (size_t)f0 * kMaxMessages, n_results + f0, c->stream));
        }
    }
    return 0;
}

static int grow(void **buf, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(buf, need));
    *cap = need;
    return 0;
}

int ft8gpu_rx_decimate(ft8gpu_ctx *c, const uint8_t *raw, int ncaptures, size_t npairs, float *iq,
                       int normalise, int flags) {
    CHECK_COMMON(c, ncaptures);
    if (ncaptures == 0) return 0;
    if (!raw || !iq) return fail("NULL array argument");
    if (npairs % 8 != 0) return fail("npairs must be a multiple of 8 (whole 16-byte units; the reference's buffers are multiples of 8 bytes)");
    const size_t nblocks = npairs / 751 > (size_t)kNSamples ? (size_t)kNSamples : npairs / 751;
    const size_t raw_bytes = (size_t)ncaptures * npairs * 2, iq_bytes = (size_t)ncaptures * 2 * kNSamples * sizeof(float);
    const size_t sums_bytes = (size_t)ncaptures * (nblocks + 1) * 16;
    const size_t p2_bytes = (size_t)ncaptures * ((nblocks + 15) / 16 + 1) * 32 + (size_t)ncaptures * 376 * 4;   // entry states + group totals + partial peaks
    const bool staged = !(flags & FT8GPU_DEVICE_PTRS);
    if (sums_bytes > c->rx_sums_cap || p2_bytes > c->rx_p2_cap || (staged && (raw_bytes > c->rx_raw_cap || iq_bytes > c->rx_iq_cap)))
        HIP_TRY(hipStreamSynchronize(c->stream));          // a buffer is regrown below: earlier launches may still use the old one
    if (grow(&c->d_rx_sums, &c->rx_sums_cap, sums_bytes)) return -1;
    if (grow(&c->d_rx_p2, &c->rx_p2_cap, p2_bytes)) return -1;
    if (flags & FT8GPU_DEVICE_PTRS) {
        if (((uintptr_t)raw & 15) != 0) return fail("raw must be 16-byte aligned");
        HIP_TRY(launch_rx(raw, ncaptures, npairs, c->d_rx_sums, c->d_rx_p2, iq, normalise, c->stream));
    } else {
        if (grow((void **)&c->d_rx_raw, &c->rx_raw_cap, raw_bytes)) return -1;
        if (grow((void **)&c->d_rx_iq, &c->rx_iq_cap, iq_bytes)) return -1;
        HIP_TRY(hipMemcpyAsync(c->d_rx_raw, raw, raw_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_rx(c->d_rx_raw, ncaptures, npairs, c->d_rx_sums, c->d_rx_p2, c->d_rx_iq, normalise, c->stream));
        HIP_TRY(hipMemcpyAsync(iq, c->d_rx_iq, iq_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// ---- f-4: PSKreporter datagrams ------------------------------------------------------------------
// The frame-independent front of the datagram (rtlsdr_ft8d.c:386-482): IPFIX message header, the
// receiver (options) template, the sender template and the receiver record.  The templates are
// generated from their field lists; the lengths that depend on the spot list are patched per frame
// by the kernel.
static int build_report_prefix(const ft8gpu_report_info *info, ReportPrefix *out) {
    struct Field { uint16_t id, len; bool enterprise; };
    static const Field rx_fields[] = { { 0x8002, 0xFFFF, true }, { 0x8004, 0xFFFF, true }, { 0x8008, 0xFFFF, true } };
    static const Field tx_fields[] = { { 0x8001, 0xFFFF, true }, { 0x8005, 4, true }, { 0x8006, 1, true }, { 0x800A, 0xFFFF, true },
                                       { 0x8003, 0xFFFF, true }, { 0x800B, 1, true }, { 0x0096, 4, false } };
    const uint32_t enterprise = 30351;                       // 0x0000768F
    unsigned char *b = out->bytes;
    size_t n = 0;
    auto be16 = [&](uint32_t v) { b[n++] = (unsigned char)(v >> 8); b[n++] = (unsigned char)v; };
    auto be32 = [&](uint32_t v) { be16(v >> 16); be16(v & 0xFFFF); };
    auto fields = [&](const Field *f, int count) {
        for (int i = 0; i < count; ++i) { be16(f[i].id); be16(f[i].len); if (f[i].enterprise) be32(enterprise); }
    };
    auto text = [&](const char *s, size_t cap) -> int {      // one length byte + characters
        const size_t len = strnlen(s, cap);
        if (len == cap) return fail("ft8gpu_report_info string is not NUL-terminated");
        b[n++] = (unsigned char)len;
        memcpy(b + n, s, len);
        n += len;
        return 0;
    };
    memset(out->bytes, 0, sizeof out->bytes);
    be16(0x000A); be16(0);                                   // version, total length (per frame)
    be32(info->unixtime); be32(info->sequence); be32(info->random_id);
    be16(3); be16(36); be16(0x9992); be16(3); be16(0);       // options template set: id, length, link, fields, scope fields
    fields(rx_fields, 3);
    be16(0);                                                 // padding
    be16(2); be16(60); be16(0x9993); be16(7);                // template set
    fields(tx_fields, 7);
    const size_t rx0 = n;                                    // receiver record
    be16(0x9992); be16(0);
    if (text(info->rcall, sizeof info->rcall) || text(info->rloc, sizeof info->rloc) ||
        text(info->app_version, sizeof info->app_version)) return -1;
    n += (4 - ((n - rx0) & 3)) & 3;                          // zero padding to 4 bytes
    b[rx0 + 2] = (unsigned char)((n - rx0) >> 8);
    b[rx0 + 3] = (unsigned char)(n - rx0);
    out->len = (int32_t)n;
    out->dial_freq = info->dial_freq;
    out->unixtime = info->unixtime;
    return 0;
}

int ft8gpu_pskreporter_datagrams(ft8gpu_ctx *c, const struct decoder_results *decodes, const int32_t *n_results,
                                 int nframes, const ft8gpu_report_info *info, const uint32_t *unixtimes,
                                 uint8_t *datagrams, int32_t *lengths, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!decodes || !n_results || !info || !datagrams || !lengths) return fail("NULL array argument");
    ReportPrefix pre;
    if (build_report_prefix(info, &pre)) return -1;
    if (flags & FT8GPU_DEVICE_PTRS) {
        if (((uintptr_t)datagrams & 15) != 0) return fail("datagrams must be 16-byte aligned");
        HIP_TRY(launch_report(decodes, n_results, nframes, pre, unixtimes, datagrams, lengths, c->stream));
        return 0;
    }
    const size_t F = (size_t)c->max_frames;
    HIP_TRY(hipStreamSynchronize(c->stream));              // staging may be regrown below
    if (grow((void **)&c->d_rep, &c->rep_cap, F * FT8GPU_DATAGRAM_STRIDE)) return -1;
    if (grow((void **)&c->d_rep_len, &c->rep_len_cap, F * sizeof(int32_t))) return -1;
    if (unixtimes && grow((void **)&c->d_rep_time, &c->rep_time_cap, F * sizeof(uint32_t))) return -1;
    for (int f0 = 0; f0 < nframes; f0 += c->max_frames) {
        const int n = (nframes - f0 < c->max_frames) ? nframes - f0 : c->max_frames;
        HIP_TRY(hipMemcpyAsync(c->d_decodes, decodes + (size_t)f0 * kMaxMessages, (size_t)n * kMaxMessages * sizeof(struct decoder_results), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_nres, n_results + f0, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        if (unixtimes) HIP_TRY(hipMemcpyAsync(c->d_rep_time, unixtimes + f0, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_report(c->d_decodes, c->d_nres, n, pre, unixtimes ? c->d_rep_time : nullptr, c->d_rep, c->d_rep_len, c->stream));
        HIP_TRY(hipMemcpyAsync(datagrams + (size_t)f0 * FT8GPU_DATAGRAM_STRIDE, c->d_rep, (size_t)n * FT8GPU_DATAGRAM_STRIDE, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(lengths + f0, c->d_rep_len, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int ft8gpu_synth_frames(ft8gpu_ctx *c, const ft8gpu_synth_signal *signals, int nframes, int nsig,
                        float noise_sigma, uint64_t seed, float *iq_dev) {
    return ft8gpu_synth_frames_at(c, signals, nframes, nsig, noise_sigma, seed, 0, iq_dev);
}

int ft8gpu_synth_frames_at(ft8gpu_ctx *c, const ft8gpu_synth_signal *signals, int nframes, int nsig,
                           float noise_sigma, uint64_t seed, uint64_t first_frame, float *iq_dev) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (nsig < 0 || nsig > 64) return fail("nsig_per_frame %d out of range [0, 64]", nsig);
    if (!iq_dev || (nsig > 0 && !signals)) return fail("NULL array argument");
    const size_t bytes = (size_t)nframes * (nsig > 0 ? nsig : 1) * sizeof(ft8gpu_synth_signal);
    if (bytes > c->sigs_cap) {
        if (c->d_sigs) (void)hipFree(c->d_sigs);
        c->d_sigs = nullptr;
        HIP_TRY(hipMalloc(&c->d_sigs, bytes));
        c->sigs_cap = bytes;
    }
    if (nsig > 0) HIP_TRY(hipMemcpyAsync(c->d_sigs, signals, (size_t)nframes * nsig * sizeof(ft8gpu_synth_signal), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_synth(c->d_sigs, nframes, nsig, noise_sigma, seed, first_frame, iq_dev, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
