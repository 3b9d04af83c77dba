// api_context.hip -- the decoder context: creation / destruction, constant tables, the co-execution probe of its streams,
// parameters, timings, memory helpers.  Replaces the process-global FFTW state of the reference (initFFTW/freeFFTW,
// rtlsdr_ft8d.c:314-347) by an explicit, re-entrant context; the reference-named drop-in symbols live in ft8_compat.c.
#include "ft8gpu_ctx.h"

#include <float.h>
#include <math.h>
#include <new>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
thread_local char g_err[kErrBytes] = "";
}

int ft8_fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}
char *ft8_err_buffer() { return g_err; }

namespace {

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
                return ft8_fail("host log10f is not monotone around quantiser threshold %d", k);
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
        if (q != ref_quant(y)) return ft8_fail("quantiser threshold table disagrees with log10f at y=%g", (double)y);
    }
    return 0;
}


// All or nothing: both new buffers exist before the old pair is released, so a failed allocation (ft8gpu_set_params growing
// the cap under memory pressure) leaves the context exactly as it was -- old buffers, old cap_candidates -- and returns -1.
int alloc_candidate_buffers(ft8gpu_ctx *c, int cap) {
    ft8gpu_candidate *cands = nullptr;
    ft8gpu_decode_status *status = nullptr;
    if (hipMalloc(&cands, (size_t)c->max_frames * cap * sizeof(ft8gpu_candidate)) != hipSuccess ||
        hipMalloc(&status, (size_t)c->max_frames * cap * sizeof(ft8gpu_decode_status)) != hipSuccess) {
        (void)hipGetLastError();
        if (cands) (void)hipFree(cands);
        return ft8_fail("out of device memory for %d candidates x %d frames (the context keeps its %d-candidate buffers)", cap, c->max_frames,
                        c->cap_candidates);
    }
    if (c->d_cands) (void)hipFree(c->d_cands);          // hipFree waits for work that still uses the old pair
    if (c->d_status) (void)hipFree(c->d_status);
    c->d_cands = cands;
    c->d_status = status;
    c->cap_candidates = cap;
    return 0;
}

int check_params(const ft8gpu_params *p) {
    if (p->max_candidates < 1 || p->max_candidates > FT8GPU_ABS_MAX_CANDIDATES)
        return ft8_fail("max_candidates %d out of range [1, %d]", p->max_candidates, FT8GPU_ABS_MAX_CANDIDATES);
    if (p->ldpc_iters < 1 || p->ldpc_iters > 1000) return ft8_fail("ldpc_iters %d out of range", p->ldpc_iters);
    if (p->min_score < -32768 || p->min_score > 32767) return ft8_fail("min_score %d out of range", p->min_score);
    return 0;
}

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

}  // namespace

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
    if (!h) return ft8_fail("out of host memory");
    if (build_tables(h)) { free(h); return -1; }
    hipError_t e = hipMalloc(&c->d_tab, sizeof(Ft8Tables));
    if (e == hipSuccess) e = hipMemcpy(c->d_tab, h, sizeof(Ft8Tables), hipMemcpyHostToDevice);
    free(h);
    if (e != hipSuccess) return ft8_fail("uploading the constant tables failed: %s", hipGetErrorString(e));
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
    if (!out) return ft8_fail("ft8gpu_create: out is NULL");
    *out = nullptr;
    if (max_frames < 1) return ft8_fail("ft8gpu_create: max_frames must be >= 1");
    if (params && check_params(params)) return -1;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return ft8_fail("ft8gpu_create: device %d not present (%d visible)", device, ndev);
    int prev = -1;
    (void)hipGetDevice(&prev);
    HIP_TRY(hipSetDevice(device));
    ft8gpu_ctx *c = new (std::nothrow) ft8gpu_ctx();
    if (!c) { if (prev >= 0 && prev != device) (void)hipSetDevice(prev); return ft8_fail("out of host memory"); }
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
    return 0;                                  // whichever pipeline the probe chose: ft8gpu_overlap_active / _reason tell
}

// pure queries: they never touch ft8gpu_last_error() (a NULL context is the one error they report)
int ft8gpu_overlap_active(ft8gpu_ctx *c) {
    if (!c) return ft8_fail("ctx is NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    return c->overlap_ok ? 1 : 0;
}

int ft8gpu_overlap_reason(ft8gpu_ctx *c, char *buf, size_t cap) {
    if (!c) return ft8_fail("ctx is NULL");
    if (!buf || cap == 0) return ft8_fail("NULL buffer");
    std::lock_guard<std::mutex> lock(c->mu);
    snprintf(buf, cap, "%s", c->overlap_ok ? "" : c->overlap_why);
    return 0;
}

void *ft8gpu_get_stream(ft8gpu_ctx *c) {
    if (!c) { ft8_fail("ctx is NULL"); return nullptr; }
    std::lock_guard<std::mutex> lock(c->mu);
    return (void *)c->stream;
}

int ft8gpu_set_params(ft8gpu_ctx *c, const ft8gpu_params *p) {
    if (!c || !p) return ft8_fail("NULL argument");
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
    if (flags & ~kDbgAccepted) return ft8_fail("ft8gpu_set_debug_flags: unknown bits 0x%x", flags & ~kDbgAccepted);
#ifdef FT8GPU_AB_FORMS
    if ((flags & FT8GPU_AB_HEAP_LANE_PER_FRAME) && (flags & FT8GPU_AB_HEAP_WAVE_PER_FRAME))
        return ft8_fail("ft8gpu_set_debug_flags: FT8GPU_AB_HEAP_LANE_PER_FRAME and FT8GPU_AB_HEAP_WAVE_PER_FRAME exclude each other");
#endif
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->debug_flags = flags;
    return 0;
}

int ft8gpu_selftest_bp_math(ft8gpu_ctx *c, uint64_t out[7]) {
    if (!out) return ft8_fail("NULL argument");
    CHECK_COMMON(c, 0);
    HIP_TRY(run_bp_math_selftest(out, c->stream));
    return 0;
}

int ft8gpu_selftest_norm_math(ft8gpu_ctx *c, uint64_t out[7]) {
    if (!out) return ft8_fail("NULL argument");
    CHECK_COMMON(c, 0);
    HIP_TRY(run_norm_math_selftest(out, c->stream));
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
    if (!c || !out) return ft8_fail("NULL argument");
    CHECK_COMMON(c, 0);
    if (!c->timing || c->runs == 0) return ft8_fail("no timed pipeline run recorded");
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
    if (!c) { ft8_fail("ctx is NULL"); return nullptr; }
    Entry entry_(c);
    void *p = nullptr;
    if (entry_.err != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { ft8_fail("hipMalloc(%zu) on device %d failed", bytes, c->device); return nullptr; }
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
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { ft8_fail("hipHostMalloc(%zu) failed", bytes); return nullptr; }
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


}  // extern "C"
