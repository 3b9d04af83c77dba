// ft8gpu_ctx.h -- the context object and the helpers every host-side translation unit of the C ABI shares
// (api_context.hip, api_pipeline.hip, api_multi.hip, api_stages.hip, api_glue.hip).  Not installed.
#pragma once
#include "ft8gpu_internal.h"

#include <mutex>
#include <stddef.h>

// ft8gpu_last_error(): thread-local text, written by ft8_fail() only (api_context.hip)
int ft8_fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
char *ft8_err_buffer();                       // the calling thread's buffer (kErrBytes), for hand-overs between threads
constexpr size_t kErrBytes = 512;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return ft8_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

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

#define CHECK_COMMON(c, n)                                                              \
    if (!(c)) return ft8_fail("ctx is NULL");                                               \
    if ((n) < 0) return ft8_fail("nframes < 0");                                            \
    Entry entry_(c);                                                                    \
    HIP_TRY(entry_.err);

inline int force_ieee(const ft8gpu_ctx *c) { return (c->debug_flags & FT8GPU_DBG_FORCE_IEEE_DIV) ? 1 : 0; }

// api_context.hip
int probe_streams(ft8gpu_ctx *c);             // (re)establishes c->overlap_ok for the current main stream
// api_pipeline.hip: the whole path on device pointers; all intermediates in the context's HBM buffers
int run_pipeline(ft8gpu_ctx *c, const float *d_iq, int n, struct decoder_results *d_dec, int32_t *d_nres);
// frames resident on the context's GPU, records to host arrays (used by the multi-GPU entries)
int decode_dev_to_host(ft8gpu_ctx *c, const float *d_iq, int nframes, struct decoder_results *decodes, int32_t *n_results);
// api_glue.hip: (re)allocates *buf when `need` exceeds *cap (the caller has synchronised the stream)
int grow_buffer(void **buf, size_t *cap, size_t need);
