// api_pipeline.hip -- the whole path, ft8_subsystem() of rtlsdr_ft8d.c:1387-1524 for a batch: the order of the kernel
// launches on the context's streams (plain, or two overlapped parts) and the host-buffer form with chunked uploads.
#include "ft8gpu_ctx.h"

namespace {

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

}  // namespace

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

static constexpr int kHostChunk = 512;      // frames per upload chunk of a host-buffer batch

extern "C" int ft8gpu_decode_batch(ft8gpu_ctx *c, const float *iq, int nframes, struct decoder_results *decodes,
                        int32_t *n_results, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!iq || !decodes || !n_results) return ft8_fail("NULL array argument");
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

// frames resident on the context's GPU, records to host arrays (used by the multi-GPU entry)
int decode_dev_to_host(ft8gpu_ctx *c, const float *d_iq, int nframes, struct decoder_results *decodes, int32_t *n_results) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!d_iq || !decodes || !n_results) return ft8_fail("NULL array argument");
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
