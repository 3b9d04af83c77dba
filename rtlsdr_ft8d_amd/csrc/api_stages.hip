// api_stages.hip -- the same data stage by stage (used by the parity tests and by the ft8_lib-level symbols of
// ft8_compat.c): waterfall, sync search, candidate decode, spot collection.
#include "ft8gpu_ctx.h"

extern "C" {

int ft8gpu_waterfall(ft8gpu_ctx *c, const float *iq, int nframes, uint8_t *mag, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!iq || !mag) return ft8_fail("NULL array argument");
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
    if (!mag || !cands || !counts) return ft8_fail("NULL array argument");
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
    if (!mag || !scores) return ft8_fail("NULL array argument");
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
    if (!mag || !cands || !counts || !status) return ft8_fail("NULL array argument");
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
    if (!cands || !counts || !status || !decodes || !n_results) return ft8_fail("NULL array argument");
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

}  // extern "C"
