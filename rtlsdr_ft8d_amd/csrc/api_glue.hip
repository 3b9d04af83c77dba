// api_glue.hip -- host glue of the stages either side of the decode path: RX front end (f-1), PSKreporter datagrams
// (f-4), synthetic frames (f-3).
#include "ft8gpu_ctx.h"

#include <string.h>

int grow_buffer(void **buf, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(buf, need));
    *cap = need;
    return 0;
}

extern "C" {

int ft8gpu_rx_decimate(ft8gpu_ctx *c, const uint8_t *raw, int ncaptures, size_t npairs, float *iq,
                       int normalise, int flags) {
    CHECK_COMMON(c, ncaptures);
    if (ncaptures == 0) return 0;
    if (!raw || !iq) return ft8_fail("NULL array argument");
    if (npairs % 8 != 0) return ft8_fail("npairs must be a multiple of 8 (whole 16-byte units; the reference's buffers are multiples of 8 bytes)");
    const size_t nblocks = npairs / 751 > (size_t)kNSamples ? (size_t)kNSamples : npairs / 751;
    const size_t raw_bytes = (size_t)ncaptures * npairs * 2, iq_bytes = (size_t)ncaptures * 2 * kNSamples * sizeof(float);
    const size_t sums_bytes = (size_t)ncaptures * (nblocks + 1) * 16;
    const size_t p2_bytes = (size_t)ncaptures * ((nblocks + 15) / 16 + 1) * 32 + (size_t)ncaptures * 376 * 4;   // entry states + group totals + partial peaks
    const bool staged = !(flags & FT8GPU_DEVICE_PTRS);
    if (sums_bytes > c->rx_sums_cap || p2_bytes > c->rx_p2_cap || (staged && (raw_bytes > c->rx_raw_cap || iq_bytes > c->rx_iq_cap)))
        HIP_TRY(hipStreamSynchronize(c->stream));          // a buffer is regrown below: earlier launches may still use the old one
    if (grow_buffer(&c->d_rx_sums, &c->rx_sums_cap, sums_bytes)) return -1;
    if (grow_buffer(&c->d_rx_p2, &c->rx_p2_cap, p2_bytes)) return -1;
    if (flags & FT8GPU_DEVICE_PTRS) {
        if (((uintptr_t)raw & 15) != 0) return ft8_fail("raw must be 16-byte aligned");
        HIP_TRY(launch_rx(raw, ncaptures, npairs, c->d_rx_sums, c->d_rx_p2, iq, normalise, c->stream));
    } else {
        if (grow_buffer((void **)&c->d_rx_raw, &c->rx_raw_cap, raw_bytes)) return -1;
        if (grow_buffer((void **)&c->d_rx_iq, &c->rx_iq_cap, iq_bytes)) return -1;
        HIP_TRY(hipMemcpyAsync(c->d_rx_raw, raw, raw_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_rx(c->d_rx_raw, ncaptures, npairs, c->d_rx_sums, c->d_rx_p2, c->d_rx_iq, normalise, c->stream));
        HIP_TRY(hipMemcpyAsync(iq, c->d_rx_iq, iq_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

}  // extern "C"

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
        if (len == cap) return ft8_fail("ft8gpu_report_info string is not NUL-terminated");
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

extern "C" {

int ft8gpu_pskreporter_datagrams(ft8gpu_ctx *c, const struct decoder_results *decodes, const int32_t *n_results,
                                 int nframes, const ft8gpu_report_info *info, const uint32_t *unixtimes,
                                 uint8_t *datagrams, int32_t *lengths, int flags) {
    CHECK_COMMON(c, nframes);
    if (nframes == 0) return 0;
    if (!decodes || !n_results || !info || !datagrams || !lengths) return ft8_fail("NULL array argument");
    ReportPrefix pre;
    if (build_report_prefix(info, &pre)) return -1;
    if (flags & FT8GPU_DEVICE_PTRS) {
        if (((uintptr_t)datagrams & 15) != 0) return ft8_fail("datagrams must be 16-byte aligned");
        HIP_TRY(launch_report(decodes, n_results, nframes, pre, unixtimes, datagrams, lengths, c->stream));
        return 0;
    }
    const size_t F = (size_t)c->max_frames;
    HIP_TRY(hipStreamSynchronize(c->stream));              // staging may be regrown below
    if (grow_buffer((void **)&c->d_rep, &c->rep_cap, F * FT8GPU_DATAGRAM_STRIDE)) return -1;
    if (grow_buffer((void **)&c->d_rep_len, &c->rep_len_cap, F * sizeof(int32_t))) return -1;
    if (unixtimes && grow_buffer((void **)&c->d_rep_time, &c->rep_time_cap, F * sizeof(uint32_t))) return -1;
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
    if (nsig < 0 || nsig > 64) return ft8_fail("nsig_per_frame %d out of range [0, 64]", nsig);
    if (!iq_dev || (nsig > 0 && !signals)) return ft8_fail("NULL array argument");
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
