// report.hip -- SURVEY.md section 8 f-4: the PSKreporter datagram of postSpots(), rtlsdr_ft8d.c:365-590,
// assembled for a whole batch of spot lists in HBM (one datagram per frame; nothing is sent).
//
// postSpots() appends one variable-length sender record per spot behind a running byte offset
// (txPtr, :494-533).  Here that running offset is an exclusive prefix sum over the 64 lanes of a wave
// (lane = spot slot, <= 50 slots), so every lane knows where its record goes and writes it on its
// own; the "start a record only while txPtr <= 1200" rule (:497) is a predicate on the scanned
// offset, which is monotone, so the kept records are a prefix exactly as with the reference's break.
// The datagram is composed in LDS and leaves as one coalesced, zero-padded store of fixed stride.
#include "ft8gpu_internal.h"

namespace {

constexpr int kStride = FT8GPU_DATAGRAM_STRIDE;
static_assert(kStride % 16 == 0, "datagram rows are stored as dwordx4");

__device__ inline int wave_excl_scan(int v, int lane, int *total) {
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    *total = __shfl(incl, 63, 64);
    return incl - v;
}

__device__ inline int bounded_len(const char *s, int cap) {      // strlen, fenced at the field size
    int n = 0;
    while (n < cap && s[n] != 0) ++n;
    return n;
}

__device__ inline void put_be32(unsigned char *p, uint32_t v) {
    p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}

// one wave per frame, four frames per workgroup
__global__ __launch_bounds__(256)
void ft8_report_kernel(const struct decoder_results *__restrict__ decodes, const int32_t *__restrict__ n_results,
                       int nframes, ReportPrefix pre, const uint32_t *__restrict__ unixtimes,
                       uint8_t *__restrict__ datagrams, int32_t *__restrict__ lengths) {
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[4][kStride];
    __shared__ __attribute__((aligned(16))) uint32_t s_rec[4][kMaxMessages * 7];      // the frame's spot records
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= nframes) return;                                             // wave-uniform
    unsigned char *buf = s_buf[wave];

    uint4 *buf4 = reinterpret_cast<uint4 *>(buf);
    for (int i = lane; i < kStride / 16; i += 64) buf4[i] = make_uint4(0, 0, 0, 0);   // rxInfoData / txInfoData = {0}
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < pre.len; i += 64) buf[i] = pre.bytes[i];           // header, both templates, receiver record

    int n = n_results[frame];
    n = n < 0 ? 0 : (n > kMaxMessages ? kMaxMessages : n);
    const uint32_t now = unixtimes ? unixtimes[frame] : pre.unixtime;

    const uint32_t *src = reinterpret_cast<const uint32_t *>(decodes + (size_t)frame * kMaxMessages);
    for (int i = lane; i < n * 7; i += 64) s_rec[wave][i] = src[i];           // coalesced, 28 B per record
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const struct decoder_results &r = reinterpret_cast<const struct decoder_results *>(s_rec[wave])[lane < n ? lane : 0];
    int lc = 0, ll = 0, rec = 0;
    if (lane < n) {
        lc = bounded_len(r.call, 12);
        ll = bounded_len(r.loc, 6);
        rec = 16 + lc + ll;                       // 1+lc | 4 freq | 1 snr | 1+3 mode | 1+ll | 1 source | 4 time
    }
    int all = 0;
    const int off = 4 + wave_excl_scan(rec, lane, &all);                      // txPtr before this record (:488-492)
    const bool keep = lane < n && off <= 1200;                                // :497
    int kept = 0;
    (void)wave_excl_scan(keep ? rec : 0, lane, &kept);
    int tx_len = 4 + kept;
    tx_len += (4 - (tx_len & 3)) & 3;                                         // :536-537

    unsigned char *tx = buf + pre.len;
    if (keep) {
        unsigned char *p = tx + off;
        *p++ = (unsigned char)lc;                                             // :501-504
        for (int i = 0; i < lc; ++i) *p++ = (unsigned char)r.call[i];
        put_be32(p, (uint32_t)r.freq + pre.dial_freq); p += 4;                // :507
        *p++ = (unsigned char)((int)(int8_t)r.snr - 20);                      // :511
        *p++ = 3; *p++ = 'F'; *p++ = 'T'; *p++ = '8';                         // :515-518
        *p++ = (unsigned char)ll;                                             // :521-524
        for (int i = 0; i < ll; ++i) *p++ = (unsigned char)r.loc[i];
        *p++ = 1;                                                             // :527
        put_be32(p, now);                                                     // :531
    }
    const int total = pre.len + tx_len;                                       // :541
    if (lane == 0) {
        tx[0] = 0x99; tx[1] = 0x93;                                           // :490
        tx[2] = (unsigned char)(tx_len >> 8); tx[3] = (unsigned char)tx_len;  // :543
        buf[2] = (unsigned char)(total >> 8); buf[3] = (unsigned char)total;  // :544
        put_be32(buf + 4, now);                                               // :445
        lengths[frame] = total;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4 *dst = reinterpret_cast<uint4 *>(datagrams + (size_t)frame * kStride);
    for (int i = lane; i < kStride / 16; i += 64) dst[i] = buf4[i];
}

}  // namespace

hipError_t launch_report(const struct decoder_results *decodes, const int32_t *n_results, int nframes,
                         const ReportPrefix &pre, const uint32_t *unixtimes, uint8_t *datagrams,
                         int32_t *lengths, hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    hipLaunchKernelGGL(ft8_report_kernel, dim3((nframes + 3) / 4), dim3(256), 0, s,
                       decodes, n_results, nframes, pre, unixtimes, datagrams, lengths);
    return hipGetLastError();
}
