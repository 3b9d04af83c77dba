// ft8gpu_internal.h -- shared between the HIP translation units of libft8gpu.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ft8gpu.h"

// rtlsdr_ft8d.h:34-56
constexpr int kNSamples   = 48000;   // SIGNAL_LENGHT * SIGNAL_SAMPLE_RATE
constexpr int kNfft       = 1024;    // NFFT
constexpr int kNumBlocks  = 92;      // NUM_BLOCKS
constexpr int kNumBin     = 256;     // NUM_BIN
constexpr int kBlockStride = 1024;   // K_TIME_OSR * K_FREQ_OSR * NUM_BIN  (rtlsdr_ft8d.c:1446)
constexpr int kMagArray   = 94208;   // MAG_ARRAY
constexpr int kRowsPerFrame = 184;   // NUM_BLOCKS * K_TIME_OSR FFT rows
constexpr int kMaxMessages = 50;     // K_MAX_MESSAGES

// sync scan geometry (ft8_lib ft8_find_sync): time_offset in [-12, 24), freq_offset in [0, 249)
constexpr int kT0Min = -12, kT0Count = 36, kF0Count = 249;
constexpr int kSegments = 4;                                   // (time_sub, freq_sub)
constexpr int kSyncWaves = 8;                                  // waves per sync workgroup
constexpr int kSyncHalves = 2;                                 // a workgroup scores half of the time offsets of a segment
constexpr int kT0PerHalf = kT0Count / kSyncHalves;             // 18
constexpr int kT0PerWaveMax = (kT0PerHalf + kSyncWaves - 1) / kSyncWaves;   // 3 (waves get 2 or 3 time offsets)
constexpr int kSublistCap = (kT0PerWaveMax * kF0Count + 3) & ~3; // 748: the worst case (747 entries) rounded up so that every sub-list starts 16-byte aligned
constexpr int kSublistsPerFrame = kSegments * kSyncHalves * kSyncWaves;     // 64, in scan order: segment, half, wave
constexpr int kScoresPerFrame = kSegments * kT0Count * kF0Count; // 35856

constexpr int kLdpcN = 174, kLdpcK = 91, kLdpcM = 83;

// waterfall kernel work decomposition
constexpr int kWfRowsPerItem = 4;                              // FFT rows per work item
constexpr int kWfItemsPerFrame = kRowsPerFrame / kWfRowsPerItem; // 46

struct Ft8Tables {                 // device-resident constant tables, built on the host at create()
    float  hann[kNfft];            // rtlsdr_ft8d.c:331-334 (sine window)
    float2 tw[kNfft];              // exp(-2 pi i k / 1024) = (cos, -sin), double -> float
    float  qthr[260];              // quantiser thresholds: q(y) = #{k in 1..255 : y >= qthr[k]}
};

// Alternative, bit-identical kernel forms exist only in the A/B build of the library (make ab -> libft8gpu_ab.so,
// -DFT8GPU_AB_FORMS): history to hold the product forms against (tools/ab_libs.py, tests/test_gpu_parity.py), not
// shipped.  The two heap bits exclude each other.
#ifdef FT8GPU_AB_FORMS
#define FT8GPU_AB_WATERFALL_LDS        8u   /* last FFT stage: second exchange through LDS instead of register transposes across the wave's rows */
#define FT8GPU_AB_HEAP_LANE_PER_FRAME 16u   /* heap replay: one lane per frame for every launch the cap allows (<= 128) */
#define FT8GPU_AB_HEAP_WAVE_PER_FRAME 32u   /* heap replay: one wave per frame for every launch */
constexpr unsigned kDbgAccepted = FT8GPU_DBG_ALL | 56u;
#else
constexpr unsigned kDbgAccepted = FT8GPU_DBG_ALL;
#endif

// kernel launchers (each enqueues on `s`, returns hipGetLastError())
// debug_flags: the context's FT8GPU_DBG_* bits (kernel-form selectors are read by the launcher that owns the form)
hipError_t launch_waterfall(const float *iq, uint8_t *mag, const Ft8Tables *tab, int nframes,
                            int num_cus, unsigned debug_flags, hipStream_t s);
hipError_t launch_sync(const uint8_t *mag, uint32_t *lists, int32_t *list_counts, int16_t *score_map,
                       int nframes, int min_score, hipStream_t s);
hipError_t launch_heap(const uint32_t *lists, const int32_t *list_counts, ft8gpu_candidate *cands,
                       int32_t *counts, int nframes, int max_candidates, unsigned debug_flags, hipStream_t s,
                       bool latency_hidden = false);
hipError_t launch_decode(const uint8_t *mag, const ft8gpu_candidate *cands, const int32_t *counts,
                         ft8gpu_decode_status *status, int nframes, int max_candidates, int ldpc_iters,
                         bool count_errors, int force_ieee_div, hipStream_t s);
hipError_t launch_spots(const ft8gpu_candidate *cands, const int32_t *counts,
                        const ft8gpu_decode_status *status, int nframes, int max_candidates,
                        int min_score, struct decoder_results *decodes, int32_t *n_results, hipStream_t s);
hipError_t launch_synth(const ft8gpu_synth_signal *sig_dev, int nframes, int nsig, float noise_sigma,
                        uint64_t seed, uint64_t first_frame, float *iq, hipStream_t s);
hipError_t run_bp_math_selftest(uint64_t out[7], hipStream_t s);   // bp_selftest.hip: exhaustive check of bp_math.h
hipError_t run_norm_math_selftest(uint64_t out[7], hipStream_t s); // bp_selftest.hip: sqrtf(24.0f / v) against exact arithmetic, every float
hipError_t decode_tables_init(hipStream_t s);   // uploads the LDPC edge tables used by the BP kernel
hipError_t launch_rx(const uint8_t *raw, int ncaptures, size_t npairs, void *scratch_sums, void *scratch_base,
                     float *iq, int normalise, hipStream_t s);

// f-4: everything of a PSKreporter datagram that does not depend on the frame (header, receiver and
// sender templates, receiver record), assembled on the host once per call and passed by value
struct ReportPrefix {
    unsigned char bytes[192];
    int32_t  len;
    uint32_t dial_freq;
    uint32_t unixtime;
};
hipError_t launch_report(const struct decoder_results *decodes, const int32_t *n_results, int nframes,
                         const ReportPrefix &pre, const uint32_t *unixtimes, uint8_t *datagrams,
                         int32_t *lengths, hipStream_t s);
