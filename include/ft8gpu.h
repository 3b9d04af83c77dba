/*
 * ft8gpu.h -- C ABI of libft8gpu.so: the FT8 decode hot path of Guenael/rtlsdr-ft8d on one
 * AMD MI355X (gfx950), as hand-written HIP kernels behind the reference's own function boundary.
 *
 * Everything here is plain C: pointers, sizes, POD structs.  No HIP, torch or C++ types.
 * Every entry point names the reference interface it replaces (file:line into the reference).
 *
 * Frame conventions (identical to the reference, rtlsdr_ft8d.h:34-56, rtlsdr_ft8d.c:274-278):
 *   one frame = 15 s at 3200 sps = 48000 complex samples, planar float32: I[48000] then Q[48000].
 *   A batch is `nframes` such frames back to back: iq[nframes][2][48000].
 *   waterfall = uint8 mag[92][2][2][256] (block, time_sub, freq_sub, bin) = 94208 bytes per frame.
 *
 * Environment: FT8GPU_DEVICE=<n> is the GPU used by the drop-in ft8_subsystem (default 0; the reference's function has no
 * device argument).  Nothing else is read from the environment: every test hook is a per-context flag
 * (ft8gpu_set_debug_flags), so that behaviour never depends on the process environment.
 */
#ifndef FT8GPU_H
#define FT8GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden; only the C ABI below is exported */
#pragma GCC visibility push(default)

/* ---- constants: rtlsdr_ft8d.h:34-56 ------------------------------------------------------- */
#define FT8GPU_NSAMPLES        48000   /* SIGNAL_LENGHT * SIGNAL_SAMPLE_RATE */
#define FT8GPU_K_MIN_SCORE     10      /* K_MIN_SCORE      rtlsdr_ft8d.h:43 */
#define FT8GPU_K_MAX_CANDIDATES 120    /* K_MAX_CANDIDATES rtlsdr_ft8d.h:44 */
#define FT8GPU_K_LDPC_ITERS    20      /* K_LDPC_ITERS     rtlsdr_ft8d.h:45 */
#define FT8GPU_K_MAX_MESSAGES  50      /* K_MAX_MESSAGES   rtlsdr_ft8d.h:46 */
#define FT8GPU_NUM_BIN         256     /* NUM_BIN          rtlsdr_ft8d.h:51 */
#define FT8GPU_NFFT            1024    /* NFFT             rtlsdr_ft8d.h:54 */
#define FT8GPU_NUM_BLOCKS      92      /* NUM_BLOCKS       rtlsdr_ft8d.h:55 */
#define FT8GPU_MAG_ARRAY       94208   /* MAG_ARRAY        rtlsdr_ft8d.h:56 */
#define FT8GPU_NN              79      /* FT8_NN (ft8_lib constants.h; used rtlsdr_ft8d.c:933,:947) */
#define FT8GPU_ABS_MAX_CANDIDATES 1024 /* upper bound accepted for ft8gpu_params.max_candidates */

/* ---- ABI structs --------------------------------------------------------------------------- */

/* struct decoder_results, rtlsdr_ft8d.h:136-141 (offsets 0/13/20/24, size 28).  The name is kept
 * so that rtlsdr_ft8d.c compiles against this header unchanged. */
#ifndef FT8GPU_NO_DECODER_RESULTS
struct decoder_results {
    char    call[13];
    char    loc[7];
    int32_t freq;
    int32_t snr;
};
#endif

/* candidate_t of ft8_lib decode.h, as used at rtlsdr_ft8d.c:1439, :1466-1470 (8 bytes) */
typedef struct {
    int16_t score;
    int16_t time_offset;
    int16_t freq_offset;
    uint8_t time_sub;
    uint8_t freq_sub;
} ft8gpu_candidate;

/* Per-candidate outcome of ft8_decode(): message_t + decode_status_t of ft8_lib decode.h
 * (rtlsdr_ft8d.c:1474-1487, :1494) folded into one 48-byte record, plus the packed 91 bits. */
typedef struct {
    int16_t  ldpc_errors;     /* decode_status_t.ldpc_errors (min parity errors seen, 0 = codeword) */
    int16_t  iters;           /* BP iterations entered before exit (diagnostic) */
    uint16_t crc_extracted;   /* valid when ldpc_errors == 0 */
    uint16_t crc_calculated;  /* valid when ldpc_errors == 0 */
    int8_t   unpack_status;   /* valid when CRCs match; < 0 = unpack77 failed */
    uint8_t  ok;              /* 1 iff ft8_decode() would have returned true */
    uint8_t  a91[12];         /* packed payload+CRC bits of the last hard decision */
    char     text[25];        /* message_t.text (valid when ok) */
    uint8_t  pad;
} ft8gpu_decode_status;

/* run-time forms of K_MIN_SCORE / K_MAX_CANDIDATES / K_LDPC_ITERS (rtlsdr_ft8d.h:43-45) */
typedef struct {
    int32_t min_score;
    int32_t max_candidates;
    int32_t ldpc_iters;
} ft8gpu_params;

typedef struct ft8gpu_ctx ft8gpu_ctx;

/* flags for the batch entry points */
#define FT8GPU_HOST_PTRS    0   /* all array arguments are host memory; copies are staged by the library */
#define FT8GPU_DEVICE_PTRS  1   /* all array arguments are device (HBM) pointers on the context's GPU */

/* per-stage kernel timings of ft8gpu_decode_batch(), hipEvent-measured on the context stream and
 * averaged over the pipeline runs recorded since ft8gpu_enable_timing(ctx, 1) (ring of 32 runs;
 * recording does not synchronise the host) */
typedef struct {
    float waterfall_ms;
    float sync_ms;        /* score + compaction */
    float heap_ms;        /* top-N selection (exact heap replay) */
    float decode_ms;      /* LLR + LDPC BP + CRC + unpack */
    float spots_ms;       /* dedup + CQ spot fill */
    float total_ms;       /* first kernel start to last kernel end */
    int32_t launches_per_stage; /* 1, or 2 when a large batch is processed as two overlapped parts (the first quarter and the
                                   rest): the heap replay of one part then runs on a side stream under the other part's
                                   kernels, and the per-stage figures are sums over both launches of a stage (the spot
                                   collection is one launch for both parts) */
} ft8gpu_timings;

/* ---- lifecycle: replaces initFFTW()/freeFFTW(), rtlsdr_ft8d.c:314-347 ----------------------- */

/* Creates a decoder context on GPU `device` with persistent buffers for up to `max_frames` frames
 * (larger batches are processed in chunks of max_frames).  `params` may be NULL (reference defaults
 * 10 / 120 / 20).  Returns 0 on success.  A context owns its intermediate buffers; every entry point
 * holds the context's mutex, so two host threads calling into ONE context serialise (use one context
 * per thread / per GPU for concurrency: contexts are independent, unlike the reference's process-global
 * FFTW state, rtlsdr_ft8d.c:57-60).  Entry points make the context's GPU current for their duration and
 * restore the caller's current device before returning. */
int  ft8gpu_create(ft8gpu_ctx **out, int device, int max_frames, const ft8gpu_params *params);
void ft8gpu_destroy(ft8gpu_ctx *ctx);
/* Use an existing hipStream_t (passed as void*) for all work of this context.  NULL = the context
 * creates its own non-blocking stream; FT8GPU_STREAM_LEGACY (= hipStreamLegacy) selects the legacy
 * null stream explicitly.  The call synchronises the old and the new stream and runs the co-execution probe of
 * ft8gpu_overlap_active on the new one (a few one-thread kernels and host synchronisations, well under a millisecond
 * when the streams co-run): do not call it on a stream that is being captured into a graph. */
int  ft8gpu_set_stream(ft8gpu_ctx *ctx, void *hip_stream);
#define FT8GPU_STREAM_LEGACY ((void *)1)
/* The hipStream_t (as void*) the context enqueues on, so that a caller can order its own work behind the decoder's
 * (hipStreamWaitEvent / an RCCL collective) without moving the decoder onto a foreign stream: the context's streams
 * are created together and map to distinct hardware queues, which the overlap of the heap / spots kernels with the
 * throughput kernels relies on (measured: the same pipeline takes 1.48 ms on a borrowed framework stream against 1.28 ms
 * on the context's own at 1024 frames, cap 480). */
void *ft8gpu_get_stream(ft8gpu_ctx *ctx);
/* 1: batches of >= 512 frames run the two-part pipeline with the serial kernels (heap replay, spot collection) on side
 * streams under the throughput kernels of the other part; 0: plain pipeline, one launch per stage.  Whether streams run
 * side by side depends on which hardware queues HIP hands out, so the context measures it at ft8gpu_create and again at
 * ft8gpu_set_stream (a 2 ms co-execution probe per pair of streams), replaces side streams that share a queue with another
 * one, and only falls back when that does not help.  Records are identical either way.  A pure query: it does not touch
 * ft8gpu_last_error().  The probe is a timing measurement: on a GPU shared with other work, or under a profiler that
 * serialises kernels, it can come out 0 -- ft8gpu_overlap_reason says why. */
int  ft8gpu_overlap_active(ft8gpu_ctx *ctx);
/* why the plain pipeline runs ("" while the overlapped one is active), copied into buf (NUL-terminated, truncated to cap) */
int  ft8gpu_overlap_reason(ft8gpu_ctx *ctx, char *buf, size_t cap);
/* test hooks, per context (any combination; 0 = product behaviour) */
#define FT8GPU_DBG_FORCE_IEEE_DIV 1u  /* LDPC kernel: the compiler's IEEE division everywhere (the guard's fallback path) */
#define FT8GPU_DBG_PIPELINE_FORM  2u  /* ft8gpu_decode_candidates runs the form of the LDPC kernel ft8gpu_decode_batch
                                         uses (no exact error count: ldpc_errors is 0 or 83) */
#define FT8GPU_DBG_NO_OVERLAP     4u  /* one launch per stage for the whole batch: no two-half overlap, no chunked upload */
#define FT8GPU_DBG_ALL            7u
/* (Alternative, bit-identical forms of the waterfall and heap kernels are not in this library: they are compiled into the
 * A/B build only -- `make -C rtlsdr_ft8d_amd/csrc ab` -> libft8gpu_ab.so, whose ft8gpu_set_debug_flags accepts the extra
 * selector bits of csrc/ft8gpu_internal.h.) */
int  ft8gpu_set_debug_flags(ft8gpu_ctx *ctx, unsigned flags);   /* unknown bits are refused */
/* Identity of the build: "<dev>.<all>[+ab]" -- 16 hex digits of SHA-256 over the device sources (csrc/ *.hip, *.h) and 16 over
 * every source of the library (those + csrc/ *.c, Makefile, include/), computed by the Makefile when the library is
 * linked.  rtlsdr_ft8d_amd.source_build_id() recomputes it from the tree: smoke(), the GPU tests and bench.py refuse a
 * library whose id differs from the sources beside it (a stale or foreign .so fails instead of producing numbers). */
const char *ft8gpu_build_id(void);
/* Proof by exhaustion behind the LDPC kernel's short division chains (csrc/bp_math.h): fast_tanh / fast_atanh of
 * ft8_lib ldpc.c (reached through ft8_decode, rtlsdr_ft8d.c:1476) are functions of one float, so all 2^32 inputs are
 * evaluated on the GPU, fast form against the compiler's IEEE-754 division, on the domain the kernel's guard
 * establishes.  out[0..6] = tanh inputs, tanh mismatches, atanh inputs, atanh mismatches, packed-form mismatches,
 * float bits of max |fast_tanh|, one offending input pattern (0 = none).  About 40 ms. */
int  ft8gpu_selftest_bp_math(ft8gpu_ctx *ctx, uint64_t out[7]);
/* The same kind of proof for the scale factor of the soft bits, sqrtf(24.0f / variance) (ftx_normalize_logl of ft8_lib decode.c,
 * reached through ft8_decode, rtlsdr_ft8d.c:1476): for every float v in [2^-60, 2^60] the quotient and the root the LDPC kernel
 * computes are checked against exact arithmetic (products and squares that are exact in double), not against another
 * division or root.  The same exact test is applied to the divisions of fast_tanh / fast_atanh themselves on their whole
 * domains -- the "IEEE quotient" ft8gpu_selftest_bp_math compares the short chains with.  out[0..6] = inputs, quotients 24/v not
 * correctly rounded, roots not correctly rounded, kernel function != sqrtf(24.0f / v), one offending input pattern (0 = none),
 * rational-function divisions tested, of those not correctly rounded.  (Round 5: HIP's __fsqrt_rn turned out to be the 1-ulp
 * native root.) */
int  ft8gpu_selftest_norm_math(ft8gpu_ctx *ctx, uint64_t out[7]);
int  ft8gpu_set_params(ft8gpu_ctx *ctx, const ft8gpu_params *params);
int  ft8gpu_enable_timing(ft8gpu_ctx *ctx, int on);
int  ft8gpu_get_timings(ft8gpu_ctx *ctx, ft8gpu_timings *out, int32_t *nruns);
int  ft8gpu_synchronize(ft8gpu_ctx *ctx);
/* The reference path has no error channel (void ft8_subsystem, rtlsdr_ft8d.h:164); failures are
 * reported here and by the int return codes of the batch API (0 = ok, <0 = error). */
const char *ft8gpu_last_error(void);
int  ft8gpu_device_count(void);

/* ---- the whole path: ft8_subsystem(), rtlsdr_ft8d.c:1387-1524, for `nframes` frames ----------
 * decodes:   [nframes][50] struct decoder_results.  Exactly as the reference (:1509-1520), slot k
 *            of a frame is written only if the k-th unique message of that frame starts with "CQ";
 *            other slots are left untouched.  n_results[f] = number of unique messages (:1523).
 * Fences (documented deviations where the reference is undefined): more than 50 unique messages
 * in a frame -> the surplus is dropped (reference: infinite loop); missing tokens -> "(null)". */
int ft8gpu_decode_batch(ft8gpu_ctx *ctx, const float *iq, int nframes,
                        struct decoder_results *decodes, int32_t *n_results, int flags);

/* ---- the same across several GPUs of one node, for a plain C caller ------
 * ctxs[0..ndev): one context per GPU (normally ft8gpu_create(&ctxs[g], g, ...); several contexts on one
 * GPU are legal).  Frames are independent, so the batch is cut into ndev contiguous shards (shard g =
 * frames [g*n/ndev, (g+1)*n/ndev)), each decoded by its own host thread on its own context, and every
 * shard's records land directly at their frame offsets in the caller's HOST arrays -- the gather of the
 * 1 404 B/frame spot records is that placement; no collective is needed when the list is consumed on
 * the host, as the daemon does.  iq / decodes / n_results are host memory (FT8GPU_HOST_PTRS layout of
 * ft8gpu_decode_batch; take them from ft8gpu_host_alloc when throughput matters: uploads from pageable memory are
 * staged and do not overlap the kernels).  Returns 0, or -1 with ft8gpu_last_error() naming the failing shard. */
int ft8gpu_decode_batch_multi(ft8gpu_ctx *const *ctxs, int ndev, const float *iq, int nframes,
                              struct decoder_results *decodes, int32_t *n_results);
/* Device-resident form: shard g's frames already sit in HBM of ctxs[g]'s GPU (iq_dev[g]: [nframes_dev[g]][2][48000],
 * e.g. synthesised or decimated there); records are gathered into the host arrays in shard order. */
int ft8gpu_decode_batch_multi_dev(ft8gpu_ctx *const *ctxs, int ndev, const float *const *iq_dev,
                                  const int *nframes_dev, struct decoder_results *decodes, int32_t *n_results);

/* Device-resident gather of the spot list over RCCL / xGMI: after every GPU g has decoded its shard
 * into its own HBM (ft8gpu_decode_batch with FT8GPU_DEVICE_PTRS: decodes_dev[g] = [frames_per_dev][50] records,
 * n_results_dev[g] = [frames_per_dev] counts), ONE grouped all-gather per buffer leaves the whole job's list, in shard
 * order, on EVERY GPU: all_decodes_dev[g] = [ndev * frames_per_dev][50], all_n_results_dev[g] = [ndev * frames_per_dev].
 * The collectives are enqueued on each context's own stream (ordered behind the kernels that produced the records; the
 * host is not synchronised: ft8gpu_synchronize(ctxs[g]) waits).  Single-process RCCL (ncclCommInitAll over the contexts'
 * devices, created on first use, one rank per GPU -- two contexts on one GPU are refused); librccl is bound at run time,
 * so -1 with "RCCL unavailable" on a box without it.  Equal shard sizes only (pad the last shard).  Replaces nothing in
 * the reference (single decoder thread, rtlsdr_ft8d.c:221-285); the host-side gather of ft8gpu_decode_batch_multi is what
 * a daemon consumes.  ft8gpu_gather_shutdown() destroys the communicators. */
int  ft8gpu_gather_spots(ft8gpu_ctx *const *ctxs, int ndev, const struct decoder_results *const *decodes_dev,
                         const int32_t *const *n_results_dev, int frames_per_dev,
                         struct decoder_results *const *all_decodes_dev, int32_t *const *all_n_results_dev);
void ft8gpu_gather_shutdown(void);
/* host worker threads the multi-GPU entries keep alive between calls (created on first use; diagnostic) */
int  ft8gpu_shard_workers(void);

/* ---- stage entries (same data, stage by stage; used by the parity tests) --------------------- */
/* rtlsdr_ft8d.c:1395-1435: window, 184 FFTs, log-magnitude, quantise.  mag: [nframes][94208] */
int ft8gpu_waterfall(ft8gpu_ctx *ctx, const float *iq, int nframes, uint8_t *mag, int flags);
/* ft8_find_sync(&power, K_MAX_CANDIDATES, candidate_list, K_MIN_SCORE), rtlsdr_ft8d.c:1450.
 * cands: [nframes][max_candidates], counts: [nframes] */
int ft8gpu_find_sync(ft8gpu_ctx *ctx, const uint8_t *mag, int nframes,
                     ft8gpu_candidate *cands, int32_t *counts, int flags);
/* every sync score of the scan, int16 [nframes][2][2][36][249] (diagnostic / parity) */
int ft8gpu_score_map(ft8gpu_ctx *ctx, const uint8_t *mag, int nframes, int16_t *scores, int flags);
/* ft8_decode(&power, cand, &message, K_LDPC_ITERS, &status) for every candidate, :1476.
 * status: [nframes][max_candidates] */
int ft8gpu_decode_candidates(ft8gpu_ctx *ctx, const uint8_t *mag, const ft8gpu_candidate *cands,
                             const int32_t *counts, int nframes, ft8gpu_decode_status *status, int flags);
/* dedup hash table + CQ filter + spot fill, rtlsdr_ft8d.c:1452-1460, :1487-1523 */
int ft8gpu_collect_spots(ft8gpu_ctx *ctx, const ft8gpu_candidate *cands, const int32_t *counts,
                         const ft8gpu_decode_status *status, int nframes,
                         struct decoder_results *decodes, int32_t *n_results, int flags);

/* ---- tooling: encoder + synthetic frames (pack77 / ft8_encode / CPFSK synth of
 *      decoderSelfTest, rtlsdr_ft8d.c:924-955) --------------------------------------------- */
/* Message text -> 77 bits in 10 bytes (pack77, :927); 0 = ok, -1 = the text fits no message type.  Tokens are separated
 * by blanks.  Tried in this order:
 *   telemetry   one token of 18 hexadecimal digits (the first 0..7)                                       i3.n3 = 0.5
 *   type 1 / 2  FIELD1 CALL2 [GRID4 | R GRID4 | +NN | -NN | R+NN | R-NN | RRR | RR73 | 73]                  i3 = 1 / 2
 *               FIELD1 = CQ | CQ nnn | CQ aaaa | DE | QRZ | call; a call is a standard call sign, optionally with
 *               /R (type 1) or /P (type 2), or <CALL> (sent as a 22-bit hash: receivers without a hash table -- the
 *               reference's ft8_lib era -- print "<...>"); reports -30 .. +99
 *   type 4      <CALL> LONGCALL [RRR | RR73 | 73]  |  LONGCALL <CALL> [...]  |  CQ LONGCALL                    i3 = 4
 *               LONGCALL: up to 11 characters of [0-9A-Z/], sent in full; the bracketed call as a 12-bit hash
 *   free text   up to 13 characters of [ 0-9A-Z+-./?]                                                   i3.n3 = 0.0
 * (ft8_lib's pack77 of the reference's era packs the type 1 forms without suffixes, brackets, "R GRID4" and CQ
 * modifiers, and turns everything else into free text.) */
int  ft8gpu_pack77(const char *msg, uint8_t payload[10]);
/* the strict subset of it: "CALL1 CALL2 [GRID4]" with plain standard calls (CQ / DE / QRZ allowed first), else -1 */
int  ft8gpu_pack77_std(const char *msg, uint8_t payload[10]);
/* payload -> 79 tone numbers (ft8_encode, :934) */
void ft8gpu_encode(const uint8_t payload[10], uint8_t tones[FT8GPU_NN]);

typedef struct {
    uint8_t tones[FT8GPU_NN];
    uint8_t pad;
    float   f0_hz;        /* frequency of tone 0 */
    float   t0_s;         /* start time within the frame */
    float   amplitude;    /* linear amplitude (noise has unit power in 3200 Hz before normalisation) */
} ft8gpu_synth_signal;

/* Synthesises frames directly in HBM: complex AWGN (variance noise_sigma^2 per component) plus
 * nsig_per_frame CPFSK signals per frame (plain FSK as rtlsdr_ft8d.c:946-955), then peak-normalises
 * each frame to 0.5 as the decoder thread does (rtlsdr_ft8d.c:248-263).
 * signals: host array [nframes][nsig_per_frame]; iq_dev: device pointer [nframes][2][48000]. */
int ft8gpu_synth_frames(ft8gpu_ctx *ctx, const ft8gpu_synth_signal *signals, int nframes,
                        int nsig_per_frame, float noise_sigma, uint64_t seed, float *iq_dev);
/* The same for a shard of a larger job: frame k of this call is global frame first_frame + k, and the
 * noise of a frame depends on (seed, global frame index) only -- any partition of the job over ranks,
 * GPUs or calls synthesises the same frames.  ft8gpu_synth_frames == first_frame 0. */
int ft8gpu_synth_frames_at(ft8gpu_ctx *ctx, const ft8gpu_synth_signal *signals, int nframes,
                           int nsig_per_frame, float noise_sigma, uint64_t seed, uint64_t first_frame,
                           float *iq_dev);

/* ---- RX front end: rtlsdr_callback(), rtlsdr_ft8d.c:76-202 ------------
 * Whole raw RTL-SDR captures (unsigned 8-bit I,Q interleaved at 2.4 Msps) -> the 15 s / ~3200 sps
 * float frames the decoder consumes: fs/4 mixer, CIC (N = 2, comb delay 2, effective ratio 751),
 * 57-tap compensation FIR, scaling; every capture starts from the reset filter state.  Samples past
 * npairs/751 are zero as after the decoder thread's tail zeroing (:243-246); normalise != 0 applies
 * its peak normalisation to 0.5 (:248-263), after which `iq` can go straight into ft8gpu_decode_batch.
 * raw: [ncaptures][2*npairs] bytes, npairs a multiple of 8, 16-byte aligned; iq: [ncaptures][2][48000]. */
int ft8gpu_rx_decimate(ft8gpu_ctx *ctx, const uint8_t *raw, int ncaptures, size_t npairs,
                       float *iq, int normalise, int flags);

/* ---- spot reporting wire formats ------------------------------------
 * The bytes postSpots() (rtlsdr_ft8d.c:365-590) assembles for report.pskreporter.info:4739 (IPFIX:
 * 16-byte header, receiver and sender template sets, receiver record, one sender record per slot
 * of decodes[0..n_results)), built for a whole batch of spot lists at once; nothing is sent.
 * Reference behaviour kept: every slot below n_results is emitted, also the untouched slots of
 * non-CQ messages (:494-533); a record is started only while the sender block is <= 1200 bytes
 * (:497); SNR byte = (int8)snr - 20 (:511); both data blocks are zero-padded to 4 bytes.
 * Fenced: call / loc are read up to 12 / 6 characters (the reference's strlen has no bound). */
#define FT8GPU_DATAGRAM_STRIDE 1408        /* >= the largest datagram (168 + 1236 bytes) */
typedef struct {
    char     rcall[13];        /* dec_options.rcall, rtlsdr_ft8d.h:132 */
    char     rloc[7];          /* dec_options.rloc,  rtlsdr_ft8d.h:133 */
    char     app_version[32];  /* pskreporter_app_version, rtlsdr_ft8d.c:72 */
    uint32_t dial_freq;        /* dec_options.freq, added to every spot's audio offset (:507) */
    uint32_t unixtime;         /* header export time and spot time (:445, :531) unless unixtimes != NULL */
    uint32_t sequence;         /* 1 in the reference (:425) */
    uint32_t random_id;        /* :430-435 */
} ft8gpu_report_info;
/* decodes: [nframes][50], n_results: [nframes], unixtimes: [nframes] or NULL,
 * datagrams: [nframes][FT8GPU_DATAGRAM_STRIDE] (bytes past the length are zero), lengths: [nframes] */
int ft8gpu_pskreporter_datagrams(ft8gpu_ctx *ctx, const struct decoder_results *decodes,
                                 const int32_t *n_results, int nframes, const ft8gpu_report_info *info,
                                 const uint32_t *unixtimes, uint8_t *datagrams, int32_t *lengths, int flags);
/* the stdout table of printSpots() (:643-663) for one frame's spot list, into `out` (NUL-terminated,
 * truncated to cap); returns the untruncated length.  Host-side text formatting, no GPU involved. */
int ft8gpu_format_spots(const struct decoder_results *decodes, int32_t n_results, uint32_t dial_freq,
                        int year, int month, int mday, int hour, int minute, char *out, size_t cap);

/* device memory helpers so that a plain C caller needs no HIP headers; they act on the context's GPU
 * (whatever device is current in the calling thread) and order after the context's enqueued work */
void *ft8gpu_dev_alloc(ft8gpu_ctx *ctx, size_t bytes);
void  ft8gpu_dev_free(ft8gpu_ctx *ctx, void *p);
int   ft8gpu_memcpy_h2d(ft8gpu_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int   ft8gpu_memcpy_d2h(ft8gpu_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Page-locked host memory (hipHostMalloc / hipHostFree) for the buffers handed to the FT8GPU_HOST_PTRS entries:
 * ft8gpu_decode_batch uploads in 512-frame chunks with asynchronous copies under the kernels of the previous chunk,
 * and ft8gpu_decode_batch_multi does the same on every GPU at once -- which only overlaps (and only reaches the
 * link rate, about 50 GB/s per GPU) from pinned memory; from malloc'ed memory every copy is staged and serialises with
 * the kernels.  The reference's own buffers are plain static arrays (rtlsdr_ft8d.c:274-278): a daemon decoding one frame
 * per 15 s does not need this, a replay that feeds thousands of frames does.  NULL + ft8gpu_last_error() on failure. */
void *ft8gpu_host_alloc(size_t bytes);
void  ft8gpu_host_free(void *p);

/* ---- drop-in symbols of the reference (rtlsdr_ft8d.h:155-156, :164) --------------------------
 * Link rtlsdr_ft8d.c against libft8gpu.so instead of its own ft8_subsystem/initFFTW/freeFFTW
 * (INTEGRATION.md).  They drive a process-global single-frame context on GPU 0
 * (env FT8GPU_DEVICE overrides). */
void initFFTW(void);
void freeFFTW(void);
void ft8_subsystem(float *iSamples, float *qSamples, uint32_t samples_len,
                   struct decoder_results *decodes, int32_t *n_results);

/* .iq / .c2 readers and writer with the reference's conventions (rtlsdr_ft8d.c:744-856):
 * interleaved float32 I,Q on disk, Q negated, peak-normalised to 0.5 on load. */
int32_t ft8gpu_read_raw_iq(float *iSamples, float *qSamples, const char *filename);
int32_t ft8gpu_read_c2(float *iSamples, float *qSamples, const char *filename, double *dialfreq);
int32_t ft8gpu_write_raw_iq(const float *iSamples, const float *qSamples, const char *filename);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* FT8GPU_H */
