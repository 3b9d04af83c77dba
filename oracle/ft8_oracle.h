/*
 * ft8_oracle.h -- CPU ORACLE for the FT8 decode hot path of Guenael/rtlsdr-ft8d.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libft8gpu.so) never links,
 * loads or calls anything in oracle/.
 *
 * What it restates (plain C, one function per reference function, file:line cited at
 * each definition in ft8_oracle.c):
 *   - rtlsdr_ft8d.c:314-335   initFFTW()        window table (sine window named "hann")
 *   - rtlsdr_ft8d.c:1387-1524 ft8_subsystem()   waterfall, candidate loop, dedup, CQ spot fill
 *   - rtlsdr_ft8d.c:890-972   decoderSelfTest() signal synthesis with libc rand()
 *   - rtlsdr_ft8d.c:744-856   .iq / .c2 file readers + writer
 *   - kgoba/ft8_lib (git submodule at ft8_lib/, EMPTY in the reference snapshot, pinned
 *     commit unrecoverable; API era: has waterfall_t.protocol / PROTO_FT8 / message_t{text,hash},
 *     i.e. between the Nov-2021 FT4 commit and the 2022 ftx_* rename):
 *       decode.c  ft8_find_sync / ft8_sync_score / heap, ft8_decode, ft8_extract_likelihood,
 *                 ftx_normalize_logl            (call sites rtlsdr_ft8d.c:1450, :1476)
 *       ldpc.c    bp_decode, ldpc_check, fast_tanh, fast_atanh
 *       crc.c     ftx_compute_crc, ftx_extract_crc, ftx_add_crc
 *       unpack.c  unpack77 (types 0.0, 0.5, 1, 2, 4), text.c helpers
 *       pack.c    pack77 (standard type-1 messages only), encode.c ft8_encode
 *     These are restated from the published algorithm.
 *
 * PARITY PINNING STATUS: "parity unpinned" for everything that lives in ft8_lib, except
 * for the two fixed points the reference itself carries:
 *   (1) the known-answer comment rtlsdr_ft8d.c:919-923 (message -> 10 packed bytes -> 79 tones),
 *       which pins pack77, CRC-14, the LDPC generator, the Gray map and the Costas layout;
 *   (2) the -t self-test pass condition rtlsdr_ft8d.c:966-971 (decode of the synthesised
 *       frame yields call "K1JT" / loc "FN20" in slot 0).
 * Both are checked in tests/test_oracle.py against tests/golden/.  The reference cannot be
 * compiled here (no ft8_lib sources, no fftw3.h, no rtl-sdr.h), so there is no oracle/_ref.
 *
 * FFT: the reference calls fftwf (FFTW_ESTIMATE plan, rtlsdr_ft8d.c:326); FFTW is absent and
 * its codelet choice is machine dependent, so no implementation can be bit-identical to it.
 * The oracle uses a 5-stage radix-4 decimation-in-frequency complex FFT in float32 with a
 * double-precision-derived twiddle table and a fully specified operation order (see
 * ft8o_fft1024); it is validated against a float64 FFT in the tests.  The product kernel
 * implements the same operation order, so waterfalls compare bit-exact.
 */
#ifndef FT8_ORACLE_H
#define FT8_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rtlsdr_ft8d.h:34-56 */
#define FT8O_SIGNAL_LENGHT      15
#define FT8O_SAMPLE_RATE        3200
#define FT8O_NSAMPLES           (FT8O_SIGNAL_LENGHT * FT8O_SAMPLE_RATE) /* 48000 */
#define FT8O_K_MIN_SCORE        10
#define FT8O_K_MAX_CANDIDATES   120
#define FT8O_K_LDPC_ITERS       20
#define FT8O_K_MAX_MESSAGES     50
#define FT8O_K_FREQ_OSR         2
#define FT8O_K_TIME_OSR         2
#define FT8O_K_FSK_DEV          6.25f
#define FT8O_NUM_BIN            256
#define FT8O_BLOCK_SIZE         512
#define FT8O_SUB_BLOCK_SIZE     256
#define FT8O_NFFT               1024
#define FT8O_NUM_BLOCKS         92
#define FT8O_MAG_ARRAY          94208
#define FT8O_BLOCK_STRIDE       (FT8O_K_TIME_OSR * FT8O_K_FREQ_OSR * FT8O_NUM_BIN) /* 1024 */

#define FT8O_NN   79
#define FT8O_ND   58
#define FT8O_LDPC_N 174
#define FT8O_LDPC_K 91
#define FT8O_LDPC_M 83
#define FT8O_LDPC_K_BYTES 12

/* rtlsdr_ft8d.h:136-141 */
struct ft8o_decoder_results {
    char    call[13];
    char    loc[7];
    int32_t freq;
    int32_t snr;
};

/* ft8_lib decode.h candidate_t (used at rtlsdr_ft8d.c:1439, :1466-1470) */
typedef struct {
    int16_t score;
    int16_t time_offset;
    int16_t freq_offset;
    uint8_t time_sub;
    uint8_t freq_sub;
} ft8o_candidate_t;

/* ft8_lib decode.h message_t (rtlsdr_ft8d.c:1454, :1474, :1487, :1494) */
typedef struct {
    char     text[25];
    uint16_t hash;
} ft8o_message_t;

/* ft8_lib decode.h decode_status_t (rtlsdr_ft8d.c:1475-1481) */
typedef struct {
    int      ldpc_errors;
    uint16_t crc_extracted;
    uint16_t crc_calculated;
    int      unpack_status;
} ft8o_decode_status_t;

/* extra per-candidate diagnostics (not part of ft8_lib's API) */
typedef struct {
    int     iters;        /* BP iterations entered (index of the iteration that broke, or max) */
    uint8_t a91[12];      /* packed 91 bits of the last hard decision */
} ft8o_decode_extra_t;

typedef struct {
    int min_score;        /* K_MIN_SCORE       rtlsdr_ft8d.h:43 */
    int max_candidates;   /* K_MAX_CANDIDATES  rtlsdr_ft8d.h:44 */
    int ldpc_iters;       /* K_LDPC_ITERS      rtlsdr_ft8d.h:45 */
} ft8o_params_t;

void ft8o_init(void);                                       /* initFFTW  rtlsdr_ft8d.c:314 */
const float *ft8o_window(void);                             /* hann[1024] */
const float *ft8o_twiddles(void);                           /* [1024][2] (cos, -sin) */

void ft8o_fft1024(float *re, float *im);                    /* in place; output digit-reversed -> natural */
void ft8o_fft1024_f64(const double *re_in, const double *im_in, double *re_out, double *im_out);
uint8_t ft8o_quantise(float mag2);                          /* rtlsdr_ft8d.c:1415-1427 for one bin (fenced for non-finite input) */
uint8_t ft8o_quantise_x86(float mag2);                      /* the same as the reference's x86 build executes it (inf, NaN -> 0) */
void ft8o_set_quantiser_x86(int on);                        /* ft8o_quantise / ft8o_waterfall follow the x86 form (default: fenced) */
void ft8o_waterfall(const float *iSamples, const float *qSamples, uint8_t *mag_power);
void ft8o_waterfall_f64(const float *iSamples, const float *qSamples, uint8_t *mag_power);
/* Optional run-time FFTW leg (the reference's own transform: fftwf_plan_dft_1d(NFFT, in, out, FFTW_FORWARD,
 * FFTW_ESTIMATE) rtlsdr_ft8d.c:326, fftwf_execute :1411), bound with dlopen when libfftw3f.so.3 is installed.
 * ft8o_fftw_init: 1 = bound (ft8o_fftw_detail() = the library found), 0 = not (detail = the names searched).
 * explicit_path is tried first (tests); bench.py passes NULL. */
int  ft8o_fftw_init(const char *explicit_path);
const char *ft8o_fftw_detail(void);
int  ft8o_waterfall_fftw(const float *iSamples, const float *qSamples, uint8_t *mag_power);   /* -1: FFTW not bound */

int  ft8o_sync_score(const uint8_t *mag, const ft8o_candidate_t *c);
int  ft8o_find_sync(const uint8_t *mag, int num_candidates, ft8o_candidate_t *heap, int min_score);
void ft8o_score_map(const uint8_t *mag, int16_t *scores /* [2][2][36][249] */);

void ft8o_extract_likelihood(const uint8_t *mag, const ft8o_candidate_t *c, float *log174);
void ft8o_normalize_logl(float *log174);
void ft8o_bp_decode(const float *codeword, int max_iters, uint8_t *plain, int *ok, int *iters_out);
int  ft8o_ldpc_check(const uint8_t *plain);
uint16_t ft8o_compute_crc(const uint8_t *message, int num_bits);
uint16_t ft8o_extract_crc(const uint8_t *a91);
void ft8o_add_crc(const uint8_t *payload, uint8_t *a91);
int  ft8o_unpack77(const uint8_t *a77, char *message);
int  ft8o_pack77(const char *msg, uint8_t *c77);
void ft8o_encode(const uint8_t *payload, uint8_t *tones);
int  ft8o_decode(const uint8_t *mag, const ft8o_candidate_t *cand, ft8o_message_t *message,
                 int max_iterations, ft8o_decode_status_t *status, ft8o_decode_extra_t *extra);
/* ft8o_find_sync for B waterfalls (OpenMP over frames): cands [B][cap], zero behind each count */
void ft8o_find_sync_batch(const uint8_t *mag, int B, int cap, int min_score, ft8o_candidate_t *cands, int32_t *counts, int nthreads);
/* ft8o_decode for every candidate of B frames as canonical 48-byte records (layout of ft8gpu_decode_status) */
void ft8o_decode_candidates_batch(const uint8_t *mag, const ft8o_candidate_t *cands, const int32_t *counts, int B, int cap,
                                  int max_iterations, uint8_t *records, int nthreads);

/* rtlsdr_ft8d.c:1387-1524; _ex takes the three compile-time constants as run-time parameters */
void ft8o_subsystem(const float *iSamples, const float *qSamples, uint32_t samples_len,
                    struct ft8o_decoder_results *decodes, int32_t *n_results);
void ft8o_subsystem_ex(const float *iSamples, const float *qSamples, const ft8o_params_t *p,
                       struct ft8o_decoder_results *decodes, int32_t *n_results);
/* same, starting from a given waterfall (stage-isolated parity) */
void ft8o_subsystem_from_waterfall(const uint8_t *mag, const ft8o_params_t *p,
                                   struct ft8o_decoder_results *decodes, int32_t *n_results);
/* the candidate loop rtlsdr_ft8d.c:1452-1523 on a given candidate list (what follows ft8_find_sync, :1450) */
void ft8o_spots_from_candidates(const uint8_t *mag, const ft8o_candidate_t *candidate_list, int num_candidates,
                                const ft8o_params_t *p, struct ft8o_decoder_results *decodes, int32_t *n_results);
/* batch forms (OpenMP over frames): waterfalls with the float32 R4DIF FFT (f64 = 0) or the float64 DFT (f64 = 1);
 * everything after the waterfall; everything after ft8_find_sync (configs[1]: cands [B][p->max_candidates]) */
void ft8o_waterfall_batch(const float *iq, int B, uint8_t *mag, int f64 /* 0 R4DIF, 1 float64, 2 FFTW (if bound) */, int nthreads);
int  ft8o_subsystem_batch_fftw(const float *iq, int B, const ft8o_params_t *p,
                               struct ft8o_decoder_results *decodes, int32_t *n_results, int nthreads);   /* -1: FFTW not bound */
void ft8o_subsystem_from_waterfall_batch(const uint8_t *mag, int B, const ft8o_params_t *p,
                                         struct ft8o_decoder_results *decodes, int32_t *n_results, int nthreads);
void ft8o_decode_from_candidates_batch(const uint8_t *mag, const ft8o_candidate_t *cands, const int32_t *counts, int B,
                                       const ft8o_params_t *p, struct ft8o_decoder_results *decodes, int32_t *n_results,
                                       int nthreads);
/* B independent frames, iq planar [B][2][48000]; nthreads OpenMP threads (cpu_baseline leg) */
void ft8o_subsystem_batch(const float *iq, int B, const ft8o_params_t *p,
                          struct ft8o_decoder_results *decodes /* [B][50] */, int32_t *n_results,
                          int nthreads);

/* rtlsdr_ft8d.c:890-955: the -t signal.  Uses libc rand() seeded with `seed` (reference: unseeded = 1). */
int  ft8o_selftest_signal(float *iSamples, float *qSamples, unsigned seed);
void ft8o_normalise(float *iSamples, float *qSamples, int n);   /* rtlsdr_ft8d.c:248-263 */
/* rtlsdr_ft8d.c:946-955 generalised: nsig plain-FSK signals (phase accumulated in double), no noise */
void ft8o_synth_cpfsk(const uint8_t *tones, const double *f_tone0_hz, const int *start_sample,
                      const double *amplitude, int nsig, float *iSamples, float *qSamples);
int32_t ft8o_write_raw_iq(const float *iSamples, const float *qSamples, const char *filename);
int32_t ft8o_read_raw_iq(float *iSamples, float *qSamples, const char *filename);
int32_t ft8o_read_c2(float *iSamples, float *qSamples, const char *filename, double *dialfreq);

/* rtlsdr_callback(), rtlsdr_ft8d.c:76-202: RX front end (fs/4 mixer, CIC N=2 with the reference's
 * effective decimation of 751, 57-tap compensation FIR).  The reference keeps the filter state in
 * function-local statics; here it is an explicit struct so that a capture can be replayed from reset. */
typedef struct {
    int32_t  Ix1, Ix2, Qx1, Qx2;
    int32_t  Iy1, It1y, It1z, Qy1, Qt1y, Qt1z;
    int32_t  Iy2, It2y, It2z, Qy2, Qt2y, Qt2z;
    uint32_t decimationIndex;
    float    firI[56], firQ[56];
} ft8o_rx_state_t;
void ft8o_rx_reset(ft8o_rx_state_t *st);
/* processes `samples_count` raw bytes IN PLACE (the reference rewrites the buffer, :129-140) and
 * appends decimated samples to iSamples/qSamples while *iq_index < 48000 (:195-200) */
void ft8o_rx_callback(ft8o_rx_state_t *st, unsigned char *samples, uint32_t samples_count,
                      float *iSamples, float *qSamples, uint32_t *iq_index);
/* whole capture from reset + the decoder thread's tail zeroing and optional peak normalisation (:243-263) */
void ft8o_rx_capture(const unsigned char *raw, size_t nbytes, float *iSamples, float *qSamples,
                     uint32_t *n_out, int normalise);

/* postSpots(), rtlsdr_ft8d.c:365-590: the PSKreporter UDP datagram for one frame's spot list (the
 * reference function returns at its first line, :380, so upstream never sends it; this restates the
 * bytes it would assemble at :386-561).  Returns the datagram length (fullBlockSize, :541); out must
 * hold 1536 bytes.  strlen() of call / loc is bounded by the field sizes. */
typedef struct {
    char     rcall[13];
    char     rloc[7];
    char     app_version[32];
    uint32_t dial_freq;
    uint32_t unixtime;
    uint32_t sequence;
    uint32_t random_id;
} ft8o_report_info;
int ft8o_pskreporter_datagram(const struct ft8o_decoder_results *dec_results, int32_t n_results,
                              const ft8o_report_info *info, unsigned char *out);
/* printSpots(), rtlsdr_ft8d.c:643-663, into a string; returns its length (year/month as printed) */
int ft8o_format_spots(const struct ft8o_decoder_results *dec_results, int32_t n_results, uint32_t dial_freq,
                      int year, int month, int mday, int hour, int minute, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
