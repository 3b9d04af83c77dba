/*
 * ft8_compat.c -- host side of libft8gpu.so in C, as the reference is C:
 *   * the reference-named drop-in symbols ft8_subsystem / initFFTW / freeFFTW
 *     (rtlsdr_ft8d.h:155-156, :164; definitions rtlsdr_ft8d.c:314-347, :1387-1524)
 *     (the encoder tooling -- pack77, ft8_encode: rtlsdr_ft8d.c:924-934 -- lives in ft8_pack.c)
 *   * the .iq / .c2 replay readers and the .iq writer (rtlsdr_ft8d.c:744-856)
 *   * the ft8_lib-level symbols the reference's own ft8_subsystem() calls (ft8_find_sync, ft8_decode:
 *     rtlsdr_ft8d.c:1450, :1476; pack77, ft8_encode: :927, :934), declared in the headers under include/ft8_lib/ft8
 * No GPU code here: everything goes through the C ABI of include/ft8gpu.h.
 */
#define _GNU_SOURCE
#include "../../include/ft8gpu.h"
#include "../../include/ft8_lib/ft8/decode.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * drop-in symbols
 * ------------------------------------------------------------------------------------------- */
static ft8gpu_ctx *g_ctx = NULL;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
/* state of the ft8_lib-level entries further down (they share the context) */
static struct {
    const uint8_t *mag;                     /* waterfall of the remembered ft8_find_sync() call */
    uint64_t mag_sum;                       /* its content checksum */
    int ncand, cap, min_score;
    int decoded_iters;                      /* max_iterations the remembered statuses were computed with, 0 = none yet */
    ft8gpu_candidate cands[FT8GPU_ABS_MAX_CANDIDATES];
    ft8gpu_decode_status st[FT8GPU_ABS_MAX_CANDIDATES];
} g_l2;

static int global_ctx_init(void) {
    if (g_ctx) return 0;
    int dev = 0;
    const char *e = getenv("FT8GPU_DEVICE");
    if (e) dev = atoi(e);
    if (ft8gpu_create(&g_ctx, dev, 1, NULL) != 0) {
        fprintf(stderr, "ft8gpu: cannot create GPU decoder context: %s\n", ft8gpu_last_error());
        g_ctx = NULL;
        return -1;
    }
    return 0;
}

/* rtlsdr_ft8d.c:314 -- called once from main() (:1026) before option parsing */
void initFFTW(void) {
    pthread_mutex_lock(&g_lock);
    (void)global_ctx_init();
    pthread_mutex_unlock(&g_lock);
}

/* rtlsdr_ft8d.c:338 */
void freeFFTW(void) {
    pthread_mutex_lock(&g_lock);
    if (g_ctx) { ft8gpu_destroy(g_ctx); g_ctx = NULL; }
    pthread_mutex_unlock(&g_lock);
}

/* rtlsdr_ft8d.c:1387.  samples_len is ignored exactly as the reference does (:1393): 48000 samples
 * are read from each array.  The reference has no error channel; on a GPU failure *n_results = 0
 * and the reason is printed to stderr / available from ft8gpu_last_error(). */
void ft8_subsystem(float *iSamples, float *qSamples, uint32_t samples_len,
                   struct decoder_results *decodes, int32_t *n_results) {
    (void)samples_len;
    pthread_mutex_lock(&g_lock);
    if (global_ctx_init() != 0) { *n_results = 0; pthread_mutex_unlock(&g_lock); return; }
    float *iq = (float *)malloc(sizeof(float) * 2 * FT8GPU_NSAMPLES);
    if (!iq) { *n_results = 0; pthread_mutex_unlock(&g_lock); return; }
    memcpy(iq, iSamples, sizeof(float) * FT8GPU_NSAMPLES);
    memcpy(iq + FT8GPU_NSAMPLES, qSamples, sizeof(float) * FT8GPU_NSAMPLES);
    int32_t n = 0;
    const ft8gpu_params ref_params = { FT8GPU_K_MIN_SCORE, FT8GPU_K_MAX_CANDIDATES, FT8GPU_K_LDPC_ITERS };   /* rtlsdr_ft8d.h:43-45 */
    g_l2.mag = NULL;                                       /* (the ft8_lib-level entries below share this context) */
    if (ft8gpu_set_params(g_ctx, &ref_params) != 0 || ft8gpu_decode_batch(g_ctx, iq, 1, decodes, &n, FT8GPU_HOST_PTRS) != 0) {
        fprintf(stderr, "ft8gpu: decode failed: %s\n", ft8gpu_last_error());
        n = 0;
    }
    *n_results = n;
    free(iq);
    pthread_mutex_unlock(&g_lock);
}

/* ---------------------------------------------------------------------------------------------
 * ft8_lib level: ft8_find_sync / ft8_decode as the reference calls them (rtlsdr_ft8d.c:1450, :1476)
 *
 * Both run on the process-global context above.  The reference calls ft8_find_sync() once per frame and
 * then ft8_decode() once per candidate of the returned list (:1465-1485); a GPU launch per candidate would be
 * all latency, so the first ft8_decode() after an ft8_find_sync() on the same waterfall decodes the WHOLE list
 * in one launch and the following calls are answered from that result.  ft8_decode() is a pure function of
 * (waterfall, candidate, max_iterations): a candidate that is not in the remembered list, another waterfall or
 * changed waterfall bytes simply take the one-candidate path.
 *
 * THE ASSUMPTION (also in include/ft8_lib/ft8/decode.h and INTEGRATION.md 1b): ft8_decode() answers from the list the last
 * ft8_find_sync() remembered when it is handed the same `mag` POINTER and the 64-bit hash of the bytes behind it is the one
 * ft8_find_sync() took.  The hash is recomputed on every call; a caller that rewrites the buffer between the two calls is
 * served a fresh decode unless the new bytes collide with the old ones in a 64-bit multiply-mix hash (chance 2^-64 per
 * call; the reference never rewrites: one stack buffer from :1450 to :1476).
 * ------------------------------------------------------------------------------------------- */

/* 64-bit multiply-mix hash of the 94 208 waterfall bytes (about 10 us).  Four chained lanes, each step
 * h = (rotl(h) ^ word) * odd constant: a bijection of the lane state for a given word AND of the word for a given state, so a
 * change confined to one 64-bit word always changes the hash, and the chaining makes it position dependent -- moving or
 * swapping bytes (which an additive sum, or sum of sums, can miss) collides only by chance, 2^-64.  Not cryptographic: it
 * guards against a caller that reuses a buffer, not against an adversary. */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t x) { x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33; return x; }

static uint64_t waterfall_checksum(const uint8_t *mag) {
    _Static_assert(FT8GPU_MAG_ARRAY % 32 == 0, "four 64-bit lanes per step");
    uint64_t h0 = 0x9E3779B97F4A7C15ull, h1 = 0xC2B2AE3D27D4EB4Full, h2 = 0x165667B19E3779F9ull, h3 = 0x27D4EB2F165667C5ull;
    for (int i = 0; i < FT8GPU_MAG_ARRAY; i += 32) {
        uint64_t w[4];
        memcpy(w, mag + i, sizeof w);                       /* any alignment */
        h0 = (rotl64(h0, 29) ^ w[0]) * 0x9FB21C651E98DF25ull;
        h1 = (rotl64(h1, 31) ^ w[1]) * 0xD6E8FEB86659FD93ull;
        h2 = (rotl64(h2, 27) ^ w[2]) * 0xA0761D6478BD642Full;
        h3 = (rotl64(h3, 33) ^ w[3]) * 0xE7037ED1A0B428DBull;
    }
    return fmix64(h0 ^ fmix64(h1 ^ fmix64(h2 ^ fmix64(h3))));
}

static int waterfall_supported(const waterfall_t *wf) {
    return wf && wf->mag && wf->num_blocks == FT8GPU_NUM_BLOCKS && wf->num_bins == FT8GPU_NUM_BIN && wf->time_osr == 2 &&
           wf->freq_osr == 2 && wf->block_stride == 4 * FT8GPU_NUM_BIN && wf->protocol == PROTO_FT8;
}

int ft8_find_sync(const waterfall_t *power, int num_candidates, candidate_t heap[], int min_score) {
    if (!waterfall_supported(power) || !heap || num_candidates < 1 || num_candidates > FT8GPU_ABS_MAX_CANDIDATES) {
        fprintf(stderr, "ft8gpu: ft8_find_sync: unsupported waterfall geometry or candidate count (92 x 2 x 2 x 256, PROTO_FT8, 1..%d)\n",
                FT8GPU_ABS_MAX_CANDIDATES);
        return 0;
    }
    int32_t count = 0;
    pthread_mutex_lock(&g_lock);
    g_l2.mag = NULL;
    if (global_ctx_init() == 0) {
        const ft8gpu_params p = { min_score, num_candidates, FT8GPU_K_LDPC_ITERS };
        if (ft8gpu_set_params(g_ctx, &p) != 0 ||
            ft8gpu_find_sync(g_ctx, power->mag, 1, g_l2.cands, &count, FT8GPU_HOST_PTRS) != 0) {
            fprintf(stderr, "ft8gpu: ft8_find_sync failed: %s\n", ft8gpu_last_error());
            count = 0;
        } else {
            memcpy(heap, g_l2.cands, sizeof(candidate_t) * (size_t)count);      /* same 8-byte layout */
            g_l2.mag = power->mag;
            g_l2.mag_sum = waterfall_checksum(power->mag);
            g_l2.ncand = count;
            g_l2.cap = num_candidates;
            g_l2.min_score = min_score;
            g_l2.decoded_iters = 0;
        }
    }
    pthread_mutex_unlock(&g_lock);
    return count;
}

bool ft8_decode(const waterfall_t *power, const candidate_t *cand, message_t *message, int max_iterations,
                decode_status_t *status) {
    decode_status_t local;
    if (!status) status = &local;
    memset(status, 0, sizeof *status);
    status->ldpc_errors = FTX_LDPC_M;                      /* "nothing satisfied": the value bp_decode starts from */
    if (!waterfall_supported(power) || !cand || !message || max_iterations < 1) {
        fprintf(stderr, "ft8gpu: ft8_decode: unsupported waterfall geometry or arguments\n");
        return false;
    }
    _Static_assert(sizeof(candidate_t) == sizeof(ft8gpu_candidate), "candidate_t layout");
    const ft8gpu_decode_status *res = NULL;
    ft8gpu_decode_status one;
    pthread_mutex_lock(&g_lock);
    if (global_ctx_init() == 0) {
        int idx = -1;
        /* The remembered list answers only for the very bytes ft8_find_sync() saw: the checksum (94 KB, about 10 us) is
         * taken on EVERY call, so a caller that rewrites the waterfall in place, or reuses the address for another one
         * without a new ft8_find_sync(), gets a fresh single-candidate decode instead of stale statuses. */
        if (g_l2.mag == power->mag && waterfall_checksum(power->mag) != g_l2.mag_sum) g_l2.mag = NULL;
        if (g_l2.mag == power->mag)
            for (int i = 0; i < g_l2.ncand; i++)
                if (!memcmp(&g_l2.cands[i], cand, sizeof *cand)) { idx = i; break; }
        if (idx >= 0 && g_l2.decoded_iters != max_iterations) {
            /* first ft8_decode() of this list (or another iteration count): one launch for all candidates */
            const ft8gpu_params p = { g_l2.min_score, g_l2.cap, max_iterations };
            const int32_t n = g_l2.ncand;
            if (ft8gpu_set_params(g_ctx, &p) != 0 ||
                     ft8gpu_decode_candidates(g_ctx, power->mag, g_l2.cands, &n, 1, g_l2.st, FT8GPU_HOST_PTRS) != 0) {
                fprintf(stderr, "ft8gpu: ft8_decode failed: %s\n", ft8gpu_last_error());
                idx = -2;
            } else g_l2.decoded_iters = max_iterations;
        }
        if (idx >= 0) res = &g_l2.st[idx];
        else if (idx == -1) {                              /* not from the remembered list: decode this one candidate */
            const ft8gpu_params p = { -32768, 1, max_iterations };
            const int32_t n = 1;
            ft8gpu_candidate c1;
            memcpy(&c1, cand, sizeof c1);
            g_l2.decoded_iters = 0;                        /* the context's parameters change: the list is re-decoded if asked again */
            if (ft8gpu_set_params(g_ctx, &p) != 0 ||
                ft8gpu_decode_candidates(g_ctx, power->mag, &c1, &n, 1, &one, FT8GPU_HOST_PTRS) != 0)
                fprintf(stderr, "ft8gpu: ft8_decode failed: %s\n", ft8gpu_last_error());
            else res = &one;
        }
    }
    bool ok = false;
    if (res) {
        status->ldpc_errors = res->ldpc_errors;
        status->crc_extracted = res->crc_extracted;
        status->crc_calculated = res->crc_calculated;
        status->unpack_status = res->unpack_status;
        if (res->ok) {
            memcpy(message->text, res->text, sizeof message->text);
            message->hash = res->crc_extracted;
            ok = true;
        }
    }
    pthread_mutex_unlock(&g_lock);
    return ok;
}

/* ---------------------------------------------------------------------------------------------
 * replay file formats (what readRawIQfile / readC2file / writeRawIQfile of rtlsdr_ft8d.c:744-856 read and write)
 *
 *   .iq  up to 48000 records of two little-endian float32: (I, -Q)   -- Q is stored negated ("convention used by
 *        wsprsim", :760); a short file gives a short frame, an odd trailing float is dropped (nread / 2, :756)
 *   .c2  the same records behind a 26-byte header: 14 B name, int32 type, float64 dial frequency (:823-825)
 * After loading, the frame is scaled so that its largest |I| or |Q| becomes 0.5 (:763-778: the peak starts at 1e-24f,
 * the factor is the DOUBLE quotient 0.5 / peak rounded to float, the products are float).
 * Streamed through a small block buffer (the reference puts 384 KB on the stack); the peak is taken while the records
 * are split, so the file is walked once and the frame twice.
 * ------------------------------------------------------------------------------------------- */
enum { kRecordsPerBlock = 2048 };

static FILE *open_or_complain(const char *path, const char *mode) {
    FILE *f = fopen(path, mode);
    if (!f) fprintf(stderr, "ft8gpu: cannot open %s (%s)\n", path, strerror(errno));
    return f;
}

/* splits (I, -Q) records from `f` into the planar frame; returns the number of whole records */
static int32_t split_records(FILE *f, float *plane_i, float *plane_q) {
    float block[2 * kRecordsPerBlock];
    int32_t have = 0;                       /* records stored so far */
    float peak = 1e-24f;
    while (have < FT8GPU_NSAMPLES) {
        size_t want = 2 * (size_t)(FT8GPU_NSAMPLES - have);            /* always an even number of floats */
        if (want > 2 * kRecordsPerBlock) want = 2 * kRecordsPerBlock;
        const size_t got = fread(block, sizeof(float), want, f);
        const int32_t whole = (int32_t)(got / 2);                       /* a dangling half record is dropped, as nread / 2 does */
        for (int32_t r = 0; r < whole; r++) {
            const float vi = block[2 * r], vq = -block[2 * r + 1];
            plane_i[have + r] = vi;
            plane_q[have + r] = vq;
            const float ai = fabsf(vi), aq = fabsf(vq);
            if (ai > peak) peak = ai;
            if (aq > peak) peak = aq;
        }
        have += whole;
        if (got < want) break;                                          /* end of file */
    }
    const float gain = (float)(0.5 / (double)peak);
    for (int32_t k = 0; k < have; k++) {
        plane_i[k] *= gain;
        plane_q[k] *= gain;
    }
    return have;
}

int32_t ft8gpu_read_raw_iq(float *iSamples, float *qSamples, const char *filename) {
    FILE *f = open_or_complain(filename, "rb");
    if (!f) return 0;
    const int32_t records = split_records(f, iSamples, qSamples);
    fclose(f);
    return records;
}

int32_t ft8gpu_read_c2(float *iSamples, float *qSamples, const char *filename, double *dialfreq) {
    FILE *f = open_or_complain(filename, "rb");
    if (!f) return 0;
    struct { char name[14]; int32_t type; double dial_hz; } head;
    memset(&head, 0, sizeof head);
    /* three reads, as the file has no padding between the fields; a truncated header leaves zeros */
    if (fread(head.name, 1, sizeof head.name, f) == sizeof head.name && fread(&head.type, sizeof head.type, 1, f) == 1)
        (void)!fread(&head.dial_hz, sizeof head.dial_hz, 1, f);
    if (dialfreq) *dialfreq = head.dial_hz;
    const int32_t records = split_records(f, iSamples, qSamples);
    fclose(f);
    return records;
}

int32_t ft8gpu_write_raw_iq(const float *iSamples, const float *qSamples, const char *filename) {
    FILE *f = open_or_complain(filename, "wb");
    if (!f) return 0;
    float block[2 * kRecordsPerBlock];
    int32_t done = 0;
    while (done < FT8GPU_NSAMPLES) {
        int32_t n = FT8GPU_NSAMPLES - done;
        if (n > kRecordsPerBlock) n = kRecordsPerBlock;
        for (int32_t r = 0; r < n; r++) {
            block[2 * r] = iSamples[done + r];
            block[2 * r + 1] = -qSamples[done + r];
        }
        if (fwrite(block, 2 * sizeof(float), (size_t)n, f) != (size_t)n) break;
        done += n;
    }
    const int closed = fclose(f);
    if (done != FT8GPU_NSAMPLES || closed != 0) {
        fprintf(stderr, "ft8gpu: short write to %s (%d of %d records)\n", filename, (int)done, FT8GPU_NSAMPLES);
        return 0;
    }
    return FT8GPU_NSAMPLES;
}

/* ---- the stdout table of printSpots(), rtlsdr_ft8d.c:643-663 --------------------------------- */
static void emit(char *out, size_t cap, size_t *at, const char *line, int len) {
    if (len < 0) return;
    if (out && *at < cap) {
        size_t room = cap - *at - 1, n = (size_t)len < room ? (size_t)len : room;
        memcpy(out + *at, line, n);
        out[*at + n] = 0;
    }
    *at += (size_t)len;
}

int ft8gpu_format_spots(const struct decoder_results *decodes, int32_t n_results, uint32_t dial_freq,
                        int year, int month, int mday, int hour, int minute, char *out, size_t cap) {
    char line[96];
    size_t at = 0;
    if (out && cap) out[0] = 0;
    if (n_results <= 0) {                                                    /* :644-653 */
        emit(out, cap, &at, line, snprintf(line, sizeof line, "No spot %04d-%02d-%02d %02d:%02dz\n", year, month, mday, hour, minute));
        return (int)at;
    }
    if (!decodes) return -1;
    emit(out, cap, &at, line, snprintf(line, sizeof line, "  Score     Freq       Call    Loc\n"));   /* :655 */
    for (int32_t i = 0; i < n_results && i < FT8GPU_K_MAX_MESSAGES; i++)     /* :656-662 */
        emit(out, cap, &at, line, snprintf(line, sizeof line, "     %2d %8d %10.12s %6.6s\n", decodes[i].snr,
                                          (int)((uint32_t)decodes[i].freq + dial_freq), decodes[i].call, decodes[i].loc));
    return (int)at;
}
