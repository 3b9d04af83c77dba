/*
 * ThreadSanitizer harness for the host logic of csrc/ft8_compat.c: the process-global context behind the drop-in
 * symbols (initFFTW / freeFFTW / ft8_subsystem, rtlsdr_ft8d.h:155-156, :164) and the remembered candidate list of the
 * ft8_lib-level entries (ft8_find_sync / ft8_decode, rtlsdr_ft8d.c:1450, :1476).  The GPU half of the library is
 * replaced by a deterministic CPU stand-in of OUR OWN C ABI (include/ft8gpu.h) whose context is deliberately not
 * thread-safe: every field access is unsynchronised, so a call that reaches it outside ft8_compat.c's lock is a data
 * race TSan reports, and every answer is a pure function of (waterfall bytes, candidate, iterations), so an answer
 * served from a stale remembered list is a wrong value the harness reports.
 * Built and run by tests/test_sanitizers.py (gcc -fsanitize=thread), CPU only.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ft8gpu.h"
#include "ft8_lib/ft8/decode.h"

/* ---- stand-in for the GPU half ------------------------------------------------------------------ */
struct ft8gpu_ctx { ft8gpu_params p; long calls; int alive; };
static int g_created, g_destroyed;                       /* touched under ft8_compat.c's lock only (if it holds) */
static int g_last_decode_n = -1, g_decode_calls;         /* `cache` mode (single-threaded): how many candidates the last launch decoded */

int ft8gpu_create(ft8gpu_ctx **out, int device, int max_frames, const ft8gpu_params *params) {
    (void)device; (void)max_frames;
    ft8gpu_ctx *c = calloc(1, sizeof *c);
    if (!c) return -1;
    c->p = params ? *params : (ft8gpu_params){ 10, 120, 20 };
    c->alive = 1;
    g_created++;
    *out = c;
    return 0;
}
void ft8gpu_destroy(ft8gpu_ctx *c) { if (c) { c->alive = 0; g_destroyed++; free(c); } }
const char *ft8gpu_last_error(void) { return "stand-in"; }
int ft8gpu_set_params(ft8gpu_ctx *c, const ft8gpu_params *p) { c->p = *p; c->calls++; return 0; }

static uint32_t mix(uint32_t h, uint32_t v) { h ^= v + 0x9E3779B9u + (h << 6) + (h >> 2); return h; }
static uint32_t mag_digest(const uint8_t *mag) {
    uint32_t h = 17;
    for (int i = 0; i < FT8GPU_MAG_ARRAY; i += 997) h = mix(h, mag[i]);
    return h;
}
static void expected_candidate(const uint8_t *mag, int k, ft8gpu_candidate *c) {
    const uint32_t h = mix(mag_digest(mag), (uint32_t)k);
    c->score = (int16_t)(10 + h % 40); c->time_offset = (int16_t)((int)(h >> 8) % 36 - 12); c->freq_offset = (int16_t)((h >> 16) % 249);
    c->time_sub = (uint8_t)((h >> 3) & 1); c->freq_sub = (uint8_t)((h >> 5) & 1);
}
static int expected_count(const uint8_t *mag, int cap) { const int n = 3 + (int)(mag_digest(mag) % 9); return n < cap ? n : cap; }
static void expected_status(const uint8_t *mag, const ft8gpu_candidate *cand, int iters, ft8gpu_decode_status *st) {
    uint32_t h = mix(mag_digest(mag), (uint32_t)iters);
    h = mix(h, (uint32_t)(uint16_t)cand->score | ((uint32_t)(uint16_t)cand->freq_offset << 16));
    h = mix(h, (uint32_t)(uint16_t)cand->time_offset | ((uint32_t)cand->time_sub << 16) | ((uint32_t)cand->freq_sub << 24));
    memset(st, 0, sizeof *st);
    st->ok = (uint8_t)(h & 1);
    st->ldpc_errors = st->ok ? 0 : (int16_t)(1 + h % 40);
    st->crc_extracted = (uint16_t)(h >> 7); st->crc_calculated = st->ok ? st->crc_extracted : (uint16_t)(h >> 9);
    st->unpack_status = 0;
    snprintf(st->text, sizeof st->text, "MSG %08X", (unsigned)h);
}

int ft8gpu_find_sync(ft8gpu_ctx *c, const uint8_t *mag, int nframes, ft8gpu_candidate *cands, int32_t *counts, int flags) {
    (void)flags;
    if (!c->alive || nframes != 1) return -1;
    c->calls++;
    counts[0] = expected_count(mag, c->p.max_candidates);
    for (int k = 0; k < counts[0]; k++) expected_candidate(mag, k, &cands[k]);
    return 0;
}
int ft8gpu_decode_candidates(ft8gpu_ctx *c, const uint8_t *mag, const ft8gpu_candidate *cands, const int32_t *counts,
                             int nframes, ft8gpu_decode_status *status, int flags) {
    (void)flags;
    if (!c->alive || nframes != 1) return -1;
    c->calls++;
    g_last_decode_n = counts[0]; g_decode_calls++;
    for (int k = 0; k < counts[0]; k++) expected_status(mag, &cands[k], c->p.ldpc_iters, &status[k]);
    return 0;
}
int ft8gpu_decode_batch(ft8gpu_ctx *c, const float *iq, int nframes, struct decoder_results *decodes, int32_t *n_results, int flags) {
    (void)flags;
    if (!c->alive || nframes != 1) return -1;
    c->calls++;
    n_results[0] = 1 + ((int)(iq[0] * 1000.0f) & 3);
    snprintf(decodes[0].call, sizeof decodes[0].call, "T%d", (int)(iq[1] * 1000.0f));
    decodes[0].snr = (int32_t)(iq[2] * 1000.0f);
    return 0;
}

/* ---- the threads --------------------------------------------------------------------------------- */
enum { kThreads = 6, kRounds = 300 };
static int g_bad[kThreads];

static void fill_mag(uint8_t *mag, unsigned seed) {
    for (int i = 0; i < FT8GPU_MAG_ARRAY; i++) { seed = seed * 1664525u + 1013904223u; mag[i] = (uint8_t)(seed >> 24); }
}

static void *worker(void *arg) {
    const int id = (int)(intptr_t)arg;
    uint8_t *mag = malloc(FT8GPU_MAG_ARRAY);
    float *iq = malloc(sizeof(float) * 2 * FT8GPU_NSAMPLES);
    if (!mag || !iq) { g_bad[id] = 1000; return NULL; }
    memset(iq, 0, sizeof(float) * 2 * FT8GPU_NSAMPLES);
    waterfall_t wf = { .num_blocks = 92, .num_bins = 256, .time_osr = 2, .freq_osr = 2, .mag = mag, .block_stride = 1024, .protocol = PROTO_FT8 };
    unsigned rs = 12345u * (unsigned)(id + 1);
    for (int r = 0; r < kRounds; r++) {
        rs = rs * 1103515245u + 12345u;
        const unsigned pick = (rs >> 16) % 10;
        if (pick == 0) { initFFTW(); continue; }
        if (pick == 1) { freeFFTW(); continue; }
        if (pick <= 3) {                                   /* the drop-in frame decode */
            struct decoder_results d[FT8GPU_K_MAX_MESSAGES];
            memset(d, 0, sizeof d);
            int32_t n = -1;
            iq[0] = 0.001f * (float)(r & 3); iq[1] = 0.001f * (float)id; iq[2] = 0.001f * (float)r;
            ft8_subsystem(iq, iq + FT8GPU_NSAMPLES, FT8GPU_NSAMPLES, d, &n);
            char want[13];
            snprintf(want, sizeof want, "T%d", (int)(iq[1] * 1000.0f));
            if (n != 1 + ((int)(iq[0] * 1000.0f) & 3) || strcmp(d[0].call, want) || d[0].snr != (int32_t)(iq[2] * 1000.0f)) g_bad[id]++;
            continue;
        }
        /* the reference's own call pattern (rtlsdr_ft8d.c:1450, :1465-1485): one ft8_find_sync, then ft8_decode per candidate --
         * while the other threads move the shared context's parameters and remembered list under this one's feet */
        fill_mag(mag, rs);
        candidate_t heap[32];
        const int cap = 4 + (int)((rs >> 8) % 20);
        const int n = ft8_find_sync(&wf, cap, heap, 10);
        if (n != expected_count(mag, cap)) { g_bad[id]++; continue; }
        const int iters = 5 + (int)((rs >> 4) % 3) * 10;
        int list_ok = 1;
        for (int k = 0; k < n; k++) {
            ft8gpu_candidate want_c;
            expected_candidate(mag, k, &want_c);
            if (memcmp(&want_c, &heap[k], sizeof want_c)) list_ok = 0;
        }
        if (!list_ok) { g_bad[id]++; continue; }
        for (int k = 0; k < n; k++) {
            if (pick == 9 && k == n / 2) mag[0] ^= 0x5A;   /* the caller rewrites the waterfall in place: the remembered list must not answer */
            message_t msg;
            decode_status_t st;
            memset(&msg, 0, sizeof msg);
            const bool ok = ft8_decode(&wf, &heap[k], &msg, iters, &st);
            ft8gpu_decode_status want;
            ft8gpu_candidate c1;
            memcpy(&c1, &heap[k], sizeof c1);
            expected_status(mag, &c1, iters, &want);
            if (ok != (want.ok != 0) || st.ldpc_errors != want.ldpc_errors || st.crc_extracted != want.crc_extracted ||
                st.crc_calculated != want.crc_calculated || (ok && (strcmp(msg.text, want.text) || msg.hash != want.crc_extracted))) g_bad[id]++;
        }
    }
    free(mag); free(iq);
    return NULL;
}

/* ---- `cache` mode: rewrites of the waterfall between ft8_find_sync and ft8_decode that an additive checksum misses ------ */
/* the checksum ft8_compat.c used through round 5 (sum and sum of sums of the 64-bit words), kept here to show that the
 * rewrites below are ones it could NOT see: the test has teeth only if they collide under it */
static uint64_t additive_checksum_r05(const uint8_t *mag) {
    uint64_t a = 0x9E3779B97F4A7C15ull, b = 0;
    for (int i = 0; i < FT8GPU_MAG_ARRAY / 8; i++) { uint64_t w; memcpy(&w, mag + 8 * i, 8); a += w; b += a; }
    return a ^ (b << 1);
}
static uint64_t plain_sum(const uint8_t *mag) {
    uint64_t a = 0;
    for (int i = 0; i < FT8GPU_MAG_ARRAY / 8; i++) { uint64_t w; memcpy(&w, mag + 8 * i, 8); a += w; }
    return a;
}

static int check_decode(const waterfall_t *wf, const candidate_t *cand, int iters, const char *what) {
    message_t msg;
    decode_status_t st;
    ft8gpu_decode_status want;
    ft8gpu_candidate c1;
    memset(&msg, 0, sizeof msg);
    memcpy(&c1, cand, sizeof c1);
    const bool ok = ft8_decode(wf, cand, &msg, iters, &st);
    expected_status(wf->mag, &c1, iters, &want);
    if (ok != (want.ok != 0) || st.ldpc_errors != want.ldpc_errors || st.crc_extracted != want.crc_extracted ||
        st.crc_calculated != want.crc_calculated || (ok && strcmp(msg.text, want.text))) {
        printf("cache: %s: ft8_decode answered for other bytes than the ones it was handed\n", what);
        return 1;
    }
    return 0;
}

static int cache_mode(void) {
    uint8_t *mag = malloc(FT8GPU_MAG_ARRAY + 8);
    if (!mag) return 1;
    int bad = 0;
    for (int trial = 0; trial < 64; trial++) {
        uint8_t *m = mag + (trial & 1 ? 3 : 0);            /* aligned and unaligned buffers */
        waterfall_t wf = { .num_blocks = 92, .num_bins = 256, .time_osr = 2, .freq_osr = 2, .mag = m, .block_stride = 1024, .protocol = PROTO_FT8 };
        fill_mag(m, 777u + (unsigned)trial);
        if (m[0] == m[7976]) m[7976] ^= 0x5A;              /* the two bytes that will change places differ */
        candidate_t heap[32];
        const int n = ft8_find_sync(&wf, 24, heap, 10);
        if (n < 3) { printf("cache: list too short\n"); bad++; continue; }
        bad += check_decode(&wf, &heap[0], 20, "first call");
        if (g_last_decode_n != n) { printf("cache: the first ft8_decode did not decode the whole list (%d of %d)\n", g_last_decode_n, n); bad++; }
        int calls = g_decode_calls;
        bad += check_decode(&wf, &heap[1], 20, "lookup");
        if (g_decode_calls != calls) { printf("cache: an unchanged waterfall was decoded again\n"); bad++; }
        const uint32_t before = mag_digest(m);
        const uint64_t add_before = additive_checksum_r05(m), sum_before = plain_sum(m);
        if (trial % 2 == 0) {
            /* two bytes of the same lane of two 64-bit words change places (offsets 0 and 997*8, both seen by the stand-in's digest):
             * the sum of the words, hence every byte-lane sum, is unchanged */
            const uint64_t s0 = plain_sum(m);
            const uint8_t t = m[0]; m[0] = m[7976]; m[7976] = t;
            if (plain_sum(m) != s0) { printf("cache: harness error: the swap changed the word sum\n"); bad++; }
        } else {
            /* w[0] += d, w[1] -= 2d, w[2] += d: keeps the sum AND the sum of sums -- a collision of the round-5 checksum by construction */
            uint64_t w[3];
            memcpy(w, m, sizeof w);
            const uint64_t d = 1 + (uint64_t)trial;
            w[0] += d; w[1] -= 2 * d; w[2] += d;
            memcpy(m, w, sizeof w);
            if (additive_checksum_r05(m) != add_before || plain_sum(m) != sum_before) { printf("cache: harness error: the pattern does not collide\n"); bad++; }
        }
        if (mag_digest(m) == before) { printf("cache: harness error: rewrite invisible to the stand-in\n"); bad++; continue; }
        calls = g_decode_calls;
        bad += check_decode(&wf, &heap[2], 20, "after the rewrite");
        if (g_decode_calls != calls + 1 || g_last_decode_n != 1) {
            printf("cache: trial %d: a rewritten waterfall was answered from the remembered list (launches %d, candidates %d)\n", trial,
                   g_decode_calls - calls, g_last_decode_n);
            bad++;
        }
    }
    freeFFTW();
    free(mag);
    printf("compat_cache %s: %d wrong\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "cache")) return cache_mode();
    pthread_t th[kThreads];
    for (int i = 0; i < kThreads; i++) pthread_create(&th[i], NULL, worker, (void *)(intptr_t)i);
    int bad = 0;
    for (int i = 0; i < kThreads; i++) { pthread_join(th[i], NULL); bad += g_bad[i]; }
    freeFFTW();
    printf("compat_tsan %s: %d threads x %d rounds, %d wrong answers, contexts created %d destroyed %d\n", bad ? "FAILED" : "ok",
           kThreads, kRounds, bad, g_created, g_destroyed);
    return (bad == 0 && g_created == g_destroyed && g_created > 0) ? 0 : 1;
}
