/*
 * ft8_lib_level.c -- a C caller written against the ft8_lib-level interface the reference's own ft8_subsystem()
 * uses (rtlsdr_ft8d.c:1438-1523): the seven "./ft8_lib/ft8/..." headers it includes (:38-44), waterfall_t filled
 * with designated initialisers, ft8_find_sync(), one ft8_decode() per candidate, the caller's own dedup table
 * and CQ filter.  With libft8gpu.so behind those symbols this is what the UNMODIFIED rtlsdr_ft8d.c does when
 * it is compiled with -I<repo>/include and linked against libft8gpu.so instead of the ft8_lib objects.
 *
 *   ft8_lib_level           self-test frame ("CQ K1JT FN20QI", as decoderSelfTest() :913-972) plus thirteen more signals
 *                           of every message shape (reports, RR73 twice, bare calls, type 4, hashed call, free text,
 *                           CQ with a modifier / without a grid); result compared with the library's own
 *                           ft8_subsystem() on the same frame
 *
 * The waterfall comes from the library's stage entry (the reference computes it with fftw3f, :1395-1435).
 * Build:  gcc -O2 -std=gnu17 -Iinclude examples/ft8_lib_level.c -Lrtlsdr_ft8d_amd -lft8gpu \
 *             -Wl,-rpath,$PWD/rtlsdr_ft8d_amd -lm -o examples/ft8_lib_level
 */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ft8gpu.h"

#include "./ft8_lib/ft8/constants.h"
#include "./ft8_lib/ft8/pack.h"
#include "./ft8_lib/ft8/unpack.h"
#include "./ft8_lib/ft8/ldpc.h"
#include "./ft8_lib/ft8/crc.h"
#include "./ft8_lib/ft8/decode.h"
#include "./ft8_lib/ft8/encode.h"

#define K_MIN_SCORE FT8GPU_K_MIN_SCORE
#define K_MAX_CANDIDATES FT8GPU_K_MAX_CANDIDATES
#define K_LDPC_ITERS FT8GPU_K_LDPC_ITERS
#define K_MAX_MESSAGES FT8GPU_K_MAX_MESSAGES
#define K_FREQ_OSR 2
#define K_TIME_OSR 2
#define K_FSK_DEV 6.25f
#define NUM_BIN FT8GPU_NUM_BIN
#define NUM_BLOCKS FT8GPU_NUM_BLOCKS

static float iSamples[FT8GPU_NSAMPLES], qSamples[FT8GPU_NSAMPLES];
static uint8_t mag_power[FT8GPU_MAG_ARRAY];

static void add_signal(const char *text, float f0, float amp, int start) {
    uint8_t packed[FTX_LDPC_K_BYTES], tones[FT8_NN];
    if (pack77(text, packed) < 0) { fprintf(stderr, "Cannot parse message %s\n", text); exit(2); }
    ft8_encode(packed, tones);
    double phi = 0.0;
    for (int i = 0; i < FT8_NN; i++) {
        const double dphi = 2.0 * M_PI * (f0 + ((double)tones[i] - 3.5) * 6.25) / 3200.0;
        for (int j = 0; j < 512; j++) {
            const int index = start + 512 * i + j;
            if (index < FT8GPU_NSAMPLES) { iSamples[index] += amp * cos(phi); qSamples[index] += amp * sin(phi); }
            phi += dphi;
        }
    }
}

/* the candidate loop of rtlsdr_ft8d.c:1438-1523 over the ft8_lib-level interface */
static int32_t decode_like_the_reference(struct decoder_results *decodes) {
    candidate_t candidate_list[K_MAX_CANDIDATES];
    waterfall_t power = {
        .num_blocks = NUM_BLOCKS, .num_bins = NUM_BIN, .time_osr = K_TIME_OSR, .freq_osr = K_FREQ_OSR,
        .mag = mag_power, .block_stride = (K_TIME_OSR * K_FREQ_OSR * NUM_BIN), .protocol = PROTO_FT8
    };
    const int num_candidates = ft8_find_sync(&power, K_MAX_CANDIDATES, candidate_list, K_MIN_SCORE);
    int num_decoded = 0;
    static message_t decoded[K_MAX_MESSAGES];
    message_t *table[K_MAX_MESSAGES] = { 0 };
    for (int idx = 0; idx < num_candidates; ++idx) {
        const candidate_t *cand = &candidate_list[idx];
        if (cand->score < K_MIN_SCORE) continue;
        const float freq_hz = (cand->freq_offset + (float)cand->freq_sub / K_FREQ_OSR) * K_FSK_DEV;
        message_t message;
        decode_status_t status;
        if (!ft8_decode(&power, cand, &message, K_LDPC_ITERS, &status)) continue;
        int slot = message.hash % K_MAX_MESSAGES, probes = 0;
        bool is_new = false;
        for (;; slot = (slot + 1) % K_MAX_MESSAGES) {
            if (!table[slot]) { is_new = true; break; }
            if (table[slot]->hash == message.hash && !strcmp(table[slot]->text, message.text)) break;
            if (++probes >= K_MAX_MESSAGES) break;
        }
        if (!is_new) continue;
        decoded[slot] = message;
        table[slot] = &decoded[slot];
        char *save = NULL, *tok = strtok_r(message.text, " ", &save);
        if (tok && !strncmp(tok, "CQ", 2)) {
            tok = strtok_r(NULL, " ", &save);
            snprintf(decodes[num_decoded].call, sizeof decodes[num_decoded].call, "%.12s", tok ? tok : "(null)");
            tok = strtok_r(NULL, " ", &save);
            snprintf(decodes[num_decoded].loc, sizeof decodes[num_decoded].loc, "%.6s", tok ? tok : "(null)");
            decodes[num_decoded].freq = (int32_t)freq_hz;
            decodes[num_decoded].snr = (int32_t)cand->score;
        }
        num_decoded++;
    }
    return num_decoded;
}

int main(void) {
    srand(7);
    for (int i = 0; i < FT8GPU_NSAMPLES; i++) {
        iSamples[i] = 0.02f * (float)(rand() / (double)RAND_MAX - 0.5);
        qSamples[i] = 0.02f * (float)(rand() / (double)RAND_MAX - 0.5);
    }
    add_signal("CQ K1JT FN20QI", 50.0f, 0.5f, 0);
    add_signal("CQ DL1ABC JO62", 700.0f, 0.2f, 1600);
    add_signal("W9XYZ K1ABC EN37", 1211.0f, 0.3f, 800);     /* not a CQ: counted, its slot stays untouched */
    /* what most of a band looks like: QSO traffic of every shape, one message on two frequencies, CQ calls without a grid */
    add_signal("K1ABC W9XYZ -11", 150.0f, 0.25f, 400);
    add_signal("W9XYZ K1ABC R-09", 250.0f, 0.22f, 2000);
    add_signal("K1ABC W9XYZ RR73", 350.0f, 0.3f, 1200);
    add_signal("K1ABC W9XYZ RR73", 1450.0f, 0.12f, 2400);    /* the same message again: one entry (rtlsdr_ft8d.c:1487-1507) */
    add_signal("W9XYZ K1ABC 73", 450.0f, 0.2f, 0);
    add_signal("G4ABC PA9XYZ", 550.0f, 0.25f, 3000);         /* two bare calls: text with a trailing blank */
    add_signal("CQ PJ4/K1ABC", 850.0f, 0.3f, 1000);          /* type 4 CQ: no locator token -> "(null)" */
    add_signal("CQ73 GL", 950.0f, 0.2f, 500);                /* free text whose first token starts with "CQ" */
    add_signal("<PJ4/K1ABC> W9XYZ RRR", 1050.0f, 0.25f, 1500);
    add_signal("TNX BOB 73 GL", 1130.0f, 0.2f, 200);
    add_signal("CQ DX VK3ABC QF22", 1310.0f, 0.3f, 2800);    /* second token is the modifier: call = "DX" */

    ft8gpu_ctx *ctx = NULL;
    if (ft8gpu_create(&ctx, 0, 1, NULL) != 0) { fprintf(stderr, "%s\n", ft8gpu_last_error()); return 2; }
    float *iq = malloc(sizeof(float) * 2 * FT8GPU_NSAMPLES);
    memcpy(iq, iSamples, sizeof iSamples);
    memcpy(iq + FT8GPU_NSAMPLES, qSamples, sizeof qSamples);
    if (ft8gpu_waterfall(ctx, iq, 1, mag_power, FT8GPU_HOST_PTRS) != 0) { fprintf(stderr, "%s\n", ft8gpu_last_error()); return 2; }
    ft8gpu_destroy(ctx);
    free(iq);

    static struct decoder_results via_ft8_lib[K_MAX_MESSAGES], via_subsystem[K_MAX_MESSAGES];
    memset(via_ft8_lib, 0x5a, sizeof via_ft8_lib);
    memset(via_subsystem, 0x5a, sizeof via_subsystem);
    const int32_t n1 = decode_like_the_reference(via_ft8_lib);
    int32_t n2 = 0;
    ft8_subsystem(iSamples, qSamples, FT8GPU_NSAMPLES, via_subsystem, &n2);
    freeFFTW();
    printf("ft8_lib level: %d messages, ft8_subsystem: %d messages\n", n1, n2);
    for (int k = 0; k < n1; k++)
        if (via_ft8_lib[k].call[0] != 0x5a) printf("  %2d %8d %10.12s %6.6s\n", via_ft8_lib[k].snr, via_ft8_lib[k].freq, via_ft8_lib[k].call, via_ft8_lib[k].loc);
    /* snprintf leaves the bytes behind the terminator alone in both paths, so whole records compare */
    if (n1 != n2 || n1 < 13 || memcmp(via_ft8_lib, via_subsystem, sizeof via_ft8_lib) != 0) { fprintf(stderr, "MISMATCH\n"); return 1; }
    int written = 0, null_loc = 0;
    for (int k = 0; k < n1; k++) {
        if ((unsigned char)via_ft8_lib[k].call[0] == 0x5a && (unsigned char)via_ft8_lib[k].call[1] == 0x5a) continue;   /* slot left untouched */
        written++;
        null_loc += !strcmp(via_ft8_lib[k].loc, "(null)");
    }
    if (written < 5 || written > n1 - 6 || null_loc < 2) { fprintf(stderr, "unexpected slot census: %d written of %d, %d without locator\n", written, n1, null_loc); return 1; }
    int k1jt = 0;                                           /* slots follow the candidates' score order: not necessarily slot 0 here */
    for (int k = 0; k < n1; k++) k1jt |= !strcmp(via_ft8_lib[k].call, "K1JT") && !strcmp(via_ft8_lib[k].loc, "FN20");
    if (!k1jt) { fprintf(stderr, "K1JT FN20 not among the spots\n"); return 1; }
    puts("ft8_lib-level path == ft8_subsystem");
    return 0;
}
