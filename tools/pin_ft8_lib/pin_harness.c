/*
 * pin_harness.c -- dumps what an implementation of the ft8_lib interface of the reference's era computes, as text that
 * can be diffed between implementations.  Written against the interface rtlsdr_ft8d.c itself uses and nothing else:
 *   waterfall_t by designated initialisers (rtlsdr_ft8d.c:1440-1448), ft8_find_sync (:1450), ft8_decode (:1476),
 *   candidate_t / message_t / decode_status_t fields (:1466-1494), pack77 (:927), ft8_encode (:934).
 * The same file is linked against
 *   (a) kgoba/ft8_lib's own sources      tools/pin_ft8_lib.sh <checkout>   -- the pin the oracle is missing today
 *   (b) the CPU oracle                   oracle_as_ft8_lib.c               -- the expected dump
 *   (c) libft8gpu.so's ft8_lib-level symbols (tests/test_pin_harness.py, -m gpu) -- proves the harness and the product
 * With -DPIN_INTERNALS the backend also provides three hooks into the stages behind the public calls (every sync
 * score, the normalised LLRs, belief propagation by itself); see pin_hooks.h.
 *
 * input : a file of N waterfalls, 94208 bytes each (uint8 mag[92][2][2][256], the layout of rtlsdr_ft8d.c:1420-1433);
 *         the reference computes the waterfall itself (:1395-1435), ft8_lib only reads it
 * output: stdout, one record per line; floats are printed as their bit patterns
 * usage : pin_harness <waterfalls.bin> [messages.txt]
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ft8/constants.h"
#include "ft8/pack.h"
#include "ft8/encode.h"
#include "ft8/decode.h"
#ifdef PIN_INTERNALS
#include "pin_hooks.h"
#endif

#ifndef FTX_LDPC_K_BYTES
#define FTX_LDPC_K_BYTES 12
#endif
#ifndef FT8_NN
#define FT8_NN 79
#endif

enum { kMag = 94208, kNumBlocks = 92, kNumBin = 256, kCapMax = 480 };

static uint64_t fnv64(const void *data, size_t n, uint64_t h) {
    const unsigned char *p = data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}

static void print_text(const char *t) {
    putchar('"');
    for (; *t; t++) {
        if (*t == '"' || *t == '\\' || (unsigned char)*t < 32 || (unsigned char)*t > 126) printf("\\x%02x", (unsigned char)*t);
        else putchar(*t);
    }
    putchar('"');
}

static void dump_frame(int k, uint8_t *mag) {
    waterfall_t power = {                                          /* rtlsdr_ft8d.c:1440-1448 */
        .num_blocks = kNumBlocks, .num_bins = kNumBin, .time_osr = 2, .freq_osr = 2,
        .mag = mag, .block_stride = 2 * 2 * kNumBin, .protocol = PROTO_FT8 };
    printf("frame %d waterfall_fnv %016llx\n", k, (unsigned long long)fnv64(mag, kMag, 0xCBF29CE484222325ull));
    static const int caps[] = { 7, 120, kCapMax };
    static const int mins[] = { 10, 10, 10 };
    candidate_t list[kCapMax];
    for (int c = 0; c < 3; c++) {
        memset(list, 0, sizeof list);
        const int n = ft8_find_sync(&power, caps[c], list, mins[c]);             /* :1450 */
        printf("sync cap %d min_score %d n %d\n", caps[c], mins[c], n);
        for (int i = 0; i < n; i++)
            printf("cand %d score %d time_offset %d freq_offset %d time_sub %d freq_sub %d\n", i, list[i].score, list[i].time_offset,
                   list[i].freq_offset, list[i].time_sub, list[i].freq_sub);
        if (caps[c] != 120) continue;
        for (int i = 0; i < n; i++) {                                            /* the reference's candidate loop, :1465-1485 */
            message_t message;
            decode_status_t status;
            memset(&message, 0, sizeof message);
            memset(&status, 0, sizeof status);
            const bool ok = ft8_decode(&power, &list[i], &message, 20, &status);  /* :1476 */
            printf("decode %d ok %d ldpc_errors %d", i, ok ? 1 : 0, status.ldpc_errors);
            if (status.ldpc_errors == 0) {
                printf(" crc_extracted %u crc_calculated %u", status.crc_extracted, status.crc_calculated);
                if (status.crc_extracted == status.crc_calculated) printf(" unpack_status %d", status.unpack_status);
            }
            if (ok) { printf(" hash %u text ", message.hash); print_text(message.text); }
            putchar('\n');
#ifdef PIN_INTERNALS
            float llr[174];
            uint8_t plain[174];
            pin_llr(&power, &list[i], llr);
            uint32_t bits[4];
            memcpy(bits, llr, sizeof bits);
            printf("llr %d fnv %016llx first %08x %08x %08x %08x\n", i, (unsigned long long)fnv64(llr, sizeof llr, 0xCBF29CE484222325ull),
                   bits[0], bits[1], bits[2], bits[3]);
            /* belief propagation by itself: the parity errors left after at most m iterations, m = 1..20 -- the first m that
             * reaches 0 is the iteration the decoder converges in (bp_decode has no iteration output of its own) */
            int first = -1, last = -1;
            for (int m = 1; m <= 20; m++) {
                float copy[174];
                memcpy(copy, llr, sizeof copy);
                last = pin_bp(copy, m, plain);
                if (last == 0) { first = m; break; }
            }
            printf("bp %d converges_with_max_iters %d errors_at_end %d plain_fnv %016llx\n", i, first, last,
                   (unsigned long long)fnv64(plain, sizeof plain, 0xCBF29CE484222325ull));
#endif
        }
    }
#ifdef PIN_INTERNALS
    /* every sync score of the scan ft8_find_sync walks: time_sub, freq_sub, time_offset -12..23, freq_offset 0..248 */
    uint64_t h = 0xCBF29CE484222325ull;
    long sum = 0;
    int best = -32768;
    for (int ts = 0; ts < 2; ts++) for (int fs = 0; fs < 2; fs++) for (int t0 = -12; t0 < 24; t0++) for (int f0 = 0; f0 < 249; f0++) {
        candidate_t c = { .score = 0, .time_offset = (int16_t)t0, .freq_offset = (int16_t)f0, .time_sub = (uint8_t)ts, .freq_sub = (uint8_t)fs };
        const int32_t s = pin_sync_score(&power, &c);
        h = fnv64(&s, sizeof s, h);
        sum += s;
        if (s > best) best = s;
    }
    printf("scoremap fnv %016llx sum %ld best %d\n", (unsigned long long)h, sum, best);
#endif
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s waterfalls.bin [messages.txt]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint8_t *mag = malloc(kMag);
    int k = 0;
    while (mag && fread(mag, 1, kMag, f) == kMag) dump_frame(k++, mag);
    fclose(f);
    free(mag);
    if (argc > 2) {                                                /* encoder side: pack77 (:927) + ft8_encode (:934) */
        FILE *m = fopen(argv[2], "r");
        if (!m) { perror(argv[2]); return 2; }
        char line[128];
        while (fgets(line, sizeof line, m)) {
            line[strcspn(line, "\r\n")] = 0;
            if (!line[0]) continue;
            uint8_t packed[FTX_LDPC_K_BYTES + 4], tones[FT8_NN];
            memset(packed, 0, sizeof packed);
            const int rc = pack77(line, packed);
            printf("pack ");
            print_text(line);
            printf(" rc %d", rc < 0 ? -1 : 0);
            if (rc >= 0) {
                packed[9] &= 0xF8;                                 /* 77 bits */
                printf(" payload ");
                for (int i = 0; i < 10; i++) printf("%02x", packed[i]);
                ft8_encode(packed, tones);
                printf(" tones ");
                for (int i = 0; i < FT8_NN; i++) putchar('0' + tones[i]);
            }
            putchar('\n');
        }
        fclose(m);
    }
    printf("end frames %d\n", k);
    return 0;
}
