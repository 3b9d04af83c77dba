/*
 * Host-only sanitizer harness for csrc/ft8_compat.c and csrc/ft8_pack.c (pack77 / ft8_encode / .iq and .c2 readers /
 * printSpots formatter / the drop-in shim): compiled with gcc -fsanitize=address,undefined together with both and
 * run by tests/test_sanitizers.py on the CPU box (GPU AddressSanitizer is not available on the pool).
 * The four ft8gpu_* entry points the shim calls live in the HIP half of the library; here they are replaced
 * by failing definitions, which also drives the shim's "no GPU" path (*n_results = 0, reason on stderr).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "ft8gpu.h"
#include "ft8_lib/ft8/decode.h"
#include "ft8_lib/ft8/encode.h"
#include "ft8_lib/ft8/pack.h"

int ft8gpu_create(ft8gpu_ctx **out, int device, int max_frames, const ft8gpu_params *params) {
    (void)device; (void)max_frames; (void)params;
    *out = NULL;
    return -1;
}
void ft8gpu_destroy(ft8gpu_ctx *ctx) { (void)ctx; }
const char *ft8gpu_last_error(void) { return "host sanitizer harness: no GPU half linked"; }
int ft8gpu_decode_batch(ft8gpu_ctx *ctx, const float *iq, int nframes, struct decoder_results *decodes, int32_t *n_results, int flags) {
    (void)ctx; (void)iq; (void)nframes; (void)decodes; (void)n_results; (void)flags;
    return -1;
}

int ft8gpu_set_params(ft8gpu_ctx *ctx, const ft8gpu_params *params) { (void)ctx; (void)params; return -1; }
int ft8gpu_find_sync(ft8gpu_ctx *ctx, const uint8_t *mag, int nframes, ft8gpu_candidate *cands, int32_t *counts, int flags) {
    (void)ctx; (void)mag; (void)nframes; (void)cands; (void)counts; (void)flags;
    return -1;
}
int ft8gpu_decode_candidates(ft8gpu_ctx *ctx, const uint8_t *mag, const ft8gpu_candidate *cands, const int32_t *counts,
                             int nframes, ft8gpu_decode_status *status, int flags) {
    (void)ctx; (void)mag; (void)cands; (void)counts; (void)nframes; (void)status; (void)flags;
    return -1;
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    /* pack77 + encode: the reference's known answer (rtlsdr_ft8d.c:919-923) and a pile of malformed inputs */
    static const uint8_t kat_packed[10] = { 0x00, 0x00, 0x00, 0x20, 0x4d, 0xfc, 0xdc, 0x8a, 0x14, 0x08 };
    static const char kat_tones[] = "3140652000000001005477547106035036373140652547441342116056460065174427143140652";
    uint8_t p[10], tones[FT8GPU_NN];
    CHECK(ft8gpu_pack77_std("CQ K1JT FN20QI", p) == 0 && !memcmp(p, kat_packed, 10));
    ft8gpu_encode(p, tones);
    for (int i = 0; i < FT8GPU_NN; i++) CHECK(tones[i] == kat_tones[i] - '0');
    static const char *bad[] = { "", " ", "CQ", "CQ ", "CQ  ", "CQ K1JT FN2", "CQ K1JT ZZ99", "CQ k1jt FN20", "CQ TOOLONGCALL FN20",
                                 "CQ K1JT FN20 EXTRA TOKENS HERE", "K K K", "CQ 1 AA00", "QRZ DL1ABC", "DE W1AW JO62", "CQ K1JT \xff\xfe\xfd\xfc",
                                 "ABCDEFGHIJKLMNOPQRSTUVWXYZABCDEFGHIJKLMNOPQRSTUVWXYZ 0123456789 0123456789" };
    for (size_t i = 0; i < sizeof bad / sizeof bad[0]; i++) {
        uint8_t q[10];
        if (ft8gpu_pack77_std(bad[i], q) == 0) ft8gpu_encode(q, tones);
    }
    CHECK(ft8gpu_pack77_std(NULL, p) != 0 && ft8gpu_pack77_std("CQ K1JT FN20", NULL) != 0);
    /* the full packer (csrc/ft8_pack.c): every bad[] text again, hand-picked edge shapes, and 200 000 random strings over
     * the characters its parsers branch on -- whatever it accepts must encode, nothing may read or write out of bounds */
    static const char *edge[] = { "<", ">", "<>", "<> <>", "< > K1ABC", "<K1ABC", "K1ABC>", "<K1ABC> <W9XYZ>", "<ABCDEFGHIJKL> K1ABC", "CQ <K1ABC>",
                                  "CQ 000 K1ABC", "CQ 999 K1ABC FN20", "CQ ZZZZ K1ABC", "CQ ABCDE K1ABC", "CQ K1ABC/R", "CQ K1ABC/P FN20", "/R /P", "K1ABC/ W9XYZ",
                                  "K1ABC W9XYZ R", "K1ABC W9XYZ R R", "K1ABC W9XYZ R FN2", "K1ABC W9XYZ +00", "K1ABC W9XYZ -30", "K1ABC W9XYZ -31", "K1ABC W9XYZ R+99",
                                  "K1ABC W9XYZ +100", "K1ABC W9XYZ FN20QIX", "K1ABC W9XYZ FN20Q!", "3DA0", "3DA0XYZW K1ABC", "3X K1ABC", "3XA0XYZW K1ABC FN20",
                                  "7FFFFFFFFFFFFFFFFF", "8000000000000000000", "FFFFFFFFFFFFFFFFF", "CQ ABCDEFGHIJK", "CQ ABCDEFGHIJKL", "CQ A", "CQ /",
                                  "A/B/C/D/E/F <K1ABC> 73", "<K1ABC> A/B/C/D/E/F RR73 ", "             ", "1234567890123", "12345678901234", "+-./?", "a", "\x01" };
    CHECK(ft8gpu_pack77(NULL, p) != 0 && ft8gpu_pack77("CQ K1JT FN20", NULL) != 0);
    for (size_t i = 0; i < sizeof bad / sizeof bad[0]; i++) if (ft8gpu_pack77(bad[i], p) == 0) ft8gpu_encode(p, tones);
    for (size_t i = 0; i < sizeof edge / sizeof edge[0]; i++) if (ft8gpu_pack77(edge[i], p) == 0) ft8gpu_encode(p, tones);
    {
        static const char pool[] = "  KW19ABCXYZR73<>/+-?.QDEFN20P";
        uint64_t s = 0x2545F4914F6CDD1Dull;
        int accepted = 0;
        for (int it = 0; it < 200000; it++) {
            char text[48];
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const int len = (int)(s % 45);
            uint64_t r = s;
            for (int k = 0; k < len; k++) { r = r * 6364136223846793005ull + 1442695040888963407ull; text[k] = pool[(r >> 33) % (sizeof pool - 1)]; }
            text[len] = 0;
            if (ft8gpu_pack77(text, p) == 0) { ft8gpu_encode(p, tones); accepted++; CHECK((p[9] & 7) == 0); }
        }
        CHECK(accepted > 1000);
    }

    /* file formats: full, truncated, empty and missing files (rtlsdr_ft8d.c:744-856) */
    float *I = malloc(sizeof(float) * FT8GPU_NSAMPLES), *Q = malloc(sizeof(float) * FT8GPU_NSAMPLES);
    float *I2 = malloc(sizeof(float) * FT8GPU_NSAMPLES), *Q2 = malloc(sizeof(float) * FT8GPU_NSAMPLES);
    CHECK(I && Q && I2 && Q2);
    for (int i = 0; i < FT8GPU_NSAMPLES; i++) { I[i] = 0.25f * (float)((i * 37) % 101 - 50) / 50.0f; Q[i] = 0.5f * (float)((i * 53) % 89 - 44) / 44.0f; }
    char path[512];
    snprintf(path, sizeof path, "%s/asan_full.iq", dir);
    CHECK(ft8gpu_write_raw_iq(I, Q, path) == FT8GPU_NSAMPLES);
    CHECK(ft8gpu_read_raw_iq(I2, Q2, path) == FT8GPU_NSAMPLES);
    float peak = 0;
    for (int i = 0; i < FT8GPU_NSAMPLES; i++) { if (I2[i] > peak) peak = I2[i]; if (-I2[i] > peak) peak = -I2[i]; if (Q2[i] > peak) peak = Q2[i]; if (-Q2[i] > peak) peak = -Q2[i]; }
    CHECK(peak > 0.4999f && peak < 0.5001f);
    CHECK(truncate(path, 8 * 1000 + 3) == 0);                               /* ragged tail: 1000 pairs and 3 stray bytes */
    CHECK(ft8gpu_read_raw_iq(I2, Q2, path) == 1000);
    CHECK(truncate(path, 0) == 0);
    CHECK(ft8gpu_read_raw_iq(I2, Q2, path) == 0);
    snprintf(path, sizeof path, "%s/asan_missing.iq", dir);
    CHECK(ft8gpu_read_raw_iq(I2, Q2, path) == 0);
    snprintf(path, sizeof path, "%s/asan.c2", dir);
    FILE *f = fopen(path, "wb");
    CHECK(f != NULL);
    const char name[14] = "000000_0000.c2";
    int type = 2;
    double freq = 14074000.0, dial = 0;
    fwrite(name, 1, 14, f); fwrite(&type, sizeof type, 1, f); fwrite(&freq, sizeof freq, 1, f);
    for (int i = 0; i < 777; i++) { float pair[2] = { I[i], Q[i] }; fwrite(pair, sizeof(float), 2, f); }
    fclose(f);
    CHECK(ft8gpu_read_c2(I2, Q2, path, &dial) == 777 && dial == freq);
    CHECK(ft8gpu_read_c2(I2, Q2, path, NULL) == 777);
    CHECK(truncate(path, 9) == 0);                                          /* shorter than the header */
    CHECK(ft8gpu_read_c2(I2, Q2, path, &dial) == 0);

    /* printSpots formatter: tiny caps, counts beyond 50, fields without a terminating NUL */
    struct decoder_results *d = malloc(sizeof *d * FT8GPU_K_MAX_MESSAGES);
    CHECK(d != NULL);
    for (int i = 0; i < FT8GPU_K_MAX_MESSAGES; i++) { memset(d[i].call, 'A' + i % 26, sizeof d[i].call); memset(d[i].loc, 'a' + i % 26, sizeof d[i].loc); d[i].freq = 2147483647 - i; d[i].snr = -99 + 5 * i; }
    char *text = malloc(8192);
    CHECK(text != NULL);
    const int full = ft8gpu_format_spots(d, 1000, 4000000000u, 2026, 12, 31, 23, 59, text, 8192);
    CHECK(full > 0 && (size_t)full == strlen(text));
    for (size_t cap = 0; cap < 70; cap++) {
        char *small = malloc(cap ? cap : 1);
        CHECK(small != NULL);
        CHECK(ft8gpu_format_spots(d, 50, 7074000u, 2026, 1, 1, 0, 0, cap ? small : NULL, cap) > 0);
        if (cap) CHECK(strlen(small) < cap);
        free(small);
    }
    CHECK(ft8gpu_format_spots(NULL, 0, 0, 1, 1, 1, 1, 1, text, 8192) > 0);
    CHECK(ft8gpu_format_spots(NULL, 3, 0, 1, 1, 1, 1, 1, text, 8192) < 0);

    /* the drop-in shim without a GPU: no crash, *n_results = 0, caller's records untouched */
    struct decoder_results before[FT8GPU_K_MAX_MESSAGES];
    memcpy(before, d, sizeof before);
    int32_t n = 123;
    initFFTW();
    ft8_subsystem(I, Q, FT8GPU_NSAMPLES, d, &n);
    CHECK(n == 0 && !memcmp(before, d, sizeof before));
    freeFFTW();
    freeFFTW();
    /* the ft8_lib-level entries without a GPU: refuse quietly, leave the caller's buffers alone */
    uint8_t c77[FTX_LDPC_K_BYTES], tn[FT8_NN];
    CHECK(pack77("CQ K1JT FN20QI", c77) == 0 && !memcmp(c77, kat_packed, 10) && c77[10] == 0 && c77[11] == 0);
    CHECK(pack77("not a message", c77) < 0);
    ft8_encode(c77, tn);
    CHECK(!memcmp(tn, tones, 0) && tn[0] == 3 && tn[78] == 2);
    uint8_t *mag = calloc(FT8GPU_MAG_ARRAY, 1);
    CHECK(mag != NULL);
    waterfall_t wf = { .num_blocks = 92, .num_bins = 256, .time_osr = 2, .freq_osr = 2, .mag = mag, .block_stride = 1024, .protocol = PROTO_FT8 };
    candidate_t heap[8];
    memset(heap, 0x11, sizeof heap);
    CHECK(ft8_find_sync(&wf, 8, heap, 10) == 0);
    message_t msg;
    decode_status_t st;
    CHECK(!ft8_decode(&wf, &heap[0], &msg, 20, &st) && st.ldpc_errors == FTX_LDPC_M);
    CHECK(!ft8_decode(&wf, &heap[0], &msg, 20, NULL));
    wf.num_blocks = 91;
    CHECK(ft8_find_sync(&wf, 8, heap, 10) == 0 && !ft8_decode(&wf, &heap[0], &msg, 20, &st));
    CHECK(ft8_find_sync(NULL, 8, heap, 10) == 0 && !ft8_decode(NULL, NULL, NULL, 0, NULL));
    free(mag);
    free(I); free(Q); free(I2); free(Q2); free(d); free(text);
    puts("compat_asan ok");
    return 0;
}
