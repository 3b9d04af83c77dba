/*
 * ft8_replay.c -- a C caller of libft8gpu.so that uses ONLY the three symbols the reference daemon
 * itself uses (initFFTW / ft8_subsystem / freeFFTW, rtlsdr_ft8d.h:155-156,164) plus the library's
 * .iq/.c2 readers and spot-table formatter.  It walks the same steps as the reference's own callers:
 *
 *   ft8_replay -t            self-test: synthesise "CQ K1JT FN20QI" exactly as decoderSelfTest()
 *                            (rtlsdr_ft8d.c:913-972: plain FSK at 50 Hz, amplitude 0.5, Box-Muller
 *                            noise from unseeded rand()), decode, check call/locator
 *   ft8_replay file.iq|.c2   file replay as decodeRecordedFile() (rtlsdr_ft8d.c:859-887)
 *
 * Spots are printed in the layout of printSpots() (rtlsdr_ft8d.c:643-663).
 * Build:  gcc -O2 -std=gnu17 -Iinclude examples/ft8_replay.c -Lrtlsdr_ft8d_amd -lft8gpu \
 *             -Wl,-rpath,$PWD/rtlsdr_ft8d_amd -lm -o examples/ft8_replay
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ft8gpu.h"

static struct decoder_results dec_results[50];       /* rtlsdr_ft8d.c:67 */

/* printSpots(), rtlsdr_ft8d.c:643-663, through the library's formatter (every counted slot is printed, as
 * upstream does, including slots of non-CQ messages that ft8_subsystem leaves untouched) */
static void print_spots(int32_t n_results, uint32_t dialfreq) {
    char text[64 + 48 * 50];
    time_t now = time(NULL);
    struct tm *gtm = gmtime(&now);
    ft8gpu_format_spots(dec_results, n_results, dialfreq, gtm->tm_year + 1900, gtm->tm_mon + 1, gtm->tm_mday,
                        gtm->tm_hour, gtm->tm_min, text, sizeof text);
    fputs(text, stdout);
}

static float white_gaussian_noise(float factor) {    /* rtlsdr_ft8d.c:890-910 */
    static double V1, V2, S;
    static int phase = 0;
    double X;
    if (phase == 0) {
        double U1, U2;
        do {
            U1 = rand() / (double)RAND_MAX;
            U2 = rand() / (double)RAND_MAX;
            V1 = 2 * U1 - 1;
            V2 = 2 * U2 - 1;
            S = V1 * V1 + V2 * V2;
        } while (S >= 1 || S == 0);
        X = V1 * sqrt(-2 * log(S) / S);
    } else {
        X = V2 * sqrt(-2 * log(S) / S);
    }
    phase = 1 - phase;
    return (float)X * factor;
}

static int self_test(void) {
    static float iSamples[FT8GPU_NSAMPLES], qSamples[FT8GPU_NSAMPLES];
    int32_t n_results = 0;
    uint8_t packed[10], tones[FT8GPU_NN];
    if (ft8gpu_pack77_std("CQ K1JT FN20QI", packed) != 0) { printf("Cannot parse message!\n"); return 0; }
    ft8gpu_encode(packed, tones);
    float f0 = 50.0, amp = 0.5, wgn = 0.02;
    double phi = 0.0, df = 3200.0 / 512.0, dt = 1 / 3200.0;
    for (int i = 0; i < FT8GPU_NN; i++) {
        double dphi = 2.0 * M_PI * dt * (f0 + ((double)tones[i] - 3.5) * df);
        for (int j = 0; j < 512; j++) {
            int index = 512 * i + j;
            iSamples[index] = amp * cos(phi) + white_gaussian_noise(wgn);
            qSamples[index] = amp * sin(phi) + white_gaussian_noise(wgn);
            phi += dphi;
        }
    }
    ft8gpu_write_raw_iq(iSamples, qSamples, "selftest.iq");
    ft8_subsystem(iSamples, qSamples, FT8GPU_NSAMPLES, dec_results, &n_results);
    print_spots(n_results, 0);
    if (strcmp(dec_results[0].call, "K1JT") && strcmp(dec_results[0].loc, "FN20")) return 0;   /* :966-971 */
    return 1;
}

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: %s -t | file.iq | file.c2\n", argv[0]); return 2; }
    initFFTW();                                      /* rtlsdr_ft8d.c:1026 */
    int rc = 0;
    if (!strcmp(argv[1], "-t")) {
        if (self_test()) { fprintf(stdout, "Self-test SUCCESS!\n"); } else { fprintf(stderr, "Self-test FAILED!\n"); rc = 1; }
    } else {
        static float iSamples[FT8GPU_NSAMPLES], qSamples[FT8GPU_NSAMPLES];
        const size_t L = strlen(argv[1]);
        int32_t samples_len = 0, n_results = 0;
        double dial = 0.0;
        if (L > 3 && !strcmp(argv[1] + L - 3, ".iq")) samples_len = ft8gpu_read_raw_iq(iSamples, qSamples, argv[1]);
        else if (L > 3 && !strcmp(argv[1] + L - 3, ".c2")) samples_len = ft8gpu_read_c2(iSamples, qSamples, argv[1], &dial);
        else { fprintf(stderr, "Not a valid extension!! (only .iq & .c2 files)\n"); freeFFTW(); return 2; }
        printf("Number of samples: %d\n", samples_len);
        if (samples_len) {
            ft8_subsystem(iSamples, qSamples, (uint32_t)samples_len, dec_results, &n_results);
            print_spots(n_results, (uint32_t)dial);
        }
    }
    freeFFTW();                                      /* rtlsdr_ft8d.c:1365 */
    return rc;
}
