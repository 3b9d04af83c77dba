/*
 * ft8_node_bench.c -- the multi-GPU path of one node driven from plain C (north_star: "host code stays C ... many
 * independent 15 s frames shard embarrassingly across the GPUs of one node with a trivial RCCL gather for the spot list").
 *
 *   ft8_node_bench [-g gpus] [-f frames_per_gpu] [-s steps] [-c signals_per_frame] [-r] [-t phase_timeout_s]
 *
 * One context per GPU (ft8gpu_create), every GPU synthesises its own contiguous shard of the job in its HBM
 * (ft8gpu_synth_frames_at: global frame g is the same samples whatever the number of GPUs), then `steps` times:
 *   default   ft8gpu_decode_batch_multi_dev -- one host thread per GPU, the 1 404 B/frame spot records of every shard land
 *             at their frame offsets of ONE host array (the gather a daemon-like consumer needs);
 *   -r        every GPU decodes into its own HBM (ft8gpu_decode_batch, FT8GPU_DEVICE_PTRS) and ft8gpu_gather_spots leaves
 *             the whole job's list on EVERY GPU with one grouped RCCL all-gather per buffer over xGMI.
 * Prints one JSON line: frames/s of the whole node, messages per frame, and a checksum of the gathered list.
 * First contact with several GPUs (RCCL's ncclCommInitAll and first grouped all-gather have never run with more than one
 * device on the boxes this was developed on): every phase runs under alarm(-t seconds, default 180); a phase that
 * outlives it ends the program with exit status 3 and "hung in phase '<name>'" on stderr instead of hanging a node.
 * Replaces nothing in the reference (a single decoder thread, rtlsdr_ft8d.c:221-285); it is the batch counterpart of its
 * decodeRecordedFile() loop (:859-887).  No HIP header, no C++: the C ABI of include/ft8gpu.h only.
 * Build:  gcc -O2 -std=gnu17 -Iinclude examples/ft8_node_bench.c -Lrtlsdr_ft8d_amd -lft8gpu \
 *             -Wl,-rpath,$PWD/rtlsdr_ft8d_amd -lm -o examples/ft8_node_bench
 */
#include <math.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "ft8gpu.h"

#define MAX_GPUS 16
#define POOL 256

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* xorshift64*: the frame descriptors only have to be reproducible, not to match the Python workload */
static uint64_t rng_next(uint64_t *s) {
    *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
    return *s * 0x2545F4914F6CDD1Dull;
}
static double rng_unit(uint64_t *s) { return (double)(rng_next(s) >> 11) / 9007199254740992.0; }

/* phase watchdog: alarm() re-armed at every phase; the handler uses async-signal-safe calls only */
static const char *volatile g_phase = "start";
static unsigned g_phase_timeout = 180;
static void on_alarm(int sig) {
    (void)sig;
    static const char a[] = "ft8_node_bench: hung in phase '", b[] = "' (limit given by -t); exiting with status 3\n";
    const char *p = g_phase;
    if (write(2, a, sizeof a - 1) < 0 || write(2, p, strlen(p)) < 0 || write(2, b, sizeof b - 1) < 0) _exit(3);
    _exit(3);
}
static void phase(const char *name) {
    g_phase = name;
    alarm(g_phase_timeout);
}

static int die(const char *what) {
    fprintf(stderr, "ft8_node_bench: %s: %s\n", what, ft8gpu_last_error());
    return 1;
}

int main(int argc, char **argv) {
    int gpus = ft8gpu_device_count(), frames = 1024, steps = 5, nsig = 20, device_gather = 0, opt;
    while ((opt = getopt(argc, argv, "g:f:s:c:rt:")) != -1) {
        if (opt == 'g') gpus = atoi(optarg);
        else if (opt == 'f') frames = atoi(optarg);
        else if (opt == 's') steps = atoi(optarg);
        else if (opt == 'c') nsig = atoi(optarg);
        else if (opt == 'r') device_gather = 1;
        else if (opt == 't') g_phase_timeout = (unsigned)atoi(optarg);
        else { fprintf(stderr, "usage: %s [-g gpus] [-f frames_per_gpu] [-s steps] [-c signals_per_frame] [-r] [-t phase_timeout_s]\n", argv[0]); return 2; }
    }
    if (gpus < 1 || gpus > MAX_GPUS || frames < 1 || steps < 1 || nsig < 0 || nsig > 64) { fprintf(stderr, "ft8_node_bench: bad arguments (%d GPUs visible)\n", ft8gpu_device_count()); return 2; }

    /* a pool of standard messages "CQ <call> <grid>" and their 79 tones (pack77 :927, ft8_encode :934) */
    static uint8_t pool_tones[POOL][FT8GPU_NN];
    for (int k = 0; k < POOL; k++) {
        char msg[32];
        uint8_t payload[10];
        snprintf(msg, sizeof msg, "CQ K%d%c%c%c %c%c%02d", k % 10, 'A' + k % 26, 'A' + (k / 26) % 26, 'A' + (k * 7) % 26, 'A' + k % 18, 'A' + (k / 18) % 18, k % 100);
        if (ft8gpu_pack77_std(msg, payload)) { fprintf(stderr, "ft8_node_bench: cannot pack '%s'\n", msg); return 1; }
        ft8gpu_encode(payload, pool_tones[k]);
    }

    signal(SIGALRM, on_alarm);
    phase("create contexts + synthesise shards");
    ft8gpu_ctx *ctx[MAX_GPUS] = { 0 };
    float *iq_dev[MAX_GPUS] = { 0 };
    struct decoder_results *dec_dev[MAX_GPUS] = { 0 }, *all_dec_dev[MAX_GPUS] = { 0 };
    int32_t *n_dev[MAX_GPUS] = { 0 }, *all_n_dev[MAX_GPUS] = { 0 };
    int nframes_dev[MAX_GPUS];
    const size_t total = (size_t)gpus * (size_t)frames;
    const size_t rec_bytes = (size_t)FT8GPU_K_MAX_MESSAGES * sizeof(struct decoder_results);
    ft8gpu_synth_signal *sig = (ft8gpu_synth_signal *)calloc((size_t)frames * (size_t)(nsig ? nsig : 1), sizeof *sig);
    if (!sig) return 1;
    for (int g = 0; g < gpus; g++) {
        if (ft8gpu_create(&ctx[g], g, frames, NULL)) return die("ft8gpu_create");
        nframes_dev[g] = frames;
        iq_dev[g] = (float *)ft8gpu_dev_alloc(ctx[g], (size_t)frames * 2 * FT8GPU_NSAMPLES * sizeof(float));
        if (!iq_dev[g]) return die("ft8gpu_dev_alloc");
        /* shard g = global frames [g * frames, (g + 1) * frames): descriptors seeded by the GLOBAL frame index */
        for (int f = 0; f < frames; f++) {
            uint64_t s = 0x46543800ull + (uint64_t)g * (uint64_t)frames + (uint64_t)f + 1;
            for (int k = 0; k < nsig; k++) {
                ft8gpu_synth_signal *x = &sig[(size_t)f * nsig + k];
                memcpy(x->tones, pool_tones[rng_next(&s) % POOL], FT8GPU_NN);
                x->f0_hz = (float)(100.0 + 1400.0 * rng_unit(&s));
                x->t0_s = (float)(1.8 * rng_unit(&s));
                const double snr_db = -18.0 + 18.0 * rng_unit(&s);           /* SNR in 2500 Hz, unit-variance complex noise over 3200 Hz */
                x->amplitude = (float)sqrt(2.0 * 2500.0 / 3200.0 * pow(10.0, snr_db / 10.0));
            }
        }
        if (ft8gpu_synth_frames_at(ctx[g], sig, frames, nsig, 1.0f, 0x46543800ull, (uint64_t)g * (uint64_t)frames, iq_dev[g])) return die("ft8gpu_synth_frames_at");
        if (device_gather) {
            dec_dev[g] = (struct decoder_results *)ft8gpu_dev_alloc(ctx[g], (size_t)frames * rec_bytes);
            n_dev[g] = (int32_t *)ft8gpu_dev_alloc(ctx[g], (size_t)frames * sizeof(int32_t));
            all_dec_dev[g] = (struct decoder_results *)ft8gpu_dev_alloc(ctx[g], total * rec_bytes);
            all_n_dev[g] = (int32_t *)ft8gpu_dev_alloc(ctx[g], total * sizeof(int32_t));
            if (!dec_dev[g] || !n_dev[g] || !all_dec_dev[g] || !all_n_dev[g]) return die("ft8gpu_dev_alloc");
        }
    }
    free(sig);

    /* the gathered list on the host: pinned, so that the downloads of the device-gather form are true DMA */
    struct decoder_results *decodes = (struct decoder_results *)ft8gpu_host_alloc(total * rec_bytes);
    int32_t *n_results = (int32_t *)ft8gpu_host_alloc(total * sizeof(int32_t));
    if (!decodes || !n_results) return die("ft8gpu_host_alloc");
    memset(decodes, 0, total * rec_bytes);

    double t0 = 0.0;
    for (int s = -1; s < steps; s++) {                     /* step -1: warm-up (buffers, RCCL communicators) */
        if (s == 0) t0 = now_s();
        phase(s < 0 ? (device_gather ? "warm-up step (ncclCommInitAll + first grouped all-gather)" : "warm-up step") : "timed steps");
        if (!device_gather) {
            if (ft8gpu_decode_batch_multi_dev(ctx, gpus, (const float *const *)iq_dev, nframes_dev, decodes, n_results)) return die("ft8gpu_decode_batch_multi_dev");
        } else {
            for (int g = 0; g < gpus; g++)
                if (ft8gpu_decode_batch(ctx[g], iq_dev[g], frames, dec_dev[g], n_dev[g], FT8GPU_DEVICE_PTRS)) return die("ft8gpu_decode_batch");
            if (ft8gpu_gather_spots(ctx, gpus, (const struct decoder_results *const *)dec_dev, (const int32_t *const *)n_dev, frames,
                                    all_dec_dev, all_n_dev)) return die("ft8gpu_gather_spots");
            for (int g = 0; g < gpus; g++)
                if (ft8gpu_synchronize(ctx[g])) return die("ft8gpu_synchronize");
        }
    }
    const double dt = now_s() - t0;
    phase("read back + checksum");
    if (device_gather) {                                   /* every GPU holds the whole list: read it back from the LAST one */
        if (ft8gpu_memcpy_d2h(ctx[gpus - 1], decodes, all_dec_dev[gpus - 1], total * rec_bytes)) return die("ft8gpu_memcpy_d2h");
        if (ft8gpu_memcpy_d2h(ctx[gpus - 1], n_results, all_n_dev[gpus - 1], total * sizeof(int32_t))) return die("ft8gpu_memcpy_d2h");
    }
    uint64_t messages = 0, sum = 1469598103934665603ull;   /* FNV-1a over counts and the used record slots */
    for (size_t f = 0; f < total; f++) {
        messages += (uint64_t)n_results[f];
        const unsigned char *p = (const unsigned char *)&decodes[f * FT8GPU_K_MAX_MESSAGES];
        const size_t used = (size_t)(n_results[f] < FT8GPU_K_MAX_MESSAGES ? n_results[f] : FT8GPU_K_MAX_MESSAGES) * sizeof(struct decoder_results);
        for (size_t i = 0; i < used; i++) sum = (sum ^ p[i]) * 1099511628211ull;
        sum = (sum ^ (uint64_t)(uint32_t)n_results[f]) * 1099511628211ull;
    }
    int overlap = 1;
    for (int g = 0; g < gpus; g++) overlap &= ft8gpu_overlap_active(ctx[g]) == 1;
    printf("{\"gpus\": %d, \"frames_per_gpu\": %d, \"steps\": %d, \"gather\": \"%s\", \"frames_per_s\": %.1f, \"ms_per_step\": %.3f, "
           "\"messages_per_frame\": %.2f, \"overlap\": %s, \"list_fnv1a\": \"%016llx\"}\n",
           gpus, frames, steps, device_gather ? "rccl all-gather, device-resident" : "host arrays (per-shard offsets)",
           (double)total * steps / dt, 1e3 * dt / steps, (double)messages / (double)total, overlap ? "true" : "false", (unsigned long long)sum);

    for (int g = 0; g < gpus; g++) {
        ft8gpu_dev_free(ctx[g], iq_dev[g]);
        if (device_gather) { ft8gpu_dev_free(ctx[g], dec_dev[g]); ft8gpu_dev_free(ctx[g], n_dev[g]); ft8gpu_dev_free(ctx[g], all_dec_dev[g]); ft8gpu_dev_free(ctx[g], all_n_dev[g]); }
    }
    phase("shutdown");
    if (device_gather) ft8gpu_gather_shutdown();
    for (int g = 0; g < gpus; g++) ft8gpu_destroy(ctx[g]);
    ft8gpu_host_free(decodes);
    ft8gpu_host_free(n_results);
    alarm(0);
    return 0;
}
