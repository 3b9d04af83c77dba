/*
 * ft8_oracle.c -- CPU ORACLE (test infrastructure only; see ft8_oracle.h for the contract,
 * the list of reference functions restated and the "parity unpinned" statement).
 *
 * Build: gcc -O3 -std=gnu17 -ffp-contract=off -fno-fast-math (oracle/Makefile; `make asan` for the sanitizer build).
 * -ffp-contract=off matters: the product kernels are compiled the same way so that every
 * float operation below is one IEEE-754 binary32 operation on both sides.
 */
#define _GNU_SOURCE
#include "ft8_oracle.h"
#include "ft8o_tables.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * initFFTW()  rtlsdr_ft8d.c:314-335
 * ---------------------------------------------------------------------------------------- */
static float g_hann[FT8O_NFFT];
static float g_tw[FT8O_NFFT][2];
static int   g_rev[FT8O_NFFT];
static int   g_init = 0;

void ft8o_init(void) {
    if (g_init) return;
    /* rtlsdr_ft8d.c:331-334 -- "hann" is a sine window: sinf((M_PI / NFFT) * i); the product
     * (M_PI / NFFT) * i is evaluated in double and converted to float by the sinf() call. */
    for (int i = 0; i < FT8O_NFFT; i++)
        g_hann[i] = sinf((M_PI / FT8O_NFFT) * i);
    /* twiddles W^k = exp(-2*pi*i*k/1024), evaluated in double, rounded once to float */
    for (int k = 0; k < FT8O_NFFT; k++) {
        double a = 2.0 * M_PI * (double)k / (double)FT8O_NFFT;
        g_tw[k][0] = (float)cos(a);
        g_tw[k][1] = (float)(-sin(a));
    }
    /* base-4 digit reversal over 5 digits */
    for (int p = 0; p < FT8O_NFFT; p++) {
        int r = 0, v = p;
        for (int d = 0; d < 5; d++) { r = (r << 2) | (v & 3); v >>= 2; }
        g_rev[p] = r;
    }
    g_init = 1;
}
const float *ft8o_window(void)   { ft8o_init(); return g_hann; }
const float *ft8o_twiddles(void) { ft8o_init(); return &g_tw[0][0]; }

/* ------------------------------------------------------------------------------------------
 * fftwf_execute(fft_plan) stand-in  (rtlsdr_ft8d.c:326, :1411)
 *
 * Forward complex DFT, N = 1024, unnormalised.  Algorithm "R4DIF-1024":
 *   5 in-place radix-4 decimation-in-frequency stages, s = 0..4, L = 1024 >> 2s, Q = L/4,
 *   twiddle stride T = 1024 / L.  For every block base (multiple of L) and j in [0,Q):
 *       a0..a3 = x[base + j + {0,Q,2Q,3Q}]
 *       t0 = a0 + a2   t1 = a0 - a2   t2 = a1 + a3   t3 = a1 - a3
 *       y0 = t0 + t2   y2 = t0 - t2
 *       y1 = (t1.re + t3.im, t1.im - t3.re)      ( = t1 - i*t3 )
 *       y3 = (t1.re - t3.im, t1.im + t3.re)      ( = t1 + i*t3 )
 *       x[base+j] = y0,  x[base+j+qQ] = y_q * W^(q*j*T), q = 1..3
 *   The complex product is (yr*wr - yi*wi, yr*wi + yi*wr): four roundings for the products,
 *   two for the sums, no fused multiply-add.  Stages 0..3 always multiply (also when the
 *   twiddle is exactly 1); stage 4 (L = 4) has j = 0 only and never multiplies.
 *   After stage 4 position p holds X[rev4(p)] (base-4 digit reversal of the 5 digits).
 * ---------------------------------------------------------------------------------------- */
static void r4dif_inplace(float *re, float *im) {
    for (int s = 0; s < 5; s++) {
        const int L = FT8O_NFFT >> (2 * s), Q = L >> 2, T = FT8O_NFFT / L;
        for (int base = 0; base < FT8O_NFFT; base += L) {
            for (int j = 0; j < Q; j++) {
                const int i0 = base + j, i1 = i0 + Q, i2 = i1 + Q, i3 = i2 + Q;
                const float t0r = re[i0] + re[i2], t0i = im[i0] + im[i2];
                const float t1r = re[i0] - re[i2], t1i = im[i0] - im[i2];
                const float t2r = re[i1] + re[i3], t2i = im[i1] + im[i3];
                const float t3r = re[i1] - re[i3], t3i = im[i1] - im[i3];
                const float y0r = t0r + t2r, y0i = t0i + t2i;
                const float y2r = t0r - t2r, y2i = t0i - t2i;
                const float y1r = t1r + t3i, y1i = t1i - t3r;
                const float y3r = t1r - t3i, y3i = t1i + t3r;
                re[i0] = y0r; im[i0] = y0i;
                if (s < 4) {
                    const float *w1 = g_tw[(1 * j * T) & 1023];
                    const float *w2 = g_tw[(2 * j * T) & 1023];
                    const float *w3 = g_tw[(3 * j * T) & 1023];
                    float p1, p2, p3, p4;
                    p1 = y1r * w1[0]; p2 = y1i * w1[1]; p3 = y1r * w1[1]; p4 = y1i * w1[0];
                    re[i1] = p1 - p2; im[i1] = p3 + p4;
                    p1 = y2r * w2[0]; p2 = y2i * w2[1]; p3 = y2r * w2[1]; p4 = y2i * w2[0];
                    re[i2] = p1 - p2; im[i2] = p3 + p4;
                    p1 = y3r * w3[0]; p2 = y3i * w3[1]; p3 = y3r * w3[1]; p4 = y3i * w3[0];
                    re[i3] = p1 - p2; im[i3] = p3 + p4;
                } else {
                    re[i1] = y1r; im[i1] = y1i;
                    re[i2] = y2r; im[i2] = y2i;
                    re[i3] = y3r; im[i3] = y3i;
                }
            }
        }
    }
}

void ft8o_fft1024(float *re, float *im) {
    ft8o_init();
    float tr[FT8O_NFFT], ti[FT8O_NFFT];
    r4dif_inplace(re, im);
    for (int p = 0; p < FT8O_NFFT; p++) { tr[g_rev[p]] = re[p]; ti[g_rev[p]] = im[p]; }
    memcpy(re, tr, sizeof tr);
    memcpy(im, ti, sizeof ti);
}

/* float64 radix-2 FFT: the "truth" the float32 FFT above is validated against in tests */
void ft8o_fft1024_f64(const double *re_in, const double *im_in, double *re, double *im) {
    const int N = FT8O_NFFT;
    for (int i = 0; i < N; i++) {
        int r = 0, v = i;
        for (int b = 0; b < 10; b++) { r = (r << 1) | (v & 1); v >>= 1; }
        re[r] = re_in[i]; im[r] = im_in[i];
    }
    for (int len = 2; len <= N; len <<= 1) {
        for (int base = 0; base < N; base += len) {
            for (int j = 0; j < len / 2; j++) {
                double a = -2.0 * M_PI * (double)j / (double)len;
                double wr = cos(a), wi = sin(a);
                int u = base + j, v = u + len / 2;
                double xr = re[v] * wr - im[v] * wi, xi = re[v] * wi + im[v] * wr;
                re[v] = re[u] - xr; im[v] = im[u] - xi;
                re[u] = re[u] + xr; im[u] = im[u] + xi;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * rtlsdr_ft8d.c:1415-1427 for one bin: log magnitude, scale to 0.5 dB steps, clamp to uint8
 * ---------------------------------------------------------------------------------------- */
/* Two behaviours, kept apart on purpose:
 *  - reference-faithful: the statement sequence of :1416, :1425, :1427 exactly as the reference's x86 build executes it.
 *    For a non-finite or huge db the (int) conversion is undefined behaviour in C; x86's cvttss2si returns INT_MIN
 *    ("integer indefinite"), which the clamp turns into 0 -- for +inf and NaN alike.
 *  - fenced (the default of this oracle and the definition the product is tested against): +inf -> 255, NaN -> 0,
 *    i.e. the saturating conversion of the reference's ARM (Raspberry Pi) targets.  THIS DEVIATES from the x86 build of
 *    the reference for non-finite |X|^2; it is a fence around undefined behaviour, not a parity claim
 *    (tests/test_oracle.py::test_quantiser_fence_is_separate_from_the_reference_expression pins the difference).
 * For every finite db both are the same expression. */
static int g_quantiser_x86 = 0;
void ft8o_set_quantiser_x86(int on) { g_quantiser_x86 = on != 0; }

uint8_t ft8o_quantise_x86(float mag2) {
    float db = 10.0f * log10f(1E-12f + mag2 * 4.0f / (float)((uint32_t)FT8O_NFFT * (uint32_t)FT8O_NFFT));   /* :1416 */
    const float v = 2 * db + 240;
    /* (int)v of :1425 where it is defined; INT_MIN where cvttss2si would report "indefinite" (NaN, |v| >= 2^31) */
    int scaled = (v != v || v >= 2147483648.0f || v < -2147483648.0f) ? (-2147483647 - 1) : (int)v;
    return (uint8_t)((scaled < 0) ? 0 : ((scaled > 255) ? 255 : scaled));                                     /* :1427 */
}

uint8_t ft8o_quantise(float mag2) {
    if (g_quantiser_x86) return ft8o_quantise_x86(mag2);
    /* :1416  NFFT*NFFT is uint32_t 1048576, converted to float for the division */
    float db = 10.0f * log10f(1E-12f + mag2 * 4.0f / (float)((uint32_t)FT8O_NFFT * (uint32_t)FT8O_NFFT));
    /* Fence (see above): non-finite db */
    if (isnan(db)) return 0;
    if (isinf(db)) return (uint8_t)(db > 0 ? 255 : 0);
    int scaled = (int)(2 * db + 240);                                   /* :1425 */
    return (uint8_t)((scaled < 0) ? 0 : ((scaled > 255) ? 255 : scaled)); /* :1427 */
}

/* rtlsdr_ft8d.c:1395-1435 */
void ft8o_waterfall(const float *iSamples, const float *qSamples, uint8_t *mag_power) {
    ft8o_init();
    int offset = 0;
    float fr[FT8O_NFFT], fi[FT8O_NFFT];
    for (int idx_block = 0; idx_block < FT8O_NUM_BLOCKS; ++idx_block) {
        for (int time_sub = 0; time_sub < FT8O_K_TIME_OSR; ++time_sub) {
            uint8_t q[FT8O_NFFT / 2];
            const int start = idx_block * FT8O_BLOCK_SIZE + time_sub * FT8O_SUB_BLOCK_SIZE;
            for (int i = 0; i < FT8O_NFFT; ++i) {                       /* :1407-1410 */
                fr[i] = iSamples[start + i] * g_hann[i];
                fi[i] = qSamples[start + i] * g_hann[i];
            }
            r4dif_inplace(fr, fi);                                      /* :1411 */
            /* :1414-1417 (the reference computes all 1024 bins, only 0..511 are consumed) */
            for (int p = 0; p < FT8O_NFFT; ++p) {
                int k = g_rev[p];
                if (k < FT8O_NFFT / 2) {
                    float mag2 = fr[p] * fr[p] + fi[p] * fi[p];
                    q[k] = ft8o_quantise(mag2);
                }
            }
            for (int freq_sub = 0; freq_sub < FT8O_K_FREQ_OSR; ++freq_sub)   /* :1420-1433 */
                for (int pos = 0; pos < FT8O_NUM_BIN; ++pos)
                    mag_power[offset++] = q[pos * FT8O_K_FREQ_OSR + freq_sub];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Optional run-time FFTW leg: the reference's OWN transform where the box has it.
 * The reference plans fftwf_plan_dft_1d(NFFT, fft_in, fft_out, FFTW_FORWARD, PATIENCE) with PATIENCE = FFTW_ESTIMATE
 * (rtlsdr_ft8d.c:326, rtlsdr_ft8d.h:65) on fftwf_malloc'ed buffers (:322-323) and runs fftwf_execute per row (:1411).
 * libfftw3f is not linked (this image has neither the library nor fftw3.h): it is bound with dlopen, its five
 * prototypes declared by hand from FFTW's documented API.  Rows are transformed with fftwf_execute_dft on per-thread
 * fftwf_malloc'ed buffers -- the documented thread-safe form of fftwf_execute, same plan, same alignment, same
 * arithmetic -- so that the batch runs one frame per thread like the other legs.  FFTW's float butterflies are not
 * R4DIF's: a waterfall byte may differ by one step where |X|^2 sits on a quantiser threshold (the same effect the
 * float64 leg measures, tools/fft_parity.py); bench.py reports the flip counts next to the rate.
 * ---------------------------------------------------------------------------------------- */
typedef float ft8o_fftwf_complex[2];
static struct {
    int tried, ok;
    void *lib, *plan;
    void *(*malloc_)(size_t);
    void (*free_)(void *);
    void *(*plan_dft_1d)(int, ft8o_fftwf_complex *, ft8o_fftwf_complex *, int, unsigned);
    void (*execute_dft)(void *, ft8o_fftwf_complex *, ft8o_fftwf_complex *);
    void (*destroy_plan)(void *);
    char detail[768];
} g_fftw;

/* 1: libfftw3f is bound and the reference's plan exists; 0: not found (ft8o_fftw_detail() lists what was searched).
 * explicit_path: tried first when not NULL (tests); NULL = the system's library only. */
int ft8o_fftw_init(const char *explicit_path) {
    if (g_fftw.tried) return g_fftw.ok;
    g_fftw.tried = 1;
    static const char *names[] = { "libfftw3f.so.3", "libfftw3f.so", "/usr/lib/x86_64-linux-gnu/libfftw3f.so.3", "/usr/lib64/libfftw3f.so.3",
                                   "/usr/local/lib/libfftw3f.so.3", "/opt/conda/lib/libfftw3f.so.3" };
    size_t at = 0;
    const char *found = NULL;
    if (explicit_path && (g_fftw.lib = dlopen(explicit_path, RTLD_NOW | RTLD_LOCAL))) found = explicit_path;
    for (size_t i = 0; !g_fftw.lib && i < sizeof names / sizeof names[0]; i++) {
        if ((g_fftw.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL))) { found = names[i]; break; }
        at += (size_t)snprintf(g_fftw.detail + at, at < sizeof g_fftw.detail ? sizeof g_fftw.detail - at : 0, "%s%s", i ? ", " : "", names[i]);
        if (at >= sizeof g_fftw.detail) at = sizeof g_fftw.detail - 1;
    }
    if (!g_fftw.lib) return 0;
    g_fftw.malloc_ = (void *(*)(size_t))dlsym(g_fftw.lib, "fftwf_malloc");
    g_fftw.free_ = (void (*)(void *))dlsym(g_fftw.lib, "fftwf_free");
    g_fftw.plan_dft_1d = (void *(*)(int, ft8o_fftwf_complex *, ft8o_fftwf_complex *, int, unsigned))dlsym(g_fftw.lib, "fftwf_plan_dft_1d");
    g_fftw.execute_dft = (void (*)(void *, ft8o_fftwf_complex *, ft8o_fftwf_complex *))dlsym(g_fftw.lib, "fftwf_execute_dft");
    g_fftw.destroy_plan = (void (*)(void *))dlsym(g_fftw.lib, "fftwf_destroy_plan");
    if (!g_fftw.malloc_ || !g_fftw.free_ || !g_fftw.plan_dft_1d || !g_fftw.execute_dft || !g_fftw.destroy_plan) {
        snprintf(g_fftw.detail, sizeof g_fftw.detail, "%s lacks a required fftwf_* symbol", found);
        return 0;
    }
    ft8o_fftwf_complex *in = (ft8o_fftwf_complex *)g_fftw.malloc_(sizeof(ft8o_fftwf_complex) * FT8O_NFFT);     /* :322-323 */
    ft8o_fftwf_complex *out = (ft8o_fftwf_complex *)g_fftw.malloc_(sizeof(ft8o_fftwf_complex) * FT8O_NFFT);
    if (in && out) g_fftw.plan = g_fftw.plan_dft_1d(FT8O_NFFT, in, out, -1 /* FFTW_FORWARD */, 1u << 6 /* FFTW_ESTIMATE */);   /* :326 */
    if (in) g_fftw.free_(in);                          /* the plan is executed on other buffers of the same alignment */
    if (out) g_fftw.free_(out);
    if (!g_fftw.plan) { snprintf(g_fftw.detail, sizeof g_fftw.detail, "%s: fftwf_plan_dft_1d failed", found); return 0; }
    snprintf(g_fftw.detail, sizeof g_fftw.detail, "%s", found);
    g_fftw.ok = 1;
    return 1;
}
const char *ft8o_fftw_detail(void) { return g_fftw.detail; }

/* rtlsdr_ft8d.c:1395-1435 with the reference's own FFT; returns 0, or -1 when FFTW is not bound */
int ft8o_waterfall_fftw(const float *iSamples, const float *qSamples, uint8_t *mag_power) {
    ft8o_init();
    if (!g_fftw.ok) return -1;
    ft8o_fftwf_complex *in = (ft8o_fftwf_complex *)g_fftw.malloc_(sizeof(ft8o_fftwf_complex) * FT8O_NFFT);
    ft8o_fftwf_complex *out = (ft8o_fftwf_complex *)g_fftw.malloc_(sizeof(ft8o_fftwf_complex) * FT8O_NFFT);
    if (!in || !out) { if (in) g_fftw.free_(in); if (out) g_fftw.free_(out); return -1; }
    int offset = 0;
    for (int idx_block = 0; idx_block < FT8O_NUM_BLOCKS; ++idx_block) {
        for (int time_sub = 0; time_sub < FT8O_K_TIME_OSR; ++time_sub) {
            const int start = idx_block * FT8O_BLOCK_SIZE + time_sub * FT8O_SUB_BLOCK_SIZE;
            for (int i = 0; i < FT8O_NFFT; ++i) {                       /* :1407-1410 */
                in[i][0] = iSamples[start + i] * g_hann[i];
                in[i][1] = qSamples[start + i] * g_hann[i];
            }
            g_fftw.execute_dft(g_fftw.plan, in, out);                   /* :1411 */
            for (int freq_sub = 0; freq_sub < FT8O_K_FREQ_OSR; ++freq_sub)   /* :1414-1433 */
                for (int pos = 0; pos < FT8O_NUM_BIN; ++pos) {
                    const int k = pos * FT8O_K_FREQ_OSR + freq_sub;
                    mag_power[offset++] = ft8o_quantise(out[k][0] * out[k][0] + out[k][1] * out[k][1]);
                }
        }
    }
    g_fftw.free_(in);
    g_fftw.free_(out);
    return 0;
}

/* same path with the DFT carried out in float64 (window product still float, as the reference) */
void ft8o_waterfall_f64(const float *iSamples, const float *qSamples, uint8_t *mag_power) {
    ft8o_init();
    int offset = 0;
    static __thread double xr[FT8O_NFFT], xi[FT8O_NFFT], yr[FT8O_NFFT], yi[FT8O_NFFT];
    for (int idx_block = 0; idx_block < FT8O_NUM_BLOCKS; ++idx_block) {
        for (int time_sub = 0; time_sub < FT8O_K_TIME_OSR; ++time_sub) {
            const int start = idx_block * FT8O_BLOCK_SIZE + time_sub * FT8O_SUB_BLOCK_SIZE;
            for (int i = 0; i < FT8O_NFFT; ++i) {
                xr[i] = (double)(iSamples[start + i] * g_hann[i]);
                xi[i] = (double)(qSamples[start + i] * g_hann[i]);
            }
            ft8o_fft1024_f64(xr, xi, yr, yi);
            for (int freq_sub = 0; freq_sub < FT8O_K_FREQ_OSR; ++freq_sub)
                for (int pos = 0; pos < FT8O_NUM_BIN; ++pos) {
                    int k = pos * FT8O_K_FREQ_OSR + freq_sub;
                    mag_power[offset++] = ft8o_quantise((float)(yr[k] * yr[k] + yi[k] * yi[k]));
                }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib decode.c: get_index / ft8_sync_score / heap / ft8_find_sync
 * (call site rtlsdr_ft8d.c:1450; waterfall_t descriptor rtlsdr_ft8d.c:1440-1448)
 * ---------------------------------------------------------------------------------------- */
static int get_index(const ft8o_candidate_t *c) {
    int offset = c->time_offset;
    offset = (offset * FT8O_K_TIME_OSR) + c->time_sub;
    offset = (offset * FT8O_K_FREQ_OSR) + c->freq_sub;
    offset = (offset * FT8O_NUM_BIN) + c->freq_offset;
    return offset;
}

int ft8o_sync_score(const uint8_t *mag, const ft8o_candidate_t *c) {
    int score = 0, num_average = 0;
    const uint8_t *mag_cand = mag + get_index(c);
    for (int m = 0; m < 3; ++m) {                     /* FT8_NUM_SYNC */
        for (int k = 0; k < 7; ++k) {                 /* FT8_LENGTH_SYNC */
            int block = (36 * m) + k;                 /* FT8_SYNC_OFFSET */
            int block_abs = c->time_offset + block;
            if (block_abs < 0) continue;
            if (block_abs >= FT8O_NUM_BLOCKS) break;
            const uint8_t *p8 = mag_cand + (block * FT8O_BLOCK_STRIDE);
            int sm = kO_Costas[k];
            if (sm > 0) { score += p8[sm] - p8[sm - 1]; ++num_average; }
            if (sm < 7) { score += p8[sm] - p8[sm + 1]; ++num_average; }
            if ((k > 0) && (block_abs > 0)) {
                score += p8[sm] - p8[sm - FT8O_BLOCK_STRIDE]; ++num_average;
            }
            if (((k + 1) < 7) && ((block_abs + 1) < FT8O_NUM_BLOCKS)) {
                score += p8[sm] - p8[sm + FT8O_BLOCK_STRIDE]; ++num_average;
            }
        }
    }
    if (num_average > 0) score /= num_average;        /* C int division: truncates toward zero */
    return score;
}

static void heapify_down(ft8o_candidate_t heap[], int heap_size) {
    int current = 0;
    while (1) {
        int largest = current;
        int left = 2 * current + 1;
        int right = left + 1;
        if (left < heap_size && heap[left].score < heap[largest].score) largest = left;
        if (right < heap_size && heap[right].score < heap[largest].score) largest = right;
        if (largest == current) break;
        ft8o_candidate_t tmp = heap[largest];
        heap[largest] = heap[current];
        heap[current] = tmp;
        current = largest;
    }
}

static void heapify_up(ft8o_candidate_t heap[], int heap_size) {
    int current = heap_size - 1;
    while (current > 0) {
        int parent = (current - 1) / 2;
        if (heap[current].score >= heap[parent].score) break;
        ft8o_candidate_t tmp = heap[parent];
        heap[parent] = heap[current];
        heap[current] = tmp;
        current = parent;
    }
}

int ft8o_find_sync(const uint8_t *mag, int num_candidates, ft8o_candidate_t heap[], int min_score) {
    int heap_size = 0;
    ft8o_candidate_t c;
    for (c.time_sub = 0; c.time_sub < FT8O_K_TIME_OSR; ++c.time_sub) {
        for (c.freq_sub = 0; c.freq_sub < FT8O_K_FREQ_OSR; ++c.freq_sub) {
            for (c.time_offset = -12; c.time_offset < 24; ++c.time_offset) {
                for (c.freq_offset = 0; (c.freq_offset + 7) < FT8O_NUM_BIN; ++c.freq_offset) {
                    c.score = (int16_t)ft8o_sync_score(mag, &c);
                    if (c.score < min_score) continue;
                    if (heap_size == num_candidates && c.score > heap[0].score) {
                        heap[0] = heap[heap_size - 1];
                        --heap_size;
                        heapify_down(heap, heap_size);
                    }
                    if (heap_size < num_candidates) {
                        heap[heap_size] = c;
                        ++heap_size;
                        heapify_up(heap, heap_size);
                    }
                }
            }
        }
    }
    int len_unsorted = heap_size;
    while (len_unsorted > 1) {
        ft8o_candidate_t tmp = heap[len_unsorted - 1];
        heap[len_unsorted - 1] = heap[0];
        heap[0] = tmp;
        len_unsorted--;
        heapify_down(heap, len_unsorted);
    }
    return heap_size;
}

/* every score of the scan, in scan order [time_sub][freq_sub][time_offset+12][freq_offset] */
void ft8o_score_map(const uint8_t *mag, int16_t *scores) {
    ft8o_candidate_t c;
    int o = 0;
    for (c.time_sub = 0; c.time_sub < 2; ++c.time_sub)
        for (c.freq_sub = 0; c.freq_sub < 2; ++c.freq_sub)
            for (c.time_offset = -12; c.time_offset < 24; ++c.time_offset)
                for (c.freq_offset = 0; (c.freq_offset + 7) < FT8O_NUM_BIN; ++c.freq_offset)
                    scores[o++] = (int16_t)ft8o_sync_score(mag, &c);
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib decode.c: ft8_extract_symbol / ft8_extract_likelihood / ftx_normalize_logl
 * ---------------------------------------------------------------------------------------- */
static float max2(float a, float b) { return (a >= b) ? a : b; }
static float max4(float a, float b, float c, float d) { return max2(max2(a, b), max2(c, d)); }

static void extract_symbol(const uint8_t *wf, float *logl) {
    float s2[8];
    for (int j = 0; j < 8; ++j) s2[j] = (float)wf[kO_Gray[j]];
    logl[0] = max4(s2[4], s2[5], s2[6], s2[7]) - max4(s2[0], s2[1], s2[2], s2[3]);
    logl[1] = max4(s2[2], s2[3], s2[6], s2[7]) - max4(s2[0], s2[1], s2[4], s2[5]);
    logl[2] = max4(s2[1], s2[3], s2[5], s2[7]) - max4(s2[0], s2[2], s2[4], s2[6]);
}

void ft8o_extract_likelihood(const uint8_t *mag, const ft8o_candidate_t *cand, float *log174) {
    const uint8_t *mag_cand = mag + get_index(cand);
    for (int k = 0; k < FT8O_ND; ++k) {
        int sym_idx = k + ((k < 29) ? 7 : 14);
        int bit_idx = 3 * k;
        int block = cand->time_offset + sym_idx;
        if ((block < 0) || (block >= FT8O_NUM_BLOCKS)) {
            log174[bit_idx + 0] = 0;
            log174[bit_idx + 1] = 0;
            log174[bit_idx + 2] = 0;
        } else {
            const uint8_t *ps = mag_cand + (sym_idx * FT8O_BLOCK_STRIDE);
            extract_symbol(ps, log174 + bit_idx);
        }
    }
}

void ft8o_normalize_logl(float *log174) {
    float sum = 0;
    float sum2 = 0;
    for (int i = 0; i < FT8O_LDPC_N; ++i) {
        sum += log174[i];
        sum2 += log174[i] * log174[i];
    }
    float inv_n = 1.0f / FT8O_LDPC_N;
    float variance = (sum2 - (sum * sum * inv_n)) * inv_n;
    float norm_factor = sqrtf(24.0f / variance);
    for (int i = 0; i < FT8O_LDPC_N; ++i) log174[i] *= norm_factor;
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib ldpc.c: fast_tanh / fast_atanh / ldpc_check / bp_decode
 * ---------------------------------------------------------------------------------------- */
static float fast_tanh(float x) {
    if (x < -4.97f) return -1.0f;
    if (x > 4.97f) return 1.0f;
    float x2 = x * x;
    float a = x * (945.0f + x2 * (105.0f + x2));
    float b = 945.0f + x2 * (420.0f + x2 * 15.0f);
    return a / b;
}

static float fast_atanh(float x) {
    float x2 = x * x;
    float a = x * (945.0f + x2 * (-735.0f + x2 * 64.0f));
    float b = (945.0f + x2 * (-1050.0f + x2 * 225.0f));
    return a / b;
}

int ft8o_ldpc_check(const uint8_t *codeword) {
    int errors = 0;
    for (int m = 0; m < FT8O_LDPC_M; ++m) {
        uint8_t x = 0;
        for (int i = 0; i < kO_Num_rows[m]; ++i) x ^= codeword[kO_Nm[m][i] - 1];
        if (x != 0) ++errors;
    }
    return errors;
}

void ft8o_bp_decode(const float *codeword, int max_iters, uint8_t *plain, int *ok, int *iters_out) {
    float tov[FT8O_LDPC_N][3];
    float toc[FT8O_LDPC_M][7];
    int min_errors = FT8O_LDPC_M;
    int iter;

    for (int n = 0; n < FT8O_LDPC_N; ++n) tov[n][0] = tov[n][1] = tov[n][2] = 0;

    for (iter = 0; iter < max_iters; ++iter) {
        int plain_sum = 0;
        for (int n = 0; n < FT8O_LDPC_N; ++n) {
            plain[n] = ((codeword[n] + tov[n][0] + tov[n][1] + tov[n][2]) > 0) ? 1 : 0;
            plain_sum += plain[n];
        }
        if (plain_sum == 0) break;                 /* converged to all-zeros, which is prohibited */

        int errors = ft8o_ldpc_check(plain);
        if (errors < min_errors) {
            min_errors = errors;
            if (errors == 0) break;
        }

        for (int m = 0; m < FT8O_LDPC_M; ++m) {
            for (int n_idx = 0; n_idx < kO_Num_rows[m]; ++n_idx) {
                int n = kO_Nm[m][n_idx] - 1;
                float Tnm = codeword[n];
                for (int m_idx = 0; m_idx < 3; ++m_idx)
                    if ((kO_Mn[n][m_idx] - 1) != m) Tnm += tov[n][m_idx];
                toc[m][n_idx] = fast_tanh(-Tnm / 2);
            }
        }
        for (int n = 0; n < FT8O_LDPC_N; ++n) {
            for (int m_idx = 0; m_idx < 3; ++m_idx) {
                int m = kO_Mn[n][m_idx] - 1;
                float Tmn = 1.0f;
                for (int n_idx = 0; n_idx < kO_Num_rows[m]; ++n_idx)
                    if ((kO_Nm[m][n_idx] - 1) != n) Tmn *= toc[m][n_idx];
                tov[n][m_idx] = -2 * fast_atanh(Tmn);
            }
        }
    }
    *ok = min_errors;
    if (iters_out) *iters_out = iter;
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib crc.c  (CRC-14, polynomial 0x2757, MSB first, init 0)
 * ---------------------------------------------------------------------------------------- */
#define CRC_WIDTH 14
#define CRC_TOPBIT (1u << (CRC_WIDTH - 1))
#define CRC_POLY 0x2757u

uint16_t ft8o_compute_crc(const uint8_t *message, int num_bits) {
    uint16_t remainder = 0;
    int idx_byte = 0;
    for (int idx_bit = 0; idx_bit < num_bits; ++idx_bit) {
        if (idx_bit % 8 == 0) {
            remainder ^= (uint16_t)(message[idx_byte] << (CRC_WIDTH - 8));
            ++idx_byte;
        }
        if (remainder & CRC_TOPBIT) remainder = (uint16_t)((remainder << 1) ^ CRC_POLY);
        else remainder = (uint16_t)(remainder << 1);
    }
    return remainder & ((CRC_TOPBIT << 1) - 1u);
}

uint16_t ft8o_extract_crc(const uint8_t *a91) {
    return (uint16_t)(((a91[9] & 0x07) << 11) | (a91[10] << 3) | (a91[11] >> 5));
}

void ft8o_add_crc(const uint8_t *payload, uint8_t *a91) {
    for (int i = 0; i < 10; i++) a91[i] = payload[i];
    a91[9] &= 0xF8u;
    a91[10] = 0;
    a91[11] = 0;
    uint16_t checksum = ft8o_compute_crc(a91, 96 - 14);
    a91[9] |= (uint8_t)(checksum >> 11);
    a91[10] = (uint8_t)(checksum >> 3);
    a91[11] = (uint8_t)(checksum << 5);
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib text.c / unpack.c
 * ---------------------------------------------------------------------------------------- */
static const char *trim_front(const char *str) {
    while (*str == ' ') str++;
    return str;
}
static void trim_back(char *str) {
    int idx = (int)strlen(str) - 1;
    while (idx >= 0 && str[idx] == ' ') str[idx--] = '\0';
}
static char *trim(char *str) {
    str = (char *)trim_front(str);
    trim_back(str);
    return str;
}

/* table 0: " 0-9A-Z+-./?"  1: " 0-9A-Z"  2: "0-9A-Z"  3: "0-9"  4: " A-Z"  5: " 0-9A-Z/" */
static char charn(int c, int table_idx) {
    if (table_idx != 2 && table_idx != 3) {
        if (c == 0) return ' ';
        c -= 1;
    }
    if (table_idx != 4) {
        if (c < 10) return (char)('0' + c);
        c -= 10;
    }
    if (table_idx != 3) {
        if (c < 26) return (char)('A' + c);
        c -= 26;
    }
    if (table_idx == 0) {
        if (c < 5) return "+-./?"[c];
    } else if (table_idx == 5) {
        if (c == 0) return '/';
    }
    return '_';
}

static int nchar(char c, int table_idx) {
    int n = 0;
    if (table_idx != 2 && table_idx != 3) {
        if (c == ' ') return n + 0;
        n += 1;
    }
    if (table_idx != 4) {
        if (c >= '0' && c <= '9') return n + (c - '0');
        n += 10;
    }
    if (table_idx != 3) {
        if (c >= 'A' && c <= 'Z') return n + (c - 'A');
        n += 26;
    }
    if (table_idx == 0) {
        if (c == '+') return n + 0;
        if (c == '-') return n + 1;
        if (c == '.') return n + 2;
        if (c == '/') return n + 3;
        if (c == '?') return n + 4;
    } else if (table_idx == 5) {
        if (c == '/') return n + 0;
    }
    return -1;
}

static void int_to_dd(char *str, int value, int width, int full_sign) {
    if (value < 0) { *str = '-'; ++str; value = -value; }
    else if (full_sign) { *str = '+'; ++str; }
    int divisor = 1;
    for (int i = 0; i < width - 1; ++i) divisor *= 10;
    while (divisor >= 1) {
        int digit = value / divisor;
        *str = (char)('0' + digit);
        ++str;
        value -= digit * divisor;
        divisor /= 10;
    }
    *str = 0;
}

#define NTOKENS  2063592u
#define MAX22    4194304u
#define MAXGRID4 32400u

static int unpack_callsign(uint32_t n28, uint8_t ip, uint8_t i3, char *result) {
    if (n28 < NTOKENS) {
        if (n28 <= 2) {
            if (n28 == 0) strcpy(result, "DE");
            if (n28 == 1) strcpy(result, "QRZ");
            if (n28 == 2) strcpy(result, "CQ");
            return 0;
        }
        if (n28 <= 1002) {
            strcpy(result, "CQ ");
            int_to_dd(result + 3, (int)n28 - 3, 3, 0);
            return 0;
        }
        if (n28 <= 532443u) {
            uint32_t n = n28 - 1003;
            char aaaa[5];
            aaaa[4] = '\0';
            for (int i = 3; /* */; --i) {
                aaaa[i] = charn((int)(n % 27), 4);
                if (i == 0) break;
                n /= 27;
            }
            strcpy(result, "CQ ");
            strcat(result, trim_front(aaaa));
            return 0;
        }
        return -1;
    }
    n28 = n28 - NTOKENS;
    if (n28 < MAX22) {
        strcpy(result, "<...>");      /* 22-bit hashed callsign: no hash table in this era */
        return 0;
    }
    uint32_t n = n28 - MAX22;
    char callsign[7];
    callsign[6] = '\0';
    callsign[5] = charn((int)(n % 27), 4); n /= 27;
    callsign[4] = charn((int)(n % 27), 4); n /= 27;
    callsign[3] = charn((int)(n % 27), 4); n /= 27;
    callsign[2] = charn((int)(n % 10), 3); n /= 10;
    callsign[1] = charn((int)(n % 36), 2); n /= 36;
    callsign[0] = charn((int)(n % 37), 1);
    strcpy(result, trim(callsign));
    if (strlen(result) == 0) return -1;
    if (ip) {
        if (i3 == 1) strcat(result, "/R");
        else if (i3 == 2) strcat(result, "/P");
    }
    return 0;
}

static int unpack_type1(const uint8_t *a77, uint8_t i3, char *call_to, char *call_de, char *extra) {
    uint32_t n28a, n28b;
    uint16_t igrid4;
    uint8_t ir;
    n28a  = ((uint32_t)a77[0] << 21);
    n28a |= ((uint32_t)a77[1] << 13);
    n28a |= ((uint32_t)a77[2] << 5);
    n28a |= ((uint32_t)a77[3] >> 3);
    n28b  = ((uint32_t)(a77[3] & 0x07) << 26);
    n28b |= ((uint32_t)a77[4] << 18);
    n28b |= ((uint32_t)a77[5] << 10);
    n28b |= ((uint32_t)a77[6] << 2);
    n28b |= ((uint32_t)a77[7] >> 6);
    ir = ((a77[7] & 0x20) >> 5);
    igrid4  = (uint16_t)((a77[7] & 0x1F) << 10);
    igrid4 |= (uint16_t)(a77[8] << 2);
    igrid4 |= (uint16_t)(a77[9] >> 6);

    if (unpack_callsign(n28a >> 1, n28a & 0x01, i3, call_to) < 0) return -1;
    if (unpack_callsign(n28b >> 1, n28b & 0x01, i3, call_de) < 0) return -2;

    char *dst = extra;
    if (igrid4 <= MAXGRID4) {
        if (ir > 0) dst = stpcpy(dst, "R ");
        uint16_t n = igrid4;
        dst[4] = '\0';
        dst[3] = (char)('0' + (n % 10)); n /= 10;
        dst[2] = (char)('0' + (n % 10)); n /= 10;
        dst[1] = (char)('A' + (n % 18)); n /= 18;
        dst[0] = (char)('A' + (n % 18));
    } else {
        int irpt = igrid4 - MAXGRID4;
        switch (irpt) {
        case 1: extra[0] = '\0'; break;
        case 2: strcpy(dst, "RRR"); break;
        case 3: strcpy(dst, "RR73"); break;
        case 4: strcpy(dst, "73"); break;
        default:
            if (ir > 0) *dst++ = 'R';
            int_to_dd(dst, irpt - 35, 2, 1);
            break;
        }
    }
    return 0;
}

static int unpack_text(const uint8_t *a71, char *text) {
    uint8_t b71[9];
    uint8_t carry = 0;
    for (int i = 0; i < 9; ++i) {
        b71[i] = carry | (a71[i] >> 1);
        carry = (a71[i] & 1) ? 0x80 : 0;
    }
    char c14[14];
    c14[13] = 0;
    for (int idx = 12; idx >= 0; --idx) {
        uint16_t rem = 0;
        for (int i = 0; i < 9; ++i) {
            rem = (uint16_t)((rem << 8) | b71[i]);
            b71[i] = (uint8_t)(rem / 42);
            rem = rem % 42;
        }
        c14[idx] = charn(rem, 0);
    }
    strcpy(text, trim(c14));
    return 0;
}

static int unpack_telemetry(const uint8_t *a71, char *telemetry) {
    uint8_t b71[9];
    uint8_t carry = 0;
    for (int i = 0; i < 9; ++i) {
        b71[i] = (uint8_t)((carry << 7) | (a71[i] >> 1));
        carry = (a71[i] & 0x01);
    }
    for (int i = 0; i < 9; ++i) {
        uint8_t nibble1 = (b71[i] >> 4);
        uint8_t nibble2 = (b71[i] & 0x0F);
        char c1 = (char)((nibble1 > 9) ? (nibble1 - 10 + 'A') : nibble1 + '0');
        char c2 = (char)((nibble2 > 9) ? (nibble2 - 10 + 'A') : nibble2 + '0');
        telemetry[i * 2] = c1;
        telemetry[i * 2 + 1] = c2;
    }
    telemetry[18] = '\0';
    return 0;
}

static int unpack_nonstandard(const uint8_t *a77, char *call_to, char *call_de, char *extra) {
    uint32_t n12, iflip, nrpt, icq;
    uint64_t n58;
    n12 = ((uint32_t)a77[0] << 4);
    n12 |= (a77[1] >> 4);
    (void)n12;
    n58  = ((uint64_t)(a77[1] & 0x0F) << 54);
    n58 |= ((uint64_t)a77[2] << 46);
    n58 |= ((uint64_t)a77[3] << 38);
    n58 |= ((uint64_t)a77[4] << 30);
    n58 |= ((uint64_t)a77[5] << 22);
    n58 |= ((uint64_t)a77[6] << 14);
    n58 |= ((uint64_t)a77[7] << 6);
    n58 |= ((uint64_t)a77[8] >> 2);
    iflip = (a77[8] >> 1) & 0x01;
    nrpt = ((a77[8] & 0x01) << 1);
    nrpt |= (a77[9] >> 7);
    icq = ((a77[9] >> 6) & 0x01);

    char c11[12];
    c11[11] = '\0';
    for (int i = 10; /* */; --i) {
        c11[i] = charn((int)(n58 % 38), 5);
        if (i == 0) break;
        n58 /= 38;
    }
    char call_3[15];
    strcpy(call_3, "<...>");          /* 12-bit hashed callsign: no hash table in this era */
    char *call_1 = (iflip) ? c11 : call_3;
    char *call_2 = (iflip) ? call_3 : c11;

    if (icq == 0) {
        strcpy(call_to, trim(call_1));
        if (nrpt == 1) strcpy(extra, "RRR");
        else if (nrpt == 2) strcpy(extra, "RR73");
        else if (nrpt == 3) strcpy(extra, "73");
        else extra[0] = '\0';
    } else {
        strcpy(call_to, "CQ");
        extra[0] = '\0';
    }
    strcpy(call_de, trim(call_2));
    return 0;
}

static int unpack77_fields(const uint8_t *a77, char *call_to, char *call_de, char *extra) {
    call_to[0] = call_de[0] = extra[0] = '\0';
    uint8_t i3 = (a77[9] >> 3) & 0x07;
    if (i3 == 0) {
        uint8_t n3 = (uint8_t)(((a77[8] << 2) & 0x04) | ((a77[9] >> 6) & 0x03));
        if (n3 == 0) return unpack_text(a77, extra);
        else if (n3 == 5) return unpack_telemetry(a77, extra);
    } else if (i3 == 1 || i3 == 2) {
        return unpack_type1(a77, i3, call_to, call_de, extra);
    } else if (i3 == 4) {
        return unpack_nonstandard(a77, call_to, call_de, extra);
    }
    return -1;
}

int ft8o_unpack77(const uint8_t *a77, char *message) {
    char call_to[14];
    char call_de[14];
    char extra[19];
    int rc = unpack77_fields(a77, call_to, call_de, extra);
    if (rc < 0) return rc;
    char *dst = message;
    dst[0] = '\0';
    if (call_to[0] != '\0') { dst = stpcpy(dst, call_to); *dst++ = ' '; }
    if (call_de[0] != '\0') { dst = stpcpy(dst, call_de); *dst++ = ' '; }
    dst = stpcpy(dst, extra);
    *dst = '\0';
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib pack.c (standard type-1 messages "CALL1 CALL2 GRID4|report" only: what
 * decoderSelfTest needs, rtlsdr_ft8d.c:924-927) and encode.c (rtlsdr_ft8d.c:934)
 * ---------------------------------------------------------------------------------------- */
static int is_digit(char c) { return c >= '0' && c <= '9'; }
static int is_letter(char c) { return c >= 'A' && c <= 'Z'; }
static int starts_with(const char *s, const char *p) { return strncmp(s, p, strlen(p)) == 0; }

static int32_t pack28(const char *callsign) {
    if (starts_with(callsign, "DE ")) return 0;
    if (starts_with(callsign, "QRZ ")) return 1;
    if (starts_with(callsign, "CQ ")) return 2;
    if (starts_with(callsign, "CQ_")) {
        int nnum = 0, nlet = 0;
        (void)nnum; (void)nlet;
        return -1;                              /* CQ_nnn / CQ_aaaa: not needed by the tooling */
    }
    char c6[6] = { ' ', ' ', ' ', ' ', ' ', ' ' };
    int length = 0;
    while (callsign[length] != ' ' && callsign[length] != 0) length++;
    if (starts_with(callsign, "3DA0") && length <= 7) {
        memcpy(c6, "3D0", 3);
        memcpy(c6 + 3, callsign + 4, (size_t)(length - 4));
    } else if (starts_with(callsign, "3X") && is_letter(callsign[2]) && length <= 7) {
        memcpy(c6, "Q", 1);
        memcpy(c6 + 1, callsign + 2, (size_t)(length - 2));
    } else {
        if (length >= 3 && is_digit(callsign[2]) && length <= 6) memcpy(c6, callsign, (size_t)length);
        else if (length >= 2 && is_digit(callsign[1]) && length <= 5) memcpy(c6 + 1, callsign, (size_t)length);
    }
    int i0, i1, i2, i3, i4, i5;
    if ((i0 = nchar(c6[0], 1)) >= 0 && (i1 = nchar(c6[1], 2)) >= 0 && (i2 = nchar(c6[2], 3)) >= 0 &&
        (i3 = nchar(c6[3], 4)) >= 0 && (i4 = nchar(c6[4], 4)) >= 0 && (i5 = nchar(c6[5], 4)) >= 0) {
        int32_t n28 = i0;
        n28 = n28 * 36 + i1;
        n28 = n28 * 10 + i2;
        n28 = n28 * 27 + i3;
        n28 = n28 * 27 + i4;
        n28 = n28 * 27 + i5;
        return (int32_t)(NTOKENS + MAX22) + n28;
    }
    return -1;
}

static int dd_to_int(const char *str, int length) {
    int result = 0, negative, i;
    if (str[0] == '-') { negative = 1; i = 1; }
    else { negative = 0; i = (str[0] == '+') ? 1 : 0; }
    while (i < length) {
        if (str[i] == 0) break;
        if (!is_digit(str[i])) break;
        result *= 10;
        result += (str[i] - '0');
        ++i;
    }
    return negative ? -result : result;
}

static uint16_t packgrid(const char *grid4) {
    if (grid4 == 0) return (uint16_t)(MAXGRID4 + 1);
    if (strcmp(grid4, "RRR") == 0) return (uint16_t)(MAXGRID4 + 2);
    if (strcmp(grid4, "RR73") == 0) return (uint16_t)(MAXGRID4 + 3);
    if (strcmp(grid4, "73") == 0) return (uint16_t)(MAXGRID4 + 4);
    /* only the first four locator characters are used ("FN20QI" -> FN20, rtlsdr_ft8d.c:920-921) */
    if (grid4[0] >= 'A' && grid4[0] <= 'R' && grid4[1] >= 'A' && grid4[1] <= 'R' &&
        is_digit(grid4[2]) && is_digit(grid4[3])) {
        uint16_t igrid4 = (uint16_t)(grid4[0] - 'A');
        igrid4 = (uint16_t)(igrid4 * 18 + (grid4[1] - 'A'));
        igrid4 = (uint16_t)(igrid4 * 10 + (grid4[2] - '0'));
        igrid4 = (uint16_t)(igrid4 * 10 + (grid4[3] - '0'));
        return igrid4;
    }
    if (grid4[0] == 'R') {
        int dd = dd_to_int(grid4 + 1, 3);
        uint16_t irpt = (uint16_t)(35 + dd);
        return (uint16_t)((MAXGRID4 + irpt) | 0x8000);
    } else {
        int dd = dd_to_int(grid4, 3);
        uint16_t irpt = (uint16_t)(35 + dd);
        return (uint16_t)(MAXGRID4 + irpt);
    }
}

int ft8o_pack77(const char *msg, uint8_t *b77) {
    /* type 1: "<call_to> <call_de> [<grid|report>]" */
    const char *s1 = strchr(msg, ' ');
    if (s1 == 0) return -1;
    const char *call1 = msg;
    const char *call2 = s1 + 1;
    int32_t n28a = pack28(call1);
    int32_t n28b = pack28(call2);
    if (n28a < 0 || n28b < 0) return -1;
    uint16_t igrid4;
    const char *s2 = strchr(s1 + 1, ' ');
    if (s2 != 0) igrid4 = packgrid(s2 + 1);
    else igrid4 = packgrid(0);
    uint8_t i3 = 1;
    n28a <<= 1;                        /* ipa = 0 */
    n28b <<= 1;                        /* ipb = 0 */
    b77[0] = (uint8_t)(n28a >> 21);
    b77[1] = (uint8_t)(n28a >> 13);
    b77[2] = (uint8_t)(n28a >> 5);
    b77[3] = (uint8_t)((uint8_t)((uint32_t)n28a << 3) | (uint8_t)(n28b >> 26));
    b77[4] = (uint8_t)(n28b >> 18);
    b77[5] = (uint8_t)(n28b >> 10);
    b77[6] = (uint8_t)(n28b >> 2);
    b77[7] = (uint8_t)((uint8_t)((uint32_t)n28b << 6) | (uint8_t)(igrid4 >> 10));
    b77[8] = (uint8_t)(igrid4 >> 2);
    b77[9] = (uint8_t)((uint8_t)(igrid4 << 6) | (uint8_t)(i3 << 3));
    b77[10] = 0;
    b77[11] = 0;
    return 0;
}

static uint8_t parity8(uint8_t x) {
    x ^= x >> 4;
    x ^= x >> 2;
    x ^= x >> 1;
    return x % 2;
}

static void encode174(const uint8_t *message, uint8_t *codeword) {
    for (int j = 0; j < 22; ++j) codeword[j] = (j < FT8O_LDPC_K_BYTES) ? message[j] : 0;
    uint8_t col_mask = (uint8_t)(0x80u >> (FT8O_LDPC_K % 8u));
    uint8_t col_idx = FT8O_LDPC_K_BYTES - 1;
    for (int i = 0; i < FT8O_LDPC_M; ++i) {
        uint8_t nsum = 0;
        for (int j = 0; j < FT8O_LDPC_K_BYTES; ++j) {
            uint8_t bits = message[j] & kO_generator[i][j];
            nsum ^= parity8(bits);
        }
        if (nsum % 2) codeword[col_idx] |= col_mask;
        col_mask >>= 1;
        if (col_mask == 0) { col_mask = 0x80u; ++col_idx; }
    }
}

void ft8o_encode(const uint8_t *payload, uint8_t *tones) {
    uint8_t a91[FT8O_LDPC_K_BYTES];
    ft8o_add_crc(payload, a91);
    uint8_t codeword[22];
    encode174(a91, codeword);
    uint8_t mask = 0x80u;
    int i_byte = 0;
    for (int i_tone = 0; i_tone < FT8O_NN; ++i_tone) {
        if ((i_tone >= 0) && (i_tone < 7)) tones[i_tone] = kO_Costas[i_tone];
        else if ((i_tone >= 36) && (i_tone < 43)) tones[i_tone] = kO_Costas[i_tone - 36];
        else if ((i_tone >= 72) && (i_tone < 79)) tones[i_tone] = kO_Costas[i_tone - 72];
        else {
            uint8_t bits3 = 0;
            if (codeword[i_byte] & mask) bits3 |= 4;
            if (0 == (mask >>= 1)) { mask = 0x80u; i_byte++; }
            if (codeword[i_byte] & mask) bits3 |= 2;
            if (0 == (mask >>= 1)) { mask = 0x80u; i_byte++; }
            if (codeword[i_byte] & mask) bits3 |= 1;
            if (0 == (mask >>= 1)) { mask = 0x80u; i_byte++; }
            tones[i_tone] = kO_Gray[bits3];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * ft8_lib decode.c: pack_bits / ft8_decode   (call site rtlsdr_ft8d.c:1476)
 * ---------------------------------------------------------------------------------------- */
static void pack_bits(const uint8_t bit_array[], int num_bits, uint8_t packed[]) {
    int num_bytes = (num_bits + 7) / 8;
    for (int i = 0; i < num_bytes; ++i) packed[i] = 0;
    uint8_t mask = 0x80;
    int byte_idx = 0;
    for (int i = 0; i < num_bits; ++i) {
        if (bit_array[i]) packed[byte_idx] |= mask;
        mask >>= 1;
        if (!mask) { mask = 0x80; ++byte_idx; }
    }
}

int ft8o_decode(const uint8_t *mag, const ft8o_candidate_t *cand, ft8o_message_t *message,
                int max_iterations, ft8o_decode_status_t *status, ft8o_decode_extra_t *extra) {
    float log174[FT8O_LDPC_N];
    ft8o_extract_likelihood(mag, cand, log174);
    ft8o_normalize_logl(log174);

    uint8_t plain174[FT8O_LDPC_N];
    int iters = 0;
    ft8o_bp_decode(log174, max_iterations, plain174, &status->ldpc_errors, &iters);
    uint8_t a91[FT8O_LDPC_K_BYTES];
    pack_bits(plain174, FT8O_LDPC_K, a91);
    if (extra) { extra->iters = iters; memcpy(extra->a91, a91, 12); }
    if (status->ldpc_errors > 0) return 0;

    status->crc_extracted = ft8o_extract_crc(a91);
    a91[9] &= 0xF8;
    a91[10] &= 0x00;
    status->crc_calculated = ft8o_compute_crc(a91, 96 - 14);
    if (status->crc_extracted != status->crc_calculated) return 0;

    status->unpack_status = ft8o_unpack77(a91, message->text);
    if (status->unpack_status < 0) return 0;

    message->hash = status->crc_extracted;
    return 1;
}

/* ft8_find_sync (rtlsdr_ft8d.c:1450) for B waterfalls: candidate lists [B][cap] (zero behind each frame's count) and counts */
void ft8o_find_sync_batch(const uint8_t *mag, int B, int cap, int min_score, ft8o_candidate_t *cands, int32_t *counts, int nthreads) {
    ft8o_init();
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++) {
        ft8o_candidate_t *c = cands + (size_t)f * cap;
        memset(c, 0, sizeof(ft8o_candidate_t) * (size_t)cap);
        counts[f] = ft8o_find_sync(mag + (size_t)f * FT8O_MAG_ARRAY, cap, c, min_score);
    }
    (void)nthreads;
}

/* ft8_decode() for every candidate of B frames, as 48-byte records in the canonical form the product's stage entry
 * ft8gpu_decode_candidates writes (include/ft8gpu.h: ft8gpu_decode_status): a field is zero unless ft8_lib's ft8_decode
 * would have set it -- CRCs only when ldpc_errors == 0, unpack_status / ok only when the CRCs match, text only when ok.
 * Records beyond counts[f] are zero.  Lets a test compare EVERY candidate of a batch -- also the messages that are not
 * CQ calls, whose text never shows in the spot records -- byte for byte (OpenMP over frames). */
void ft8o_decode_candidates_batch(const uint8_t *mag, const ft8o_candidate_t *cands, const int32_t *counts, int B, int cap,
                                  int max_iterations, uint8_t *records /* [B][cap][48] */, int nthreads) {
    ft8o_init();
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++) {
        uint8_t *rec = records + (size_t)f * cap * 48;
        memset(rec, 0, (size_t)cap * 48);
        const int n = counts[f] < cap ? counts[f] : cap;
        for (int c = 0; c < n; c++, rec += 48) {
            ft8o_message_t m;
            ft8o_decode_status_t st;
            ft8o_decode_extra_t ex;
            memset(&m, 0, sizeof m);
            memset(&st, 0, sizeof st);
            const int ok = ft8o_decode(mag + (size_t)f * FT8O_MAG_ARRAY, &cands[(size_t)f * cap + c], &m, max_iterations, &st, &ex);
            const int16_t e16 = (int16_t)st.ldpc_errors, i16 = (int16_t)ex.iters;
            memcpy(rec + 0, &e16, 2);
            memcpy(rec + 2, &i16, 2);
            memcpy(rec + 10, ex.a91, 12);
            if (st.ldpc_errors == 0) {
                memcpy(rec + 4, &st.crc_extracted, 2);
                memcpy(rec + 6, &st.crc_calculated, 2);
                if (st.crc_extracted == st.crc_calculated) {
                    rec[8] = (uint8_t)(int8_t)st.unpack_status;
                    rec[9] = ok ? 1 : 0;
                    if (ok) memcpy(rec + 22, m.text, strnlen(m.text, 25));
                }
            }
        }
    }
    (void)nthreads;
}

/* ------------------------------------------------------------------------------------------
 * ft8_subsystem()  rtlsdr_ft8d.c:1387-1524
 *
 * Deliberate fences around reference behaviour that is undefined or non-terminating
 * (SURVEY.md Appendix C), applied identically by the product:
 *   Q7  >50 unique messages: the reference's probe loop :1490-1502 never terminates once the
 *       table is full; here a new message that finds the table full is dropped.
 *   Q8  strtok() returning NULL for a missing 2nd/3rd token: glibc's snprintf prints "(null)"
 *       for a NULL %s argument (precision >= 6); reproduced literally.  A message whose text has
 *       no token at all (empty free text) would crash the reference at :1510; it is counted as a
 *       non-CQ message here.
 * ---------------------------------------------------------------------------------------- */
/* the candidate loop :1452-1523 for a given candidate list (what follows ft8_find_sync at :1450) */
void ft8o_spots_from_candidates(const uint8_t *mag_power, const ft8o_candidate_t *candidate_list, int num_candidates,
                                const ft8o_params_t *p, struct ft8o_decoder_results *decodes, int32_t *n_results) {
    int num_decoded = 0;
    ft8o_message_t decoded[FT8O_K_MAX_MESSAGES];
    ft8o_message_t *decoded_hashtable[FT8O_K_MAX_MESSAGES];
    for (int i = 0; i < FT8O_K_MAX_MESSAGES; ++i) decoded_hashtable[i] = NULL;                      /* :1458 */

    for (int idx = 0; idx < num_candidates; ++idx) {                                                 /* :1465 */
        const ft8o_candidate_t *cand = &candidate_list[idx];
        if (cand->score < p->min_score) continue;                                                    /* :1467 */
        float freq_hz = (cand->freq_offset + (float)cand->freq_sub / FT8O_K_FREQ_OSR) * FT8O_K_FSK_DEV; /* :1470 */

        ft8o_message_t message;
        ft8o_decode_status_t status;
        memset(&message, 0, sizeof message);
        memset(&status, 0, sizeof status);
        if (!ft8o_decode(mag_power, cand, &message, p->ldpc_iters, &status, NULL)) continue;        /* :1476-1485 */

        int idx_hash = message.hash % FT8O_K_MAX_MESSAGES;                                           /* :1487 */
        int found_empty_slot = 0, found_duplicate = 0, probes = 0;
        do {
            if (decoded_hashtable[idx_hash] == NULL) {
                found_empty_slot = 1;
            } else if ((decoded_hashtable[idx_hash]->hash == message.hash) &&
                       (0 == strcmp(decoded_hashtable[idx_hash]->text, message.text))) {
                found_duplicate = 1;
            } else {
                idx_hash = (idx_hash + 1) % FT8O_K_MAX_MESSAGES;
                if (++probes >= FT8O_K_MAX_MESSAGES) break;          /* fence Q7: table full */
            }
        } while (!found_empty_slot && !found_duplicate);

        if (found_empty_slot) {                                                                      /* :1505 */
            memcpy(&decoded[idx_hash], &message, sizeof(message));
            decoded_hashtable[idx_hash] = &decoded[idx_hash];

            char *save = NULL;
            char *strPtr = strtok_r(message.text, " ", &save);                                       /* :1509 */
            if (strPtr != NULL && !strncmp(strPtr, "CQ", 2)) {                                       /* :1510 */
                strPtr = strtok_r(NULL, " ", &save);
                snprintf(decodes[num_decoded].call, sizeof(decodes[num_decoded].call), "%.12s",
                         strPtr ? strPtr : "(null)");                                                /* :1512, fence Q8 */
                strPtr = strtok_r(NULL, " ", &save);
                snprintf(decodes[num_decoded].loc, sizeof(decodes[num_decoded].loc), "%.6s",
                         strPtr ? strPtr : "(null)");                                                /* :1514 */
                decodes[num_decoded].freq = (int32_t)freq_hz;                                        /* :1516 */
                decodes[num_decoded].snr = (int32_t)cand->score;                                     /* :1517 */
            }
            num_decoded++;                                                                           /* :1520 */
        }
    }
    *n_results = num_decoded;                                                                        /* :1523 */
}

static void spots_from_waterfall(const uint8_t *mag_power, const ft8o_params_t *p,
                                 struct ft8o_decoder_results *decodes, int32_t *n_results) {
    ft8o_candidate_t *candidate_list = (ft8o_candidate_t *)malloc(sizeof(ft8o_candidate_t) * (size_t)p->max_candidates);
    int num_candidates = ft8o_find_sync(mag_power, p->max_candidates, candidate_list, p->min_score);   /* :1450 */
    ft8o_spots_from_candidates(mag_power, candidate_list, num_candidates, p, decodes, n_results);
    free(candidate_list);
}

void ft8o_subsystem_from_waterfall(const uint8_t *mag, const ft8o_params_t *p,
                                   struct ft8o_decoder_results *decodes, int32_t *n_results) {
    spots_from_waterfall(mag, p, decodes, n_results);
}

void ft8o_subsystem_ex(const float *iSamples, const float *qSamples, const ft8o_params_t *p,
                       struct ft8o_decoder_results *decodes, int32_t *n_results) {
    uint8_t *mag_power = (uint8_t *)malloc(FT8O_MAG_ARRAY);
    ft8o_waterfall(iSamples, qSamples, mag_power);                                                   /* :1395-1435 */
    spots_from_waterfall(mag_power, p, decodes, n_results);
    free(mag_power);
}

void ft8o_subsystem(const float *iSamples, const float *qSamples, uint32_t samples_len,
                    struct ft8o_decoder_results *decodes, int32_t *n_results) {
    (void)samples_len;                                               /* ignored by the reference, :1393 */
    ft8o_params_t p = { FT8O_K_MIN_SCORE, FT8O_K_MAX_CANDIDATES, FT8O_K_LDPC_ITERS };
    ft8o_subsystem_ex(iSamples, qSamples, &p, decodes, n_results);
}

void ft8o_subsystem_batch(const float *iq, int B, const ft8o_params_t *p,
                          struct ft8o_decoder_results *decodes, int32_t *n_results, int nthreads) {
    ft8o_init();
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++) {
        const float *I = iq + (size_t)f * 2 * FT8O_NSAMPLES;
        ft8o_subsystem_ex(I, I + FT8O_NSAMPLES, p, decodes + (size_t)f * FT8O_K_MAX_MESSAGES, n_results + f);
    }
    (void)nthreads;
}

/* ft8o_subsystem_batch with the reference's own FFT (fftw3f bound at run time): the CPU baseline "reference-fft" of
 * bench.py.  Returns -1 without touching the outputs when FFTW is not bound. */
int ft8o_subsystem_batch_fftw(const float *iq, int B, const ft8o_params_t *p,
                              struct ft8o_decoder_results *decodes, int32_t *n_results, int nthreads) {
    ft8o_init();
    if (!g_fftw.ok) return -1;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++) {
        const float *I = iq + (size_t)f * 2 * FT8O_NSAMPLES;
        uint8_t *mag_power = (uint8_t *)malloc(FT8O_MAG_ARRAY);
        if (mag_power && ft8o_waterfall_fftw(I, I + FT8O_NSAMPLES, mag_power) == 0)
            spots_from_waterfall(mag_power, p, decodes + (size_t)f * FT8O_K_MAX_MESSAGES, n_results + f);
        else n_results[f] = -1;
        free(mag_power);
    }
    (void)nthreads;
    return 0;
}

/* Batch forms of the two halves of the path, for the FFT-divergence study (tools/fft_parity.py) and the
 * configs[1] bench leg (GPU waterfall + sync, LDPC on the host cores): the waterfall of B frames with the
 * float32 R4DIF FFT or the float64 DFT, and everything after the waterfall (rtlsdr_ft8d.c:1438-1523). */
void ft8o_waterfall_batch(const float *iq, int B, uint8_t *mag, int f64, int nthreads) {
    ft8o_init();
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++) {
        const float *I = iq + (size_t)f * 2 * FT8O_NSAMPLES;
        if (f64 == 2) (void)ft8o_waterfall_fftw(I, I + FT8O_NSAMPLES, mag + (size_t)f * FT8O_MAG_ARRAY);      /* the reference's FFTW (when bound) */
        else if (f64) ft8o_waterfall_f64(I, I + FT8O_NSAMPLES, mag + (size_t)f * FT8O_MAG_ARRAY);
        else ft8o_waterfall(I, I + FT8O_NSAMPLES, mag + (size_t)f * FT8O_MAG_ARRAY);
    }
    (void)nthreads;
}

void ft8o_subsystem_from_waterfall_batch(const uint8_t *mag, int B, const ft8o_params_t *p,
                                         struct ft8o_decoder_results *decodes, int32_t *n_results, int nthreads) {
    ft8o_init();
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++)
        ft8o_subsystem_from_waterfall(mag + (size_t)f * FT8O_MAG_ARRAY, p, decodes + (size_t)f * FT8O_K_MAX_MESSAGES, n_results + f);
    (void)nthreads;
}

/* ft8_find_sync + the candidate loop's ft8_decode calls for B frames whose candidate lists are given (the
 * configs[1] split: the GPU hands over waterfall and candidates, the host runs LLR / BP / CRC / unpack / dedup). */
void ft8o_decode_from_candidates_batch(const uint8_t *mag, const ft8o_candidate_t *cands, const int32_t *counts, int B,
                                       const ft8o_params_t *p, struct ft8o_decoder_results *decodes, int32_t *n_results,
                                       int nthreads) {
    ft8o_init();
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int f = 0; f < B; f++)
        ft8o_spots_from_candidates(mag + (size_t)f * FT8O_MAG_ARRAY, cands + (size_t)f * p->max_candidates, counts[f], p,
                                   decodes + (size_t)f * FT8O_K_MAX_MESSAGES, n_results + f);
    (void)nthreads;
}

/* ------------------------------------------------------------------------------------------
 * decoderSelfTest() signal  rtlsdr_ft8d.c:890-955
 * ---------------------------------------------------------------------------------------- */
static double wgn_V1, wgn_V2, wgn_S;
static int wgn_phase = 0;
static float whiteGaussianNoise(float factor) {                      /* :890-910 */
    double U1, U2, X;
    if (wgn_phase == 0) {
        do {
            U1 = rand() / (double)RAND_MAX;
            U2 = rand() / (double)RAND_MAX;
            wgn_V1 = 2 * U1 - 1;
            wgn_V2 = 2 * U2 - 1;
            wgn_S = wgn_V1 * wgn_V1 + wgn_V2 * wgn_V2;
        } while (wgn_S >= 1 || wgn_S == 0);
        X = wgn_V1 * sqrt(-2 * log(wgn_S) / wgn_S);
    } else {
        X = wgn_V2 * sqrt(-2 * log(wgn_S) / wgn_S);
    }
    wgn_phase = 1 - wgn_phase;
    return (float)X * factor;
}

int ft8o_selftest_signal(float *iSamples, float *qSamples, unsigned seed) {
    const char message[] = "CQ K1JT FN20QI";                         /* :924 */
    uint8_t packed[FT8O_LDPC_K_BYTES];
    if (ft8o_pack77(message, packed) < 0) return 0;                  /* :927 */
    uint8_t tones[FT8O_NN];
    ft8o_encode(packed, tones);                                      /* :934 */

    srand(seed);                                                     /* reference never seeds: glibc default = srand(1) */
    wgn_phase = 0;
    memset(iSamples, 0, sizeof(float) * FT8O_NSAMPLES);              /* static arrays, zero-initialised, :914-915 */
    memset(qSamples, 0, sizeof(float) * FT8O_NSAMPLES);

    float  f0  = 50.0;                                               /* :938-944 */
    float  t0  = 0.0;
    float  amp = 0.5;
    float  wgn = 0.02;
    double phi = 0.0;
    double df  = 3200.0 / 512.0;
    double dt  = 1 / 3200.0;
    for (int i = 0; i < FT8O_NN; i++) {                              /* :947-955 */
        double dphi = 2.0 * M_PI * dt * (f0 + ((double)tones[i] - 3.5) * df);
        for (int j = 0; j < 512; j++) {
            int index = t0 / dt + 512 * i + j;
            iSamples[index] = amp * cos(phi) + whiteGaussianNoise(wgn);
            qSamples[index] = amp * sin(phi) + whiteGaussianNoise(wgn);
            phi += dphi;
        }
    }
    return 1;
}

/* f-3: the modulation loop of decoderSelfTest(), rtlsdr_ft8d.c:946-955 -- plain FSK,
 * 512 samples per symbol, the phase accumulated in double by `phi += dphi` -- generalised to S signals with
 * their own start sample, level and frequency, summed in double and rounded to float once.  `f_tone0_hz` is
 * the frequency of tone 0, i.e. the reference's `f0 - 3.5 * df` (its f0 = 50 Hz puts tone 0 at 28.125 Hz).
 * No noise here: the product's generator adds its own counter-based AWGN, which has no CPU counterpart. */
void ft8o_synth_cpfsk(const uint8_t *tones /* [nsig][79] */, const double *f_tone0_hz, const int *start_sample,
                      const double *amplitude, int nsig, float *iSamples, float *qSamples) {
    double *aI = (double *)calloc(FT8O_NSAMPLES, sizeof(double)), *aQ = (double *)calloc(FT8O_NSAMPLES, sizeof(double));
    const double df = 3200.0 / 512.0, dt = 1 / 3200.0;                 /* :942-943 */
    for (int s = 0; s < nsig; s++) {
        double phi = 0.0;                                                /* :941 */
        for (int i = 0; i < FT8O_NN; i++) {                              /* :946 */
            double dphi = 2.0 * M_PI * dt * (f_tone0_hz[s] + (double)tones[s * FT8O_NN + i] * df);   /* :947 */
            for (int j = 0; j < 512; j++) {
                int index = start_sample[s] + 512 * i + j;               /* :949 */
                if (index >= 0 && index < FT8O_NSAMPLES) {
                    aI[index] += amplitude[s] * cos(phi);                /* :950 */
                    aQ[index] += amplitude[s] * sin(phi);                /* :951 */
                }
                phi += dphi;                                             /* :952 */
            }
        }
    }
    for (int i = 0; i < FT8O_NSAMPLES; i++) { iSamples[i] = (float)aI[i]; qSamples[i] = (float)aQ[i]; }
    free(aI);
    free(aQ);
}

/* rtlsdr_ft8d.c:248-263 (decoder thread) == :763-778 (file readers) */
void ft8o_normalise(float *iSamples, float *qSamples, int n) {
    float maxSig = 1e-24f;
    for (int i = 0; i < n; i++) {
        float absI = fabs(iSamples[i]);
        float absQ = fabs(qSamples[i]);
        if (absI > maxSig) maxSig = absI;
        if (absQ > maxSig) maxSig = absQ;
    }
    maxSig = 0.5 / maxSig;
    for (int i = 0; i < n; i++) {
        iSamples[i] *= maxSig;
        qSamples[i] *= maxSig;
    }
}

/* rtlsdr_ft8d.c:784-806 */
int32_t ft8o_write_raw_iq(const float *iSamples, const float *qSamples, const char *filename) {
    float *filebuffer = (float *)malloc(sizeof(float) * 2 * FT8O_NSAMPLES);
    FILE *fd = fopen(filename, "wb");
    if (fd == NULL) { free(filebuffer); return 0; }
    for (int32_t i = 0; i < FT8O_NSAMPLES; i++) {
        filebuffer[2 * i] = iSamples[i];
        filebuffer[2 * i + 1] = -qSamples[i];
    }
    int32_t nwrite = (int32_t)fwrite(filebuffer, sizeof(float), 2 * FT8O_NSAMPLES, fd);
    fclose(fd);
    free(filebuffer);
    if (nwrite != 2 * FT8O_NSAMPLES) return 0;
    return FT8O_NSAMPLES;
}

static int32_t deinterleave_normalise(float *iSamples, float *qSamples, const float *filebuffer, int32_t nread) {
    int32_t recsize = nread / 2;
    for (int32_t i = 0; i < recsize; i++) {
        iSamples[i] = filebuffer[2 * i];
        qSamples[i] = -filebuffer[2 * i + 1];
    }
    ft8o_normalise(iSamples, qSamples, recsize);
    return recsize;
}

/* rtlsdr_ft8d.c:744-781 */
int32_t ft8o_read_raw_iq(float *iSamples, float *qSamples, const char *filename) {
    float *filebuffer = (float *)malloc(sizeof(float) * 2 * FT8O_NSAMPLES);
    FILE *fd = fopen(filename, "rb");
    if (fd == NULL) { free(filebuffer); return 0; }
    int32_t nread = (int32_t)fread(filebuffer, sizeof(float), 2 * FT8O_NSAMPLES, fd);
    fclose(fd);
    int32_t r = deinterleave_normalise(iSamples, qSamples, filebuffer, nread);
    free(filebuffer);
    return r;
}

/* rtlsdr_ft8d.c:809-856 */
int32_t ft8o_read_c2(float *iSamples, float *qSamples, const char *filename, double *dialfreq) {
    float *filebuffer = (float *)malloc(sizeof(float) * 2 * FT8O_NSAMPLES);
    FILE *fd = fopen(filename, "rb");
    if (fd == NULL) { free(filebuffer); return 0; }
    char name[15];
    int type;
    double frequency = 0;
    size_t nr;
    nr = fread(name, sizeof(char), 14, fd);
    nr = fread(&type, sizeof(int), 1, fd);
    nr = fread(&frequency, sizeof(double), 1, fd);
    (void)nr;
    if (dialfreq) *dialfreq = frequency;
    int32_t nread = (int32_t)fread(filebuffer, sizeof(float), 2 * FT8O_NSAMPLES, fd);
    fclose(fd);
    int32_t r = deinterleave_normalise(iSamples, qSamples, filebuffer, nread);
    free(filebuffer);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * rtlsdr_callback()  rtlsdr_ft8d.c:76-202   
 * ---------------------------------------------------------------------------------------- */
#define RX_DOWNSAMPLING 750            /* SAMPLING_RATE / SIGNAL_SAMPLE_RATE, rtlsdr_ft8d.h:38 */
#define RX_FIR_TAPS 56                 /* rtlsdr_ft8d.h:40 */

/* rtlsdr_ft8d.c:94-110 */
static const float rx_zCoef[RX_FIR_TAPS + 1] = {
    -0.0025719973,  0.0010118403,  0.0009110571, -0.0034940765,
     0.0069713409, -0.0114242790,  0.0167023466, -0.0223683056,
     0.0276808966, -0.0316243672,  0.0329894230, -0.0305042011,
     0.0230074504, -0.0096499429, -0.0098950502,  0.0352349632,
    -0.0650990428,  0.0972406918, -0.1284211497,  0.1544893973,
    -0.1705667465,  0.1713383321, -0.1514501610,  0.1060148823,
    -0.0312560926, -0.0745846391,  0.2096088743, -0.3638689868,
     0.5000000000,
    -0.3638689868,  0.2096088743, -0.0745846391, -0.0312560926,
     0.1060148823, -0.1514501610,  0.1713383321, -0.1705667465,
     0.1544893973, -0.1284211497,  0.0972406918, -0.0650990428,
     0.0352349632, -0.0098950502, -0.0096499429,  0.0230074504,
    -0.0305042011,  0.0329894230, -0.0316243672,  0.0276808966,
    -0.0223683056,  0.0167023466, -0.0114242790,  0.0069713409,
    -0.0034940765,  0.0009110571,  0.0010118403, -0.0025719973
};

void ft8o_rx_reset(ft8o_rx_state_t *st) { memset(st, 0, sizeof *st); }

void ft8o_rx_callback(ft8o_rx_state_t *st, unsigned char *samples, uint32_t samples_count,
                      float *iSamples, float *qSamples, uint32_t *iq_index) {
    int8_t *sigIn = (int8_t *)samples;
    int8_t tmp;
    /* :129-140 economic mixer @ fs/4 (int8 stores wrap: -(-128) stays -128) */
    for (uint32_t i = 0; i < samples_count; i += 8) {
        sigIn[i    ] ^= 0x80;
        sigIn[i + 1] ^= 0x80;
        tmp          = (sigIn[i + 3] ^ 0x80);
        sigIn[i + 3] = (sigIn[i + 2] ^ 0x80);
        sigIn[i + 2] = -tmp;
        sigIn[i + 4] = -(sigIn[i + 4] ^ 0x80);
        sigIn[i + 5] = -(sigIn[i + 5] ^ 0x80);
        tmp          = (sigIn[i + 6] ^ 0x80);
        sigIn[i + 6] = (sigIn[i + 7] ^ 0x80);
        sigIn[i + 7] = -tmp;
    }
    /* :148-201 CIC decimator (N = 2), compensation FIR, scaling.  Integrators use wrapping 32-bit
     * arithmetic (signed overflow wraps on every target the reference runs on; made explicit here). */
    for (int32_t i = 0; i < (int32_t)(samples_count / 2); i++) {
        st->Ix1 = (int32_t)((uint32_t)st->Ix1 + (uint32_t)(int32_t)sigIn[i * 2]);
        st->Qx1 = (int32_t)((uint32_t)st->Qx1 + (uint32_t)(int32_t)sigIn[i * 2 + 1]);
        st->Ix2 = (int32_t)((uint32_t)st->Ix2 + (uint32_t)st->Ix1);
        st->Qx2 = (int32_t)((uint32_t)st->Qx2 + (uint32_t)st->Qx1);

        st->decimationIndex++;
        if (st->decimationIndex <= RX_DOWNSAMPLING) continue;        /* :157: effective ratio 751 */
        st->decimationIndex = 0;

        st->Iy1 = (int32_t)((uint32_t)st->Ix2 - (uint32_t)st->It1z);  st->It1z = st->It1y;  st->It1y = st->Ix2;
        st->Qy1 = (int32_t)((uint32_t)st->Qx2 - (uint32_t)st->Qt1z);  st->Qt1z = st->Qt1y;  st->Qt1y = st->Qx2;
        st->Iy2 = (int32_t)((uint32_t)st->Iy1 - (uint32_t)st->It2z);  st->It2z = st->It2y;  st->It2y = st->Iy1;
        st->Qy2 = (int32_t)((uint32_t)st->Qy1 - (uint32_t)st->Qt2z);  st->Qt2z = st->Qt2y;  st->Qt2y = st->Qy1;

        float Isum = 0.0, Qsum = 0.0;
        for (uint32_t j = 0; j < RX_FIR_TAPS; j++) {
            Isum += st->firI[j] * rx_zCoef[j];
            Qsum += st->firQ[j] * rx_zCoef[j];
            if (j < RX_FIR_TAPS - 1) {
                st->firI[j] = st->firI[j + 1];
                st->firQ[j] = st->firQ[j + 1];
            }
        }
        st->firI[RX_FIR_TAPS - 1] = (float)st->Iy2;
        st->firQ[RX_FIR_TAPS - 1] = (float)st->Qy2;
        Isum += st->firI[RX_FIR_TAPS - 1] * rx_zCoef[RX_FIR_TAPS];
        Qsum += st->firQ[RX_FIR_TAPS - 1] * rx_zCoef[RX_FIR_TAPS];

        if (*iq_index < (uint32_t)FT8O_NSAMPLES) {                    /* :196 */
            iSamples[*iq_index] = Isum / (32768.0 * RX_DOWNSAMPLING);
            qSamples[*iq_index] = Qsum / (32768.0 * RX_DOWNSAMPLING);
            (*iq_index)++;
        }
    }
}

void ft8o_rx_capture(const unsigned char *raw, size_t nbytes, float *iSamples, float *qSamples,
                     uint32_t *n_out, int normalise) {
    ft8o_rx_state_t st;
    ft8o_rx_reset(&st);
    uint32_t idx = 0;
    const size_t chunk = 4 * 16384;                                   /* DEFAULT_BUF_LENGTH, rtlsdr_ft8d.h:39 */
    unsigned char *buf = (unsigned char *)malloc(chunk);
    for (size_t off = 0; off < nbytes; off += chunk) {
        size_t n = nbytes - off < chunk ? nbytes - off : chunk;
        n &= ~(size_t)7;
        memcpy(buf, raw + off, n);
        ft8o_rx_callback(&st, buf, (uint32_t)n, iSamples, qSamples, &idx);
    }
    free(buf);
    for (uint32_t i = idx; i < (uint32_t)FT8O_NSAMPLES; i++) { iSamples[i] = 0.0; qSamples[i] = 0.0; }   /* :243-246 */
    if (normalise) ft8o_normalise(iSamples, qSamples, FT8O_NSAMPLES);                                     /* :248-263 */
    if (n_out) *n_out = idx;
}

/* ================= spot reporting wire formats ================================= */

/* the two IPFIX template sets of rtlsdr_ft8d.c:386-423 as big-endian 16-bit words
 * (set id, set length, link id, field count, [scope count]; then id, length, enterprise hi, lo per field) */
static const uint16_t rep_rx_template[18] = {
    0x0003, 36, 0x9992, 3, 0,
    0x8002, 0xFFFF, 0x0000, 0x768F,       /* receiver callsign, variable */
    0x8004, 0xFFFF, 0x0000, 0x768F,       /* receiver locator, variable */
    0x8008, 0xFFFF, 0x0000, 0x768F,       /* decoder software, variable */
    0x0000 };                             /* padding */
static const uint16_t rep_tx_template[30] = {
    0x0002, 60, 0x9993, 7,
    0x8001, 0xFFFF, 0x0000, 0x768F,       /* sender callsign, variable */
    0x8005, 4,      0x0000, 0x768F,       /* frequency */
    0x8006, 1,      0x0000, 0x768F,       /* SNR */
    0x800A, 0xFFFF, 0x0000, 0x768F,       /* mode, variable */
    0x8003, 0xFFFF, 0x0000, 0x768F,       /* sender locator, variable */
    0x800B, 1,      0x0000, 0x768F,       /* information source */
    0x0096, 4 };                          /* flowStartSeconds */

static uint32_t rep_w16(unsigned char *p, uint32_t at, uint16_t v) { p[at] = (unsigned char)(v >> 8); p[at + 1] = (unsigned char)v; return at + 2; }
static uint32_t rep_w32(unsigned char *p, uint32_t at, uint32_t v) { at = rep_w16(p, at, (uint16_t)(v >> 16)); return rep_w16(p, at, (uint16_t)v); }
static uint32_t rep_str(unsigned char *p, uint32_t at, const char *s, size_t cap) {
    size_t len = 0;
    while (len < cap && s[len]) len++;
    p[at++] = (unsigned char)len;
    memcpy(p + at, s, len);
    return at + (uint32_t)len;
}

int ft8o_pskreporter_datagram(const struct ft8o_decoder_results *dec_results, int32_t n_results,
                              const ft8o_report_info *info, unsigned char *out) {
    unsigned char header[16], rx[256], tx[1500];
    memset(rx, 0, sizeof rx);                                         /* :454 */
    memset(tx, 0, sizeof tx);                                         /* :486 */

    uint32_t h = 0;                                                   /* :439-450 */
    h = rep_w16(header, h, 0x000A);
    h = rep_w16(header, h, 0);
    h = rep_w32(header, h, info->unixtime);
    h = rep_w32(header, h, info->sequence);
    h = rep_w32(header, h, info->random_id);

    uint32_t rxPtr = rep_w16(rx, 0, 0x9992) + 2;                      /* :458-460 */
    rxPtr = rep_str(rx, rxPtr, info->rcall, 12);                      /* :463-466 */
    rxPtr = rep_str(rx, rxPtr, info->rloc, 6);                        /* :469-472 */
    rxPtr = rep_str(rx, rxPtr, info->app_version, 31);                /* :475-478 */
    if (rxPtr % 4) rxPtr += 4 - rxPtr % 4;                            /* :481-482 */

    uint32_t txPtr = rep_w16(tx, 0, 0x9993) + 2;                      /* :490-492 */
    if (n_results > FT8O_K_MAX_MESSAGES) n_results = FT8O_K_MAX_MESSAGES;
    for (int32_t i = 0; i < n_results; i++) {                         /* :494 */
        if (txPtr > 1200) break;                                      /* :497 */
        txPtr = rep_str(tx, txPtr, dec_results[i].call, 12);          /* :501-504 */
        txPtr = rep_w32(tx, txPtr, (uint32_t)dec_results[i].freq + info->dial_freq);   /* :507 */
        { int8_t v = (int8_t)((int8_t)dec_results[i].snr - 20); tx[txPtr++] = (unsigned char)v; }   /* :511 */
        txPtr = rep_str(tx, txPtr, "FT8", 3);                         /* :515-518 */
        txPtr = rep_str(tx, txPtr, dec_results[i].loc, 6);            /* :521-524 */
        tx[txPtr++] = 1;                                              /* :527 */
        txPtr = rep_w32(tx, txPtr, info->unixtime);                   /* :531 */
    }
    if (txPtr % 4) txPtr += 4 - txPtr % 4;                            /* :536-537 */

    const uint32_t full = 16 + 36 + 60 + rxPtr + txPtr;               /* :541 */
    rep_w16(rx, 2, (uint16_t)rxPtr);                                  /* :542 */
    rep_w16(tx, 2, (uint16_t)txPtr);                                  /* :543 */
    rep_w16(header, 2, (uint16_t)full);                               /* :544 */

    uint32_t at = 0;                                                  /* :547-553 */
    memcpy(out + at, header, 16); at += 16;
    for (int i = 0; i < 18; i++) at = rep_w16(out, at, rep_rx_template[i]);
    for (int i = 0; i < 30; i++) at = rep_w16(out, at, rep_tx_template[i]);
    memcpy(out + at, rx, rxPtr); at += rxPtr;
    memcpy(out + at, tx, txPtr); at += txPtr;
    return (int)at;
}

int ft8o_format_spots(const struct ft8o_decoder_results *dec_results, int32_t n_results, uint32_t dial_freq,
                      int year, int month, int mday, int hour, int minute, char *out, size_t cap) {
    size_t at = 0;
    int k;
    if (cap) out[0] = 0;
    if (n_results <= 0) {                                             /* :644-653 */
        k = snprintf(out, cap, "No spot %04d-%02d-%02d %02d:%02dz\n", year, month, mday, hour, minute);
        return k;
    }
    k = snprintf(out, cap, "  Score     Freq       Call    Loc\n");   /* :655 */
    at += (size_t)k;
    for (int32_t i = 0; i < n_results && i < FT8O_K_MAX_MESSAGES; i++) {   /* :656-662 */
        char call[13], loc[7];
        memcpy(call, dec_results[i].call, 12); call[12] = 0;
        memcpy(loc, dec_results[i].loc, 6); loc[6] = 0;
        k = snprintf(at < cap ? out + at : NULL, at < cap ? cap - at : 0, "     %2d %8d %10s %6s\n",
                     dec_results[i].snr, (int)((uint32_t)dec_results[i].freq + dial_freq), call, loc);
        at += (size_t)k;
    }
    return (int)at;
}
