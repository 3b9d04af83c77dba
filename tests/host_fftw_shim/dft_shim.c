/*
 * TEST-ONLY shim, NOT FFTW: five functions under FFTW's names (fftwf_malloc, fftwf_free, fftwf_plan_dft_1d,
 * fftwf_execute_dft, fftwf_destroy_plan) around a plain float64 DFT, so that the dlopen plumbing of the oracle's optional
 * FFTW leg (oracle/ft8_oracle.c: ft8o_fftw_init with an explicit path) can be executed on a box without libfftw3f.
 * It is handed to ft8o_fftw_init by path from tests/test_oracle.py only; bench.py binds the system's libfftw3f.so.3 or
 * reports that there is none -- this file is never a baseline and never stands in for the reference's FFT.
 */
#include <math.h>
#include <stdlib.h>

typedef float cplx[2];
struct plan { int n, sign; };

void *fftwf_malloc(size_t n) { void *p = NULL; return posix_memalign(&p, 64, n) == 0 ? p : NULL; }
void fftwf_free(void *p) { free(p); }
void *fftwf_plan_dft_1d(int n, cplx *in, cplx *out, int sign, unsigned flags) {
    (void)in; (void)out; (void)flags;
    if (n != 1024) return NULL;
    struct plan *p = malloc(sizeof *p);
    if (p) { p->n = n; p->sign = sign; }
    return p;
}
void fftwf_destroy_plan(void *p) { free(p); }

void fftwf_execute_dft(void *pl, cplx *in, cplx *out) {
    const struct plan *p = pl;
    const int N = p->n;
    static __thread double re[1024], im[1024];
    for (int i = 0; i < N; i++) {
        int r = 0, v = i;
        for (int b = 0; b < 10; b++) { r = (r << 1) | (v & 1); v >>= 1; }
        re[r] = in[i][0]; im[r] = in[i][1];
    }
    for (int len = 2; len <= N; len <<= 1)
        for (int base = 0; base < N; base += len)
            for (int j = 0; j < len / 2; j++) {
                const double a = (double)p->sign * 2.0 * M_PI * (double)j / (double)len;
                const double wr = cos(a), wi = sin(a);
                const int u = base + j, v = u + len / 2;
                const double xr = re[v] * wr - im[v] * wi, xi = re[v] * wi + im[v] * wr;
                re[v] = re[u] - xr; im[v] = im[u] - xi;
                re[u] = re[u] + xr; im[u] = im[u] + xi;
            }
    for (int i = 0; i < N; i++) { out[i][0] = (float)re[i]; out[i][1] = (float)im[i]; }
}
