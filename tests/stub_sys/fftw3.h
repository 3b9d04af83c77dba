/* syntax-check stand-in for FFTW 3's public header, single-precision interface only (see README.md here) */
#ifndef STUB_FFTW3_H
#define STUB_FFTW3_H
#include <stddef.h>
#include <stdio.h>
typedef float fftwf_complex[2];
typedef struct fftwf_plan_s *fftwf_plan;
#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_MEASURE (0U)
#define FFTW_EXHAUSTIVE (1U << 3)
#define FFTW_PATIENT (1U << 5)
#define FFTW_ESTIMATE (1U << 6)
void *fftwf_malloc(size_t n);
void fftwf_free(void *p);
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);
int fftwf_import_wisdom_from_file(FILE *input_file);
void fftwf_export_wisdom_to_file(FILE *output_file);
#endif
