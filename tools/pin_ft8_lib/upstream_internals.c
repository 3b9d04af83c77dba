/*
 * upstream_internals.c -- the three pin_hooks.h hooks over kgoba/ft8_lib's OWN functions, for tools/pin_ft8_lib.sh.
 * ft8_sync_score, ft8_extract_likelihood and ftx_normalize_logl are file-local in upstream's decode.c, so this file
 * includes that source (it is compiled INSTEAD of ft8/decode.c) and wraps them.  It cannot be compiled in this
 * repository's image -- the submodule is empty (.gitmodules:1-3 of the reference) -- and the names below are those of
 * the ft8_lib revisions whose interface the reference calls (rtlsdr_ft8d.c:1450, :1476); if a checkout spells one
 * differently the build stops here with the compiler naming it: adjust the three macros, nothing else depends on them.
 */
#ifndef PIN_UPSTREAM_SYNC_SCORE
#define PIN_UPSTREAM_SYNC_SCORE   ft8_sync_score             /* static int ft8_sync_score(const waterfall_t*, const candidate_t*) */
#endif
#ifndef PIN_UPSTREAM_EXTRACT_LLR
#define PIN_UPSTREAM_EXTRACT_LLR  ft8_extract_likelihood     /* static void ft8_extract_likelihood(const waterfall_t*, const candidate_t*, float*) */
#endif
#ifndef PIN_UPSTREAM_NORMALIZE
#define PIN_UPSTREAM_NORMALIZE    ftx_normalize_logl         /* static void ftx_normalize_logl(float*) */
#endif

#include "ft8/decode.c"          /* upstream's source: brings its statics into this translation unit */
#include "ft8/ldpc.h"            /* void bp_decode(float codeword[], int max_iters, uint8_t plain[], int* ok) */
#include "pin_hooks.h"

int pin_sync_score(const waterfall_t *wf, const candidate_t *c) { return PIN_UPSTREAM_SYNC_SCORE(wf, c); }

void pin_llr(const waterfall_t *wf, const candidate_t *c, float log174[174]) {
    PIN_UPSTREAM_EXTRACT_LLR(wf, c, log174);
    PIN_UPSTREAM_NORMALIZE(log174);
}

int pin_bp(float log174[174], int max_iters, uint8_t plain[174]) {
    int errors = 0;
    bp_decode(log174, max_iters, plain, &errors);
    return errors;
}
