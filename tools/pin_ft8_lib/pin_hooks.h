/* pin_hooks.h -- what a backend of pin_harness.c provides under -DPIN_INTERNALS: the three stages behind ft8_find_sync /
 * ft8_decode that the public interface does not show.  oracle_as_ft8_lib.c implements them over the oracle;
 * upstream_internals.c implements them over kgoba/ft8_lib's own (static) functions. */
#ifndef PIN_HOOKS_H
#define PIN_HOOKS_H
#include <stdint.h>
#include "ft8/decode.h"
int  pin_sync_score(const waterfall_t *wf, const candidate_t *c);                 /* ft8_sync_score */
void pin_llr(const waterfall_t *wf, const candidate_t *c, float log174[174]);     /* ft8_extract_likelihood + ftx_normalize_logl */
int  pin_bp(float log174[174], int max_iters, uint8_t plain[174]);                /* bp_decode; returns the parity errors left */
#endif
