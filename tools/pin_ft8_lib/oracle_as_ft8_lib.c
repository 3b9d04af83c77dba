/*
 * oracle_as_ft8_lib.c -- the CPU oracle (oracle/libft8oracle.so) under the ft8_lib names pin_harness.c calls, so that
 * the harness prints the dump the oracle EXPECTS from any other implementation of that interface.  Test tooling
 * (tools/pin_ft8_lib.sh, tests/test_pin_harness.py); nothing in the product links it.
 * Built with -I include/ft8_lib (the interface headers of this repo) and -I oracle.
 */
#include <string.h>

#include "ft8/constants.h"
#include "ft8/decode.h"
#include "ft8/encode.h"
#include "ft8/pack.h"
#include "ft8_oracle.h"
#include "pin_hooks.h"

_Static_assert(sizeof(candidate_t) == sizeof(ft8o_candidate_t), "candidate_t layout");

int ft8_find_sync(const waterfall_t *power, int num_candidates, candidate_t heap[], int min_score) {
    return ft8o_find_sync(power->mag, num_candidates, (ft8o_candidate_t *)heap, min_score);
}

bool ft8_decode(const waterfall_t *power, const candidate_t *cand, message_t *message, int max_iterations, decode_status_t *status) {
    ft8o_message_t m;
    ft8o_decode_status_t st;
    ft8o_decode_extra_t ex;
    memset(&m, 0, sizeof m);
    const int ok = ft8o_decode(power->mag, (const ft8o_candidate_t *)cand, &m, max_iterations, &st, &ex);
    status->ldpc_errors = st.ldpc_errors;
    status->crc_extracted = st.crc_extracted;
    status->crc_calculated = st.crc_calculated;
    status->unpack_status = st.unpack_status;
    if (ok) { memcpy(message->text, m.text, sizeof message->text); message->hash = m.hash; }
    return ok != 0;
}

int pack77(const char *msg, uint8_t *c77) {
    uint8_t b[12];
    const int rc = ft8o_pack77(msg, b);
    if (rc == 0) memcpy(c77, b, 10);
    return rc;
}

void ft8_encode(const uint8_t *payload, uint8_t *tones) { ft8o_encode(payload, tones); }

int pin_sync_score(const waterfall_t *wf, const candidate_t *c) { return ft8o_sync_score(wf->mag, (const ft8o_candidate_t *)c); }

void pin_llr(const waterfall_t *wf, const candidate_t *c, float log174[174]) {
    ft8o_extract_likelihood(wf->mag, (const ft8o_candidate_t *)c, log174);
    ft8o_normalize_logl(log174);
}

int pin_bp(float log174[174], int max_iters, uint8_t plain[174]) {
    int errors = 0, iters = 0;
    ft8o_bp_decode(log174, max_iters, plain, &errors, &iters);
    return errors;
}
