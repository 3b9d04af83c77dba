/* bp_recurrence.c -- measurement helper of tools/bp_recurrence.py (NOT product code, NOT the oracle).
 *
 * Question: among the candidates for which ft8_lib's bp_decode() (call site
 * rtlsdr_ft8d.c:1476, 20 iterations, success iff 0 parity errors) never reaches a codeword, how many reach an
 * EXACT recurrence of the message state -- the 522 floats tov[n][m_idx] at the top of iteration k equal, bit
 * for bit, those at the top of iteration k-p for a small period p?  From there on the iteration is a pure
 * function of a state it has already been in, so every later hard decision repeats one that has already failed
 * its parity check: leaving the loop at k is output-exact for a caller that only consumes "codeword or not".
 *
 * The loop below is the same float32 operation sequence as oracle/ft8_oracle.c: ft8o_bp_decode (which restates
 * ft8_lib ldpc.c), compiled with -ffp-contract=off; tools/bp_recurrence.py cross-checks iteration count and
 * success flag of every candidate against the oracle's own function.  On top of it, the state of the last
 * MAXP iterations is kept and compared.
 *
 * Also recorded, for orientation only (not an exact exit): the first iteration whose HARD DECISION repeats the
 * one of iteration k-1 or k-2, and whether all toc magnitudes are saturated (|tanh| == 1).
 */
#include <stdint.h>
#include <string.h>
#include "../oracle/ft8o_tables.h"

#define N 174
#define M 83
#define MAXP 8

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

typedef struct {
    int32_t iters;          /* iterations entered (as the oracle reports it) */
    int32_t min_errors;     /* 0 = codeword */
    int32_t state_rec_k;    /* first k with tov(k) == tov(k - p) bitwise for some p <= MAXP; -1 = none */
    int32_t state_rec_p;    /* that p (smallest) */
    int32_t hard_rec_k;     /* first k >= 1 whose hard decision equals the one of k-1 or k-2; -1 = none */
    int32_t hard_rec_p;
    int32_t errors_last;    /* failed checks of the last hard decision */
    int32_t sat_edges_last; /* number of the 522 toc with |toc| == 1 in the last message update */
    int32_t last_flip_k;    /* last iteration whose hard decision differs from the previous one */
    int32_t errors_min_k;   /* iteration at which min_errors was first reached */
} bp_track_t;

static int check(const uint8_t *plain) {
    int errors = 0;
    for (int m = 0; m < M; ++m) {
        uint8_t x = 0;
        for (int i = 0; i < kO_Num_rows[m]; ++i) x ^= plain[kO_Nm[m][i] - 1];
        if (x) ++errors;
    }
    return errors;
}

void bp_track(const float *codeword, int max_iters, bp_track_t *out) {
    float tov[N][3];
    float toc[M][7];
    static _Thread_local float hist[MAXP][N][3];
    static _Thread_local uint8_t hard_hist[2][N];
    uint8_t plain[N];
    int min_errors = M, iter;
    memset(tov, 0, sizeof tov);
    out->state_rec_k = out->hard_rec_k = -1;
    out->state_rec_p = out->hard_rec_p = 0;
    out->errors_last = M;
    out->sat_edges_last = 0;
    out->last_flip_k = 0;
    out->errors_min_k = 0;
    for (iter = 0; iter < max_iters; ++iter) {
        /* exact recurrence of the state at the top of the iteration */
        if (out->state_rec_k < 0)
            for (int p = 1; p <= MAXP && p <= iter; ++p)
                if (memcmp(hist[(iter - p) % MAXP], tov, sizeof tov) == 0) { out->state_rec_k = iter; out->state_rec_p = p; break; }
        memcpy(hist[iter % MAXP], tov, sizeof tov);

        int plain_sum = 0;
        for (int n = 0; n < N; ++n) {
            plain[n] = ((codeword[n] + tov[n][0] + tov[n][1] + tov[n][2]) > 0) ? 1 : 0;
            plain_sum += plain[n];
        }
        if (out->hard_rec_k < 0)
            for (int p = 1; p <= 2 && p <= iter; ++p)
                if (memcmp(hard_hist[(iter - p) & 1], plain, N) == 0) { out->hard_rec_k = iter; out->hard_rec_p = p; break; }
        if (iter >= 1 && memcmp(hard_hist[(iter - 1) & 1], plain, N) != 0) out->last_flip_k = iter;
        memcpy(hard_hist[iter & 1], plain, N);
        if (plain_sum == 0) break;
        int errors = check(plain);
        out->errors_last = errors;
        if (errors < min_errors) {
            min_errors = errors;
            out->errors_min_k = iter;
            if (errors == 0) break;
        }
        int sat = 0;
        for (int m = 0; m < M; ++m)
            for (int n_idx = 0; n_idx < kO_Num_rows[m]; ++n_idx) {
                int n = kO_Nm[m][n_idx] - 1;
                float Tnm = codeword[n];
                for (int m_idx = 0; m_idx < 3; ++m_idx)
                    if ((kO_Mn[n][m_idx] - 1) != m) Tnm += tov[n][m_idx];
                toc[m][n_idx] = fast_tanh(-Tnm / 2);
                sat += (toc[m][n_idx] == 1.0f || toc[m][n_idx] == -1.0f);
            }
        out->sat_edges_last = sat;
        for (int n = 0; n < N; ++n)
            for (int m_idx = 0; m_idx < 3; ++m_idx) {
                int m = kO_Mn[n][m_idx] - 1;
                float Tmn = 1.0f;
                for (int n_idx = 0; n_idx < kO_Num_rows[m]; ++n_idx)
                    if ((kO_Nm[m][n_idx] - 1) != n) Tmn *= toc[m][n_idx];
                tov[n][m_idx] = -2 * fast_atanh(Tmn);
            }
    }
    out->iters = iter;
    out->min_errors = min_errors;
}

/* many candidates: llr [count][174] */
void bp_track_many(const float *llr, int count, int max_iters, bp_track_t *out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < count; ++i) bp_track(llr + (size_t)i * N, max_iters, out + i);
}
