/*
 * ft8_lib/ft8/decode.h -- drop-in interface header taking the place of the header of the same name of kgoba/ft8_lib, the git submodule
 * rtlsdr_ft8d.c includes at :43 ("./ft8_lib/ft8/decode.h") and whose sources are absent from the reference
 * snapshot.  It declares exactly what the reference uses at rtlsdr_ft8d.c:1439-1494 -- waterfall_t (filled by
 * designated initialisers, :1440-1448), candidate_t, message_t, decode_status_t, ft8_find_sync (:1450) and
 * ft8_decode (:1476) -- with libft8gpu.so behind the two functions.  With `-I<this repo>/include` the UNMODIFIED
 * rtlsdr_ft8d.c compiles against these seven headers and links against libft8gpu.so instead of
 * the ft8_lib objects (Makefile:9); its own ft8_subsystem() then runs its fftw3f waterfall on the host and the
 * Costas search and the LDPC decode on the GPU.
 *
 * Only the waterfall geometry the reference builds is accepted (92 blocks x 2 x 2 x 256 bins, stride 1024,
 * PROTO_FT8: rtlsdr_ft8d.h:51-56); anything else returns 0 / false and sets ft8gpu_last_error().
 */
#ifndef FT8GPU_COMPAT_FT8_DECODE_H
#define FT8GPU_COMPAT_FT8_DECODE_H

#include <stdbool.h>
#include <stdint.h>

#include "constants.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* magnitude waterfall: mag[block][time_sub][freq_sub][bin], one byte per cell (rtlsdr_ft8d.c:1440-1448) */
typedef struct {
    int max_blocks;          /* present in upstream revisions of this era; the reference leaves it 0 */
    int num_blocks;
    int num_bins;
    int time_osr;
    int freq_osr;
    uint8_t *mag;
    int block_stride;
    ftx_protocol_t protocol;
} waterfall_t;

/* rtlsdr_ft8d.c:1439, :1466-1470 */
typedef struct {
    int16_t score;
    int16_t time_offset;
    int16_t freq_offset;
    uint8_t time_sub;
    uint8_t freq_sub;
} candidate_t;

/* rtlsdr_ft8d.c:1454, :1474, :1487, :1494 */
typedef struct {
    char text[25];
    uint16_t hash;
} message_t;

/* rtlsdr_ft8d.c:1475-1481 */
typedef struct {
    int ldpc_errors;
    uint16_t crc_extracted;
    uint16_t crc_calculated;
    int unpack_status;
} decode_status_t;

/* rtlsdr_ft8d.c:1450: the `num_candidates` best Costas-sync positions with score >= min_score, best first;
 * returns how many.  GPU: ft8_sync_kernel + the exact heap replay. */
int ft8_find_sync(const waterfall_t *power, int num_candidates, candidate_t heap[], int min_score);

/* rtlsdr_ft8d.c:1476: LLR extraction, normalisation, LDPC BP (max_iterations), CRC-14, unpack77 for one
 * candidate.  GPU: ft8_decode_kernel; the first call after ft8_find_sync() on the same waterfall decodes the
 * whole candidate list in one launch and later calls for candidates of that list are answered from it.
 *
 * ASSUMPTION the remembered list rests on: "the same waterfall" means the same `power->mag` POINTER whose 94 208 bytes still
 * hash (64-bit multiply-mix, recomputed on every call, about 10 us) to what the last ft8_find_sync() saw.  Another pointer,
 * changed bytes, a candidate that is not in the list, or no preceding ft8_find_sync() all take the one-candidate launch:
 * the answer is always the pure function of (bytes, candidate, max_iterations) unless rewritten bytes collide with the
 * old ones in that hash (2^-64 per call).  The reference never rewrites the buffer between the two calls
 * (rtlsdr_ft8d.c:1450-1476 work on one stack array).  Both functions serialise on one process-wide lock. */
bool ft8_decode(const waterfall_t *power, const candidate_t *cand, message_t *message, int max_iterations,
                decode_status_t *status);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
