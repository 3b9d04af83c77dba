/* ft8_lib/ft8/pack.h -- drop-in interface header (see decode.h): pack77, used by the reference's self-test at rtlsdr_ft8d.c:927. */
#ifndef FT8GPU_COMPAT_FT8_PACK_H
#define FT8GPU_COMPAT_FT8_PACK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
/* "CALL1 CALL2 [GRID4]" standard (i3 = 1) messages -> 77 bits in c77[0..9] (c77 must hold FTX_LDPC_K_BYTES);
 * returns 0, or -1 if the text is not such a message (other message types are not needed on this path) */
int pack77(const char *msg, uint8_t *c77);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
