/* ft8_lib/ft8/pack.h -- drop-in interface header (see decode.h): pack77, used by the reference's self-test at rtlsdr_ft8d.c:927. */
#ifndef FT8GPU_COMPAT_FT8_PACK_H
#define FT8GPU_COMPAT_FT8_PACK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
/* message text -> 77 bits in c77[0..9] (c77 must hold FTX_LDPC_K_BYTES); returns 0, or -1 if the text fits no message
 * type.  Everything ft8_lib's pack77 packs (standard calls with grid / report / RRR / RR73 / 73, else free text) and
 * more: see ft8gpu_pack77 in ft8gpu.h */
int pack77(const char *msg, uint8_t *c77);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
