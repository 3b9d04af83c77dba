/* ft8_lib/ft8/encode.h -- drop-in interface header (see decode.h): ft8_encode, used by the reference's self-test at rtlsdr_ft8d.c:934. */
#ifndef FT8GPU_COMPAT_FT8_ENCODE_H
#define FT8GPU_COMPAT_FT8_ENCODE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
/* 77-bit payload -> CRC-14 -> LDPC(174,91) -> FT8_NN = 79 tone numbers 0..7 */
void ft8_encode(const uint8_t *payload, uint8_t *tones);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
