/* ft8_lib/ft8/unpack.h -- drop-in interface header (see decode.h).  rtlsdr_ft8d.c includes this header (:38-44) but calls nothing
 * from it: on this path its routines run inside ft8_decode(), i.e. inside the GPU kernel. */
#ifndef FT8GPU_COMPAT_FT8_UNPACK_H
#define FT8GPU_COMPAT_FT8_UNPACK_H
#endif
