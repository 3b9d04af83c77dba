/* ft8_lib/ft8/constants.h -- drop-in interface header (see decode.h): the constants rtlsdr_ft8d.c takes from the absent
 * submodule: FT8_NN (:933, :947), FTX_LDPC_K_BYTES (:925), PROTO_FT8 (:1447). */
#ifndef FT8GPU_COMPAT_FT8_CONSTANTS_H
#define FT8GPU_COMPAT_FT8_CONSTANTS_H

#define FT8_ND 58                 /* data symbols */
#define FT8_NN 79                 /* channel symbols: 58 data + 3 x 7 Costas */
#define FT8_LENGTH_SYNC 7
#define FT8_NUM_SYNC 3
#define FT8_SYNC_OFFSET 36

#define FTX_LDPC_N 174
#define FTX_LDPC_K 91
#define FTX_LDPC_M 83
#define FTX_LDPC_N_BYTES ((FTX_LDPC_N + 7) / 8)
#define FTX_LDPC_K_BYTES ((FTX_LDPC_K + 7) / 8)

typedef enum { PROTO_FT4, PROTO_FT8 } ftx_protocol_t;

#endif
