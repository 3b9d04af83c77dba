// decode.hip -- stage a7 of the hot path: ft8_decode() of ft8_lib decode.c,
// call site rtlsdr_ft8d.c:1476: soft-symbol extraction (ft8_extract_likelihood), LLR normalisation
// (ftx_normalize_logl), LDPC(174,91) sum-product decoding (bp_decode / ldpc_check / fast_tanh /
// fast_atanh of ft8_lib ldpc.c), CRC-14 (crc.c) and unpack77 (unpack.c).
//
// One wave64 per candidate codeword.
//   * LLR: lane k < 58 owns data symbol k: 8 waterfall bytes -> three max-log differences.
//     The values are small integers, so the reference's sequential float sums in
//     ftx_normalize_logl (sum <= 174*255, sum of squares <= 174*255^2 < 2^24) are exact in any
//     order; they are reduced across the wave in integer arithmetic.
//   * BP: lane l owns variable nodes l, l+64, l+128.  Variable-to-check messages (tov) stay in
//     registers; the check-to-variable products need the tanh terms of the other edges of a
//     check row, which are exchanged through an 83 x 8 float LDS tile (one ds_write_b32 per edge,
//     two ds_read_b128 per row).  Unused row slots hold 1.0f so the product loop is branch-free;
//     multiplying by 1.0f is exact, so the product equals the reference's skip-self loop in order.
//   * hard decisions are gathered with three __ballot()s; a parity check is popcount(word & row
//     mask); the error count is the popcount of a ballot.  Exit conditions are wave-uniform.
//   * every float expression is written in the reference's operation order and compiled with
//     -ffp-contract=off; divisions and sqrt are IEEE correctly rounded.
#include "ft8gpu_internal.h"
#include "ft8_tables.h"
#include "unpack_dev.h"

namespace {

struct DecodeTables {
    uint16_t edge_slot[3][64][3];     // [r][lane][m_idx] -> LDS float index m*8 + pos   (0xFFFF = no variable)
    uint8_t  edge_pos[3][64][3];      // position of the variable inside its check row (0..6)
    uint64_t rowmask[2][64][3];       // [r][lane][word] bit mask of the variables of check m = lane + 64 r
    uint8_t  row_valid[2][64];
};

__device__ DecodeTables d_tab;
__constant__ uint8_t c_gray[8] = { 0, 1, 3, 2, 5, 6, 4, 7 };

// branch-free: the rational is always evaluated (finite for every finite x) and the two clamp
// tests of the reference select afterwards, in the reference's order
__device__ __forceinline__ float fast_tanh(float x) {
    const float x2 = x * x;
    const float a = x * (945.0f + x2 * (105.0f + x2));
    const float b = 945.0f + x2 * (420.0f + x2 * 15.0f);
    float r = __fdiv_rn(a, b);
    r = (x > 4.97f) ? 1.0f : r;
    r = (x < -4.97f) ? -1.0f : r;
    return r;
}

__device__ __forceinline__ float fast_atanh(float x) {
    const float x2 = x * x;
    const float a = x * (945.0f + x2 * (-735.0f + x2 * 64.0f));
    const float b = (945.0f + x2 * (-1050.0f + x2 * 225.0f));
    return __fdiv_rn(a, b);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ftx_compute_crc(a91 with bits 77.. cleared, 82 bits): CRC-14, polynomial 0x2757
__device__ inline uint32_t crc14_82(const uint8_t *msg) {
    uint32_t rem = 0;
    int idx_byte = 0;
    for (int bit = 0; bit < 82; ++bit) {
        if ((bit & 7) == 0) rem ^= (uint32_t)msg[idx_byte++] << 6;
        if (rem & 0x2000u) rem = ((rem << 1) ^ 0x2757u) & 0xFFFFu;
        else rem = (rem << 1) & 0xFFFFu;
    }
    return rem & 0x3FFFu;
}

constexpr int kTocFloats = 84 * 8;        // 83 rows x 8 slots (+1 spare row)
constexpr int kWaveLds = kTocFloats + 192;

__global__ __launch_bounds__(256)
void ft8_decode_kernel(const uint8_t *__restrict__ mag, const ft8gpu_candidate *__restrict__ cands,
                       const int32_t *__restrict__ counts, ft8gpu_decode_status *__restrict__ status,
                       int nframes, int max_candidates, int max_iters) {
    __shared__ __attribute__((aligned(16))) float s_mem[4][kWaveLds];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wave;
    const int frame = (int)(gw / max_candidates);
    const int ci = (int)(gw - (long)frame * max_candidates);
    if (frame >= nframes) return;
    if (ci >= counts[frame]) return;                        // wave-uniform

    float *toc = s_mem[wave];
    float *llr = toc + kTocFloats;

    const ft8gpu_candidate cand = cands[(size_t)frame * max_candidates + ci];

    // ---- ft8_extract_likelihood ------------------------------------------------------------
    if (lane < 58) {
        const int k = lane;
        const int sym = k + ((k < 29) ? 7 : 14);
        const int block = cand.time_offset + sym;
        int l0 = 0, l1 = 0, l2 = 0;
        if (block >= 0 && block < kNumBlocks) {
            const int index = ((cand.time_offset * 2 + cand.time_sub) * 2 + cand.freq_sub) * kNumBin + cand.freq_offset;
            const uint8_t *ps = mag + (size_t)frame * kMagArray + index + sym * kBlockStride;
            int s2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s2[j] = ps[c_gray[j]];
            l0 = max(max(s2[4], s2[5]), max(s2[6], s2[7])) - max(max(s2[0], s2[1]), max(s2[2], s2[3]));
            l1 = max(max(s2[2], s2[3]), max(s2[6], s2[7])) - max(max(s2[0], s2[1]), max(s2[4], s2[5]));
            l2 = max(max(s2[1], s2[3]), max(s2[5], s2[7])) - max(max(s2[0], s2[2]), max(s2[4], s2[6]));
        }
        llr[3 * k + 0] = (float)l0;
        llr[3 * k + 1] = (float)l1;
        llr[3 * k + 2] = (float)l2;
    }
    // check-row tile: padding slots stay 1.0f for the whole decode
    for (int i = lane; i < kTocFloats; i += 64) toc[i] = 1.0f;
    wave_lds_sync();

    // ---- ftx_normalize_logl ----------------------------------------------------------------
    float cw[3];
    bool has[3];
    int isum = 0, isum2 = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int n = lane + 64 * r;
        has[r] = n < kLdpcN;
        cw[r] = has[r] ? llr[n] : 0.0f;
        const int v = (int)cw[r];
        isum += v;
        isum2 += v * v;
    }
    const float sum = (float)wave_sum(isum);
    const float sum2 = (float)wave_sum(isum2);
    const float inv_n = 1.0f / 174;
    const float variance = (sum2 - (sum * sum * inv_n)) * inv_n;
    const float norm_factor = __fsqrt_rn(__fdiv_rn(24.0f, variance));
#pragma unroll
    for (int r = 0; r < 3; ++r) cw[r] = has[r] ? cw[r] * norm_factor : 0.0f;

    // ---- per-lane constant edge data ---------------------------------------------------------
    int slot[3][3], epos[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            slot[r][e] = d_tab.edge_slot[r][lane][e];
            epos[r][e] = d_tab.edge_pos[r][lane][e];
        }
    uint64_t rmask[2][3];
    bool rvalid[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        rvalid[r] = d_tab.row_valid[r][lane] != 0;
#pragma unroll
        for (int w = 0; w < 3; ++w) rmask[r][w] = d_tab.rowmask[r][lane][w];
    }

    // ---- bp_decode ---------------------------------------------------------------------------
    float tov[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tov[r][0] = tov[r][1] = tov[r][2] = 0.0f;
    int min_errors = kLdpcM;
    uint64_t B0 = 0, B1 = 0, B2 = 0;
    int iter = 0;
    for (; iter < max_iters; ++iter) {
        // hard decision (tov = 0 in iteration 0)
        bool bit[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            bit[r] = has[r] && ((((cw[r] + tov[r][0]) + tov[r][1]) + tov[r][2]) > 0.0f);
        B0 = __ballot(bit[0]);
        B1 = __ballot(bit[1]);
        B2 = __ballot(bit[2]);
        if ((B0 | B1 | B2) == 0ull) break;              // all-zero word is prohibited

        // ldpc_check
        int errors = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int par = (__popcll(B0 & rmask[r][0]) + __popcll(B1 & rmask[r][1]) + __popcll(B2 & rmask[r][2])) & 1;
            errors += __popcll(__ballot(rvalid[r] && par));
        }
        if (errors < min_errors) {
            min_errors = errors;
            if (errors == 0) break;
        }

        // messages from bits to check nodes: toc[m][n_idx] = fast_tanh(-Tnm / 2)
        // (lanes without a third variable compute on zeros and write to the spare row 83)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float t0 = (cw[r] + tov[r][1]) + tov[r][2];
            const float t1 = (cw[r] + tov[r][0]) + tov[r][2];
            const float t2 = (cw[r] + tov[r][0]) + tov[r][1];
            toc[slot[r][0]] = fast_tanh(-t0 / 2);
            toc[slot[r][1]] = fast_tanh(-t1 / 2);
            toc[slot[r][2]] = fast_tanh(-t2 / 2);
        }
        wave_lds_sync();
        // messages from check nodes to variable nodes: tov[n][m_idx] = -2 * fast_atanh(prod of the others)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float4 *row = reinterpret_cast<const float4 *>(toc + (slot[r][e] & ~7));
                const float4 lo = row[0], hi = row[1];
                const float v[7] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z };
                const int self = epos[r][e];
                float Tmn = 1.0f;
#pragma unroll
                for (int j = 0; j < 7; ++j) Tmn *= (j == self) ? 1.0f : v[j];
                tov[r][e] = -2 * fast_atanh(Tmn);
            }
        }
        wave_lds_sync();
    }

    // ---- pack_bits / CRC / unpack77 (one lane) -----------------------------------------------
    if (lane == 0) {
        ft8gpu_decode_status st;
        st.ldpc_errors = (int16_t)min_errors;
        st.iters = (int16_t)iter;
        st.crc_extracted = 0;
        st.crc_calculated = 0;
        st.unpack_status = 0;
        st.ok = 0;
        st.pad = 0;
        for (int i = 0; i < 25; ++i) st.text[i] = 0;
        // codeword bit i is bit (i & 63) of B{i >> 6}; packed MSB first
        const uint64_t w0 = __brevll(B0);                         // bits 0..63, MSB first
        const uint64_t w1 = __brevll(B1) & 0xFFFFFFE000000000ull; // bits 64..90
        uint8_t a91[12];
#pragma unroll
        for (int i = 0; i < 8; ++i) a91[i] = (uint8_t)(w0 >> (56 - 8 * i));
#pragma unroll
        for (int i = 0; i < 4; ++i) a91[8 + i] = (uint8_t)(w1 >> (56 - 8 * i));
#pragma unroll
        for (int i = 0; i < 12; ++i) st.a91[i] = a91[i];
        if (min_errors == 0) {
            st.crc_extracted = (uint16_t)(((a91[9] & 0x07) << 11) | (a91[10] << 3) | (a91[11] >> 5));
            a91[9] &= 0xF8;
            a91[10] = 0;
            a91[11] = 0;
            st.crc_calculated = (uint16_t)crc14_82(a91);
            if (st.crc_extracted == st.crc_calculated) {
                const int rc = ft8dev::unpack77(a91, st.text);
                st.unpack_status = (int8_t)rc;
                st.ok = rc >= 0 ? 1 : 0;
                if (rc < 0) for (int i = 0; i < 25; ++i) st.text[i] = 0;
            }
        }
        status[(size_t)frame * max_candidates + ci] = st;
    }
}

}  // namespace

hipError_t decode_tables_init(hipStream_t s) {
    static DecodeTables h;
    for (int r = 0; r < 3; ++r)
        for (int l = 0; l < 64; ++l) {
            const int n = l + 64 * r;
            for (int e = 0; e < 3; ++e) {
                if (n >= kLdpcN) { h.edge_slot[r][l][e] = (uint16_t)(83 * 8 + e); h.edge_pos[r][l][e] = 7; continue; }
                const int m = kFT8_Mn[n][e] - 1;
                int pos = -1;
                for (int j = 0; j < kFT8_Num_rows[m]; ++j)
                    if (kFT8_Nm[m][j] - 1 == n) pos = j;
                h.edge_slot[r][l][e] = (uint16_t)(m * 8 + pos);
                h.edge_pos[r][l][e] = (uint8_t)pos;
            }
        }
    for (int r = 0; r < 2; ++r)
        for (int l = 0; l < 64; ++l) {
            const int m = l + 64 * r;
            h.row_valid[r][l] = m < kLdpcM;
            h.rowmask[r][l][0] = h.rowmask[r][l][1] = h.rowmask[r][l][2] = 0;
            if (m >= kLdpcM) continue;
            for (int j = 0; j < kFT8_Num_rows[m]; ++j) {
                const int n = kFT8_Nm[m][j] - 1;
                h.rowmask[r][l][n >> 6] |= 1ull << (n & 63);
            }
        }
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(d_tab), &h, sizeof(h), 0, hipMemcpyHostToDevice, s);
}

hipError_t launch_decode(const uint8_t *mag, const ft8gpu_candidate *cands, const int32_t *counts,
                         ft8gpu_decode_status *status, int nframes, int max_candidates, int ldpc_iters,
                         hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    const long nwaves = (long)nframes * max_candidates;
    const unsigned grid = (unsigned)((nwaves + 3) / 4);
    hipLaunchKernelGGL(ft8_decode_kernel, dim3(grid), dim3(256), 0, s,
                       mag, cands, counts, status, nframes, max_candidates, ldpc_iters);
    return hipGetLastError();
}
