// decode.hip -- stage a7 of the hot path: ft8_decode() of ft8_lib decode.c,
// call site rtlsdr_ft8d.c:1476: soft-symbol extraction (ft8_extract_likelihood), LLR normalisation
// (ftx_normalize_logl), LDPC(174,91) sum-product decoding (bp_decode / ldpc_check / fast_tanh /
// fast_atanh of ft8_lib ldpc.c), CRC-14 (crc.c) and unpack77 (unpack.c).
//
// One wave64 per candidate codeword.  The kernel is bound by the number of VALU instructions per BP
// iteration (gfx950 issues a scalar-f32 wave64 VALU op in ~4 cycles, a packed v_pk_* op carrying
// two results in ~5), so the iteration is organised to minimise and to pack them:
//   * LLR: lane k < 58 owns data symbol k: 8 waterfall bytes -> three max-log differences.
//     The values are small integers, so the reference's sequential float sums in
//     ftx_normalize_logl (sum <= 174*255, sum of squares <= 174*255^2 < 2^24) are exact in any
//     order; they are reduced across the wave in integer arithmetic.
//   * BP, variable side: lane l owns variable nodes l, l+64, l+128, i.e. 9 edges, processed as
//     packed pairs (float2 -> v_pk_mul/add/fma_f32).  Variable-to-check messages stay in registers.
//   * BP, check side: lane m owns check rows m and m+64.  The reference's "product of all the other
//     edges of the row, in row order" is produced for all 7 members of a row at once from shared
//     prefixes (25 multiplies per row, no selects); unused 7th slots hold 1.0f (x * 1.0f is exact),
//     so every product has exactly the reference's multiplication order.
//     Exchange goes through one 84 x 8 float LDS tile stored as two float4 planes (row-owner
//     accesses are conflict-free ds_read/write_b128).
//   * hard decisions are gathered with three __ballot()s; a parity check is popcount(word & row
//     mask); the error count is the popcount of a ballot.  Exit conditions are wave-uniform.
//   * divisions: the IEEE-754 correctly rounded quotient is required for parity.  bp_math.h holds the two
//     rational functions with the shortest rcp/fma chains that were shown BY EXHAUSTION over all 2^32 inputs to
//     return the IEEE quotient on the domain a wave-uniform guard (see guard_key) establishes -- three operations
//     after v_rcp_f32 for fast_atanh, five for fast_tanh (round 2: seven each); outside that domain the
//     compiler's division runs.  (The only difference: a zero quotient may carry the other sign, which no later
//     operation can observe.)
//   * every other float expression is written in the reference's operation order and compiled
//     with -ffp-contract=off (fused operations appear only inside the division chain above).
#include "ft8gpu_internal.h"
#include "ft8_tables.h"
#include "unpack_dev.h"
#include "bp_math.h"
#include "ldpc_lds_layout.h"
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

namespace {

using bpm::f2;
using bpm::tanh_pair;
using bpm::tanh_one;
using bpm::atanh_pair;
using bpm::atanh_one;

// The pipeline only needs to know WHETHER a hard decision satisfies all 83 checks (the error count
// matters to nobody once it is non-zero).  The XOR of any set of parity rows is itself a parity
// condition every codeword satisfies, so the rows are folded into kCheckGroups wave-uniform masks
// that are tested on the scalar unit against the ballot words; only a word that passes all of them
// (every codeword does; a non-codeword with e failed rows does so with probability ~2^-(groups-1)
// when e is even and never when e is odd) goes through the exact per-row check on the vector unit.
// The number of masks is a template parameter of the kernel: 3 keeps the kernel at 78 SGPRs, and 256-thread
// workgroups are admitted 8 per CU only up to 80 (MI355X_MICROARCH.md, residency); 4 masks (84 SGPRs, 7
// workgroups per CU) screen twice as well but hold one wave per SIMD less.  Measured in one session: 5.000 ms
// against 5.002 ms per 4096-frame batch -- the kernel is bound by VALU issue either way; 3 is kept.
constexpr int kMaxCheckGroups = 4;

struct DecodeTables {
    uint16_t edge_slot[3][64][3];     // [r][lane][m_idx] -> float index of slot (m, pos) in the LDS tile
    uint64_t rowmask[2][64][3];       // [rr][lane][word] bit mask of the variables of check m = lane + 64 rr
    uint8_t  row_valid[2][64];
    uint8_t  own6[64], own7[64];      // product ownership: lane l computes the products of 6-member row own6[l] (59 rows) and of
                                      // 7-member row own7[l] (24 rows); kRows - 1 (the spare row) = none
    uint64_t group_mask[kMaxCheckGroups + 1][kMaxCheckGroups][3];   // [G][g]: XOR of the row masks of check rows m with m % G == g
    uint16_t crc_bit[77];             // CRC-14 (over 82 bits) of the message whose only set bit is payload bit i
};

__device__ DecodeTables d_tab;
__constant__ uint8_t c_gray[8] = { 0, 1, 3, 2, 5, 6, 4, 7 };

constexpr int kRows = 84;                     // 83 check rows + 1 spare row for idle lanes
constexpr int kTocFloats = kRows * 8;         // plane LO: [84] float4 (slots 0..3), plane HI: [84] float4 (slots 4..7)
constexpr int kWaveLds = kTocFloats + 192;    // + 174 LLRs

__host__ __device__ constexpr int slot_index(int m, int pos) {
    return pos < 4 ? 4 * m + pos : 4 * kRows + 4 * m + (pos - 4);
}

// min(|a|, |b|, |c|) in one instruction (no canonicalising copies of the operands)
__device__ __forceinline__ float min3_abs(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a + b as one v_add_f32 the vectoriser cannot see through
__device__ __forceinline__ float add_f32(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Guard key of a value: (bits << 1) - 1 as unsigned.  Zero maps to 0xFFFFFFFF, every other value to
// twice its magnitude bits minus one, so "minimum key over a set >= key(T)" says: each member is
// zero or at least T in magnitude.
//
// ONE guard per iteration, on the nine row products P of the lane, with T = 2^-59, covers both
// division sites of the following work:
//   * fast_atanh(P): numerator P*(945 - 735P^2 + 64P^4) with |P| <= 1.0072^6, so |numerator| >= 200|P|;
//   * the state ah = fast_atanh(P) (tov = -2*ah) then satisfies ah == 0 or |ah| >= 2^-59 (|atanh_r(P)| >= |P|),
//     and the halved LLRs cwh are 0 or >= 0.0095 (an integer times sqrt(24/variance)/2, variance <= 255^2).
//     Any sum of two or three floats that are each 0 or >= 2^-59 is 0 or >= 2^-82 (all are multiples of
//     2^-82), hence the next iteration's x is 0 or >= 2^-82, and fast_tanh's numerator x*(945 + ...) >= 945|x|.
// Both are far above v_div_scale's 2^-103 rescaling threshold.  Large, infinite and NaN values need
// no guard: |x| > 4.97 is overridden by fast_tanh's clamp in either division form, the products are
// bounded, and NaN stays NaN through both forms.  Iteration 0 starts from tov = 0.
__device__ __forceinline__ uint32_t guard_key(float v) { return (__float_as_uint(v) << 1) - 1u; }
constexpr uint32_t kGuardMin = ((127u - 59u) << 24) - 1u;       // guard_key(0x1p-59f)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 64 lanes with DPP adds (no LDS crossbar round trips); the total comes back uniform
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// the same reduction with XOR (CRC contributions of the set payload bits)
__device__ __forceinline__ uint32_t wave_xor(uint32_t x) {
    int v = (int)x;
    v ^= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// ftx_compute_crc(a91 with bits 77.. cleared, 82 bits): CRC-14, polynomial 0x2757.  Bit-serial
// restatement (only the first 77 bits can be set; five zero bits follow).  Runs on the host when the tables are
// built: the kernel uses the linearity of the CRC (init 0, no final XOR) -- the CRC of a message is the XOR of the
// CRCs of its set bits -- so each lane contributes the table entries of the payload bits it holds and one DPP
// reduction replaces 82 dependent shift/xor steps on a single lane.
__host__ __device__ inline uint32_t crc14_82(const uint8_t *msg) {
    uint32_t rem = 0;
    int idx_byte = 0;
    for (int bit = 0; bit < 82; ++bit) {
        if ((bit & 7) == 0) rem ^= (uint32_t)msg[idx_byte++] << 6;
        if (rem & 0x2000u) rem = ((rem << 1) ^ 0x2757u) & 0xFFFFu;
        else rem = (rem << 1) & 0xFFFFu;
    }
    return rem & 0x3FFFu;
}

// COUNT_ERRORS: ldpc_check() on every iteration with the exact number of failed rows (status
// ldpc_errors = minimum seen, as ft8_lib's decode_status_t reports it).  Otherwise (the batch pipeline,
// which consumes only ok / crc / text) the exact check runs only on words that pass the scalar group
// test, and ldpc_errors is 0 for a codeword and 83 otherwise; the exit iteration and every other field
// are the same in both forms.
template <bool COUNT_ERRORS, int kCheckGroups>
__global__ __launch_bounds__(256)
void ft8_decode_kernel(const uint8_t *__restrict__ mag, const ft8gpu_candidate *__restrict__ cands,
                       const int32_t *__restrict__ counts, ft8gpu_decode_status *__restrict__ status,
                       int nframes, int max_candidates, int max_iters, int force_ieee_div,
                       unsigned blocks_per_frame, unsigned bpf_magic) {
    __shared__ __attribute__((aligned(16))) float s_mem[4][kWaveLds];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: keep it in an SGPR
    // XCD-aware block order: the dispatcher places block b on XCD b % 8 (each XCD has its own L2), so
    // blocks are renumbered to give every XCD a contiguous range of candidates -- the 30 blocks that
    // gather from one frame's 94 KB waterfall then share one L2 instead of pulling it into all eight.
    // (Placement only affects cache traffic; any mapping is correct.)
    const unsigned nb = gridDim.x, per = nb >> 3, main_blocks = per << 3;
    const unsigned vb = blockIdx.x < main_blocks ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    // vb = frame * blocks_per_frame + group; bpf_magic = floor(2^32 / blocks_per_frame) + 1 makes the
    // quotient one multiply-high (exact for vb * blocks_per_frame < 2^32, checked at launch).  One block per frame
    // (max_candidates <= 4) has no such constant -- 2^32 + 1 does not fit 32 bits; as the truncated 1 it sent every block to
    // frame 0, found in round 6 by a soak over the whole accepted range of max_candidates -- and needs no division.
    const int frame = blocks_per_frame == 1u ? (int)vb : (int)__umulhi(vb, bpf_magic);
    const int ci = (int)(vb - (unsigned)frame * blocks_per_frame) * 4 + wave;
    if (frame >= nframes || ci >= max_candidates) return;
    if (ci >= counts[frame]) return;                        // wave-uniform

    float *toc = s_mem[wave];
    float *llr = toc + kTocFloats;
    float4 *planeLO = reinterpret_cast<float4 *>(toc);
    float4 *planeHI = planeLO + kRows;

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
    const float norm_factor = bpm::llr_norm_factor(variance);       // sqrtf(24.0f / variance), both correctly rounded (bp_math.h)
#pragma unroll
    for (int r = 0; r < 3; ++r) cw[r] = has[r] ? cw[r] * norm_factor : 0.0f;

    // ---- per-lane constant edge / row data ---------------------------------------------------
    int slot[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int e = 0; e < 3; ++e) slot[3 * r + e] = d_tab.edge_slot[r][lane][e];
    uint64_t gmask[kCheckGroups][3];
    if (!COUNT_ERRORS) {
#pragma unroll
        for (int g = 0; g < kCheckGroups; ++g)
#pragma unroll
            for (int w = 0; w < 3; ++w) gmask[g][w] = d_tab.group_mask[kCheckGroups][g][w];      // wave-uniform: scalar loads
    }
    uint64_t rmask[2][3];
    bool rvalid[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        rvalid[rr] = d_tab.row_valid[rr][lane] != 0;
        // the row masks are only needed by the exact check: the counting form uses them every iteration and
        // keeps them in registers; the pipeline form runs that check on about one iteration in eight and
        // fetches them from the (cache-resident) table then, which frees 12 VGPRs -- one more wave per SIMD
#pragma unroll
        for (int w = 0; w < 3; ++w) rmask[rr][w] = COUNT_ERRORS ? d_tab.rowmask[rr][lane][w] : 0ull;
    }

    // Ownership of the check-row products.  24 rows have seven members and 59 have six; a lane computes one row of
    // each kind with a program of exactly that length (25 and 18 multiplications) instead of two passes of the
    // seven-member program in which a six-member row multiplies by a stored 1.0f six times.
    const int row6 = d_tab.own6[lane], row7 = d_tab.own7[lane];
    const bool has6 = row6 != kRows - 1, has7 = row7 != kRows - 1;
    // the spare row only needs finite content (idle lanes of the variable side read and write it)
    if (lane < 8) toc[slot_index(kRows - 1, lane)] = 1.0f;
    wave_lds_sync();

    // ---- bp_decode ---------------------------------------------------------------------------
    // Half domain.  The reference adds tov = -2*atanh(.) to the LLR and then forms x = -Tnm/2.  Scaling
    // by a power of two commutes with every rounding as long as nothing is subnormal, so with
    // ah = atanh(.) and cwh = -cw/2 the very same bits come out of x = (cwh + ah_a) + ah_b, and the
    // hard decision (cw + tov0 + tov1 + tov2 > 0) is ((cwh + ah0) + ah1) + ah2 < 0.  fast_ok (all state
    // values 0 or >= 2^-59, see guard_key) guarantees the "nothing subnormal" premise; when it does
    // not hold the sums are formed in the reference's own domain from tov = -2*ah (an exact product).
    // Register layout of the nine edge states of a lane (variables r = 0,1,2; edges e = 0,1,2 of each):
    //   A[r] = (ah[r][2], ah[r][1])   B = (ah[0][0], ah[1][0])   c2 = ah[2][0]
    // chosen so that the sums below are packed adds on whole register pairs without any shuffling:
    //   u_r = cwh_r + ah[r][0]                    (u_0, u_1) = cwh01 + B
    //   X[r] = (x[r][1], x[r][2]) = u_r + A[r]    Tnm of edge 1 omits ah[r][1], of edge 2 omits ah[r][2]
    //   x[r][0] = (cwh_r + ah[r][1]) + ah[r][2]   -> Y = (x[0][0], x[1][0]),  z = x[2][0] (scalar)
    //   hard decision: (u_r + ah[r][1]) + ah[r][2] = X[r].y + A[r].x
    // Which values share a register pair is free (every value is computed by the same operations in
    // the same order whatever its neighbour is); the LDS slot of every edge is a per-lane constant.
    float cwh[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cwh[r] = cw[r] * -0.5f;
    const f2 cwh01 = { cwh[0], cwh[1] };
    const uint64_t has2_mask = __ballot(has[2]);         // lanes that own a third variable (n = lane + 128 < 174)
    int min_errors = kLdpcM;
    uint64_t B0 = 0, B1 = 0, B2 = 0;
    int iter = 0;
    bool fast_ok = !force_ieee_div;     // every state value is 0 or >= 2^-59 (true for the initial zeros)
    // the nine row products of the previous iteration, in the pairing of the states they turn into
    f2 PA[3] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } }, PB = { 0.0f, 0.0f };
    float pc = 0.0f;

    // One iteration up to the stores of toc, in the arithmetic form FAST selects (packed rcp/fma division and
    // half-domain sums, or IEEE division and the reference's own domain).  The two forms are complete,
    // separate instruction streams behind ONE wave-uniform branch per iteration: when they were chosen phase by
    // phase, the values live across the three branch points cost 14 VGPRs and a wave of occupancy.
    // Returns true when the loop ends here.
    auto first_half = [&](auto fast_tag) -> bool {
        constexpr bool FAST = decltype(fast_tag)::value;
        // ---- checks -> bits: tov[n][m_idx] = -2 * fast_atanh(product of the other toc of the row); the state
        // kept is fast_atanh(...) itself, i.e. tov = -2 * state exactly (half domain).  No messages before
        // the first iteration.
        f2 A[3], B;
        float c2;
        if (iter > 0) {                                  // wave-uniform
#pragma unroll
            for (int r = 0; r < 2; ++r) A[r] = atanh_pair<FAST>(PA[r]);
            B = atanh_pair<FAST>(PB);
            // The 18 lanes without a third variable (n = lane + 128 >= 174) sit its three edges out under the EXEC
            // mask, as do the 45 lanes without a second check row further down.  The instruction count is the same;
            // what it saves is switching power -- the kernel runs power-limited at about 2.0 GHz, and idle lanes
            // multiplying spare-row garbage cost clock: 1.7 % of the step in interleaved A/B runs.
            A[2] = f2{ 0.0f, 0.0f };
            c2 = 0.0f;
            if (has[2]) {
                A[2] = atanh_pair<FAST>(PA[2]);
                c2 = atanh_one<FAST>(pc);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) A[r] = f2{ 0.0f, 0.0f };
            B = f2{ 0.0f, 0.0f };
            c2 = 0.0f;
        }
        // ---- hard decision and Tnm / x for the lane's nine edges
        // (lanes without a third variable compute on spare-row content and write to the spare row)
        f2 X[3], Y;
        float z;
        if (FAST) {
            const f2 u01 = cwh01 + B;
            const float u2 = cwh[2] + c2;
            X[0] = f2{ u01.x, u01.x } + A[0];
            X[1] = f2{ u01.y, u01.y } + A[1];
            X[2] = f2{ u2, u2 } + A[2];
            // (scalar adds: written opaquely, otherwise the vectoriser pairs them up behind four v_mov)
            Y.x = add_f32(add_f32(cwh[0], A[0].y), A[0].x);
            Y.y = add_f32(add_f32(cwh[1], A[1].y), A[1].x);
            z = add_f32(add_f32(cwh[2], A[2].y), A[2].x);
            B0 = __ballot((X[0].y + A[0].x) < 0.0f);               // lanes 0..63 all own variables 0..127
            B1 = __ballot((X[1].y + A[1].x) < 0.0f);
            B2 = __ballot((X[2].y + A[2].x) < 0.0f) & has2_mask;
        } else {
            const float ah0[3] = { B.x, B.y, c2 };
            float x0[3];
            bool bit[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float v0 = -2 * ah0[r], v1 = -2 * A[r].y, v2 = -2 * A[r].x;   // tov
                const float u = cw[r] + v0;
                bit[r] = has[r] && (((u + v1) + v2) > 0.0f);
                x0[r] = ((cw[r] + v1) + v2) * -0.5f;            // == -Tnm / 2 bit for bit (scaling by a power of two)
                X[r].x = (u + v2) * -0.5f;
                X[r].y = (u + v1) * -0.5f;
            }
            Y.x = x0[0];
            Y.y = x0[1];
            z = x0[2];
            B0 = __ballot(bit[0]);
            B1 = __ballot(bit[1]);
            B2 = __ballot(bit[2]);
        }
        if ((B0 | B1 | B2) == 0ull) return true;        // all-zero word is prohibited

        // ldpc_check
        bool full_check = true;
        if (!COUNT_ERRORS) {
            uint32_t odd = 0;
#pragma unroll
            for (int g = 0; g < kCheckGroups; ++g)
                odd |= (uint32_t)__popcll((B0 & gmask[g][0]) ^ (B1 & gmask[g][1]) ^ (B2 & gmask[g][2]));
            full_check = (odd & 1u) == 0;               // wave-uniform (ballot words and masks are scalars)
        }
        if (full_check) {
            int errors = 0;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const uint64_t m0 = COUNT_ERRORS ? rmask[rr][0] : d_tab.rowmask[rr][lane][0];
                const uint64_t m1 = COUNT_ERRORS ? rmask[rr][1] : d_tab.rowmask[rr][lane][1];
                const uint64_t m2 = COUNT_ERRORS ? rmask[rr][2] : d_tab.rowmask[rr][lane][2];
                const int par = (__popcll(B0 & m0) + __popcll(B1 & m1) + __popcll(B2 & m2)) & 1;
                errors += __popcll(__ballot(rvalid[rr] && par));
            }
            if (COUNT_ERRORS) {
                if (errors < min_errors) {
                    min_errors = errors;
                    if (errors == 0) return true;
                }
            } else if (errors == 0) {
                min_errors = 0;
                return true;
            }
        }
        // The reference's last iteration still updates both message arrays, which nothing reads afterwards:
        // the last hard decision has been taken and checked at this point.
        if (iter + 1 >= max_iters) { iter = max_iters; return true; }

        // ---- bits -> checks: toc[m][n_idx] = fast_tanh(-Tnm / 2)
        f2 t[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) t[r] = tanh_pair<FAST>(X[r]);
        t[3] = tanh_pair<FAST>(Y);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            toc[slot[3 * r + 1]] = t[r].x;
            toc[slot[3 * r + 2]] = t[r].y;
        }
        toc[slot[0]] = t[3].x;
        toc[slot[3]] = t[3].y;
        if (has[2]) {
            t[2] = tanh_pair<FAST>(X[2]);
            const float tz = tanh_one<FAST>(z);
            toc[slot[7]] = t[2].x;
            toc[slot[8]] = t[2].y;
            toc[slot[6]] = tz;
        }
        return false;
    };

    for (;; ++iter) {
        if (iter >= max_iters) break;                    // (max_iters >= 1 is checked by the API; the loop leaves through first_half)
        const bool stop = fast_ok ? first_half(std::true_type{}) : first_half(std::false_type{});
        if (stop) break;
        wave_lds_sync();

        // ---- check rows: ordered products that skip one member, for all members ---------------
        // o_k = product of the row's other members in row order: the prefixes p_k = v0 * ... * v(k-1) are shared, every
        // o_k then continues its own chain (a shared suffix would change the association, i.e. the roundings).  Chains of
        // equal length run as packed pairs: (o0, o1), (o2, o3) from (p2 * v3, p3), (o4, o5).
        if (has6) {                                                           // 59 lanes (EXEC mask)
            const float4 lo = planeLO[row6], hi = planeHI[row6];
            const float v0 = lo.x, v1 = lo.y, v2 = lo.z, v3 = lo.w, v4 = hi.x, v5 = hi.y;
            const f2 o01 = (((f2{ v1, v0 } * v2) * v3) * v4) * v5;           // skip 0 | skip 1
            const float p2 = v0 * v1;
            const float p3 = p2 * v2;
            const float p4 = p3 * v3;
            const f2 o23 = (f2{ p2 * v3, p3 } * v4) * v5;                     // skip 2 | skip 3
            const float o4 = p4 * v5, o5 = p4 * v4;
            planeLO[row6] = make_float4(o01.x, o01.y, o23.x, o23.y);
            *reinterpret_cast<float2 *>(planeHI + row6) = make_float2(o4, o5);
        }
        if (has7) {                                                           // 24 lanes
            const float4 lo = planeLO[row7], hi = planeHI[row7];
            const float v0 = lo.x, v1 = lo.y, v2 = lo.z, v3 = lo.w, v4 = hi.x, v5 = hi.y, v6 = hi.z;
            const f2 o01 = ((((f2{ v1, v0 } * v2) * v3) * v4) * v5) * v6;
            const float p2 = v0 * v1;
            const float p3 = p2 * v2;
            const float p4 = p3 * v3;
            const float p5 = p4 * v4;
            const f2 o23 = ((f2{ p2 * v3, p3 } * v4) * v5) * v6;
            const f2 o45 = f2{ p4 * v5, p5 } * v6;                            // skip 4 | skip 5
            const float o6 = p5 * v5;
            planeLO[row7] = make_float4(o01.x, o01.y, o23.x, o23.y);
            planeHI[row7] = make_float4(o45.x, o45.y, o6, 1.0f);
        }
        wave_lds_sync();

        // ---- the lane's nine products come back; they become messages at the top of the next iteration -------
#pragma unroll
        for (int r = 0; r < 2; ++r) PA[r] = f2{ toc[slot[3 * r + 2]], toc[slot[3 * r + 1]] };
        PB = f2{ toc[slot[0]], toc[slot[3]] };
        PA[2] = f2{ toc[slot[8]], toc[slot[7]] };
        pc = toc[slot[6]];
        // Lanes without a third variable run these three edges on whatever the spare row holds; nothing
        // they compute leaves the spare row or their own registers (their decision bit is masked), so the
        // guard ignores them.
        // Quick form of the guard: the smallest magnitude of the nine products (four v_min3_f32 with |.|
        // modifiers; a NaN operand is skipped).  If it is at least 2^-59 there is neither a tiny value nor a
        // zero and the guard holds; otherwise (exact zeros are common in the first two iterations, rare
        // afterwards) the exact key test below decides.
        float mabs = min3_abs(pc, PA[2].x, PA[2].y);
        mabs = has[2] ? mabs : __builtin_inff();
        mabs = min3_abs(mabs, PB.x, PB.y);
        mabs = min3_abs(mabs, PA[0].x, PA[0].y);
        mabs = min3_abs(mabs, PA[1].x, PA[1].y);
        bool guard_ok = __all(mabs >= 0x1p-59f);
        if (!guard_ok) {                                              // wave-uniform
            uint32_t g2 = min(guard_key(pc), min(guard_key(PA[2].x), guard_key(PA[2].y)));
            g2 = has[2] ? g2 : 0xFFFFFFFFu;
            uint32_t gmin = min(g2, min(guard_key(PB.x), guard_key(PB.y)));
#pragma unroll
            for (int r = 0; r < 2; ++r) gmin = min(gmin, min(guard_key(PA[r].x), guard_key(PA[r].y)));
            guard_ok = __all(gmin >= kGuardMin);
        }
        fast_ok = guard_ok && !force_ieee_div;                        // wave-uniform; governs the whole next iteration
        // (the next iteration's toc stores hit only this lane's own slots; LDS is in order per wave)
    }

    // ---- pack_bits / CRC / unpack77 -----------------------------------------------------------
    // codeword bit i is bit (i & 63) of B{i >> 6}: lane l holds payload bits l and (for l < 13) 64 + l
    uint32_t crc_calc = 0;
    if (min_errors == 0) {                                       // wave-uniform
        uint32_t c = ((B0 >> lane) & 1ull) ? d_tab.crc_bit[lane] : 0u;
        if (lane < 13 && ((B1 >> lane) & 1ull)) c ^= d_tab.crc_bit[64 + lane];
        crc_calc = wave_xor(c);
    }
    // The 48-byte record is composed in the wave's LDS tile (the LLR area is free now) and leaves as one contiguous
    // 12-dword burst.  Nothing of it lives in private memory: min_errors, iter and the ballot words are wave-uniform
    // scalars, the bytes of a91 are shifts of two 64-bit words, and unpack77 works on those words and stores the characters
    // it computes (in registers) straight into the record's text -- the kernel has no scratch segment and the epilogue
    // never waits for a load (tools/kernel_resources.py, tests/test_kernel_resources.py).
    static_assert(sizeof(ft8gpu_decode_status) == 48, "record is 12 dwords");
    static_assert(offsetof(ft8gpu_decode_status, a91) == 10 && offsetof(ft8gpu_decode_status, text) == 22, "record layout");
    uint32_t *rec32 = reinterpret_cast<uint32_t *>(llr);
    char *rec = reinterpret_cast<char *>(llr);
    // codeword bit i is bit (i & 63) of B{i >> 6}; packed MSB first
    const uint64_t w0 = __brevll(B0);                                 // bits 0..63, MSB first
    const uint64_t w1 = __brevll(B1) & 0xFFFFFFE000000000ull;         // bits 64..90
    if (lane < 12) {
        // dwords 0..5 hold the fixed fields and a91 (bytes 10..21), the rest is text / pad: zero
        const uint32_t hi0 = (uint32_t)(w0 >> 32), lo0 = (uint32_t)w0, hi1 = (uint32_t)(w1 >> 32);
        // a91[k] = byte k of (w0, w1) in big-endian order; record byte 10 + k
        uint32_t v = 0;
        if (lane == 0) v = ((uint32_t)min_errors & 0xFFFFu) | ((uint32_t)iter << 16);
        else if (lane == 2) v = (__builtin_bswap32(hi0) & 0xFFFFu) << 16;                                    // a91[0..1]
        else if (lane == 3) v = (__builtin_bswap32(hi0) >> 16) | (__builtin_bswap32(lo0) << 16);              // a91[2..5]
        else if (lane == 4) v = (__builtin_bswap32(lo0) >> 16) | (__builtin_bswap32(hi1) << 16);              // a91[6..9]
        else if (lane == 5) v = __builtin_bswap32(hi1) >> 16;                                                 // a91[10..11]
        rec32[lane] = v;
    }
    if (min_errors == 0) {                                            // wave-uniform
        const uint32_t crc_extracted = (uint32_t)(w1 >> 37) & 0x3FFFu;   // bits 77..90
        // lane 0 now rewrites dwords other lanes have just stored (1, 5, and the text behind them): order the two across the
        // wave explicitly instead of leaning on a wave's LDS instructions issuing in program order (converged candidates only)
        wave_lds_sync();
        if (lane == 0) {
            rec32[1] = crc_extracted | (crc_calc << 16);
            if (crc_extracted == crc_calc) {
                const int rc = ft8dev::unpack77(w0, w1 & 0xFFF8000000000000ull, rec + offsetof(ft8gpu_decode_status, text));   // (text bytes are zero: see above)
                if (rc < 0) {
                    // a failed unpack may have left characters behind: text is all zeros unless ok
                    rec32[5] &= 0xFFFFu;
                    for (int i = 6; i < 12; ++i) rec32[i] = 0;
                }
                rec[offsetof(ft8gpu_decode_status, unpack_status)] = (char)rc;
                rec[offsetof(ft8gpu_decode_status, ok)] = rc >= 0 ? 1 : 0;
                rec[offsetof(ft8gpu_decode_status, pad)] = 0;
            }
        }
    }
    wave_lds_sync();
    if (lane < 12)
        reinterpret_cast<uint32_t *>(status + (size_t)frame * max_candidates + ci)[lane] = rec32[lane];
}

}  // namespace

hipError_t decode_tables_init(hipStream_t s) {
    static DecodeTables h;
    for (int r = 0; r < 3; ++r)
        for (int l = 0; l < 64; ++l) {
            const int n = l + 64 * r;
            for (int e = 0; e < 3; ++e) {
                if (n >= kLdpcN) { h.edge_slot[r][l][e] = (uint16_t)slot_index(kRows - 1, e); continue; }
                const int m = kFT8_Mn[n][e] - 1;
                int pos = -1;
                for (int j = 0; j < kFT8_Num_rows[m]; ++j)
                    if (kFT8_Nm[m][j] - 1 == n) pos = j;
                // the row's place in the LDS tile comes from the conflict-minimising layout (ldpc_lds_layout.h); its members keep their order
                h.edge_slot[r][l][e] = (uint16_t)slot_index(kLdsRowPos[m], pos);
            }
        }
    for (int rr = 0; rr < 2; ++rr)
        for (int l = 0; l < 64; ++l) {
            const int m = l + 64 * rr;
            h.row_valid[rr][l] = m < kLdpcM;
            h.rowmask[rr][l][0] = h.rowmask[rr][l][1] = h.rowmask[rr][l][2] = 0;
            if (m >= kLdpcM) continue;
            for (int j = 0; j < kFT8_Num_rows[m]; ++j) {
                const int n = kFT8_Nm[m][j] - 1;
                h.rowmask[rr][l][n >> 6] |= 1ull << (n & 63);
            }
        }
    {
        // which lane multiplies which row: also from the layout search (the float4 accesses of the owners want distinct positions
        // mod 16 within their lane groups).  Checked here: every row is owned exactly once by a lane of the right kind, every
        // position is used once, the spare position stays spare.
        int owned[kLdpcM] = { 0 }, used[kRows] = { 0 };
        for (int l = 0; l < 64; ++l) {
            h.own6[l] = h.own7[l] = (uint8_t)(kRows - 1);
            const int m6 = kOwn6Row[l], m7 = kOwn7Row[l];
            if (m6 != 255) { if (m6 >= kLdpcM || kFT8_Num_rows[m6] != 6) abort(); h.own6[l] = kLdsRowPos[m6]; ++owned[m6]; }
            if (m7 != 255) { if (m7 >= kLdpcM || kFT8_Num_rows[m7] != 7) abort(); h.own7[l] = kLdsRowPos[m7]; ++owned[m7]; }
        }
        for (int m = 0; m < kLdpcM; ++m) { if (owned[m] != 1 || kLdsRowPos[m] >= kRows - 1 || used[kLdsRowPos[m]]++) abort(); }
        if (kLdsRowPos[kRows - 1] != kRows - 1) abort();
        // The search can also move variable nodes between lanes (34 instead of 66 extra cycles); the kernel keeps variable n on lane
        // n mod 64, slot n / 64 -- codeword order falls out of the ballots for free -- so the header must have been generated with
        // --fixed-variables (DESIGN.md section 4 has the estimate that decided it).
        for (int r = 0; r < 3; ++r)
            for (int l = 0; l < 64; ++l)
                if (kVarOf[r][l] != (l + 64 * r < kLdpcN ? l + 64 * r : 255)) abort();
    }
    for (int G = 1; G <= kMaxCheckGroups; ++G) {
        for (int g = 0; g < kMaxCheckGroups; ++g)
            for (int w = 0; w < 3; ++w) h.group_mask[G][g][w] = 0;
        for (int m = 0; m < kLdpcM; ++m)
            for (int j = 0; j < kFT8_Num_rows[m]; ++j) {
                const int n = kFT8_Nm[m][j] - 1;
                h.group_mask[G][m % G][n >> 6] ^= 1ull << (n & 63);
            }
    }
    for (int i = 0; i < 77; ++i) {
        uint8_t m[12] = { 0 };
        m[i >> 3] = (uint8_t)(0x80u >> (i & 7));                 // payload bit i, MSB first (pack_bits order)
        h.crc_bit[i] = (uint16_t)crc14_82(m);
    }
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(d_tab), &h, sizeof(h), 0, hipMemcpyHostToDevice, s);
}

hipError_t launch_decode(const uint8_t *mag, const ft8gpu_candidate *cands, const int32_t *counts,
                         ft8gpu_decode_status *status, int nframes, int max_candidates, int ldpc_iters,
                         bool count_errors, int force_ieee_div, hipStream_t s) {
    // force_ieee_div (FT8GPU_DBG_FORCE_IEEE_DIV) routes every BP division through the compiler's IEEE
    // expansion (the path the guard falls back to); used by the parity tests to cover that path
    if (nframes < 1) return hipSuccess;
    const unsigned bpf = (unsigned)(max_candidates + 3) / 4;                  // blocks (of 4 candidate waves) per frame
    const unsigned long long nblocks = (unsigned long long)nframes * bpf;
    if (nblocks * bpf >= (1ull << 32)) return hipErrorInvalidValue;           // keeps the multiply-high division exact
    const unsigned magic = bpf == 1u ? 0u : (unsigned)((1ull << 32) / bpf) + 1u;          // bpf >= 2: at most 2^31 + 1
    if (count_errors)
        hipLaunchKernelGGL((ft8_decode_kernel<true, 1>), dim3((unsigned)nblocks), dim3(256), 0, s,
                           mag, cands, counts, status, nframes, max_candidates, ldpc_iters, force_ieee_div, bpf, magic);
    else
        hipLaunchKernelGGL((ft8_decode_kernel<false, 3>), dim3((unsigned)nblocks), dim3(256), 0, s,
                           mag, cands, counts, status, nframes, max_candidates, ldpc_iters, force_ieee_div, bpf, magic);
    return hipGetLastError();
}
