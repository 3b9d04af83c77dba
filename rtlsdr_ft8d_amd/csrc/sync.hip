// sync.hip -- stage a5 of the hot path: ft8_find_sync() of ft8_lib decode.c,
// call site rtlsdr_ft8d.c:1450 (waterfall descriptor rtlsdr_ft8d.c:1440-1448).
//
// Two kernels:
//   ft8_sync_kernel  two workgroups per (frame, time_sub, freq_sub), each scoring half of the time offsets.  ft8_sync_score() sums, over three
//                    Costas blocks m and seven symbols k, neighbour contrasts of the waterfall at block
//                    b = t0 + 36 m + k.  Which terms exist depends on b and k only, never on the
//                    frequency, so the sum over m is taken ONCE per (t0 + k) while the slice streams
//                    through registers ("collapsed maps" in LDS), and a position's score numerator is the
//                    sum of seven map cells.  A lane scores four adjacent frequency offsets at once with
//                    packed int16 adds; whether trunc(numerator / terms) reaches min_score is decided
//                    exactly on the numerator, and only surviving positions get their quotient.  Survivors
//                    are compacted in the reference's scan order with __ballot/popcount prefix sums (no atomics).
//   ft8_heap_kernel  replays the reference's bounded min-heap (strict '>' replacement, its
//                    heapify tie rules and the final heap sort) over the compacted list, one
//                    wave per frame with the heap in LDS, so that candidate order is bit-identical
//                    to ft8_find_sync().
#include "ft8gpu_internal.h"
#include <stdlib.h>

namespace {

// For a block b and column f of the (time_sub, freq_sub) slice p[92][256]:
//   dl = [f>0](p - p[f-1])   dr = p - p[f+1]   du = [b>0](p - p[b-1])   dd = [b<91](p - p[b+1])
// A sync symbol k at block b contributes (Costas tones are 0..6, so the upper bin always exists):
//   k in {1,2,4,5}:  dl + dr + du + dd          k = 0 (no look-back inside a Costas block):  dl + dr + dd
//   k = 6 (no look-ahead):  dl + dr + du         k = 3 (tone 0 has no lower bin):             dr + du + dd
// and nothing when b is outside [0, 92).  With t' = t0 + k the block is b = t' + 36 m, hence
//   score numerator(t0, f0) = M0[t0][f0+3] + MA[t0+1][f0+1] + MA[t0+2][f0+4] + M3[t0+3][f0]
//                           + MA[t0+4][f0+6] + MA[t0+5][f0+5] + M6[t0+6][f0+2]          (pattern 3,1,4,0,6,5,2)
// where each map is the sum of its symbol's contribution over the (up to three) m with b in range:
//   MA[t'] for t' in [-11, 29)   M0[t'] for t' in [-12, 24)   M3[t'] for t' in [-9, 27)   M6[t'] for t' in [-6, 30)
// Magnitudes: |cell| <= 3 * 4 * 255 = 3060, so int16 holds a cell and packed 16-bit arithmetic is exact.
// A workgroup scores the time offsets t0 in [h0, h0 + 18) of one (time_sub, freq_sub) slice, h0 = -12 or 6 (two
// workgroups per slice): it needs MA[t'] for t' in [h0 + 1, h0 + 23), M0 for [h0, h0 + 18), M3 for [h0 + 3, h0 + 21),
// M6 for [h0 + 6, h0 + 24) -- 76 rows = 38.9 KB instead of the 148 rows (75.8 KB) of the whole range, i.e. four
// resident workgroups (32 waves) per CU instead of two, for 14 % more map rows built in total.
constexpr int kMapPitch = 256;                            // int16 per map row
constexpr int kRowsA = kT0PerHalf + 4, kRowsK = kT0PerHalf;   // rows of MA (22) / of M0, M3, M6 (18)
constexpr int kOffA = 0, kOff0 = kRowsA, kOff3 = kRowsA + kRowsK, kOff6 = kRowsA + 2 * kRowsK;
constexpr int kMapRows = kRowsA + 3 * kRowsK;             // 76 rows
constexpr int kTqCount = kT0PerHalf + 6;                  // t' = h0 + tq for tq in [0, 24): 3 per wave

typedef short s16x2 __attribute__((ext_vector_type(2)));

// two waterfall bytes -> two zero-extended 16-bit lanes (v_perm_b32; selector bytes 0..3 pick from
// `lo`, 4..7 from `hi`, 0x0c yields 0)
__device__ __forceinline__ s16x2 bytes2(uint32_t hi, uint32_t lo, uint32_t selector) {
    const uint32_t v = __builtin_amdgcn_perm(hi, lo, selector);
    return __builtin_bit_cast(s16x2, v);
}

// number of neighbour terms ft8_sync_score() averages over, for time offset t0 (independent of f0)
__device__ __forceinline__ int sync_navg(int t0) {
    int n = 0;
    for (int m = 0; m < 3; ++m)
        for (int k = 0; k < 7; ++k) {
            const int b = t0 + 36 * m + k;
            if (b < 0) continue;
            if (b >= kNumBlocks) break;
            n += (k != 3) + 1;                          // sm > 0 ; sm < 7 always (Costas tones are 0..6)
            n += (k > 0 && b > 0);
            n += (k + 1 < 7 && b + 1 < kNumBlocks);
        }
    return n;
}

// ---- the sync kernel -----------------------------------------------------------------------------------
//   * build: every row a wave needs (its 5-6 consecutive t' plus one halo row either side, for the three
//     Costas blocks: 24 dwords per lane) is requested up front; each row is widened to int16 once and slides
//     through a (previous, current, next) window; the horizontal neighbours are taken from the SUM over the
//     Costas blocks (a column shift commutes with that sum), not per block;
//   * score: a lane owns FOUR adjacent frequency offsets (4 x 64 = 256 >= 249: one pass per time offset
//     instead of four), read as 64-bit groups of int16 cells and added as packed pairs.  The offsets 4, 6, 2
//     and 0 of the Costas pattern land on dword boundaries; 1, 5 and 3 are funnel-shifted out of two
//     neighbouring dwords (v_alignbit).  The integer division by the number of averaged terms is NOT
//     evaluated per position: trunc(num / n) >= min_score is decided exactly on the numerator
//     (num >= T with T = min_score * n for min_score > 0, (min_score - 1) * n + 1 otherwise; saturating
//     packed subtraction, sign bits), and only rows with a survivor -- about half of them hold one or two
//     -- compute quotients, ranks (ballot + prefix popcount per cell column, scan order preserved) and
//     store list entries.
typedef uint32_t u32;

__device__ __forceinline__ s16x2 as_s16x2(u32 v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u32 as_u32(s16x2 v) { return __builtin_bit_cast(u32, v); }
// ({hi, lo} >> 16) & 0xFFFFFFFF: (lo.hi16, hi.lo16) as (low half, high half)
__device__ __forceinline__ u32 funnel16(u32 hi, u32 lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

template <bool SCORE_MAP>
__global__ __launch_bounds__(64 * kSyncWaves)
void ft8_sync_kernel(const uint8_t *__restrict__ mag, uint32_t *__restrict__ lists,
                        int32_t *__restrict__ list_counts, int16_t *__restrict__ score_map, int min_score) {
    __shared__ __attribute__((aligned(16))) int16_t s_map[kMapRows * kMapPitch + 16];   // + the two dwords lane 63 reads past a row
    __shared__ int s_navg[kT0Count];                             // terms averaged by ft8_sync_score() per time offset

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The two workgroups of a slice read the same waterfall rows (the second one 60 % of them); the dispatcher places
    // block x on XCD x % 8 and every XCD has its own L2, so the halves of slice u sit 8 blocks apart -- same XCD, second
    // read from L2 -- whenever the grid is a whole number of such groups of 16.  (Placement only affects cache traffic.)
    const unsigned x = blockIdx.x;
    const bool paired = (gridDim.x & 15u) == 0u;
    const unsigned unit = paired ? (((x >> 4) << 3) | (x & 7u)) : (x >> 1);
    const int half = paired ? (int)((x >> 3) & 1u) : (int)(x & 1u);
    const int frame = (int)(unit >> 2), seg = (int)(unit & 3u);
    const int h0i = half * kT0PerHalf, h0 = h0i + kT0Min;        // first time offset of this workgroup: index and value
    const int ts = seg >> 1, fs = seg & 1;
    if (tid < kT0Count) s_navg[tid] = sync_navg(tid + kT0Min);   // (a loop of scalar branches when evaluated per row)

    // ---- build the collapsed maps ------------------------------------------------------------------
    {
        const u32 *rows = reinterpret_cast<const u32 *>(mag + (size_t)frame * kMagArray + ts * 512 + fs * 256) + lane;
        const int tq_begin = (wave * kTqCount) / kSyncWaves, tq_end = ((wave + 1) * kTqCount) / kSyncWaves;
        const int ntq = tq_end - tq_begin;                       // 3
        constexpr int kMaxTq = (kTqCount + kSyncWaves - 1) / kSyncWaves;   // 3
        // row j of Costas block m is block b = h0 + tq_begin + 36 m - 1 + j, j in [0, ntq + 2)
        u32 raw[3][kMaxTq + 2];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int j = 0; j < kMaxTq + 2; ++j) {
                const int b = h0 + tq_begin + 36 * m - 1 + j;   // wave-uniform
                raw[m][j] = (j < ntq + 2 && b >= 0 && b < kNumBlocks) ? rows[(size_t)b * (kBlockStride / 4)] : 0u;
            }
        // (c0, c1) and (c2, c3) of a row as int16 pairs
        auto widen = [](u32 v, s16x2 (&e)[2]) { e[0] = bytes2(0, v, 0x0c010c00u); e[1] = bytes2(0, v, 0x0c030c02u); };
        s16x2 prev[3][2], cur[3][2], next[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m) { widen(raw[m][0], prev[m]); widen(raw[m][1], cur[m]); }
#pragma unroll
        for (int i = 0; i < kMaxTq; ++i) {
            if (i < ntq) {                                       // wave-uniform
                const int tq = tq_begin + i, tp = h0 + tq;       // t'
                const s16x2 zero = { 0, 0 };
                s16x2 C[2] = { zero, zero }, Up[2] = { zero, zero }, Dn[2] = { zero, zero };
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    widen(raw[m][i + 2], next[m]);
                    const int b = tp + 36 * m;
                    if (b < 0 || b >= kNumBlocks) continue;      // wave-uniform
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        C[h] += cur[m][h];
                        Up[h] += b > 0 ? prev[m][h] : cur[m][h];                  // missing time neighbour: p - p = 0
                        Dn[h] += b + 1 < kNumBlocks ? next[m][h] : cur[m][h];
                    }
                }
                // horizontal neighbours of the summed cells (c0..c3 of this lane): left (c-1, c0, c1, c2), right (c1, c2, c3, c4)
                const u32 c01 = as_u32(C[0]), c23 = as_u32(C[1]);
                // lane 0 has no lower bin (p - p = 0: its left neighbour is c0 itself); lane 63's right neighbour (column 256) is never used
                const u32 lprev = (u32)__builtin_amdgcn_update_dpp((int)(c01 << 16), (int)c23, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                const u32 rnext = (u32)__builtin_amdgcn_update_dpp(0, (int)c01, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
                const s16x2 mid12 = as_s16x2(funnel16(c23, c01));                 // (c1, c2)
                const s16x2 L[2] = { as_s16x2(funnel16(c01, lprev)), mid12 };     // (c-1, c0), (c1, c2)
                const s16x2 R[2] = { mid12, as_s16x2(funnel16(rnext, c23)) };     // (c1, c2), (c3, c4)
                s16x2 S[2], U[2], V[2], W[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    W[h] = C[h] - L[h];                                           // sum of dl
                    U[h] = C[h] - Up[h];                                          // sum of du
                    V[h] = C[h] - Dn[h];                                          // sum of dd
                    S[h] = (W[h] + (C[h] - R[h])) + (U[h] + V[h]);                // sum of dl + dr + du + dd
                }
                auto store = [&](int row, s16x2 a, s16x2 b2) {
                    uint2 v;
                    v.x = as_u32(a);
                    v.y = as_u32(b2);
                    *reinterpret_cast<uint2 *>(s_map + row * kMapPitch + 4 * lane) = v;
                };
                if (tq >= 1 && tq < kRowsA + 1) store(kOffA + tq - 1, S[0], S[1]);                        // MA[t'], t' in [h0 + 1, h0 + 23)
                if (tq < kRowsK)                store(kOff0 + tq, S[0] - U[0], S[1] - U[1]);              // M0[t'], [h0, h0 + 18)
                if (tq >= 3 && tq < kRowsK + 3) store(kOff3 + tq - 3, S[0] - W[0], S[1] - W[1]);          // M3[t'], [h0 + 3, h0 + 21)
                if (tq >= 6)                    store(kOff6 + tq - 6, S[0] - V[0], S[1] - V[1]);          // M6[t'], [h0 + 6, h0 + 24)
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { prev[m][h] = cur[m][h]; cur[m][h] = next[m][h]; }
            }
        }
        if (tid < 8) reinterpret_cast<u32 *>(s_map + kMapRows * kMapPitch)[tid] = 0u;   // the pad behind the last row
    }
    __syncthreads();

    // ---- score all positions: lane = frequency offsets 4 lane .. 4 lane + 3 ----------------------------
    const int sub = (seg * kSyncHalves + half) * kSyncWaves + wave;      // scan order: segment, then time offset
    uint32_t *my_list = lists + ((size_t)frame * kSublistsPerFrame + sub) * kSublistCap;
    int count = 0;
    const int t_begin = (wave * kT0PerHalf) / kSyncWaves, t_end = ((wave + 1) * kT0PerHalf) / kSyncWaves;   // 2 or 3 local offsets
    // sign-bit masks of the cells that are real positions (f0 < 249): all four up to lane 61, one in lane 62
    const u32 vm_lo = lane < 62 ? 0x80008000u : (lane == 62 ? 0x00008000u : 0u);
    const u32 vm_hi = lane < 62 ? 0x80008000u : 0u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (int tl = t_begin; tl < t_end; ++tl) {                   // scan order: time_offset ascending
        const int t0i = h0i + tl;                                // index into the 36 time offsets
        const int navg = __builtin_amdgcn_readfirstlane(s_navg[t0i]);
        // trunc(num / navg) >= min_score  <=>  num >= T   (C division truncates toward zero; navg = 0 leaves the sum undivided)
        int T = navg > 0 ? (min_score > 0 ? min_score * navg : (min_score - 1) * navg + 1) : min_score;
        T = T > 32767 ? 32767 : (T < -32768 ? -32768 : T);       // |num| <= 21420: saturated thresholds mean never / always
        const s16x2 Tpk = { (short)T, (short)T };
        const u32 *pa0 = reinterpret_cast<const u32 *>(s_map + (kOffA + tl + 0) * kMapPitch) + 2 * lane;   // t' = t0 + 1
        const u32 *pa1 = reinterpret_cast<const u32 *>(s_map + (kOffA + tl + 1) * kMapPitch) + 2 * lane;   // t0 + 2
        const u32 *pa3 = reinterpret_cast<const u32 *>(s_map + (kOffA + tl + 3) * kMapPitch) + 2 * lane;   // t0 + 4
        const u32 *pa4 = reinterpret_cast<const u32 *>(s_map + (kOffA + tl + 4) * kMapPitch) + 2 * lane;   // t0 + 5
        const u32 *p0 = reinterpret_cast<const u32 *>(s_map + (kOff0 + tl) * kMapPitch) + 2 * lane;
        const u32 *p3 = reinterpret_cast<const u32 *>(s_map + (kOff3 + tl) * kMapPitch) + 2 * lane;
        const u32 *p6 = reinterpret_cast<const u32 *>(s_map + (kOff6 + tl) * kMapPitch) + 2 * lane;
        // cells f0 + off .. f0 + off + 3 of a row as two packed pairs; dword d of the lane's pointer holds cells 2d, 2d + 1
        const u32 a0 = pa0[0], a1 = pa0[1], a2 = pa0[2];         // MA[t0+1], offset 1
        const u32 b2 = pa1[2], b3 = pa1[3];                      // MA[t0+2], offset 4
        const u32 c3 = pa3[3], c4 = pa3[4];                      // MA[t0+4], offset 6
        const u32 d2 = pa4[2], d3 = pa4[3], d4 = pa4[4];         // MA[t0+5], offset 5
        const u32 e1 = p0[1], e2 = p0[2], e3 = p0[3];            // M0[t0],   offset 3
        const u32 f0_ = p3[0], f1 = p3[1];                       // M3[t0+3], offset 0
        const u32 g1 = p6[1], g2 = p6[2];                        // M6[t0+6], offset 2
        const s16x2 n_lo = ((as_s16x2(funnel16(e2, e1)) + as_s16x2(funnel16(a1, a0))) + as_s16x2(b2)) +
                           ((as_s16x2(f0_) + as_s16x2(c3)) + as_s16x2(funnel16(d3, d2))) + as_s16x2(g1);
        const s16x2 n_hi = ((as_s16x2(funnel16(e3, e2)) + as_s16x2(funnel16(a2, a1))) + as_s16x2(b3)) +
                           ((as_s16x2(f1) + as_s16x2(c4)) + as_s16x2(funnel16(d4, d3))) + as_s16x2(g2);
        // keep bit of a cell = NOT sign(num - T), restricted to real positions
        const u32 k_lo = ~as_u32(__builtin_elementwise_sub_sat(n_lo, Tpk)) & vm_lo;
        const u32 k_hi = ~as_u32(__builtin_elementwise_sub_sat(n_hi, Tpk)) & vm_hi;
        const bool any = __builtin_amdgcn_ballot_w64((k_lo | k_hi) != 0u) != 0ull;
        if (SCORE_MAP || any) {                                   // wave-uniform
            const float rnavg = navg > 0 ? 1.0f / (float)navg : 1.0f;
            // score /= navg (C int division, truncating toward zero) without an integer divide: with
            // |score| <= 21*4*255 and navg <= 84 the quotient is either an integer or at least 1/84 away from
            // one, while float(score)*fl(1/navg) is within 2e-3 of it, so adding 0.004 away from zero and
            // truncating is exact.
            auto quotient = [&](int num) {
                const float fs_ = (float)num;
                return (int)(fs_ * rnavg + __builtin_copysignf(0.004f, fs_));
            };
            const int num[4] = { (int)n_lo.x, (int)n_lo.y, (int)n_hi.x, (int)n_hi.y };
            const bool keep[4] = { (k_lo & 0x8000u) != 0, (k_lo & 0x80000000u) != 0, (k_hi & 0x8000u) != 0, (k_hi & 0x80000000u) != 0 };
            int score[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) score[j] = quotient(num[j]);
            if (SCORE_MAP) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f0 = 4 * lane + j;
                    if (f0 < kF0Count)
                        score_map[(size_t)frame * kScoresPerFrame + (seg * kT0Count + t0i) * kF0Count + f0] = (int16_t)score[j];
                }
            }
            if (any) {
                // scan order inside the row is f0 = 4 lane + j ascending: entries of lower lanes first, then lower j
                unsigned long long bal[4];
                int before = 0, total = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bal[j] = __builtin_amdgcn_ballot_w64(keep[j]);
                    before += __popcll(bal[j] & lt_mask);
                    total += __popcll(bal[j]);
                }
                int pos = count + before;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (keep[j]) {
                        my_list[pos] = ((uint32_t)(score[j] & 0xFFFF) << 16) | ((uint32_t)t0i << 8) | (uint32_t)(4 * lane + j);
                        ++pos;
                    }
                }
                count += total;
            }
        }
    }
    if (lane == 0) list_counts[(size_t)frame * kSublistsPerFrame + sub] = count;
}

// ---- exact top-N selection -----------------------------------------------------------------
// One wave per frame.  The min-heap lives in LDS as 64-bit words whose low 16 bits are the score
// (the little-endian image of candidate_t).  Lanes fetch 64 list entries at a time; because the heap
// minimum never decreases once the heap is full, entries that cannot beat the minimum seen at the
// start of a chunk are dropped with one ballot, and only the survivors are replayed, in list order,
// by lane 0 through the reference's insertion / eviction / heapify rules.  The final heap sort is
// replayed the same way, so ties come out in exactly the reference's order.
__device__ __forceinline__ int sc(uint64_t e) { return (int)(int16_t)(e & 0xFFFFu); }
__device__ __forceinline__ void wave_lds_sync_heap() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void heapify_down(uint64_t *heap, int heap_size) {
    int current = 0;
    uint64_t cur = heap[0];
    while (true) {
        int largest = current;
        uint64_t lv = cur;
        const int left = 2 * current + 1, right = left + 1;
        if (left < heap_size) { const uint64_t l = heap[left]; if (sc(l) < sc(lv)) { largest = left; lv = l; } }
        if (right < heap_size) { const uint64_t r = heap[right]; if (sc(r) < sc(lv)) { largest = right; lv = r; } }
        if (largest == current) break;
        heap[current] = lv;                      // swap: the child value moves up ...
        current = largest;                       // ... and `cur` continues down
    }
    heap[current] = cur;
}

__device__ __forceinline__ void heapify_up(uint64_t *heap, int heap_size) {
    int current = heap_size - 1;
    const uint64_t cur = heap[current];
    while (current > 0) {
        const int parent = (current - 1) / 2;
        const uint64_t pv = heap[parent];
        if (sc(cur) >= sc(pv)) break;
        heap[current] = pv;
        current = parent;
    }
    heap[current] = cur;
}

// ---- register-resident form of the same replay (max_candidates <= 128) -----------------------------
// The replay is a chain of dependent compare-and-move steps; with the heap in LDS every step pays an LDS
// round trip from a single active lane.  Here the heap array lives in two VGPRs across the wave as 32-bit
// keys  score << 16 | entry id  -- node j in lane j >> 1 of kE (j even) or kO (j odd), so the root is
// kE[0] and the children of node c are kO[c] and kE[c + 1] -- every index is wave-uniform, and the whole
// control flow runs on the scalar unit with v_readlane / v_writelane, free of data-dependent branches
// except the loop exits; LDS only keeps the 64-bit entries, addressed by id.  Same comparisons and moves in
// the same order as heapify_up / heapify_down above.
}  // namespace
// v_writelane_b32 (this hipcc declares no builtin for it; the LLVM intrinsic is bound by name)
extern "C" __device__ int ft8_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
namespace {

struct HeapRegs {
    uint32_t kE, kO;                                             // nodes 2*lane and 2*lane + 1
    __device__ __forceinline__ uint32_t get(int j) const {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)kE, j >> 1), o = (uint32_t)__builtin_amdgcn_readlane((int)kO, j >> 1);
        return (j & 1) ? o : e;
    }
    __device__ __forceinline__ void set(int j, uint32_t v) {     // rewrites both registers' lane, one of them with its own value
        const int l = j >> 1;
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)kE, l), o = (uint32_t)__builtin_amdgcn_readlane((int)kO, l);
        kE = (uint32_t)ft8_writelane_i32((int)((j & 1) ? e : v), l, (int)kE);
        kO = (uint32_t)ft8_writelane_i32((int)((j & 1) ? v : o), l, (int)kO);
    }
    __device__ __forceinline__ uint32_t root() const { return (uint32_t)__builtin_amdgcn_readlane((int)kE, 0); }
};
__device__ __forceinline__ int hk_sc(uint32_t key) { return (int)key >> 16; }

__device__ __forceinline__ void hk_sift_down(HeapRegs &h, int heap_size, uint32_t cur) {   // cur enters at the root
    int current = 0;
    while (true) {
        const int left = 2 * current + 1, right = left + 1;
        if (left >= heap_size) break;                                        // no child: (largest == current)
        const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)h.kO, current);       // node 2c + 1
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)h.kE, current + 1);   // node 2c + 2 (ignored if outside)
        const bool take_l = hk_sc(l) < hk_sc(cur);
        const uint32_t lv = take_l ? l : cur;
        const bool take_r = right < heap_size && hk_sc(r) < hk_sc(lv);
        if (!take_l && !take_r) break;
        h.set(current, take_r ? r : l);
        current = take_r ? right : left;
    }
    h.set(current, cur);
}

__device__ __forceinline__ void hk_sift_up(HeapRegs &h, int heap_size, uint32_t cur) {     // cur enters at the last node
    int current = heap_size - 1;
    while (current > 0) {
        const int parent = (current - 1) / 2;
        const uint32_t pv = h.get(parent);
        if (hk_sc(cur) >= hk_sc(pv)) break;
        h.set(current, pv);
        current = parent;
    }
    h.set(current, cur);
}

__device__ __forceinline__ void heap_select_regs(const uint32_t *__restrict__ frame_lists, const int32_t *__restrict__ frame_counts,
                                                 uint64_t *ent, uint64_t *out, int32_t *count_out, int max_candidates, int lane) {
    HeapRegs h = { 0u, 0u };
    int heap_size = 0;                                           // wave-uniform, like every index below
    // Memory latency must stay out of the dependent chain: the 32 sub-list lengths arrive with one load, and
    // the entries of the next 64-entry chunk are requested before the current chunk is replayed.
    static_assert(kSublistsPerFrame <= 64, "one lane per sub-list length");
    const int my_n = lane < kSublistsPerFrame ? frame_counts[lane] : 0;
    auto count_of = [&](int sub) { return __builtin_amdgcn_readlane(my_n, sub); };
    auto advance = [&](int &sub, int &base) {                    // next non-empty chunk in (sub-list, offset) order
        base += 64;
        while (sub < kSublistsPerFrame && (sub < 0 || base >= count_of(sub))) { ++sub; base = 0; }
    };
    auto fetch = [&](int sub, int base) -> uint32_t {
        if (sub >= kSublistsPerFrame) return 0u;
        const int e = base + lane;
        return e < count_of(sub) ? frame_lists[(size_t)sub * kSublistCap + e] : 0u;
    };
    int sub = -1, base = 0;
    advance(sub, base);
    uint32_t v_cur = fetch(sub, base);
    while (sub < kSublistsPerFrame) {                            // (time_sub, freq_sub, time_offset) order
        int nsub = sub, nbase = base;
        advance(nsub, nbase);
        const uint32_t v_next = fetch(nsub, nbase);              // in flight while this chunk is replayed
        const int n = count_of(sub);
        const uint32_t seg = (uint32_t)(sub / (kSyncHalves * kSyncWaves));
        uint64_t c = 0;
        bool live = base + lane < n;
        if (live) {
            const uint32_t v = v_cur;
            const uint32_t score = v >> 16;
            const uint32_t t0 = (uint32_t)(int)((int)((v >> 8) & 0xFF) + kT0Min) & 0xFFFFu;
            c = (uint64_t)score | ((uint64_t)t0 << 16) | ((uint64_t)(v & 0xFF) << 32) |
                ((uint64_t)(seg >> 1) << 48) | ((uint64_t)(seg & 1) << 56);
            // prefilter against the current minimum (it can only grow while this chunk is replayed)
            if (heap_size == max_candidates && !(sc(c) > hk_sc(h.root()))) live = false;
        }
        const uint32_t my_score16 = (uint32_t)(c & 0xFFFFu);
        for (unsigned long long mask = __ballot(live); mask != 0ull; mask &= mask - 1) {
            const int j = __builtin_ctzll(mask);                                  // list order
            const uint32_t s16 = (uint32_t)__builtin_amdgcn_readlane((int)my_score16, j);
            const int score = (int)(int16_t)s16;
            int id = heap_size;                                                   // ids 0..cap-1 are handed out in order while filling
            if (heap_size == max_candidates && score > hk_sc(h.root())) {
                id = (int)(h.root() & 0xFFFFu);                                   // the evicted root's entry slot is reused
                const uint32_t last = h.get(heap_size - 1);
                --heap_size;
                hk_sift_down(h, heap_size, last);
            }
            if (heap_size < max_candidates) {
                if (lane == j) ent[id] = c;
                ++heap_size;
                hk_sift_up(h, heap_size, (s16 << 16) | (uint32_t)id);
            }
        }
        sub = nsub;
        base = nbase;
        v_cur = v_next;
    }
    // heap sort (descending), replayed on the keys
    for (int len_unsorted = heap_size; len_unsorted > 1;) {
        const uint32_t tmp = h.get(len_unsorted - 1);
        h.set(len_unsorted - 1, h.root());
        len_unsorted--;
        hk_sift_down(h, len_unsorted, tmp);
    }
    if (lane == 0) *count_out = heap_size;
    wave_lds_sync_heap();
    // node j sits in lane j >> 1: lane l stores nodes 2l and 2l + 1 (8-byte stores, pairs adjacent)
    if (2 * lane < max_candidates) out[2 * lane] = 2 * lane < heap_size ? ent[h.kE & 0xFFFFu] : 0ull;            // deterministic tail
    if (2 * lane + 1 < max_candidates) out[2 * lane + 1] = 2 * lane + 1 < heap_size ? ent[h.kO & 0xFFFFu] : 0ull;
}

// ---- the same replay, one LANE per frame (large launches, candidate caps up to 128) ------------------------------------
// The forms above give a frame a whole wave: 64 entries are prefiltered at once, but the replay itself runs on one
// lane (or on the scalar unit), so every one of its instructions occupies a 64-lane issue slot for one frame -- about
// 16 000 VALU instructions per frame, 3 % of all VALU instructions of a batch, spent beside the VALU-bound LDPC kernel.
// Here each lane replays the heap of its own frame with the plain reference algorithm; control flow diverges between
// lanes, but every issued instruction now works for up to 64 frames.  A heap entry is one 32-bit word -- score (16
// bits, signed) | segment (2) | time-offset index (6) | frequency offset (8) -- so that comparisons read one word and
// the heap of a lane is max_candidates words, stored element-major ([element][lane]: lanes touching the same element,
// the common case, hit 64 different banks).  List entries are read four at a time (every sub-list is 16-byte aligned).
// Same insertion / eviction / heapify / heap-sort rules in the same order: identical candidate lists.
__device__ __forceinline__ int sk(uint32_t key) { return (int)key >> 16; }

__global__ __launch_bounds__(64)
void ft8_heap_simt_kernel(const uint32_t *__restrict__ lists, const int32_t *__restrict__ list_counts,
                          ft8gpu_candidate *__restrict__ cands, int32_t *__restrict__ counts, int nframes, int max_candidates) {
    extern __shared__ uint32_t s_keys[];                          // [max_candidates][64]
    const int lane = threadIdx.x;
    const int frame = blockIdx.x * 64 + lane;
    if (frame >= nframes) return;                                // (no barrier in this kernel)
    uint32_t *h = s_keys + lane;                                 // element i of this lane: h[64 * i]
    const int cap = max_candidates;
    int heap_size = 0;

    auto down = [&](int size) {                                  // heapify_down from the root
        int current = 0;
        const uint32_t cur = h[0];
        while (true) {
            int smallest = current;
            uint32_t sv = cur;
            const int left = 2 * current + 1, right = left + 1;
            if (left < size) { const uint32_t l = h[64 * left]; if (sk(l) < sk(sv)) { smallest = left; sv = l; } }
            if (right < size) { const uint32_t r = h[64 * right]; if (sk(r) < sk(sv)) { smallest = right; sv = r; } }
            if (smallest == current) break;
            h[64 * current] = sv;
            current = smallest;
        }
        h[64 * current] = cur;
    };
    auto up = [&](int size) {                                    // heapify_up from the last node
        int current = size - 1;
        const uint32_t cur = h[64 * current];
        while (current > 0) {
            const int parent = (current - 1) / 2;
            const uint32_t pv = h[64 * parent];
            if (sk(cur) >= sk(pv)) break;
            h[64 * current] = pv;
            current = parent;
        }
        h[64 * current] = cur;
    };

    const uint32_t *fl = lists + (size_t)frame * kSublistsPerFrame * kSublistCap;
    const int32_t *fc = list_counts + (size_t)frame * kSublistsPerFrame;
    int root_score = 0;                                          // score of h[0] while the heap is full
    for (int sub = 0; sub < kSublistsPerFrame; ++sub) {          // (time_sub, freq_sub, time_offset) order
        const int n = fc[sub];
        const uint32_t seg_bits = (uint32_t)(sub / (kSyncHalves * kSyncWaves)) << 14;
        const uint4 *src = reinterpret_cast<const uint4 *>(fl + (size_t)sub * kSublistCap);
        for (int e0 = 0; e0 < n; e0 += 4) {
            const uint4 v4 = src[e0 >> 2];
            const uint32_t vs[4] = { v4.x, v4.y, v4.z, v4.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (e0 + j >= n) break;
                const uint32_t key = (vs[j] & 0xFFFF3FFFu) | seg_bits;       // t0 index < 36: bits 14, 15 of the list entry are free
                const int score = sk(key);
                if (heap_size == cap) {
                    if (!(score > root_score)) continue;                     // the common case once the heap is full
                    h[0] = h[64 * (heap_size - 1)];
                    --heap_size;
                    down(heap_size);
                }
                h[64 * heap_size] = key;
                ++heap_size;
                up(heap_size);
                if (heap_size == cap) root_score = sk(h[0]);
            }
        }
    }
    // heap sort (descending)
    for (int len = heap_size; len > 1;) {
        const uint32_t tmp = h[64 * (len - 1)];
        h[64 * (len - 1)] = h[0];
        h[0] = tmp;
        --len;
        down(len);
    }
    counts[frame] = heap_size;
    uint64_t *out = reinterpret_cast<uint64_t *>(cands) + (size_t)frame * max_candidates;
    for (int i = 0; i < max_candidates; ++i) {
        uint64_t c = 0;                                                      // deterministic tail
        if (i < heap_size) {
            const uint32_t key = h[64 * i];
            const uint32_t seg = (key >> 14) & 3u;
            const uint32_t t0 = (uint32_t)((int)((key >> 8) & 0x3Fu) + kT0Min) & 0xFFFFu;
            c = (uint64_t)(key >> 16) | ((uint64_t)t0 << 16) | ((uint64_t)(key & 0xFFu) << 32) |
                ((uint64_t)(seg >> 1) << 48) | ((uint64_t)(seg & 1u) << 56);
        }
        out[i] = c;
    }
}

__global__ __launch_bounds__(256)
void ft8_heap_kernel(const uint32_t *__restrict__ lists, const int32_t *__restrict__ list_counts,
                     ft8gpu_candidate *__restrict__ cands, int32_t *__restrict__ counts,
                     int nframes, int max_candidates, int register_form) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_heap_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= nframes) return;                                // wave-uniform
    uint64_t *heap = s_heap_all + (size_t)wave * (max_candidates + 64);
    if (register_form) {                                         // wave-uniform
        heap_select_regs(lists + (size_t)frame * kSublistsPerFrame * kSublistCap, list_counts + (size_t)frame * kSublistsPerFrame,
                         heap, reinterpret_cast<uint64_t *>(cands) + (size_t)frame * max_candidates, counts + frame, max_candidates, lane);
        return;
    }
    uint64_t *stage = heap + max_candidates;                     // 64 staged survivors
    int heap_size = 0;

    for (int sub = 0; sub < kSublistsPerFrame; ++sub) {          // (time_sub, freq_sub, time_offset) order
        const int n = list_counts[(size_t)frame * kSublistsPerFrame + sub];
        const uint32_t *l = lists + ((size_t)frame * kSublistsPerFrame + sub) * kSublistCap;
        const uint32_t seg = (uint32_t)(sub / (kSyncHalves * kSyncWaves));
        for (int base = 0; base < n; base += 64) {
            const int e = base + lane;
            uint64_t c = 0;
            bool live = e < n;
            if (live) {
                const uint32_t v = l[e];
                const uint32_t score = v >> 16;
                const uint32_t t0 = (uint32_t)(int)((int)((v >> 8) & 0xFF) + kT0Min) & 0xFFFFu;
                c = (uint64_t)score | ((uint64_t)t0 << 16) | ((uint64_t)(v & 0xFF) << 32) |
                    ((uint64_t)(seg >> 1) << 48) | ((uint64_t)(seg & 1) << 56);
                // prefilter against the current minimum (it can only grow while this chunk is replayed)
                if (heap_size == max_candidates && !(sc(c) > sc(heap[0]))) live = false;
            }
            const unsigned long long mask = __ballot(live);
            if (mask == 0ull) continue;
            if (live) stage[__popcll(mask & ((1ull << lane) - 1ull))] = c;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int m = __popcll(mask);
            if (lane == 0) {
                for (int i = 0; i < m; ++i) {
                    const uint64_t cc = stage[i];
                    if (heap_size == max_candidates && sc(cc) > sc(heap[0])) {
                        heap[0] = heap[heap_size - 1];
                        --heap_size;
                        heapify_down(heap, heap_size);
                    }
                    if (heap_size < max_candidates) {
                        heap[heap_size] = cc;
                        ++heap_size;
                        heapify_up(heap, heap_size);
                    }
                }
            }
            heap_size = __shfl(heap_size, 0, 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (lane == 0) {
        int len_unsorted = heap_size;
        while (len_unsorted > 1) {
            const uint64_t tmp = heap[len_unsorted - 1];
            heap[len_unsorted - 1] = heap[0];
            heap[0] = tmp;
            len_unsorted--;
            heapify_down(heap, len_unsorted);
        }
        counts[frame] = heap_size;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint64_t *out = reinterpret_cast<uint64_t *>(cands) + (size_t)frame * max_candidates;
    for (int i = lane; i < max_candidates; i += 64) out[i] = (i < heap_size) ? heap[i] : 0ull;   // deterministic tail
}

}  // namespace

hipError_t launch_sync(const uint8_t *mag, uint32_t *lists, int32_t *list_counts, int16_t *score_map,
                       int nframes, int min_score, hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    if (score_map)
        hipLaunchKernelGGL(ft8_sync_kernel<true>, dim3(nframes * kSegments * kSyncHalves), dim3(64 * kSyncWaves), 0, s,
                           mag, lists, list_counts, score_map, min_score);
    else
        hipLaunchKernelGGL(ft8_sync_kernel<false>, dim3(nframes * kSegments * kSyncHalves), dim3(64 * kSyncWaves), 0, s,
                           mag, lists, list_counts, score_map, min_score);
    return hipGetLastError();
}

hipError_t launch_heap(const uint32_t *lists, const int32_t *list_counts, ft8gpu_candidate *cands,
                       int32_t *counts, int nframes, int max_candidates, unsigned debug_flags, hipStream_t s, bool latency_hidden) {
    if (nframes < 1) return hipSuccess;
    // One lane per frame (ft8_heap_simt_kernel) issues a tenth of the instructions but takes about 0.55 ms whatever the
    // number of frames (0.1-0.3 ms for the forms below): the batch pipeline asks for it when the kernels it runs beside
    // are long enough to cover that (latency_hidden), everybody else gets the short chain.
    // (A/B build only: FT8GPU_AB_HEAP_LANE_PER_FRAME takes it whenever the cap allows, FT8GPU_AB_HEAP_WAVE_PER_FRAME never.)
    const bool lane_form_possible = max_candidates <= 128;
    bool want_lane_form = latency_hidden && nframes >= 256;
#ifdef FT8GPU_AB_FORMS
    if (debug_flags & FT8GPU_AB_HEAP_LANE_PER_FRAME) want_lane_form = true;
    if (debug_flags & FT8GPU_AB_HEAP_WAVE_PER_FRAME) want_lane_form = false;
#endif
    (void)debug_flags;
    if (lane_form_possible && want_lane_form) {
        hipLaunchKernelGGL(ft8_heap_simt_kernel, dim3((nframes + 63) / 64), dim3(64), (size_t)max_candidates * 64 * sizeof(uint32_t), s,
                           lists, list_counts, cands, counts, nframes, max_candidates);
        return hipGetLastError();
    }
    const size_t lds = (size_t)4 * (max_candidates + 64) * sizeof(uint64_t);
    // Two forms of the same replay.  The register form has the shorter dependent chain (a frame takes about
    // 0.11 ms instead of 0.14) and is what small launches wait for; it issues more instructions, though, and a
    // large launch runs next to the sync / LDPC kernels of the other half-batch, where the LDS form (which
    // mostly waits) finishes earlier -- measured at 2048 frames per launch: 0.29 against 0.245 ms.
    const int register_form = (max_candidates <= 128 && nframes <= 1024) ? 1 : 0;
    hipLaunchKernelGGL(ft8_heap_kernel, dim3((nframes + 3) / 4), dim3(256), lds, s,
                       lists, list_counts, cands, counts, nframes, max_candidates, register_form);
    return hipGetLastError();
}
