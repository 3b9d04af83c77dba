// sync.hip -- stage a5 of the hot path: ft8_find_sync() of ft8_lib decode.c,
// call site rtlsdr_ft8d.c:1450 (waterfall descriptor rtlsdr_ft8d.c:1440-1448).
//
// Two kernels:
//   ft8_sync_kernel  one workgroup per (frame, time_sub, freq_sub): stages that 92x256-byte slice
//                    of the waterfall in LDS, derives a 5-point-stencil int16 map from it (every
//                    sync symbol is then one map read, 9 of the 21 with a two-byte correction), scores all
//                    36 x 249 (time_offset, freq_offset) positions with the integer Costas
//                    neighbour-contrast score, and compacts the positions with score >= min_score,
//                    in the reference's scan order, with __ballot/popcount prefix sums (no atomics,
//                    no barriers after the map is built).
//   ft8_heap_kernel  replays the reference's bounded min-heap (strict '>' replacement, its
//                    heapify tie rules and the final heap sort) over the compacted list, one
//                    wave per frame with the heap in LDS, so that candidate order is bit-identical
//                    to ft8_find_sync().
#include "ft8gpu_internal.h"

namespace {

// LDS layout of one (frame, time_sub, freq_sub) slice, in this order, one allocation:
//   guard (13 rows of bytes) | P: uint8 [92][256] waterfall slice | S5: int16 [92][256] stencil map | guard (10 int16 rows)
// A sync symbol at an out-of-range block (b < 0 or b >= 92) is addressed like any other, lands in a
// guard or in the neighbouring map, and is multiplied by a wave-uniform weight of 0 -- so every LDS
// read of a position has a compile-time offset from one base register and no address arithmetic.
constexpr int kPitch = 256;                              // bytes per waterfall row
constexpr int kSPitch = 256;                             // int16 per stencil row
constexpr int kGuardP = 13 * kPitch;                     // rows -13..-1 (b - 1 for b = t0 = -12)
constexpr int kOffP = kGuardP;
constexpr int kOffS = kOffP + kNumBlocks * kPitch;       // byte offset of S5
constexpr int kGuardS = 10 * kSPitch * 2;
constexpr int kSyncLds = kOffS + kNumBlocks * kSPitch * 2 + kGuardS;   // 79104 bytes: two workgroups per CU

// Contribution of one sync symbol (Costas index K, tone column C) of ft8_sync_score() at absolute
// block b = t0 + 36 m + K (wave-uniform) and bin column f = f0 + C.  The 5-point stencil map
//   S5[b][f] = [f>0](p-p[f-1]) + (p-p[f+1]) + [b>0](p-p[b-1][f]) + [b<91](p-p[b+1][f])
// already omits a missing time neighbour at the first/last block exactly as the reference does, and
// for K in {1,2,4,5} it IS the symbol's contribution.  The other three symbols lack one term:
//   K = 0 (first of a Costas block) has no look-back:    S5 - [b>0](p-p[b-1][f])
//   K = 6 (last of a Costas block) has no look-ahead:     S5 - [b<91](p-p[b+1][f])
//   K = 3 (tone 0) has no lower bin:                      S5 - [f>0](p-p[f-1])
// so every symbol is one S5 read, three of them with a two-byte correction.  pb / sb point at
// (row t0, column f0) of P / S5.  The byte left of column 0 of every row is a copy of column 0
// (see the kernel), so the K = 3 correction needs no test for f = 0.
//
// Everything that depends on b is a wave-uniform 0/1 weight, and at most one of the three Costas
// blocks of a position can touch the edge of the waterfall.  MODE names it, so that the other two
// (or, for the interior time offsets, all three) are plain sums without weights:
//   MODE 0: 1 <= t0 <= 12, every block and every time neighbour exists
//   MODE 1: t0 <= 0, Costas block m = 0 is weighted          MODE 2: t0 >= 13, block m = 2 is weighted
template <int K, int C, int MODE>
__device__ __forceinline__ int sync_symbol(const uint8_t *pb, const int16_t *sb, int t0) {
    int acc = 0;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int rel = 36 * m + K;                                      // compile-time row offset from t0
        const bool weighted = (MODE == 1 && m == 0) || (MODE == 2 && m == 2);
        if (!weighted) {
            acc += (int)sb[rel * kSPitch + C];
            if (K == 0) acc -= (int)pb[rel * kPitch + C] - (int)pb[(rel - 1) * kPitch + C];
            else if (K == 6) acc -= (int)pb[rel * kPitch + C] - (int)pb[(rel + 1) * kPitch + C];
            else if (K == 3) acc -= (int)pb[rel * kPitch + C] - (int)pb[rel * kPitch + C - 1];
        } else {
            const int b = t0 + rel;                                      // wave-uniform
            const int w = (b >= 0 && b < kNumBlocks) ? 1 : 0;
            acc += w * (int)sb[rel * kSPitch + C];
            if (K == 0) {
                const int wm = (w && b > 0) ? 1 : 0;
                acc -= wm * ((int)pb[rel * kPitch + C] - (int)pb[(rel - 1) * kPitch + C]);
            } else if (K == 6) {
                const int wp = (w && b + 1 < kNumBlocks) ? 1 : 0;
                acc -= wp * ((int)pb[rel * kPitch + C] - (int)pb[(rel + 1) * kPitch + C]);
            } else if (K == 3) {
                acc -= w * ((int)pb[rel * kPitch + C] - (int)pb[rel * kPitch + C - 1]);
            }
        }
    }
    return acc;
}

// ft8_sync_score() numerator for one position; Costas pattern {3,1,4,0,6,5,2}
template <int MODE>
__device__ __forceinline__ int sync_sum(const uint8_t *pb, const int16_t *sb, int t0) {
    return sync_symbol<0, 3, MODE>(pb, sb, t0) + sync_symbol<1, 1, MODE>(pb, sb, t0) + sync_symbol<2, 4, MODE>(pb, sb, t0) +
           sync_symbol<3, 0, MODE>(pb, sb, t0) + sync_symbol<4, 6, MODE>(pb, sb, t0) + sync_symbol<5, 5, MODE>(pb, sb, t0) +
           sync_symbol<6, 2, MODE>(pb, sb, t0);
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

__global__ __launch_bounds__(64 * kSyncWaves)
void ft8_sync_kernel(const uint8_t *__restrict__ mag, uint32_t *__restrict__ lists,
                     int32_t *__restrict__ list_counts, int16_t *__restrict__ score_map, int min_score) {
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kSyncLds];
    uint8_t *s_wf = s_lds + kOffP;
    int16_t *s_s5 = reinterpret_cast<int16_t *>(s_lds + kOffS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keep it in an SGPR
    const int frame = blockIdx.x >> 2, seg = blockIdx.x & 3;
    const int ts = seg >> 1, fs = seg & 1;

    // mag[block][time_sub][freq_sub][bin]: one 256-byte run per block for this (ts, fs)
    const uint8_t *src = mag + (size_t)frame * kMagArray + ts * 512 + fs * 256;
    for (int i = tid; i < kNumBlocks * 16; i += 64 * kSyncWaves) {
        const int row = i >> 4, col = (i & 15) * 16;
        *reinterpret_cast<uint4 *>(s_wf + row * kPitch + col) =
            *reinterpret_cast<const uint4 *>(src + (size_t)row * kBlockStride + col);
    }
    __syncthreads();
    // 5-point stencil map, four cells per thread from dword reads; column 0 has no lower bin,
    // column 255 is never addressed (f0 + tone <= 254) and is left 0
    for (int i = tid; i < kNumBlocks * 64; i += 64 * kSyncWaves) {
        const int row = i >> 6, col = (i & 63) * 4;
        const uint8_t *p = s_wf + row * kPitch + col;
        const uint32_t mid = *reinterpret_cast<const uint32_t *>(p);
        const uint32_t up = row > 0 ? *reinterpret_cast<const uint32_t *>(p - kPitch) : mid;              // missing neighbour:
        const uint32_t dn = row + 1 < kNumBlocks ? *reinterpret_cast<const uint32_t *>(p + kPitch) : mid;  // p - p = 0
        const int left = col > 0 ? p[-1] : (int)(mid & 0xFF) /* p - p = 0: no lower bin */, right = col + 4 < 256 ? p[4] : 0;
        int c[6];
        c[0] = left;
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j + 1] = (mid >> (8 * j)) & 0xFF;
        c[5] = right;
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = c[j + 1];
            v[j] = (cc - c[j]) + (cc - c[j + 2]) + (cc - (int)((up >> (8 * j)) & 0xFF)) + (cc - (int)((dn >> (8 * j)) & 0xFF));
        }
        if (col == 252) v[3] = 0;
        uint2 packed;
        packed.x = (uint32_t)(v[0] & 0xFFFF) | ((uint32_t)v[1] << 16);
        packed.y = (uint32_t)(v[2] & 0xFFFF) | ((uint32_t)v[3] << 16);
        *reinterpret_cast<uint2 *>(s_s5 + row * kSPitch + col) = packed;
    }
    __syncthreads();
    // The map is built; column 255 of P is not read again (f0 + tone <= 254), so the byte left of every
    // row's column 0 (= column 255 of the row above, or the last guard byte) can take a copy of column 0:
    // p[f] - p[f-1] is then 0 at f = 0, which is the reference's "no lower bin" case.
    if (tid < kNumBlocks) s_wf[tid * kPitch - 1] = s_wf[tid * kPitch];
    __syncthreads();

    const int sub = seg * kSyncWaves + wave;
    uint32_t *my_list = lists + ((size_t)frame * kSublistsPerFrame + sub) * kSublistCap;
    int count = 0;
    const int t_begin = (wave * kT0Count) / kSyncWaves, t_end = ((wave + 1) * kT0Count) / kSyncWaves;

    for (int t0i = t_begin; t0i < t_end; ++t0i) {        // scan order: time_offset ascending
        const int t0 = t0i + kT0Min;
        const int navg = sync_navg(t0);
        // score /= navg (C int division, truncating toward zero) without an integer divide: with
        // |score| <= 21*255 and navg <= 84 the quotient is either an integer or at least 1/84 away from
        // one, while float(score)*fl(1/navg) is within 1e-3 of it, so adding 0.004 away from zero and
        // truncating is exact.
        const float rnavg = navg > 0 ? 1.0f / (float)navg : 1.0f;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int f0 = pass * 64 + lane;            // then freq_offset ascending
            const bool valid = f0 < kF0Count;
            const int fc = valid ? f0 : 0;
            const uint8_t *pb = s_wf + t0 * kPitch + fc;
            const int16_t *sb = s_s5 + t0 * kSPitch + fc;
            int score;                                  // t0 is wave-uniform: one of three straight-line variants
            if (t0 >= 1 && t0 <= 12) score = sync_sum<0>(pb, sb, t0);
            else if (t0 <= 0) score = sync_sum<1>(pb, sb, t0);
            else score = sync_sum<2>(pb, sb, t0);
            {
                const float fs_ = (float)score;
                score = (int)(fs_ * rnavg + __builtin_copysignf(0.004f, fs_));
            }
            if (valid && score_map)
                score_map[(size_t)frame * kScoresPerFrame + (seg * kT0Count + t0i) * kF0Count + f0] = (int16_t)score;
            const bool keep = valid && score >= min_score;
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
                my_list[pos] = ((uint32_t)(score & 0xFFFF) << 16) | ((uint32_t)t0i << 8) | (uint32_t)f0;
            }
            count += __popcll(mask);
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

__global__ __launch_bounds__(256)
void ft8_heap_kernel(const uint32_t *__restrict__ lists, const int32_t *__restrict__ list_counts,
                     ft8gpu_candidate *__restrict__ cands, int32_t *__restrict__ counts,
                     int nframes, int max_candidates) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_heap_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= nframes) return;                                // wave-uniform
    uint64_t *heap = s_heap_all + (size_t)wave * (max_candidates + 64);
    uint64_t *stage = heap + max_candidates;                     // 64 staged survivors
    int heap_size = 0;

    for (int sub = 0; sub < kSublistsPerFrame; ++sub) {          // (time_sub, freq_sub, time_offset) order
        const int n = list_counts[(size_t)frame * kSublistsPerFrame + sub];
        const uint32_t *l = lists + ((size_t)frame * kSublistsPerFrame + sub) * kSublistCap;
        const uint32_t seg = (uint32_t)(sub / kSyncWaves);
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
    hipLaunchKernelGGL(ft8_sync_kernel, dim3(nframes * kSegments), dim3(64 * kSyncWaves), 0, s,
                       mag, lists, list_counts, score_map, min_score);
    return hipGetLastError();
}

hipError_t launch_heap(const uint32_t *lists, const int32_t *list_counts, ft8gpu_candidate *cands,
                       int32_t *counts, int nframes, int max_candidates, hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    const size_t lds = (size_t)4 * (max_candidates + 64) * sizeof(uint64_t);
    hipLaunchKernelGGL(ft8_heap_kernel, dim3((nframes + 3) / 4), dim3(256), lds, s,
                       lists, list_counts, cands, counts, nframes, max_candidates);
    return hipGetLastError();
}
