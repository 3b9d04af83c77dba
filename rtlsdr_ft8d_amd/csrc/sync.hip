// sync.hip -- stage a5 of the hot path: ft8_find_sync() of ft8_lib decode.c,
// call site rtlsdr_ft8d.c:1450 (waterfall descriptor rtlsdr_ft8d.c:1440-1448).
//
// Two kernels:
//   ft8_sync_kernel  one workgroup per (frame, time_sub, freq_sub): stages that 92x256-byte slice
//                    of the waterfall in LDS, scores all 36 x 249 (time_offset, freq_offset)
//                    positions with the integer Costas neighbour-contrast score, and compacts the
//                    positions with score >= min_score, in the reference's scan order, with
//                    __ballot/popcount prefix sums (no atomics, no barriers after the load).
//   ft8_heap_kernel  replays the reference's bounded min-heap (strict '>' replacement, its
//                    heapify tie rules and the final heap sort) over the compacted list, one
//                    wave per frame with the heap in LDS, so that candidate order is bit-identical
//                    to ft8_find_sync().
#include "ft8gpu_internal.h"

namespace {

__constant__ uint8_t c_costas[7] = { 3, 1, 4, 0, 6, 5, 2 };

constexpr int kSliceRows = kNumBlocks;                  // 92 rows of 256 bytes
constexpr int kSlicePitch = 256 + 16;                   // bytes; +16 keeps rows 16-byte aligned and
                                                        // staggers rows across LDS banks

__global__ __launch_bounds__(256)
void ft8_sync_kernel(const uint8_t *__restrict__ mag, uint32_t *__restrict__ lists,
                     int32_t *__restrict__ list_counts, int16_t *__restrict__ score_map, int min_score) {
    __shared__ __attribute__((aligned(16))) uint8_t s_wf[kSliceRows * kSlicePitch];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frame = blockIdx.x >> 2, seg = blockIdx.x & 3;
    const int ts = seg >> 1, fs = seg & 1;

    // mag[block][time_sub][freq_sub][bin]: one 256-byte run per block for this (ts, fs)
    const uint8_t *src = mag + (size_t)frame * kMagArray + ts * 512 + fs * 256;
    for (int i = tid; i < kSliceRows * 16; i += 256) {
        const int row = i >> 4, col = (i & 15) * 16;
        *reinterpret_cast<uint4 *>(s_wf + row * kSlicePitch + col) =
            *reinterpret_cast<const uint4 *>(src + (size_t)row * kBlockStride + col);
    }
    __syncthreads();

    uint32_t *my_list = lists + ((size_t)frame * kSublistsPerFrame + seg * kSyncWaves + wave) * kSublistCap;
    int count = 0;

    for (int r = 0; r < kT0PerWave; ++r) {
        const int t0i = wave * kT0PerWave + r;          // scan order: time_offset ascending
        const int t0 = t0i + kT0Min;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int f0 = pass * 64 + lane;            // then freq_offset ascending
            const bool valid = f0 < kF0Count;
            int score = 0, navg = 0;
            if (valid) {
                // ft8_sync_score(): neighbours of the expected Costas tone, frequency- and time-wise
                for (int m = 0; m < 3; ++m) {
                    for (int k = 0; k < 7; ++k) {
                        const int block_abs = t0 + 36 * m + k;
                        if (block_abs < 0) continue;
                        if (block_abs >= kNumBlocks) break;
                        const int sm = c_costas[k];
                        const uint8_t *p8 = s_wf + block_abs * kSlicePitch + f0;
                        const int c = p8[sm];
                        if (sm > 0) { score += c - p8[sm - 1]; ++navg; }
                        if (sm < 7) { score += c - p8[sm + 1]; ++navg; }
                        if (k > 0 && block_abs > 0) { score += c - p8[sm - kSlicePitch]; ++navg; }
                        if (k + 1 < 7 && block_abs + 1 < kNumBlocks) { score += c - p8[sm + kSlicePitch]; ++navg; }
                    }
                }
                if (navg > 0) score /= navg;            // C int division, truncates toward zero
                if (score_map)
                    score_map[(size_t)frame * kScoresPerFrame + (seg * kT0Count + t0i) * kF0Count + f0] = (int16_t)score;
            }
            const bool keep = valid && score >= min_score;
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
                my_list[pos] = ((uint32_t)(score & 0xFFFF) << 16) | ((uint32_t)t0i << 8) | (uint32_t)f0;
            }
            count += __popcll(mask);
        }
    }
    if (lane == 0) list_counts[(size_t)frame * kSublistsPerFrame + seg * kSyncWaves + wave] = count;
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    hipLaunchKernelGGL(ft8_sync_kernel, dim3(nframes * kSegments), dim3(256), 0, s,
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
