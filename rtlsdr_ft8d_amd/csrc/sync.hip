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
//                    frame per lane, so that candidate order is bit-identical to ft8_find_sync().
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

struct __attribute__((aligned(8))) Cand { int16_t score, time_offset, freq_offset; uint8_t time_sub, freq_sub; };

__device__ __forceinline__ void heapify_down(Cand *heap, int heap_size) {
    int current = 0;
    while (true) {
        int largest = current;
        const int left = 2 * current + 1, right = left + 1;
        if (left < heap_size && heap[left].score < heap[largest].score) largest = left;
        if (right < heap_size && heap[right].score < heap[largest].score) largest = right;
        if (largest == current) break;
        const Cand tmp = heap[largest];
        heap[largest] = heap[current];
        heap[current] = tmp;
        current = largest;
    }
}

__device__ __forceinline__ void heapify_up(Cand *heap, int heap_size) {
    int current = heap_size - 1;
    while (current > 0) {
        const int parent = (current - 1) / 2;
        if (heap[current].score >= heap[parent].score) break;
        const Cand tmp = heap[parent];
        heap[parent] = heap[current];
        heap[current] = tmp;
        current = parent;
    }
}

__global__ __launch_bounds__(64)
void ft8_heap_kernel(const uint32_t *__restrict__ lists, const int32_t *__restrict__ list_counts,
                     ft8gpu_candidate *__restrict__ cands, int32_t *__restrict__ counts,
                     int nframes, int max_candidates) {
    const int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= nframes) return;
    Cand *heap = reinterpret_cast<Cand *>(cands) + (size_t)frame * max_candidates;
    int heap_size = 0;
    for (int sub = 0; sub < kSublistsPerFrame; ++sub) {          // (time_sub, freq_sub, time_offset) order
        const int n = list_counts[(size_t)frame * kSublistsPerFrame + sub];
        const uint32_t *l = lists + ((size_t)frame * kSublistsPerFrame + sub) * kSublistCap;
        const int seg = sub / kSyncWaves;
        for (int e = 0; e < n; ++e) {
            const uint32_t v = l[e];
            Cand c;
            c.score = (int16_t)(v >> 16);
            c.time_offset = (int16_t)(((v >> 8) & 0xFF) + kT0Min);
            c.freq_offset = (int16_t)(v & 0xFF);
            c.time_sub = (uint8_t)(seg >> 1);
            c.freq_sub = (uint8_t)(seg & 1);
            if (heap_size == max_candidates && c.score > heap[0].score) {
                heap[0] = heap[heap_size - 1];
                --heap_size;
                heapify_down(heap, heap_size);
            }
            if (heap_size < max_candidates) {
                heap[heap_size] = c;
                ++heap_size;
                heapify_up(heap, heap_size);
            }
        }
    }
    int len_unsorted = heap_size;
    while (len_unsorted > 1) {
        const Cand tmp = heap[len_unsorted - 1];
        heap[len_unsorted - 1] = heap[0];
        heap[0] = tmp;
        len_unsorted--;
        heapify_down(heap, len_unsorted);
    }
    counts[frame] = heap_size;
    const Cand zero = { 0, 0, 0, 0, 0 };
    for (int i = heap_size; i < max_candidates; ++i) heap[i] = zero;      // deterministic tail
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
    hipLaunchKernelGGL(ft8_heap_kernel, dim3((nframes + 63) / 64), dim3(64), 0, s,
                       lists, list_counts, cands, counts, nframes, max_candidates);
    return hipGetLastError();
}
