// spots.hip -- stages a6, a8, a9 of the hot path, rtlsdr_ft8d.c:1452-1523:
// candidate loop with score gate and freq_hz (:1465-1470), open-addressing dedup table of
// K_MAX_MESSAGES = 50 slots keyed by message.hash % 50 with strcmp on the text (:1487-1507),
// CQ-only spot fill via strtok semantics (:1509-1518), num_decoded++ for every unique message (:1520).
//
// The table walk is inherently sequential in candidate order (first duplicate wins, which fixes the
// reported freq / snr), tiny, and byte-oriented.
#include "ft8gpu_internal.h"

namespace {

// strcmp(a, b) == 0 on two message_t.text[25] fields in HBM: all 50 bytes are requested before the
// first comparison, so the walk costs one memory round trip instead of one per character
__device__ inline bool text_equal(const char *a, const char *b) {
    char ta[25], tb[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) { ta[i] = a[i]; tb[i] = b[i]; }
    bool equal = true, open = true;                 // open: no terminator seen yet
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        equal = equal && (!open || ta[i] == tb[i]);
        open = open && ta[i] != 0;
    }
    return equal;
}

// strtok(text, " ") semantics: returns start index of the next token at or after *pos, or -1;
// *len receives the token length, *pos is advanced past the token and one delimiter
__device__ inline int next_token(const char *s, int *pos, int *len) {
    int p = *pos;
    while (s[p] == ' ') ++p;
    if (s[p] == 0) { *pos = p; return -1; }
    const int start = p;
    while (s[p] != 0 && s[p] != ' ') ++p;
    *len = p - start;
    *pos = (s[p] == 0) ? p : p + 1;
    return start;
}

// snprintf(dst, cap, "%.<prec>s", tok) ; tok == NULL prints "(null)" (glibc)
__device__ inline void put_field(char *dst, int cap, int prec, const char *tok, int len) {
    const char null_str[7] = { '(', 'n', 'u', 'l', 'l', ')', 0 };
    if (tok == nullptr) { tok = null_str; len = 6; }
    int n = len < prec ? len : prec;
    if (n > cap - 1) n = cap - 1;
    for (int i = 0; i < n; ++i) dst[i] = tok[i];
    dst[n] = 0;
}

// One wave per frame.  Lanes read the frame's candidate scores, `ok` flags and hashes in parallel
// (coalesced) into LDS and reduce them to a bit mask of the candidates that reach the table code
// (score gate :1467, ft8_decode() true :1476); lane 0 then walks only those, in candidate order,
// through the reference's open-addressing table.  Texts are compared from HBM only on a hash match.
__global__ __launch_bounds__(256)
void ft8_spots_kernel(const ft8gpu_candidate *__restrict__ cands, const int32_t *__restrict__ counts,
                      const ft8gpu_decode_status *__restrict__ status, int nframes, int max_candidates,
                      int min_score, struct decoder_results *__restrict__ decodes,
                      int32_t *__restrict__ n_results) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= nframes) return;                                             // wave-uniform
    const int words = (max_candidates + 63) / 64;
    // per wave: live mask words | hash[max_candidates] | table[50]
    const size_t per_wave = (size_t)words * 8 + (size_t)max_candidates * 2 + 2 * (kMaxMessages + 2);
    unsigned char *base = s_dyn + (size_t)wave * ((per_wave + 15) & ~(size_t)15);
    unsigned long long *live = reinterpret_cast<unsigned long long *>(base);
    uint16_t *hashes = reinterpret_cast<uint16_t *>(base + (size_t)words * 8);
    uint16_t *table = hashes + max_candidates;

    const ft8gpu_candidate *fc = cands + (size_t)frame * max_candidates;
    const ft8gpu_decode_status *fs = status + (size_t)frame * max_candidates;
    struct decoder_results *out = decodes + (size_t)frame * kMaxMessages;
    const int num_candidates = counts[frame];

    for (int w = 0; w < words; ++w) {
        const int idx = w * 64 + lane;
        bool ok = false;
        if (idx < num_candidates) {
            ok = fc[idx].score >= min_score && fs[idx].ok != 0;               // :1467, :1476-1485
            hashes[idx] = fs[idx].crc_extracted;                              // message.hash
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) live[w] = m;
    }
    if (lane < kMaxMessages) table[lane] = 0;                                 // :1458-1460
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane != 0) return;

    int num_decoded = 0;
    for (int w = 0; w < words; ++w) {
        unsigned long long m = live[w];
        while (m) {                                                           // :1465, candidate order
            const int idx = w * 64 + __builtin_ctzll(m);
            m &= m - 1;
            const uint16_t hash = hashes[idx];
            int idx_hash = hash % kMaxMessages;                               // :1487
            bool found_empty_slot = false, found_duplicate = false;
            int probes = 0;
            do {
                const int t = table[idx_hash];
                if (t == 0) {
                    found_empty_slot = true;
                } else if (hashes[t - 1] == hash && text_equal(fs[t - 1].text, fs[idx].text)) {
                    found_duplicate = true;
                } else {
                    idx_hash = (idx_hash + 1) % kMaxMessages;
                    if (++probes >= kMaxMessages) break;  // table full: drop (reference never terminates here)
                }
            } while (!found_empty_slot && !found_duplicate);

            if (found_empty_slot) {                                           // :1505
                table[idx_hash] = (uint16_t)(idx + 1);
                char text[25];
                for (int i = 0; i < 25; ++i) text[i] = fs[idx].text[i];
                text[24] = 0;
                int pos = 0, len = 0;
                const int t0 = next_token(text, &pos, &len);                  // :1509
                if (t0 >= 0 && len >= 2 && text[t0] == 'C' && text[t0 + 1] == 'Q') {   // :1510 strncmp(.., "CQ", 2)
                    const ft8gpu_candidate cand = fc[idx];
                    const float freq_hz = (cand.freq_offset + (float)cand.freq_sub / 2) * 6.25f;   // :1470
                    int l1 = 0, l2 = 0;
                    const int t1 = next_token(text, &pos, &l1);
                    put_field(out[num_decoded].call, 13, 12, t1 >= 0 ? text + t1 : nullptr, l1);   // :1512
                    const int t2 = next_token(text, &pos, &l2);
                    put_field(out[num_decoded].loc, 7, 6, t2 >= 0 ? text + t2 : nullptr, l2);      // :1514
                    out[num_decoded].freq = (int32_t)freq_hz;                 // :1516
                    out[num_decoded].snr = (int32_t)cand.score;               // :1517
                }
                num_decoded++;                                                // :1520
            }
        }
    }
    n_results[frame] = num_decoded;                                           // :1523
}

}  // namespace

hipError_t launch_spots(const ft8gpu_candidate *cands, const int32_t *counts,
                        const ft8gpu_decode_status *status, int nframes, int max_candidates,
                        int min_score, struct decoder_results *decodes, int32_t *n_results, hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    const int words = (max_candidates + 63) / 64;
    const size_t per_wave = ((size_t)words * 8 + (size_t)max_candidates * 2 + 2 * (kMaxMessages + 2) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(ft8_spots_kernel, dim3((nframes + 3) / 4), dim3(256), 4 * per_wave, s,
                       cands, counts, status, nframes, max_candidates, min_score, decodes, n_results);
    return hipGetLastError();
}
