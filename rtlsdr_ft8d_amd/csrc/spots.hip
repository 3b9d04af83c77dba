// spots.hip -- stages a6, a8, a9 of the hot path, rtlsdr_ft8d.c:1452-1523:
// candidate loop with score gate and freq_hz (:1465-1470), open-addressing dedup table of
// K_MAX_MESSAGES = 50 slots keyed by message.hash % 50 with strcmp on the text (:1487-1507),
// CQ-only spot fill via strtok semantics (:1509-1518), num_decoded++ for every unique message (:1520).
//
// The reference's table is a set: a message is new iff no EARLIER candidate of the frame carried the
// same hash and strcmp-equal text, new messages are numbered in candidate order, and the table index
// itself is never observable.  (With the table-full fence: the first 50 new messages are kept.)  That
// is a data-parallel formulation: one wave per frame, lane = candidate, 64 candidates at a time.
// Every lane checks its message against the messages kept so far and against the earlier lanes of its
// chunk (16-bit hash first, text only on a hash match), a ballot + prefix popcount numbers the new
// ones, and every new message parses its own text and writes its own spot record.  Texts, hashes and
// candidates are staged in LDS, so the only global traffic is one coalesced fetch per candidate
// record and the stores of the spot records.
#include "ft8gpu_internal.h"

namespace {

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kTextDw = 7;                    // message_t.text[25] in 7 aligned dwords (bytes 25..27 are 0)

// Staged texts are kept in CANONICAL form: every byte behind the first NUL is zero (what strcmp never looks at), so
// strcmp(a, b) == 0 is equality of the seven dwords -- a dozen instructions instead of a 25-step byte loop, and that
// comparison runs once per (lane, kept message) and once per (lane, unique message of the chunk).  The LDPC kernel
// writes its texts into zero-filled records, but collect_spots also takes caller-made records: canonicalising here
// keeps the reference's semantics for any input.
__device__ __forceinline__ void canonical_text(uint32_t (&w)[kTextDw]) {
    bool open = true;                               // no terminator seen yet
#pragma unroll
    for (int k = 0; k < kTextDw; ++k) {
        const uint32_t v = open ? w[k] : 0u;
        const uint32_t z = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;   // 0x80 in every zero byte (exact, no carries between bytes)
        const uint32_t low = z & (0u - z);          // the first one: 0x80 << 8 i  (0 if none)
        w[k] = v & ((low >> 7) - 1u);               // bytes below it (all four if none)
        open = open && z == 0u;
    }
}
// strcmp == 0 between the lane's own text (registers) and a staged one
__device__ __forceinline__ bool text_equal(const uint32_t (&mine)[kTextDw], const uint32_t *other) {
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < kTextDw; ++k) d |= mine[k] ^ other[k];
    return d == 0u;
}

// strtok(text, " ") semantics on message_t.text with text[24] taken as the terminator (:1509 works on a
// NUL-terminated char[25]): returns start index of the next token at or after *pos, or -1; *len
// receives the token length, *pos is advanced past the token and one delimiter
__device__ inline int next_token(const char *s, int *pos, int *len) {
    int p = *pos;
    while (p < 24 && s[p] == ' ') ++p;
    if (p >= 24 || s[p] == 0) { *pos = p; return -1; }
    const int start = p;
    while (p < 24 && s[p] != 0 && s[p] != ' ') ++p;
    *len = p - start;
    *pos = (p >= 24 || s[p] == 0) ? p : p + 1;
    return start;
}

// snprintf(dst, cap, "%.<prec>s", tok) ; tok == NULL prints "(null)" (glibc).  tok points into LDS.
__device__ inline void put_field(char *dst, int cap, int prec, const char *tok, int len) {
    int n;
    if (tok == nullptr) {
        const uint64_t null_str = 0x00296C6C756E28ull;          // "(null)" little-endian
        n = 6 < prec ? 6 : prec;
        if (n > cap - 1) n = cap - 1;
        for (int i = 0; i < n; ++i) dst[i] = (char)((null_str >> (8 * i)) & 0xFF);
    } else {
        n = len < prec ? len : prec;
        if (n > cap - 1) n = cap - 1;
        for (int i = 0; i < n; ++i) dst[i] = tok[i];
    }
    dst[n] = 0;
}

// LDS per wave: staged texts [64][7 dw] and hashes [64] of the current chunk; texts and hashes of the
// messages kept so far [50]
struct SpotsWaveLds {
    uint32_t ctext[64][kTextDw];
    uint32_t ttext[kMaxMessages][kTextDw];
    uint16_t chash[64];
    uint16_t thash[kMaxMessages];
};

__global__ __launch_bounds__(256)
void ft8_spots_kernel(const ft8gpu_candidate *__restrict__ cands, const int32_t *__restrict__ counts,
                      const ft8gpu_decode_status *__restrict__ status, int nframes, int max_candidates,
                      int min_score, struct decoder_results *__restrict__ decodes,
                      int32_t *__restrict__ n_results) {
    __shared__ __attribute__((aligned(16))) SpotsWaveLds s_all[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= nframes) return;                                             // wave-uniform
    SpotsWaveLds &L = s_all[wave];

    const ft8gpu_candidate *fc = cands + (size_t)frame * max_candidates;
    const ft8gpu_decode_status *fs = status + (size_t)frame * max_candidates;
    struct decoder_results *out = decodes + (size_t)frame * kMaxMessages;
    const int words = (max_candidates + 63) / 64;
    const unsigned long long below = (1ull << lane) - 1ull;

    // One chunk of 64 candidates per round; the candidate and the nine record dwords a lane needs are fetched a chunk
    // AHEAD (and the first chunk together with the frame's candidate count: rows below max_candidates always exist),
    // so the kernel waits for memory once, not four times in a row -- it is nothing but a chain of latencies.
    struct Fetched { uint64_t cand; uint32_t hash_dw, st2, t[7]; };
    auto fetch = [&](int w) {
        Fetched f = {};
        const int idx = w * 64 + lane;
        if (idx < max_candidates) {
            f.cand = reinterpret_cast<const uint64_t *>(fc)[idx];
            const uint32_t *rec = reinterpret_cast<const uint32_t *>(fs + idx);                   // 48-byte record, 12 dwords
            f.hash_dw = rec[1];                                                                   // crc_extracted | crc_calculated << 16
            f.st2 = rec[2];                                                                       // unpack_status | ok << 8 | a91[0..1]
#pragma unroll
            for (int k = 0; k < 7; ++k) f.t[k] = rec[5 + k];                                      // bytes 20..47; text starts at byte 22
        }
        return f;
    };
    Fetched nxt = fetch(0);
    const int num_candidates = counts[frame];

    int num_decoded = 0;                                                      // wave-uniform
    for (int w = 0; w < words; ++w) {                                         // :1465, candidate order
        const int idx = w * 64 + lane;
        const Fetched cur = nxt;
        if (w + 1 < words) nxt = fetch(w + 1);
        const uint64_t cand_bits = cur.cand;
        const bool ok = idx < num_candidates && (int16_t)(cand_bits & 0xFFFFu) >= min_score && ((cur.st2 >> 8) & 0xFFu) != 0;   // :1467, :1476-1485
        uint32_t my_hash = 0;
        uint32_t mine[kTextDw] = {};                                          // the lane's text, canonical
        const unsigned long long live = __ballot(ok);
        if (live == 0ull) continue;                                           // wave-uniform
        if (ok) {
            my_hash = cur.hash_dw & 0xFFFFu;                                                      // crc_extracted = message.hash
            L.chash[lane] = (uint16_t)my_hash;
#pragma unroll
            for (int k = 0; k < kTextDw; ++k) {
                mine[k] = (cur.t[k] >> 16) | ((k + 1 < 7 ? cur.t[k + 1] : 0u) << 16);
                if (k == kTextDw - 1) mine[k] &= 0xFFu;                       // text[24] only (byte 47 is the record's pad)
            }
            canonical_text(mine);
#pragma unroll
            for (int k = 0; k < kTextDw; ++k) L.ctext[lane][k] = mine[k];
        }
        wave_lds_sync();

        // :1487-1503 -- is the message already known?  First against the messages kept from earlier chunks.
        bool dup = false;
        for (int t = 0; t < num_decoded; ++t)
            if (ok && L.thash[t] == my_hash && text_equal(mine, L.ttext[t])) dup = true;
        // Then inside the chunk, leader by leader: the first lane that is still undecided cannot have an equal
        // message before it (that one would be a leader, and would have struck it), so it is NEW; it strikes every
        // later lane carrying its message.  One round per UNIQUE message of the chunk (about a dozen) instead of
        // one per decoded candidate (about forty), and a text comparison only where the 16-bit hashes agree.
        unsigned long long pending = __ballot(ok && !dup), fresh = 0ull;
        while (pending != 0ull) {                                             // wave-uniform
            const int j = __builtin_ctzll(pending);
            fresh |= 1ull << j;
            const bool same = ok && !dup && lane > j && L.chash[j] == my_hash && text_equal(mine, L.ctext[j]);
            dup = dup || same;
            pending &= ~((1ull << j) | __ballot(same));
        }
        const int rank = num_decoded + __popcll(fresh & below);               // position among the frame's unique messages
        const bool keep = ((fresh >> lane) & 1ull) != 0ull && rank < kMaxMessages;                  // table full: drop (the reference never terminates there)
        if (keep) {                                                           // :1505-1520
#pragma unroll
            for (int k = 0; k < kTextDw; ++k) L.ttext[rank][k] = mine[k];
            L.thash[rank] = (uint16_t)my_hash;
            const char *text = reinterpret_cast<const char *>(L.ctext[lane]);
            int pos = 0, len = 0;
            const int t0 = next_token(text, &pos, &len);                      // :1509
            if (t0 >= 0 && len >= 2 && text[t0] == 'C' && text[t0 + 1] == 'Q') {   // :1510 strncmp(.., "CQ", 2)
                const int score = (int16_t)(cand_bits & 0xFFFFu), freq_offset = (int16_t)((cand_bits >> 32) & 0xFFFFu);
                const int freq_sub = (int)((cand_bits >> 56) & 0xFFu);
                const float freq_hz = (freq_offset + (float)freq_sub / 2) * 6.25f;                  // :1470
                int l1 = 0, l2 = 0;
                const int t1 = next_token(text, &pos, &l1);
                put_field(out[rank].call, 13, 12, t1 >= 0 ? text + t1 : nullptr, l1);              // :1512
                const int t2 = next_token(text, &pos, &l2);
                put_field(out[rank].loc, 7, 6, t2 >= 0 ? text + t2 : nullptr, l2);                 // :1514
                out[rank].freq = (int32_t)freq_hz;                            // :1516
                out[rank].snr = (int32_t)score;                               // :1517
            }
        }
        num_decoded += __popcll(__ballot(keep));                              // :1520
        wave_lds_sync();                                                      // staging rows are rewritten by the next 64
    }
    if (lane == 0) n_results[frame] = num_decoded;                            // :1523
}

}  // namespace

hipError_t launch_spots(const ft8gpu_candidate *cands, const int32_t *counts,
                        const ft8gpu_decode_status *status, int nframes, int max_candidates,
                        int min_score, struct decoder_results *decodes, int32_t *n_results, hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    hipLaunchKernelGGL(ft8_spots_kernel, dim3((nframes + 3) / 4), dim3(256), 0, s,
                       cands, counts, status, nframes, max_candidates, min_score, decodes, n_results);
    return hipGetLastError();
}
