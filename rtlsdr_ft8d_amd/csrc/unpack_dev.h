// unpack_dev.h -- device-side unpack77 (ft8_lib unpack.c / text.c; reached through ft8_decode(),
// rtlsdr_ft8d.c:1476).  77-bit payload -> message text, message types 0.0 (free text),
// 0.5 (telemetry), 1 / 2 (standard, "/R" "/P"), 4 (non-standard call); everything else fails,
// as in the ft8_lib era the reference links (hashed calls print "<...>").
// Runs on one lane per decoded codeword; plain integer code.
//
// No private memory, no loads, and almost no vector instructions.  The payload arrives as two 64-bit words (bit i of
// the message, MSB first, is bit 63 - (i & 63) of word i >> 6 -- the ballot words of the BP kernel, bit-reversed), so
// every field is a shift and a mask instead of an index into a byte array.  The characters of a call sign or of a free
// text are computed into REGISTERS (arrays with compile-time indices only) and stored straight into the text at a
// running position: nothing is ever read back -- upstream's intermediate strings (call_to, call_de, extra, the trimmed
// copies) existed only to be copied again.  Everything is force-inlined into the kernel: the two words are wave-uniform
// there (SGPRs), so the compiler keeps the whole field arithmetic -- the divisions by 27, 37, 38, 42 ... -- on the SCALAR
// unit, which has slack beside the VALU-bound BP loop; only the byte stores are vector instructions.
// History: byte arrays on the stack cost the kernel a 128-byte scratch segment and 5.8 x its record bytes in HBM writes
// (round 3); LDS buffers behind an out-of-line call removed the scratch but ran about 300 vector instructions per
// decoded message on one lane (0.045 ms per 4096 frames, measured by compiling unpack77 out); this form gives 0.028 ms
// of that back (profiles/r04_ab_zero_scratch.json).
// The destination must be zero-filled by the caller (the text is NUL-terminated by what is already there).
#pragma once
#include <stdint.h>

namespace ft8dev {

__device__ __forceinline__ char charn(int c, int table_idx) {
    // 0: " 0-9A-Z+-./?"  1: " 0-9A-Z"  2: "0-9A-Z"  3: "0-9"  4: " A-Z"  5: " 0-9A-Z/"
    if (table_idx != 2 && table_idx != 3) {
        if (c == 0) return ' ';
        c -= 1;
    }
    if (table_idx != 4) {
        if (c < 10) return (char)('0' + c);
        c -= 10;
    }
    if (table_idx != 3) {
        if (c < 26) return (char)('A' + c);
        c -= 26;
    }
    if (table_idx == 0) {
        if (c == 0) return '+';
        if (c < 4) return (char)('-' + (c - 1));          // "-./" are consecutive codes
        if (c == 4) return '?';
    } else if (table_idx == 5) {
        if (c == 0) return '/';
    }
    return '_';
}

// appends a string literal at dst (constant stores), returns the new end
template <int N>
__device__ __forceinline__ char *put(char *dst, const char (&lit)[N]) {
#pragma unroll
    for (int i = 0; i < N - 1; ++i) dst[i] = lit[i];
    return dst + (N - 1);
}

// appends c[lo .. hi) -- the array is indexed with compile-time constants only (registers), the position runs
template <int N>
__device__ __forceinline__ char *put_range(char *dst, const char (&c)[N], int lo, int hi) {
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i >= lo && i < hi) *dst++ = c[i];
    return dst;
}

// [lo, hi) of c[0 .. N) without leading / trailing blanks (trim_front / trim_back of upstream's text.c)
template <int N>
__device__ __forceinline__ void trim_bounds(const char (&c)[N], int &lo, int &hi) {
    lo = 0;
    bool lead = true;
#pragma unroll
    for (int i = 0; i < N; ++i) { lead = lead && c[i] == ' '; lo += lead ? 1 : 0; }
    hi = N;
    bool trail = true;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) { trail = trail && c[i] == ' ' && i >= lo; hi -= trail ? 1 : 0; }
}

__device__ __forceinline__ char *int_to_dd(char *str, int value, int width, bool full_sign) {
    if (value < 0) { *str++ = '-'; value = -value; }
    else if (full_sign) { *str++ = '+'; }
    int divisor = 1;
    for (int i = 0; i < width - 1; ++i) divisor *= 10;
    while (divisor >= 1) {
        const int digit = value / divisor;
        *str++ = (char)('0' + digit);       // may leave '0'..'9' when value >= 10^width, as upstream
        value -= digit * divisor;
        divisor /= 10;
    }
    return str;
}

constexpr uint32_t NTOKENS = 2063592u, MAX22 = 4194304u, MAXGRID4 = 32400u;

// writes the call sign at dst; returns the new end, or nullptr on failure
__device__ __forceinline__ char *unpack_callsign(uint32_t n28, uint32_t ip, int i3, char *dst) {
    if (n28 < NTOKENS) {
        if (n28 == 0) return put(dst, "DE");
        if (n28 == 1) return put(dst, "QRZ");
        if (n28 == 2) return put(dst, "CQ");
        if (n28 <= 1002) return int_to_dd(put(dst, "CQ "), (int)n28 - 3, 3, false);
        if (n28 <= 532443u) {
            uint32_t n = n28 - 1003;
            char c[4];
            c[3] = charn((int)(n % 27), 4); n /= 27;
            c[2] = charn((int)(n % 27), 4); n /= 27;
            c[1] = charn((int)(n % 27), 4); n /= 27;
            c[0] = charn((int)(n % 27), 4);
            int lo, hi;
            trim_bounds(c, lo, hi);
            return put_range(put(dst, "CQ "), c, lo, 4);   // trim_front only
        }
        return nullptr;
    }
    n28 -= NTOKENS;
    if (n28 < MAX22) return put(dst, "<...>");
    uint32_t n = n28 - MAX22;
    char c[6];
    c[5] = charn((int)(n % 27), 4); n /= 27;
    c[4] = charn((int)(n % 27), 4); n /= 27;
    c[3] = charn((int)(n % 27), 4); n /= 27;
    c[2] = charn((int)(n % 10), 3); n /= 10;
    c[1] = charn((int)(n % 36), 2); n /= 36;
    c[0] = charn((int)(n % 37), 1);
    int lo, hi;
    trim_bounds(c, lo, hi);
    if (hi <= lo) return nullptr;
    char *p = put_range(dst, c, lo, hi);
    if (ip) {
        if (i3 == 1) p = put(p, "/R");
        else if (i3 == 2) p = put(p, "/P");
    }
    return p;
}

// w0: message bits 0..63, w1: bits 64..76 in its top 13 bits (anything below is ignored), MSB first.
// text: >= 25 zero-filled bytes.  returns 0 or a negative code (the text may then hold leftovers: the caller clears it)
// Upstream assembles "call_to call_de extra" from three strings, with a blank behind each non-empty call: here the
// three parts are written in place, and the blank follows a part that wrote at least one character.
__device__ __forceinline__ int unpack77(uint64_t w0, uint64_t w1, char *text) {
    const int i3 = (int)(w1 >> 51) & 7;                                  // bits 74..76
    if (i3 == 0) {
        const int n3 = (int)(w1 >> 54) & 7;                              // bits 71..73
        if (n3 != 0 && n3 != 5) return -1;
        // the first 71 bits as one number: top 7 bits | 64 bits
        uint32_t hi = (uint32_t)(w0 >> 57);
        uint64_t lo = (w0 << 7) | (w1 >> 57);
        if (n3 == 0) {                                                   // free text, base 42
            char c[13];
#pragma unroll
            for (int idx = 12; idx >= 0; --idx) {
                // long division of (hi, lo) by 42 over 32-bit limbs: the same quotient and remainder as
                // upstream's byte-wise division of the nine bytes
                uint64_t cur = hi;
                hi = (uint32_t)(cur / 42);
                cur = ((cur % 42) << 32) | (lo >> 32);
                const uint64_t q1 = cur / 42;
                cur = ((cur % 42) << 32) | (lo & 0xFFFFFFFFu);
                const uint64_t q0 = cur / 42;
                lo = (q1 << 32) | q0;
                c[idx] = charn((int)(cur % 42), 0);
            }
            int a, b;
            trim_bounds(c, a, b);
            put_range(text, c, a, b);
        } else {                                                         // telemetry, 18 hex digits
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                const int nib = k < 2 ? (int)(hi >> (4 - 4 * k)) & 15 : (int)(lo >> (60 - 4 * (k - 2))) & 15;
                text[k] = (char)(nib > 9 ? nib - 10 + 'A' : nib + '0');
            }
        }
        return 0;
    }
    if (i3 == 1 || i3 == 2) {
        const uint32_t n29a = (uint32_t)(w0 >> 35);                      // bits 0..28
        const uint32_t n29b = (uint32_t)(w0 >> 6) & 0x1FFFFFFFu;         // bits 29..57
        const int ir = (int)(w0 >> 5) & 1;                               // bit 58
        const uint32_t igrid4 = ((uint32_t)(w0 & 0x1F) << 10) | (uint32_t)(w1 >> 54);   // bits 59..73
        char *p = unpack_callsign(n29a >> 1, n29a & 1, i3, text);
        if (!p) return -1;
        *p++ = ' ';
        p = unpack_callsign(n29b >> 1, n29b & 1, i3, p);
        if (!p) return -2;
        *p++ = ' ';
        if (igrid4 <= MAXGRID4) {
            if (ir) p = put(p, "R ");
            uint32_t n = igrid4;
            p[3] = (char)('0' + n % 10); n /= 10;
            p[2] = (char)('0' + n % 10); n /= 10;
            p[1] = (char)('A' + n % 18); n /= 18;
            p[0] = (char)('A' + n % 18);
        } else {
            const int irpt = (int)igrid4 - (int)MAXGRID4;
            if (irpt == 1) { /* no third field: the blank behind the second call stays, as upstream leaves it */ }
            else if (irpt == 2) put(p, "RRR");
            else if (irpt == 3) put(p, "RR73");
            else if (irpt == 4) put(p, "73");
            else {
                if (ir) *p++ = 'R';
                int_to_dd(p, irpt - 35, 2, true);
            }
        }
        return 0;
    }
    if (i3 == 4) {
        uint64_t n58 = ((w0 & 0x000FFFFFFFFFFFFFull) << 6) | (w1 >> 58);  // bits 12..69
        const int iflip = (int)(w1 >> 57) & 1;                           // bit 70
        const int nrpt = (int)(w1 >> 55) & 3;                            // bits 71..72
        const int icq = (int)(w1 >> 54) & 1;                             // bit 73
        char c[11];
#pragma unroll
        for (int i = 10; i >= 0; --i) { c[i] = charn((int)(n58 % 38), 5); if (i) n58 /= 38; }
        int a, b;
        trim_bounds(c, a, b);
        // call_1 / call_2 of upstream: the plain call and the hashed one ("<...>"), swapped by iflip
        char *p = text;
        if (icq) p = put(p, "CQ");
        else if (iflip) p = put_range(p, c, a, b);
        else p = put(p, "<...>");
        if (p != text) *p++ = ' ';
        char *q = p;
        if (iflip) q = put(q, "<...>");
        else q = put_range(q, c, a, b);
        if (q != p) *q++ = ' ';
        if (icq == 0) {
            if (nrpt == 1) put(q, "RRR");
            else if (nrpt == 2) put(q, "RR73");
            else if (nrpt == 3) put(q, "73");
        }
        return 0;
    }
    return -1;
}

}  // namespace ft8dev
