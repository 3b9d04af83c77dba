// unpack_dev.h -- device-side unpack77 (ft8_lib unpack.c / text.c; reached through ft8_decode(),
// rtlsdr_ft8d.c:1476).  77-bit payload -> message text, message types 0.0 (free text),
// 0.5 (telemetry), 1 / 2 (standard, "/R" "/P"), 4 (non-standard call); everything else fails,
// as in the ft8_lib era the reference links (hashed calls print "<...>").
// Runs on one lane per decoded codeword; plain integer and byte code.
//
// No private memory.  The payload arrives as two 64-bit words (bit i of the message, MSB first, is bit
// 63 - (i & 63) of word i >> 6 -- the ballot words of the BP kernel, bit-reversed), so every field is a shift
// and a mask instead of an index into a byte array; the character buffers whose indices depend on the data
// (trimmed call signs, the assembled text) live in a caller-provided work area, which the BP kernel takes from
// the wave's own LDS tile.  With byte arrays on the stack the kernel needed 128 bytes of scratch per lane, and
// scratch lines that are written are written back to HBM: 5.8 x the bytes of the status records themselves
// (profiles/pmc_traffic.json of round 3).
#pragma once
#include <stdint.h>

namespace ft8dev {

struct UnpackWork {                  // 72 bytes of the caller's (LDS) work area
    char call_to[16], call_de[16], extra[24], tmp[16];
};

__device__ inline char charn(int c, int table_idx) {
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

// appends a string literal at dst (constant stores, no loads), returns pointer to the new NUL
template <int N>
__device__ inline char *put(char *dst, const char (&lit)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = lit[i];
    return dst + (N - 1);
}

// appends the NUL-terminated string at src (work area), returns pointer to the new NUL
__device__ inline char *put_str(char *dst, const char *src) {
    while (*src) *dst++ = *src++;
    *dst = 0;
    return dst;
}

// copies buf[0..n) without leading/trailing blanks, returns new end (NUL written)
__device__ inline char *put_trimmed(char *dst, const char *buf, int n) {
    int a = 0, b = n;
    while (a < n && buf[a] == ' ') ++a;
    while (b > a && buf[b - 1] == ' ') --b;
    for (int i = a; i < b; ++i) *dst++ = buf[i];
    *dst = 0;
    return dst;
}

__device__ inline char *int_to_dd(char *str, int value, int width, bool full_sign) {
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
    *str = 0;
    return str;
}

constexpr uint32_t NTOKENS = 2063592u, MAX22 = 4194304u, MAXGRID4 = 32400u;

// returns new end pointer, or nullptr on failure; tmp: >= 6 bytes of work area
__device__ inline char *unpack_callsign(uint32_t n28, uint32_t ip, int i3, char *result, char *tmp) {
    if (n28 < NTOKENS) {
        if (n28 == 0) return put(result, "DE");
        if (n28 == 1) return put(result, "QRZ");
        if (n28 == 2) return put(result, "CQ");
        if (n28 <= 1002) {
            char *p = put(result, "CQ ");
            return int_to_dd(p, (int)n28 - 3, 3, false);
        }
        if (n28 <= 532443u) {
            uint32_t n = n28 - 1003;
            for (int i = 3; i >= 0; --i) { tmp[i] = charn((int)(n % 27), 4); if (i) n /= 27; }
            char *p = put(result, "CQ ");
            int a = 0;
            while (a < 4 && tmp[a] == ' ') ++a;           // trim_front only
            for (int i = a; i < 4; ++i) *p++ = tmp[i];
            *p = 0;
            return p;
        }
        return nullptr;
    }
    n28 -= NTOKENS;
    if (n28 < MAX22) return put(result, "<...>");
    uint32_t n = n28 - MAX22;
    tmp[5] = charn((int)(n % 27), 4); n /= 27;
    tmp[4] = charn((int)(n % 27), 4); n /= 27;
    tmp[3] = charn((int)(n % 27), 4); n /= 27;
    tmp[2] = charn((int)(n % 10), 3); n /= 10;
    tmp[1] = charn((int)(n % 36), 2); n /= 36;
    tmp[0] = charn((int)(n % 37), 1);
    char *p = put_trimmed(result, tmp, 6);
    if (p == result) return nullptr;
    if (ip) {
        if (i3 == 1) p = put(p, "/R");
        else if (i3 == 2) p = put(p, "/P");
    }
    return p;
}

// w0: message bits 0..63, w1: bits 64..76 in its top 13 bits (anything below is ignored), MSB first.
// text: >= 25 bytes.  returns 0 or a negative code
__device__ inline int unpack77(uint64_t w0, uint64_t w1, char *text, UnpackWork *wk) {
    char *call_to = wk->call_to, *call_de = wk->call_de, *extra = wk->extra;
    call_to[0] = call_de[0] = extra[0] = 0;
    const int i3 = (int)(w1 >> 51) & 7;                                  // bits 74..76
    int rc = -1;
    if (i3 == 0) {
        const int n3 = (int)(w1 >> 54) & 7;                              // bits 71..73
        if (n3 == 0 || n3 == 5) {
            // the first 71 bits as one number: top 7 bits | 64 bits
            uint32_t hi = (uint32_t)(w0 >> 57);
            uint64_t lo = (w0 << 7) | (w1 >> 57);
            if (n3 == 0) {                                               // free text, base 42
                char *c13 = wk->tmp;
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
                    c13[idx] = charn((int)(cur % 42), 0);
                }
                put_trimmed(extra, c13, 13);
            } else {                                                     // telemetry, 18 hex digits
#pragma unroll
                for (int k = 0; k < 18; ++k) {
                    const int nib = k < 2 ? (int)(hi >> (4 - 4 * k)) & 15 : (int)(lo >> (60 - 4 * (k - 2))) & 15;
                    extra[k] = (char)(nib > 9 ? nib - 10 + 'A' : nib + '0');
                }
                extra[18] = 0;
            }
            rc = 0;
        }
    } else if (i3 == 1 || i3 == 2) {
        const uint32_t n29a = (uint32_t)(w0 >> 35);                      // bits 0..28
        const uint32_t n29b = (uint32_t)(w0 >> 6) & 0x1FFFFFFFu;         // bits 29..57
        const int ir = (int)(w0 >> 5) & 1;                               // bit 58
        const uint32_t igrid4 = ((uint32_t)(w0 & 0x1F) << 10) | (uint32_t)(w1 >> 54);   // bits 59..73
        if (!unpack_callsign(n29a >> 1, n29a & 1, i3, call_to, wk->tmp)) return -1;
        if (!unpack_callsign(n29b >> 1, n29b & 1, i3, call_de, wk->tmp)) return -2;
        char *dst = extra;
        if (igrid4 <= MAXGRID4) {
            if (ir) dst = put(dst, "R ");
            uint32_t n = igrid4;
            dst[4] = 0;
            dst[3] = (char)('0' + n % 10); n /= 10;
            dst[2] = (char)('0' + n % 10); n /= 10;
            dst[1] = (char)('A' + n % 18); n /= 18;
            dst[0] = (char)('A' + n % 18);
        } else {
            const int irpt = (int)igrid4 - (int)MAXGRID4;
            if (irpt == 1) extra[0] = 0;
            else if (irpt == 2) put(dst, "RRR");
            else if (irpt == 3) put(dst, "RR73");
            else if (irpt == 4) put(dst, "73");
            else {
                if (ir) *dst++ = 'R';
                int_to_dd(dst, irpt - 35, 2, true);
            }
        }
        rc = 0;
    } else if (i3 == 4) {
        uint64_t n58 = ((w0 & 0x000FFFFFFFFFFFFFull) << 6) | (w1 >> 58);  // bits 12..69
        const int iflip = (int)(w1 >> 57) & 1;                           // bit 70
        const int nrpt = (int)(w1 >> 55) & 3;                            // bits 71..72
        const int icq = (int)(w1 >> 54) & 1;                             // bit 73
        char *c11 = wk->tmp;
        for (int i = 10; i >= 0; --i) { c11[i] = charn((int)(n58 % 38), 5); if (i) n58 /= 38; }
        // call_1 / call_2 of upstream: the plain call and the hashed one ("<...>"), swapped by iflip
        if (icq == 0) {
            if (iflip) put_trimmed(call_to, c11, 11); else put(call_to, "<...>");
            if (nrpt == 1) put(extra, "RRR");
            else if (nrpt == 2) put(extra, "RR73");
            else if (nrpt == 3) put(extra, "73");
        } else {
            put(call_to, "CQ");
        }
        if (iflip) put(call_de, "<...>"); else put_trimmed(call_de, c11, 11);
        rc = 0;
    }
    if (rc < 0) return rc;
    char *dst = text;
    dst[0] = 0;
    if (call_to[0]) { dst = put_str(dst, call_to); *dst++ = ' '; }
    if (call_de[0]) { dst = put_str(dst, call_de); *dst++ = ' '; }
    dst = put_str(dst, extra);
    *dst = 0;
    return 0;
}

}  // namespace ft8dev
