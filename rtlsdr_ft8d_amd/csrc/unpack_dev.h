// unpack_dev.h -- device-side unpack77 (ft8_lib unpack.c / text.c; reached through ft8_decode(),
// rtlsdr_ft8d.c:1476).  77-bit payload -> message text, message types 0.0 (free text),
// 0.5 (telemetry), 1 / 2 (standard, "/R" "/P"), 4 (non-standard call); everything else fails,
// as in the ft8_lib era the reference links (hashed calls print "<...>").
// Runs on one lane per decoded codeword; plain integer and byte code.
#pragma once
#include <stdint.h>

namespace ft8dev {

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
        if (c < 5) {
            const char t[5] = { '+', '-', '.', '/', '?' };
            return t[c];
        }
    } else if (table_idx == 5) {
        if (c == 0) return '/';
    }
    return '_';
}

// appends src (NUL terminated) at dst, returns pointer to the new NUL
__device__ inline char *put(char *dst, const char *src) {
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

// returns new end pointer, or nullptr on failure
__device__ inline char *unpack_callsign(uint32_t n28, uint32_t ip, int i3, char *result) {
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
            char aaaa[4];
            for (int i = 3; i >= 0; --i) { aaaa[i] = charn((int)(n % 27), 4); if (i) n /= 27; }
            char *p = put(result, "CQ ");
            int a = 0;
            while (a < 4 && aaaa[a] == ' ') ++a;          // trim_front only
            for (int i = a; i < 4; ++i) *p++ = aaaa[i];
            *p = 0;
            return p;
        }
        return nullptr;
    }
    n28 -= NTOKENS;
    if (n28 < MAX22) return put(result, "<...>");
    uint32_t n = n28 - MAX22;
    char cs[6];
    cs[5] = charn((int)(n % 27), 4); n /= 27;
    cs[4] = charn((int)(n % 27), 4); n /= 27;
    cs[3] = charn((int)(n % 27), 4); n /= 27;
    cs[2] = charn((int)(n % 10), 3); n /= 10;
    cs[1] = charn((int)(n % 36), 2); n /= 36;
    cs[0] = charn((int)(n % 37), 1);
    char *p = put_trimmed(result, cs, 6);
    if (p == result) return nullptr;
    if (ip) {
        if (i3 == 1) p = put(p, "/R");
        else if (i3 == 2) p = put(p, "/P");
    }
    return p;
}

// a77: 10 bytes (bits 77..79 already cleared).  text: >= 25 bytes.  returns 0 or a negative code
__device__ inline int unpack77(const uint8_t *a77, char *text) {
    char call_to[14], call_de[14], extra[20];
    call_to[0] = call_de[0] = extra[0] = 0;
    const int i3 = (a77[9] >> 3) & 7;
    int rc = -1;
    if (i3 == 0) {
        const int n3 = ((a77[8] << 2) & 4) | ((a77[9] >> 6) & 3);
        if (n3 == 0 || n3 == 5) {
            uint8_t b71[9];
            uint8_t carry = 0;
            for (int i = 0; i < 9; ++i) {
                b71[i] = (uint8_t)(carry | (a77[i] >> 1));
                carry = (a77[i] & 1) ? 0x80 : 0;
            }
            if (n3 == 0) {                                   // free text, base 42
                char c13[13];
                for (int idx = 12; idx >= 0; --idx) {
                    uint32_t rem = 0;
                    for (int i = 0; i < 9; ++i) {
                        rem = (rem << 8) | b71[i];
                        b71[i] = (uint8_t)(rem / 42);
                        rem = rem % 42;
                    }
                    c13[idx] = charn((int)rem, 0);
                }
                put_trimmed(extra, c13, 13);
            } else {                                         // telemetry, 18 hex digits
                for (int i = 0; i < 9; ++i) {
                    const int n1 = b71[i] >> 4, n2 = b71[i] & 15;
                    extra[2 * i] = (char)(n1 > 9 ? n1 - 10 + 'A' : n1 + '0');
                    extra[2 * i + 1] = (char)(n2 > 9 ? n2 - 10 + 'A' : n2 + '0');
                }
                extra[18] = 0;
            }
            rc = 0;
        }
    } else if (i3 == 1 || i3 == 2) {
        uint32_t n29a = ((uint32_t)a77[0] << 21) | ((uint32_t)a77[1] << 13) | ((uint32_t)a77[2] << 5) | (a77[3] >> 3);
        uint32_t n29b = ((uint32_t)(a77[3] & 7) << 26) | ((uint32_t)a77[4] << 18) | ((uint32_t)a77[5] << 10) |
                        ((uint32_t)a77[6] << 2) | (a77[7] >> 6);
        const int ir = (a77[7] >> 5) & 1;
        const uint32_t igrid4 = ((uint32_t)(a77[7] & 0x1F) << 10) | ((uint32_t)a77[8] << 2) | (a77[9] >> 6);
        if (!unpack_callsign(n29a >> 1, n29a & 1, i3, call_to)) return -1;
        if (!unpack_callsign(n29b >> 1, n29b & 1, i3, call_de)) return -2;
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
        uint64_t n58 = ((uint64_t)(a77[1] & 0x0F) << 54) | ((uint64_t)a77[2] << 46) | ((uint64_t)a77[3] << 38) |
                       ((uint64_t)a77[4] << 30) | ((uint64_t)a77[5] << 22) | ((uint64_t)a77[6] << 14) |
                       ((uint64_t)a77[7] << 6) | ((uint64_t)a77[8] >> 2);
        const int iflip = (a77[8] >> 1) & 1;
        const int nrpt = ((a77[8] & 1) << 1) | (a77[9] >> 7);
        const int icq = (a77[9] >> 6) & 1;
        char c11[11];
        for (int i = 10; i >= 0; --i) { c11[i] = charn((int)(n58 % 38), 5); if (i) n58 /= 38; }
        char t11[12];
        put_trimmed(t11, c11, 11);
        const char *hashed = "<...>";
        const char *call_1 = iflip ? t11 : hashed;
        const char *call_2 = iflip ? hashed : t11;
        if (icq == 0) {
            put(call_to, call_1);
            if (nrpt == 1) put(extra, "RRR");
            else if (nrpt == 2) put(extra, "RR73");
            else if (nrpt == 3) put(extra, "73");
        } else {
            put(call_to, "CQ");
        }
        put(call_de, call_2);
        rc = 0;
    }
    if (rc < 0) return rc;
    char *dst = text;
    dst[0] = 0;
    if (call_to[0]) { dst = put(dst, call_to); *dst++ = ' '; }
    if (call_de[0]) { dst = put(dst, call_de); *dst++ = ' '; }
    dst = put(dst, extra);
    *dst = 0;
    return 0;
}

}  // namespace ft8dev
