/*
 * ft8_pack.c -- encoder tooling of libft8gpu.so (host C, no GPU): message text -> 77-bit payload -> CRC-14 ->
 * LDPC(174,91) -> 79 tones.  The reference reaches it once, in its self-test (pack77 rtlsdr_ft8d.c:927, ft8_encode :934);
 * here it also feeds the synthetic workloads, which is why it covers the message types a
 * receiver meets on the air and not only the self-test's "CQ K1JT FN20QI".
 *
 * Written from the published protocol (Franke, Somerville, Taylor: "The FT4 and FT8 Communication Protocols", QEX
 * July/August 2020: field widths of the message types, the 28-bit call sign code, the 15-bit grid / report code, the
 * 71-bit free text, the hashed-call multiplier), not from ft8_lib's pack.c (absent from the reference tree).  What this
 * packer accepts is a superset of what ft8_lib's pack77 of the reference's era packs (standard calls with grid, report,
 * RRR, RR73, 73, else free text) -- see pack77() at the end.
 */
#include "../../include/ft8gpu.h"
#include "../../include/ft8_lib/ft8/pack.h"
#include "../../include/ft8_lib/ft8/encode.h"
#include "ft8_tables.h"

#include <string.h>

#define NTOKENS  2063592
#define MAX22    4194304
#define MAXGRID4 32400

static const char A_ALNUM_SP[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";       /* 37: first character of a call */
static const char A_ALNUM[]    = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";        /* 36: second */
static const char A_DIGIT[]    = "0123456789";                                  /* 10: third */
static const char A_LETTER_SP[] = " ABCDEFGHIJKLMNOPQRSTUVWXYZ";                /* 27: suffix letters, CQ modifiers */
static const char A_TEXT[]     = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ+-./?";  /* 42: free text */
static const char A_CALL11[]   = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/";      /* 38: non-standard calls and hashes */

static int idx_in(const char *alphabet, char c) {
    const char *p = c ? strchr(alphabet, c) : NULL;
    return p ? (int)(p - alphabet) : -1;
}

static int is_digit(char c) { return c >= '0' && c <= '9'; }
static int is_letter(char c) { return c >= 'A' && c <= 'Z'; }

/* ---- bit writer: fields MSB first into 10 bytes (77 bits used) -------------------------------- */
typedef struct { uint8_t b[10]; int at; } bitbuf;

static void put_bits(bitbuf *w, uint64_t value, int width) {
    for (int i = width - 1; i >= 0; --i, ++w->at)
        if ((value >> i) & 1u) w->b[w->at >> 3] |= (uint8_t)(0x80u >> (w->at & 7));
}

/* ---- hashed calls: m-bit hash of a call of up to 11 characters (left-justified, base 38) -------- */
static int call_hash(const char *call, int len, int bits, uint32_t *out) {
    if (len < 1 || len > 11) return -1;
    uint64_t n = 0;
    for (int i = 0; i < 11; ++i) {
        const int j = idx_in(A_CALL11, i < len ? call[i] : ' ');
        if (j < 0) return -1;
        n = 38u * n + (uint64_t)j;
    }
    *out = (uint32_t)((47055833459ull * n) >> (64 - bits));
    return 0;
}

/* ---- 28-bit code of a standard call (no suffix): -1 if the text is not one ---------------------- */
static int32_t pack_basecall(const char *call, int len, int workarounds) {
    char c6[7] = "      ";
    if (workarounds && len >= 4 && len <= 7 && !strncmp(call, "3DA0", 4)) {                 /* Eswatini: 3DA0XYZ travels as 3D0XYZ */
        memcpy(c6, "3D0", 3);
        memcpy(c6 + 3, call + 4, (size_t)(len - 4));
    } else if (workarounds && len >= 3 && len <= 7 && !strncmp(call, "3X", 2) && is_letter(call[2])) {   /* Guinea: 3XA0XYZ travels as QA0XYZ */
        c6[0] = 'Q';
        memcpy(c6 + 1, call + 2, (size_t)(len - 2));
    } else if (len >= 3 && len <= 6 && is_digit(call[2])) memcpy(c6, call, (size_t)len);
    else if (len >= 2 && len <= 5 && is_digit(call[1])) memcpy(c6 + 1, call, (size_t)len);
    else return -1;
    const int i0 = idx_in(A_ALNUM_SP, c6[0]), i1 = idx_in(A_ALNUM, c6[1]), i2 = idx_in(A_DIGIT, c6[2]);
    const int i3 = idx_in(A_LETTER_SP, c6[3]), i4 = idx_in(A_LETTER_SP, c6[4]), i5 = idx_in(A_LETTER_SP, c6[5]);
    if (i0 < 0 || i1 < 0 || i2 < 0 || i3 < 0 || i4 < 0 || i5 < 0) return -1;
    int32_t n = i0;
    n = n * 36 + i1;
    n = n * 10 + i2;
    n = n * 27 + i3;
    n = n * 27 + i4;
    n = n * 27 + i5;
    return NTOKENS + MAX22 + n;
}

/* one call field of a type 1 / 2 message: special token, <hashed call>, or standard call with optional /R or /P.
 * *suffix: 0 none, 'R', 'P'.  -1: not packable in 28 bits. */
static int32_t pack_call_field(const char *tok, int len, int allow_token, char *suffix) {
    *suffix = 0;
    if (allow_token) {
        if (len == 2 && !strncmp(tok, "DE", 2)) return 0;
        if (len == 3 && !strncmp(tok, "QRZ", 3)) return 1;
        if (len == 2 && !strncmp(tok, "CQ", 2)) return 2;
    }
    if (len >= 3 && tok[0] == '<' && tok[len - 1] == '>') {
        uint32_t h;
        if (call_hash(tok + 1, len - 2, 22, &h) != 0) return -1;
        return NTOKENS + (int32_t)h;
    }
    if (len > 2 && tok[len - 2] == '/' && (tok[len - 1] == 'R' || tok[len - 1] == 'P')) {
        *suffix = tok[len - 1];
        len -= 2;
    }
    return pack_basecall(tok, len, 1);
}

/* "CQ nnn" / "CQ aaaa": the modifier token -> 28-bit code, or -1 */
static int32_t pack_cq_modifier(const char *tok, int len) {
    if (len == 3 && is_digit(tok[0]) && is_digit(tok[1]) && is_digit(tok[2]))
        return 3 + (tok[0] - '0') * 100 + (tok[1] - '0') * 10 + (tok[2] - '0');
    if (len < 1 || len > 4) return -1;
    int32_t m = 0;
    for (int i = 0; i < 4; ++i) {                          /* right-justified in four characters */
        const int k = i - (4 - len);
        if (k >= 0 && !is_letter(tok[k])) return -1;
        m = m * 27 + (k < 0 ? 0 : tok[k] - 'A' + 1);
    }
    return 1003 + m;
}

static int is_grid4(const char *t, int len) {
    return len == 4 && t[0] >= 'A' && t[0] <= 'R' && t[1] >= 'A' && t[1] <= 'R' && is_digit(t[2]) && is_digit(t[3]);
}

/* "+NN" / "-NN" -> value; 0 on success */
static int parse_report(const char *t, int len, int *value) {
    if (len != 3 || (t[0] != '+' && t[0] != '-') || !is_digit(t[1]) || !is_digit(t[2])) return -1;
    const int v = (t[1] - '0') * 10 + (t[2] - '0');
    *value = t[0] == '-' ? -v : v;
    return 0;
}

/* ---- tokens ---------------------------------------------------------------------------------- */
enum { kMaxTokens = 6 };
typedef struct { const char *p; int len; } token;

static int split_tokens(const char *msg, token *tok) {
    int n = 0;
    const char *s = msg;
    while (*s) {
        while (*s == ' ') ++s;
        if (!*s) break;
        if (n == kMaxTokens) return -1;
        tok[n].p = s;
        while (*s && *s != ' ') ++s;
        tok[n].len = (int)(s - tok[n].p);
        ++n;
    }
    return n;
}

static int tok_is(const token *t, const char *word) { return (int)strlen(word) == t->len && !strncmp(t->p, word, (size_t)t->len); }

/* ---- type 1 (i3 = 1, optional /R) and type 2 (i3 = 2, optional /P) ---------------------------- */
static int pack_type1(const token *tok, int n, uint8_t out[10]) {
    if (n < 2) return -1;
    int k = 0;
    int32_t na;
    char sa = 0, sb = 0;
    /* "CQ nnn CALL ..." / "CQ aaaa CALL ...": the modifier belongs to the first field when a call follows it */
    if (n >= 3 && tok_is(&tok[0], "CQ") && pack_cq_modifier(tok[1].p, tok[1].len) >= 0 &&
        pack_call_field(tok[2].p, tok[2].len, 0, &sb) >= 0) {
        na = pack_cq_modifier(tok[1].p, tok[1].len);
        k = 2;
    } else {
        na = pack_call_field(tok[0].p, tok[0].len, 1, &sa);
        k = 1;
    }
    const int32_t nb = pack_call_field(tok[k].p, tok[k].len, 0, &sb);
    ++k;
    if (na < 0 || nb < 0) return -1;
    if (sa && sb && sa != sb) return -1;                   /* /R and /P live in different message types */
    const int i3 = (sa == 'P' || sb == 'P') ? 2 : 1;
    uint32_t ir = 0, igrid4 = MAXGRID4 + 1;                /* no third field */
    const int rest = n - k;
    int rpt;
    if (rest == 0) {
    } else if (rest == 1 && tok_is(&tok[k], "RRR")) igrid4 = MAXGRID4 + 2;
    else if (rest == 1 && tok_is(&tok[k], "RR73")) igrid4 = MAXGRID4 + 3;
    else if (rest == 1 && tok_is(&tok[k], "73")) igrid4 = MAXGRID4 + 4;
    else if (rest == 1 && tok[k].len >= 4 && is_grid4(tok[k].p, 4)) {
        /* only the first four locator characters travel ("FN20QI" -> FN20, rtlsdr_ft8d.c:920-921) */
        const char *g = tok[k].p;
        for (int i = 4; i < tok[k].len; ++i) if (!is_letter(g[i]) && !is_digit(g[i])) return -1;
        if (tok[k].len > 6) return -1;
        igrid4 = (uint32_t)((((g[0] - 'A') * 18 + (g[1] - 'A')) * 10 + (g[2] - '0')) * 10 + (g[3] - '0'));
    } else if (rest == 2 && tok_is(&tok[k], "R") && is_grid4(tok[k + 1].p, tok[k + 1].len)) {
        const char *g = tok[k + 1].p;
        ir = 1;
        igrid4 = (uint32_t)((((g[0] - 'A') * 18 + (g[1] - 'A')) * 10 + (g[2] - '0')) * 10 + (g[3] - '0'));
    } else if (rest == 1 && parse_report(tok[k].p, tok[k].len, &rpt) == 0) {
        if (rpt < -30) return -1;                          /* 35 + rpt would collide with the RRR / RR73 / 73 codes */
        igrid4 = (uint32_t)(MAXGRID4 + 35 + rpt);
    } else if (rest == 1 && tok[k].len == 4 && tok[k].p[0] == 'R' && parse_report(tok[k].p + 1, 3, &rpt) == 0) {
        if (rpt < -30) return -1;
        ir = 1;
        igrid4 = (uint32_t)(MAXGRID4 + 35 + rpt);
    } else return -1;
    bitbuf w;
    memset(&w, 0, sizeof w);
    put_bits(&w, (uint64_t)na, 28);
    put_bits(&w, sa ? 1 : 0, 1);
    put_bits(&w, (uint64_t)nb, 28);
    put_bits(&w, sb ? 1 : 0, 1);
    put_bits(&w, ir, 1);
    put_bits(&w, igrid4, 15);
    put_bits(&w, (uint64_t)i3, 3);
    memcpy(out, w.b, 10);
    return 0;
}

/* ---- type 4 (i3 = 4): one call of up to 11 characters in full, the other as a 12-bit hash ------ */
static int pack_c11(const char *call, int len, uint64_t *out) {
    if (len < 3 || len > 11) return -1;
    uint64_t n = 0;
    for (int i = 0; i < 11; ++i) {                         /* right-justified */
        const int k = i - (11 - len);
        const int j = k < 0 ? 0 : idx_in(A_CALL11, call[k]);
        if (j < 0 || (k >= 0 && call[k] == ' ')) return -1;
        n = 38u * n + (uint64_t)j;
    }
    *out = n;
    return 0;
}

static int pack_type4(const token *tok, int n, uint8_t out[10]) {
    if (n < 2 || n > 3) return -1;
    uint32_t h12 = 0, iflip = 0, nrpt = 0, icq = 0;
    uint64_t n58 = 0;
    const int a_hashed = tok[0].len >= 3 && tok[0].p[0] == '<' && tok[0].p[tok[0].len - 1] == '>';
    const int b_hashed = tok[1].len >= 3 && tok[1].p[0] == '<' && tok[1].p[tok[1].len - 1] == '>';
    if (tok_is(&tok[0], "CQ")) {
        if (n != 2 || b_hashed || pack_c11(tok[1].p, tok[1].len, &n58) != 0) return -1;
        icq = 1;
        if (call_hash(tok[1].p, tok[1].len, 12, &h12) != 0) return -1;      /* the sender's own hash, ignored by receivers */
    } else if (a_hashed != b_hashed) {
        const token *hashed = a_hashed ? &tok[0] : &tok[1], *full = a_hashed ? &tok[1] : &tok[0];
        iflip = a_hashed ? 0 : 1;                          /* 0: "<hash> CALL", 1: "CALL <hash>" */
        if (call_hash(hashed->p + 1, hashed->len - 2, 12, &h12) != 0 || pack_c11(full->p, full->len, &n58) != 0) return -1;
        if (n == 3) {
            if (tok_is(&tok[2], "RRR")) nrpt = 1;
            else if (tok_is(&tok[2], "RR73")) nrpt = 2;
            else if (tok_is(&tok[2], "73")) nrpt = 3;
            else return -1;
        }
    } else return -1;
    bitbuf w;
    memset(&w, 0, sizeof w);
    put_bits(&w, h12, 12);
    put_bits(&w, n58, 58);
    put_bits(&w, iflip, 1);
    put_bits(&w, nrpt, 2);
    put_bits(&w, icq, 1);
    put_bits(&w, 4, 3);
    memcpy(out, w.b, 10);
    return 0;
}

/* ---- type 0.5: telemetry, 18 hexadecimal digits = 71 bits (the first digit is 0..7) ------------ */
static int pack_telemetry(const char *t, int len, uint8_t out[10]) {
    if (len != 18) return -1;
    bitbuf w;
    memset(&w, 0, sizeof w);
    for (int i = 0; i < 18; ++i) {
        const int v = is_digit(t[i]) ? t[i] - '0' : (t[i] >= 'A' && t[i] <= 'F') ? t[i] - 'A' + 10 : -1;
        if (v < 0 || (i == 0 && v > 7)) return -1;
        put_bits(&w, (uint64_t)v, i == 0 ? 3 : 4);
    }
    put_bits(&w, 5, 3);                                    /* n3 = 5 */
    put_bits(&w, 0, 3);                                    /* i3 = 0 */
    memcpy(out, w.b, 10);
    return 0;
}

/* ---- type 0.0: free text, up to 13 characters of the 42-character alphabet, right-justified ----- */
static int pack_free_text(const char *text, int len, uint8_t out[10]) {
    if (len < 1 || len > 13) return -1;
    uint8_t num[9];                                        /* 71-bit number, big-endian in 72 bits */
    memset(num, 0, sizeof num);
    for (int i = 0; i < 13; ++i) {
        const int k = i - (13 - len);
        const int j = k < 0 ? 0 : idx_in(A_TEXT, text[k]);
        if (j < 0) return -1;
        unsigned carry = (unsigned)j;
        for (int b = 8; b >= 0; --b) {                     /* num = num * 42 + j */
            const unsigned v = (unsigned)num[b] * 42u + carry;
            num[b] = (uint8_t)v;
            carry = v >> 8;
        }
    }
    bitbuf w;
    memset(&w, 0, sizeof w);
    put_bits(&w, num[0] & 0x7Fu, 7);
    for (int b = 1; b < 9; ++b) put_bits(&w, num[b], 8);
    put_bits(&w, 0, 3);                                    /* n3 = 0 */
    put_bits(&w, 0, 3);                                    /* i3 = 0 */
    memcpy(out, w.b, 10);
    return 0;
}

/* Text -> 77-bit payload.  Tried in this order: telemetry (one token of 18 hex digits), type 1 / 2, type 4, free text. */
int ft8gpu_pack77(const char *msg, uint8_t payload[10]) {
    if (!msg || !payload) return -1;
    token tok[kMaxTokens];
    const size_t total = strlen(msg);
    if (total > 40) return -1;
    const int n = split_tokens(msg, tok);
    if (n < 1) return -1;
    if (n == 1 && pack_telemetry(tok[0].p, tok[0].len, payload) == 0) return 0;
    if (pack_type1(tok, n, payload) == 0) return 0;
    if (pack_type4(tok, n, payload) == 0) return 0;
    /* free text: the message without its outer blanks, inner blanks kept */
    const char *first = tok[0].p, *last = tok[n - 1].p + tok[n - 1].len;
    return pack_free_text(first, (int)(last - first), payload);
}

/* The strict subset of earlier rounds, kept for its callers: "CALL1 CALL2 [GRID4]" with plain standard calls
 * (or CQ / DE / QRZ first), nothing else. */
int ft8gpu_pack77_std(const char *msg, uint8_t payload[10]) {
    if (!msg || !payload) return -1;
    const char *s1 = strchr(msg, ' ');
    if (!s1) return -1;
    const char *c2 = s1 + 1;
    const char *s2 = strchr(c2, ' ');
    const int len1 = (int)(s1 - msg);
    const int len2 = s2 ? (int)(s2 - c2) : (int)strlen(c2);
    int32_t na;
    if (len1 == 2 && !strncmp(msg, "DE", 2)) na = 0;
    else if (len1 == 3 && !strncmp(msg, "QRZ", 3)) na = 1;
    else if (len1 == 2 && !strncmp(msg, "CQ", 2)) na = 2;
    else if (len1 <= 6) na = pack_basecall(msg, len1, 0);
    else na = -1;
    const int32_t nb = len2 <= 6 ? pack_basecall(c2, len2, 0) : -1;
    if (na < 0 || nb < 0) return -1;
    uint32_t igrid4 = MAXGRID4 + 1;                       /* no grid */
    if (s2) {
        const char *g = s2 + 1;                           /* only the first four locator characters count */
        if (strlen(g) < 4 || !is_grid4(g, 4)) return -1;
        igrid4 = (uint32_t)((((g[0] - 'A') * 18 + (g[1] - 'A')) * 10 + (g[2] - '0')) * 10 + (g[3] - '0'));
    }
    bitbuf w;
    memset(&w, 0, sizeof w);
    put_bits(&w, (uint64_t)na, 28);
    put_bits(&w, 0, 1);
    put_bits(&w, (uint64_t)nb, 28);
    put_bits(&w, 0, 1);
    put_bits(&w, 0, 1);
    put_bits(&w, igrid4, 15);
    put_bits(&w, 1, 3);
    memcpy(payload, w.b, 10);
    return 0;
}

/* ---- CRC-14 + LDPC(174,91) generator + tone mapping: ft8_encode, rtlsdr_ft8d.c:934 -------------- */
static uint16_t crc14(const uint8_t *msg, int nbits) {
    uint32_t rem = 0;
    for (int bit = 0, byte = 0; bit < nbits; ++bit) {
        if ((bit & 7) == 0) rem ^= (uint32_t)msg[byte++] << 6;
        rem = (rem & 0x2000u) ? (((rem << 1) ^ 0x2757u) & 0xFFFFu) : ((rem << 1) & 0xFFFFu);
    }
    return (uint16_t)(rem & 0x3FFFu);
}

void ft8gpu_encode(const uint8_t payload[10], uint8_t tones[FT8GPU_NN]) {
    uint8_t a91[12];
    memcpy(a91, payload, 10);
    a91[9] &= 0xF8u;
    a91[10] = a91[11] = 0;
    const uint16_t crc = crc14(a91, 82);
    a91[9] |= (uint8_t)(crc >> 11);
    a91[10] = (uint8_t)(crc >> 3);
    a91[11] = (uint8_t)(crc << 5);
    uint8_t bits[174];
    for (int i = 0; i < 91; ++i) bits[i] = (a91[i >> 3] >> (7 - (i & 7))) & 1;
    for (int m = 0; m < 83; ++m) {
        unsigned acc = 0;
        for (int j = 0; j < 12; ++j) acc ^= (unsigned)(a91[j] & kFT8_generator[m][j]);
        acc ^= acc >> 4; acc ^= acc >> 2; acc ^= acc >> 1;
        bits[91 + m] = acc & 1;
    }
    int k = 0;
    for (int t = 0; t < FT8GPU_NN; ++t) {
        if (t < 7) tones[t] = kFT8_Costas[t];
        else if (t >= 36 && t < 43) tones[t] = kFT8_Costas[t - 36];
        else if (t >= 72) tones[t] = kFT8_Costas[t - 72];
        else { tones[t] = kFT8_Gray[(bits[k] << 2) | (bits[k + 1] << 1) | bits[k + 2]]; k += 3; }
    }
}

/* ---- ft8_lib-level names (include/ft8_lib/ft8/pack.h, encode.h) --------------------------------
 * ft8_lib's pack77 of the reference's era packs "CALL1 CALL2 [GRID4 | +NN | -NN | R+NN | R-NN | RRR | RR73 | 73]" with
 * standard calls and falls back to free text for everything else, returning 0 always (text that fits neither is
 * mangled).  This one packs the same inputs to the same meaning and, beyond them, /R /P, "R GRID4", CQ modifiers,
 * hashed calls, type 4 and telemetry; text that fits no type returns -1 instead of being mangled. */
int pack77(const char *msg, uint8_t *c77) {
    uint8_t p[10];
    if (ft8gpu_pack77(msg, p) != 0) return -1;
    memcpy(c77, p, 10);                                    /* the 77-bit payload: 10 bytes, as upstream's pack77 fills (a caller may
                                                              pass uint8_t[10]); ft8_encode reads no further */
    return 0;
}

void ft8_encode(const uint8_t *payload, uint8_t *tones) { ft8gpu_encode(payload, tones); }
