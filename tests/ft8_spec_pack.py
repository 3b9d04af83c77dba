"""An encoder for 77-bit FT8 payloads written from the published protocol description (Franke, Somerville,
Taylor: "The FT4 and FT8 Communication Protocols", QEX 2020, section "Source encoding"; field widths of its
table of message types) -- independently of the oracle's and the kernel's unpack77, which are two writings
of one recollection of ft8_lib's unpack.c.  Used by tests only: unpack(pack(text)) == text exercises the
branches the reference's own vectors never reach (free text, telemetry, non-standard calls, reports, /R /P,
CQ with a number or letters).  Returns 10 bytes, 77 bits MSB first."""

NTOKENS, MAX22, MAXGRID4 = 2063592, 4194304, 32400
A_ALNUM_SP = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"      # 37
A_ALNUM = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"          # 36
A_DIGIT = "0123456789"
A_LETTER_SP = " ABCDEFGHIJKLMNOPQRSTUVWXYZ"               # 27
A_TEXT = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ+-./?"     # 42
A_CALL11 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/"       # 38


def _bits(fields):
    """fields: [(value, width)] MSB first -> 10 bytes"""
    v = 0
    n = 0
    for val, w in fields:
        assert 0 <= val < (1 << w), (val, w)
        v = (v << w) | val
        n += w
    assert n == 77
    return (v << 3).to_bytes(10, "big")


def call_hash(call, bits):
    """the m-bit hash of a call sign: the call left-justified in 11 characters as a base-38 number, multiplied by
    47055833459 modulo 2^64, top m bits (m = 22 in type 1 / 2 messages, 12 in type 4)"""
    n = 0
    for c in call.ljust(11):
        n = n * 38 + A_CALL11.index(c)
    return ((47055833459 * n) & 0xFFFFFFFFFFFFFFFF) >> (64 - bits)


def pack28(token):
    if token.startswith("<") and token.endswith(">"):
        return NTOKENS + call_hash(token[1:-1], 22)
    if token == "DE":
        return 0
    if token == "QRZ":
        return 1
    if token == "CQ":
        return 2
    if token.startswith("CQ "):
        t = token[3:]
        if t.isdigit() and len(t) == 3:
            return 3 + int(t)
        assert 1 <= len(t) <= 4 and t.isalpha()
        m = 0
        for c in t.rjust(4):
            m = m * 27 + A_LETTER_SP.index(c)
        return 1003 + m
    call = token
    if call.startswith("3DA0"):
        call = "3D0" + call[4:]
    elif call.startswith("3X") and call[2].isalpha():
        call = "Q" + call[2:]
    if len(call) >= 3 and call[2].isdigit():
        c6 = call.ljust(6)
    else:
        assert call[1].isdigit(), call
        c6 = (" " + call).ljust(6)
    assert len(c6) == 6
    n = A_ALNUM_SP.index(c6[0])
    n = n * 36 + A_ALNUM.index(c6[1])
    n = n * 10 + A_DIGIT.index(c6[2])
    for c in c6[3:]:
        n = n * 27 + A_LETTER_SP.index(c)
    return NTOKENS + MAX22 + n


def pack_standard(to, de, extra="", i3=1):
    """'<to> <de> <grid4 | report | RRR | RR73 | 73 | ''>' ; a call may carry /R (i3 = 1) or /P (i3 = 2)"""
    suffix = "/R" if i3 == 1 else "/P"
    ipa = ipb = 0
    if to.endswith(suffix):
        to, ipa = to[:-2], 1
    if de.endswith(suffix):
        de, ipb = de[:-2], 1
    ir = 0
    if extra == "":
        ig = MAXGRID4 + 1
    elif extra == "RRR":
        ig = MAXGRID4 + 2
    elif extra == "RR73":
        ig = MAXGRID4 + 3
    elif extra == "73":
        ig = MAXGRID4 + 4
    elif len(extra) >= 4 and extra[-4].isalpha() and extra[-3].isalpha() and extra[-2:].isdigit():
        if extra.startswith("R "):
            ir, extra = 1, extra[2:]
        g = extra
        ig = ((ord(g[0]) - 65) * 18 + (ord(g[1]) - 65)) * 100 + int(g[2:4])
    else:                                              # signal report, optionally preceded by R
        if extra.startswith("R"):
            ir, extra = 1, extra[1:]
        ig = MAXGRID4 + 35 + int(extra)
    return _bits([(pack28(to), 28), (ipa, 1), (pack28(de), 28), (ipb, 1), (ir, 1), (ig, 15), (i3, 3)])


def pack_free_text(text):
    assert len(text) <= 13
    v = 0
    for c in text.rjust(13):
        v = v * 42 + A_TEXT.index(c)
    return _bits([(v, 71), (0, 3), (0, 3)])


def pack_telemetry(hex18):
    assert len(hex18) == 18
    v = int(hex18, 16)
    return _bits([(v, 71), (5, 3), (0, 3)])


def pack_nonstandard(call, hash12=0, flip=0, nrpt=0, icq=0):
    assert len(call) <= 11
    v = 0
    for c in call.rjust(11):
        v = v * 38 + A_CALL11.index(c)
    return _bits([(hash12, 12), (v, 58), (flip, 1), (nrpt, 2), (icq, 1), (4, 3)])
