"""The product's encoder (csrc/ft8_pack.c: ft8gpu_pack77, pack77, ft8gpu_encode) against
  * the reference's own known answer (rtlsdr_ft8d.c:919-923),
  * tests/ft8_spec_pack.py, a second writing of the published field layout, bit for bit,
  * the oracle's unpack77 (ft8_lib unpack.c as SURVEY Appendix A.6 restates it): unpack(pack(text)) == text.
Host C only: no GPU."""
import ctypes as C

import numpy as np
import pytest

import ft8_spec_pack as P
import rtlsdr_ft8d_amd as ft8

LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"


def _call(rng):
    pfx = str(rng.choice(["K", "W", "G", "F", "DL", "JA", "VK", "9A", "A6", "3D", "EA", "OH"]))
    return pfx + str(rng.integers(0, 10)) + "".join(rng.choice(list(LETTERS), size=int(rng.integers(1, 4))))


def _grid(rng):
    return LETTERS[rng.integers(0, 18)] + LETTERS[rng.integers(0, 18)] + f"{rng.integers(0, 100):02d}"


def _roundtrip(oracle, text, expect=None):
    p = ft8.pack77(text).tobytes()
    rc, got = oracle.unpack77(p)
    want = text if expect is None else expect
    # (unpack77 leaves a trailing blank when the third field is empty)
    assert rc >= 0 and got.rstrip(" ") == want and len(got) - len(want) <= 1, (text, p.hex(), got)
    return p


def test_reference_known_answer():
    assert ft8.pack77("CQ K1JT FN20QI").tobytes() == bytes.fromhex("000000204dfcdc8a1408")          # rtlsdr_ft8d.c:921
    assert ft8.pack77("CQ K1JT FN20").tobytes() == ft8.pack77_std("CQ K1JT FN20").tobytes()


def test_type1_forms_bit_exact_and_round_trip(oracle):
    rng = np.random.default_rng(5)
    for _ in range(200):
        a, b, g = _call(rng), _call(rng), _grid(rng)
        rpt = int(rng.integers(-30, 50))
        nnn, word = int(rng.integers(0, 1000)), "".join(rng.choice(list(LETTERS), size=int(rng.integers(1, 5))))
        cases = [(f"{a} {b} {g}", P.pack_standard(a, b, g)), (f"{a} {b} R {g}", P.pack_standard(a, b, "R " + g)),
                 (f"{a} {b} {rpt:+03d}", P.pack_standard(a, b, f"{rpt:+03d}")), (f"{a} {b} R{rpt:+03d}", P.pack_standard(a, b, f"R{rpt:+03d}")),
                 (f"{a} {b} RRR", P.pack_standard(a, b, "RRR")), (f"{a} {b} RR73", P.pack_standard(a, b, "RR73")),
                 (f"{a} {b} 73", P.pack_standard(a, b, "73")), (f"{a} {b}", P.pack_standard(a, b, "")),
                 (f"{a}/R {b} {g}", P.pack_standard(a + "/R", b, g, i3=1)), (f"{a} {b}/R R {g}", P.pack_standard(a, b + "/R", "R " + g, i3=1)),
                 (f"{a} {b}/P {g}", P.pack_standard(a, b + "/P", g, i3=2)), (f"{a}/P {b}/P 73", P.pack_standard(a + "/P", b + "/P", "73", i3=2)),
                 (f"CQ {b} {g}", P.pack_standard("CQ", b, g)), (f"QRZ {b} {g}", P.pack_standard("QRZ", b, g)), (f"DE {b} {g}", P.pack_standard("DE", b, g)),
                 (f"CQ {nnn:03d} {b} {g}", P.pack_standard(f"CQ {nnn:03d}", b, g)), (f"CQ {word} {b} {g}", P.pack_standard("CQ " + word, b, g)),
                 (f"CQ {b}", P.pack_standard("CQ", b, ""))]
        for text, want in cases:
            assert _roundtrip(oracle, text) == want, text
    # a grid with subsquare letters travels as its first four characters (the reference's own self-test message)
    assert ft8.pack77("K1ABC W9XYZ EN37AB").tobytes() == P.pack_standard("K1ABC", "W9XYZ", "EN37")
    # work-arounds of the 28-bit code: 3DA0 and 3X prefixes
    assert ft8.pack77("CQ 3DA0XY KG53").tobytes() == P.pack_standard("CQ", "3DA0XY", "KG53")
    assert ft8.pack77("CQ 3XA0XY IJ39").tobytes() == P.pack_standard("CQ", "3XA0XY", "IJ39")


def test_hashed_calls_print_as_unknown(oracle):
    """<CALL> in a type 1 message is a 22-bit hash; the reference's ft8_lib era has no hash table: "<...>" """
    for text, want, spec in (("<PJ4/K1ABC> W9XYZ -11", "<...> W9XYZ -11", P.pack_standard("<PJ4/K1ABC>", "W9XYZ", "-11")),
                             ("K1ABC <YW18FIFA> RR73", "K1ABC <...> RR73", P.pack_standard("K1ABC", "<YW18FIFA>", "RR73")),
                             ("<W9XYZ> K1ABC EN37", "<...> K1ABC EN37", P.pack_standard("<W9XYZ>", "K1ABC", "EN37"))):
        assert _roundtrip(oracle, text, want) == spec
    # published example of the hash: the 22- and 12-bit hashes are prefixes of one product
    assert P.call_hash("K1ABC", 22) >> 10 == P.call_hash("K1ABC", 12)


def test_type4_free_text_telemetry(oracle):
    assert _roundtrip(oracle, "CQ PJ4/K1ABC") == P.pack_nonstandard("PJ4/K1ABC", P.call_hash("PJ4/K1ABC", 12), icq=1)
    assert _roundtrip(oracle, "<W9XYZ> PJ4/K1ABC RRR", "<...> PJ4/K1ABC RRR") == P.pack_nonstandard("PJ4/K1ABC", P.call_hash("W9XYZ", 12), flip=0, nrpt=1)
    assert _roundtrip(oracle, "KH1/KH7Z <K1ABC> 73", "KH1/KH7Z <...> 73") == P.pack_nonstandard("KH1/KH7Z", P.call_hash("K1ABC", 12), flip=1, nrpt=3)
    assert _roundtrip(oracle, "YW18FIFA <K1ABC>", "YW18FIFA <...>") == P.pack_nonstandard("YW18FIFA", P.call_hash("K1ABC", 12), flip=1, nrpt=0)
    for text in ("TNX BOB 73 GL", "A", "+-./?0123 ZY", "HELLO WORLD", "CQ73 GL", "CQ", "73", "K1ABC", "RR73 CUL"):
        assert _roundtrip(oracle, text) == P.pack_free_text(text), text
    p = ft8.pack77("0123456789ABCDEF01").tobytes()
    assert p == P.pack_telemetry("0123456789ABCDEF01") and oracle.unpack77(p) == (0, "0123456789ABCDEF01")
    assert ft8.pack77("7FFFFFFFFFFFFFFFFF").tobytes() == P.pack_telemetry("7FFFFFFFFFFFFFFFFF")
    rng = np.random.default_rng(9)
    for _ in range(300):
        n = int(rng.integers(1, 14))
        text = " ".join("".join(rng.choice(list(P.A_TEXT), size=n)).split())
        if not text:
            continue
        p = ft8.pack77(text).tobytes()
        rc, got = oracle.unpack77(p)
        assert rc >= 0 and got.strip() == text.strip(), (text, got)       # whatever type the text fell into, it reads back


def test_unpackable_texts_are_refused():
    for bad in ("", " ", "   ", "THIS TEXT IS FAR TOO LONG FOR ANY TYPE", "lowercase", "K1ABC W9XYZ -31", "K1ABC W9XYZ ZZ99 EXTRA",
                "8FFFFFFFFFFFFFFFFF", "K1ABC/R W9XYZ/P EN37", "<K1ABC> <W9XYZ> RRR!", "A B C D E F G", "CQ \xff\xfe"):
        with pytest.raises(ValueError):
            ft8.pack77(bad)
    L = ft8.load_library()
    out = (C.c_uint8 * 10)()
    assert L.ft8gpu_pack77(None, out) != 0 and L.ft8gpu_pack77(b"CQ K1JT FN20", None) != 0


def test_ft8_lib_level_pack77_is_the_same_packer():
    L = ft8.load_library()
    L.pack77.argtypes = [C.c_char_p, C.c_void_p]
    for text in ("CQ K1JT FN20QI", "K1ABC W9XYZ R-09", "TNX BOB 73 GL", "CQ PJ4/K1ABC"):
        a = np.zeros(12, np.uint8)
        assert L.pack77(text.encode(), a.ctypes.data) == 0 and a[:10].tobytes() == ft8.pack77(text).tobytes() and not a[10:].any()
    a = np.zeros(12, np.uint8)
    assert L.pack77(b"lowercase", a.ctypes.data) == -1
