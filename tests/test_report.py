"""Spot reporting wire formats: the PSKreporter datagram of postSpots()
(rtlsdr_ft8d.c:365-590) and the stdout table of printSpots() (:643-663).

CPU: the oracle against an independent struct.pack construction of the datagram and the committed
golden bytes; the product's host-side table formatter against the oracle.
GPU: the batched datagram kernel against the oracle, byte for byte, on random spot lists that cover
empty frames, stale (non-CQ) slots, unterminated fields, the 1200-byte cut and per-frame times."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTERPRISE = 30351


def _info(cls, rcall=b"N0CALL", rloc=b"FN20", app=b"rtlsdr-ft8d_v0.3.6", dial=14074000, now=1700000000, seq=1, rid=0x12345678):
    return cls(rcall=rcall, rloc=rloc, app_version=app, dial_freq=dial, unixtime=now, sequence=seq, random_id=rid)


def _vstr(b):
    return bytes([len(b)]) + b


def _pad4(b):
    return b + b"\0" * (-len(b) % 4)


def _expected(spots, rcall, rloc, app, dial, now, seq, rid):
    """the datagram from the protocol description: IPFIX header, two template sets, two data sets"""
    def tmpl(fields):
        return b"".join(struct.pack(">HH", i, l) + (struct.pack(">I", ENTERPRISE) if i & 0x8000 else b"") for i, l in fields)
    rx_t = struct.pack(">HHHHH", 3, 36, 0x9992, 3, 0) + tmpl([(0x8002, 0xFFFF), (0x8004, 0xFFFF), (0x8008, 0xFFFF)]) + b"\0\0"
    tx_t = struct.pack(">HHHH", 2, 60, 0x9993, 7) + tmpl([(0x8001, 0xFFFF), (0x8005, 4), (0x8006, 1), (0x800A, 0xFFFF),
                                                       (0x8003, 0xFFFF), (0x800B, 1), (0x0096, 4)])
    assert len(rx_t) == 36 and len(tx_t) == 60
    rx = _pad4(b"\x99\x92\0\0" + _vstr(rcall) + _vstr(rloc) + _vstr(app))
    rx = rx[:2] + struct.pack(">H", len(rx)) + rx[4:]
    tx = b"\x99\x93\0\0"
    for call, loc, freq, snr in spots:
        if len(tx) > 1200:
            break
        tx += _vstr(call) + struct.pack(">I", (freq + dial) & 0xFFFFFFFF) + struct.pack("B", (((snr + 128) % 256 - 128) - 20) % 256)
        tx += _vstr(b"FT8") + _vstr(loc) + b"\x01" + struct.pack(">I", now)
    tx = _pad4(tx)
    tx = tx[:2] + struct.pack(">H", len(tx)) + tx[4:]
    body = rx_t + tx_t + rx + tx
    return struct.pack(">HHIII", 10, 16 + len(body), now, seq, rid) + body


def _records(oracle, spots):
    d = np.zeros(50, oracle.RESULT_DTYPE)
    for k, (call, loc, freq, snr) in enumerate(spots):
        d[k] = (call, loc, freq, snr)
    return d


def test_oracle_datagram_matches_protocol_construction(oracle):
    spots = [(b"K1JT", b"FN20", 28, 34), (b"", b"", 0, 0), (b"VE2ABC/P", b"FN35", 1503, 12), (b"PA0XYZ", b"JO22", -5, 200)]
    info = _info(oracle.ReportInfo)
    got = oracle.pskreporter_datagram(_records(oracle, spots), len(spots), info).tobytes()
    assert got == _expected(spots, b"N0CALL", b"FN20", b"rtlsdr-ft8d_v0.3.6", 14074000, 1700000000, 1, 0x12345678)
    assert len(got) % 4 == 0 and struct.unpack(">H", got[2:4])[0] == len(got)
    # no spots: header + templates + receiver record + an empty, padded sender set
    empty = oracle.pskreporter_datagram(_records(oracle, []), 0, info).tobytes()
    assert empty == _expected([], b"N0CALL", b"FN20", b"rtlsdr-ft8d_v0.3.6", 14074000, 1700000000, 1, 0x12345678)
    assert empty.endswith(b"\x99\x93\x00\x04")


def test_oracle_datagram_cut_at_1200_bytes(oracle):
    """:497 -- a record is started only while the sender set is <= 1200 bytes long"""
    spots = [(b"AB1CDE/PX%03d" % k, b"AA00aa", k, 15) for k in range(50)]          # 34 bytes each
    info = _info(oracle.ReportInfo, rcall=b"", rloc=b"", app=b"x")
    got = oracle.pskreporter_datagram(_records(oracle, spots), 50, info).tobytes()
    assert got == _expected(spots, b"", b"", b"x", 14074000, 1700000000, 1, 0x12345678)
    tx = got[16 + 36 + 60 + 8:]
    assert struct.unpack(">H", tx[2:4])[0] == len(tx) == 4 + 36 * 34        # 4 + 35*34 = 1194 <= 1200 -> a 36th record


def test_oracle_datagram_golden(oracle):
    with open(os.path.join(ROOT, "tests", "golden", "report.json")) as f:
        g = json.load(f)
    for case in g["datagrams"]:
        spots = [(s["call"].encode(), s["loc"].encode(), s["freq"], s["snr"]) for s in case["spots"]]
        i = case["info"]
        info = _info(oracle.ReportInfo, i["rcall"].encode(), i["rloc"].encode(), i["app_version"].encode(),
                     i["dial_freq"], i["unixtime"], i["sequence"], i["random_id"])
        got = oracle.pskreporter_datagram(_records(oracle, spots), case["n_results"], info).tobytes()
        assert got.hex() == case["hex"]
    for case in g["tables"]:
        spots = [(s["call"].encode(), s["loc"].encode(), s["freq"], s["snr"]) for s in case["spots"]]
        assert oracle.format_spots(_records(oracle, spots), case["n_results"], case["dial_freq"], *case["when"]) == case["text"]


def test_format_spots_product_vs_oracle(oracle):
    """host-side text formatting of the library (no GPU involved) == oracle == printf semantics"""
    import rtlsdr_ft8d_amd as ft8
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(0, 51))
        spots = []
        for _ in range(n):
            call = bytes(rng.integers(48, 91, int(rng.integers(0, 14))).astype(np.uint8))       # up to 13: unterminated field
            loc = bytes(rng.integers(48, 91, int(rng.integers(0, 8))).astype(np.uint8))
            spots.append((call, loc, int(rng.integers(-100, 3200)), int(rng.integers(-40, 300))))
        d = _records(oracle, spots)
        when = (2025, 2, 12, 7, 5)
        want = oracle.format_spots(d, n, 14074000, *when)
        assert ft8.format_spots(d.view(ft8.RESULT_DTYPE), n, 14074000, *when) == want
        if n == 0:
            assert want == "No spot 2025-02-12 07:05z\n"
        else:
            lines = want.split("\n")
            assert lines[0] == "  Score     Freq       Call    Loc" and len(lines) == n + 2
            c, l, f, s = spots[0]
            assert lines[1] == "     %2d %8d %10s %6s" % (s, f + 14074000, c[:12].decode(), l[:6].decode())


def test_report_info_layout():
    import rtlsdr_ft8d_amd as ft8
    import oracle_lib
    for cls in (ft8.ReportInfo, oracle_lib.ReportInfo):
        assert C.sizeof(cls) == 68 and cls.dial_freq.offset == 52 and cls.random_id.offset == 64


# ---------------------------------------------------------------------------------------------------
def _random_lists(oracle, rng, nframes):
    d = np.zeros((nframes, 50), oracle.RESULT_DTYPE)
    raw = d.view(np.uint8).reshape(nframes, 50, 28)
    n = rng.integers(0, 51, nframes).astype(np.int32)
    n[0], n[1], n[2], n[3] = 0, 50, 50, 1
    for f in range(nframes):
        kind = f % 4
        for k in range(50):
            if kind == 2:                                   # stale caller bytes everywhere (no terminators in the fields)
                raw[f, k, :20] = rng.integers(1, 256, 20)
            else:
                lc = 12 if kind == 1 else int(rng.integers(0, 13))
                ll = 6 if kind == 1 else int(rng.integers(0, 7))
                raw[f, k, :lc] = rng.integers(48, 91, lc)
                raw[f, k, 13:13 + ll] = rng.integers(48, 91, ll)
            d[f, k]["freq"] = int(rng.integers(-2**31, 2**31))
            d[f, k]["snr"] = int(rng.integers(-300, 300))
    return d, n


@pytest.mark.gpu
def test_gpu_datagrams_match_oracle(oracle):
    import rtlsdr_ft8d_amd as ft8
    import torch
    rng = np.random.default_rng(2025)
    F = 203
    d, n = _random_lists(oracle, rng, F)
    n[5], n[6] = -3, 77                                     # fenced: treated as 0 and 50
    times = rng.integers(0, 2**32, F, dtype=np.uint64).astype(np.uint32)
    info = _info(ft8.ReportInfo, rcall=b"VE2XYZ/QRP12", rloc=b"FN35ab", app=b"rtlsdr-ft8d_v0.3.6", dial=4294960000)
    with ft8.Decoder(device=0, max_frames=64) as dec:       # 203 frames in chunks of 64
        out, lens = dec.pskreporter_datagrams(d.view(ft8.RESULT_DTYPE), n, info, times)
        out1, lens1 = dec.pskreporter_datagrams(d.view(ft8.RESULT_DTYPE), n, info)          # one time for all
        # device-pointer form
        dd = torch.from_numpy(d.view(np.uint8).reshape(F, -1)).cuda()
        dn = torch.from_numpy(n).cuda()
        dt = torch.from_numpy(times.view(np.int32)).cuda()
        dout = torch.full((F, ft8.DATAGRAM_STRIDE), 0xEE, dtype=torch.uint8, device="cuda")
        dlen = torch.zeros(F, dtype=torch.int32, device="cuda")
        dec.pskreporter_datagrams_dev(dd, dn, F, info, dt, dout, dlen)
        dec.synchronize()
    assert np.array_equal(dout.cpu().numpy(), out) and np.array_equal(dlen.cpu().numpy(), lens)
    longest = 0
    for f in range(F):
        oi = _info(oracle.ReportInfo, rcall=b"VE2XYZ/QRP12", rloc=b"FN35ab", app=b"rtlsdr-ft8d_v0.3.6", dial=4294960000, now=int(times[f]))
        want = oracle.pskreporter_datagram(d[f], max(0, int(n[f])), oi)
        assert lens[f] == want.size, f
        assert out[f, :lens[f]].tobytes() == want.tobytes(), f
        assert not out[f, lens[f]:].any()
        longest = max(longest, want.size)
        oi.unixtime = 1700000000
        want1 = oracle.pskreporter_datagram(d[f], max(0, int(n[f])), oi)
        assert lens1[f] == want1.size and out1[f, :lens1[f]].tobytes() == want1.tobytes()
    assert longest == 16 + 36 + 60 + 44 + 1228              # frame 1: 36 records of 34 bytes
    assert longest <= ft8.DATAGRAM_STRIDE


@pytest.mark.gpu
def test_gpu_decode_to_datagram_end_to_end(oracle):
    """self-test frame -> spots -> datagram, all on the GPU; the datagram carries K1JT / FN20"""
    import rtlsdr_ft8d_amd as ft8
    i, q = oracle.selftest_signal()
    iq = np.stack([i, q])[None]
    with ft8.Decoder(device=0, max_frames=1) as dec:
        dec_out, nres = dec.decode_batch(iq)
        info = _info(ft8.ReportInfo)
        out, lens = dec.pskreporter_datagrams(dec_out, nres, info)
    assert nres[0] == 1
    want = _expected([(b"K1JT", b"FN20", int(dec_out[0, 0]["freq"]), int(dec_out[0, 0]["snr"]))],
                     b"N0CALL", b"FN20", b"rtlsdr-ft8d_v0.3.6", 14074000, 1700000000, 1, 0x12345678)
    assert out[0, :lens[0]].tobytes() == want


@pytest.mark.gpu
def test_gpu_report_api_errors():
    import rtlsdr_ft8d_amd as ft8
    d = np.zeros((1, 50), ft8.RESULT_DTYPE)
    n = np.zeros(1, np.int32)
    with ft8.Decoder(device=0, max_frames=1) as dec:
        bad = ft8.ReportInfo(rcall=b"A" * 13)               # not NUL-terminated
        with pytest.raises(ft8.Ft8GpuError, match="NUL"):
            dec.pskreporter_datagrams(d, n, bad)
