"""ctypes binding of the CPU oracle (oracle/libft8oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NSAMPLES = 48000
MAG_ARRAY = 94208
MAX_MESSAGES = 50


class DecoderResults(C.Structure):
    _fields_ = [("call", C.c_char * 13), ("loc", C.c_char * 7), ("freq", C.c_int32), ("snr", C.c_int32)]


class Candidate(C.Structure):
    _fields_ = [("score", C.c_int16), ("time_offset", C.c_int16), ("freq_offset", C.c_int16),
                ("time_sub", C.c_uint8), ("freq_sub", C.c_uint8)]


class Message(C.Structure):
    _fields_ = [("text", C.c_char * 25), ("hash", C.c_uint16)]


class DecodeStatus(C.Structure):
    _fields_ = [("ldpc_errors", C.c_int), ("crc_extracted", C.c_uint16), ("crc_calculated", C.c_uint16),
                ("unpack_status", C.c_int)]


class DecodeExtra(C.Structure):
    _fields_ = [("iters", C.c_int), ("a91", C.c_uint8 * 12)]


class Params(C.Structure):
    _fields_ = [("min_score", C.c_int), ("max_candidates", C.c_int), ("ldpc_iters", C.c_int)]


RESULT_DTYPE = np.dtype([("call", "S13"), ("loc", "S7"), ("freq", "<i4"), ("snr", "<i4")], align=True)
CAND_DTYPE = np.dtype([("score", "<i2"), ("time_offset", "<i2"), ("freq_offset", "<i2"),
                       ("time_sub", "u1"), ("freq_sub", "u1")])
assert RESULT_DTYPE.itemsize == 28 and CAND_DTYPE.itemsize == 8

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("FT8O_LIB") or os.path.join(ORACLE_DIR, "libft8oracle.so")      # FT8O_LIB: the sanitizer build
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    u8p = C.POINTER(C.c_uint8)
    L.ft8o_init.restype = None
    L.ft8o_window.restype = fp
    L.ft8o_twiddles.restype = fp
    L.ft8o_fft1024.argtypes = [fp, fp]
    L.ft8o_quantise.argtypes = [C.c_float]
    L.ft8o_quantise.restype = C.c_uint8
    L.ft8o_quantise_x86.argtypes = [C.c_float]
    L.ft8o_quantise_x86.restype = C.c_uint8
    L.ft8o_set_quantiser_x86.argtypes = [C.c_int]
    L.ft8o_set_quantiser_x86.restype = None
    L.ft8o_waterfall.argtypes = [fp, fp, u8p]
    L.ft8o_waterfall_f64.argtypes = [fp, fp, u8p]
    L.ft8o_find_sync.argtypes = [u8p, C.c_int, C.c_void_p, C.c_int]
    L.ft8o_find_sync.restype = C.c_int
    L.ft8o_score_map.argtypes = [u8p, C.c_void_p]
    L.ft8o_extract_likelihood.argtypes = [u8p, C.c_void_p, fp]
    L.ft8o_normalize_logl.argtypes = [fp]
    L.ft8o_bp_decode.argtypes = [fp, C.c_int, u8p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ft8o_compute_crc.argtypes = [u8p, C.c_int]
    L.ft8o_compute_crc.restype = C.c_uint16
    L.ft8o_unpack77.argtypes = [u8p, C.c_char_p]
    L.ft8o_unpack77.restype = C.c_int
    L.ft8o_pack77.argtypes = [C.c_char_p, u8p]
    L.ft8o_pack77.restype = C.c_int
    L.ft8o_encode.argtypes = [u8p, u8p]
    L.ft8o_decode.argtypes = [u8p, C.c_void_p, C.POINTER(Message), C.c_int, C.POINTER(DecodeStatus),
                              C.POINTER(DecodeExtra)]
    L.ft8o_decode.restype = C.c_int
    L.ft8o_subsystem.argtypes = [fp, fp, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32)]
    L.ft8o_subsystem_ex.argtypes = [fp, fp, C.POINTER(Params), C.c_void_p, C.POINTER(C.c_int32)]
    L.ft8o_subsystem_from_waterfall.argtypes = [u8p, C.POINTER(Params), C.c_void_p, C.POINTER(C.c_int32)]
    L.ft8o_subsystem_batch.argtypes = [fp, C.c_int, C.POINTER(Params), C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    L.ft8o_selftest_signal.argtypes = [fp, fp, C.c_uint]
    L.ft8o_selftest_signal.restype = C.c_int
    L.ft8o_normalise.argtypes = [fp, fp, C.c_int]
    L.ft8o_write_raw_iq.argtypes = [fp, fp, C.c_char_p]
    L.ft8o_write_raw_iq.restype = C.c_int32
    L.ft8o_read_raw_iq.argtypes = [fp, fp, C.c_char_p]
    L.ft8o_read_raw_iq.restype = C.c_int32
    L.ft8o_read_c2.argtypes = [fp, fp, C.c_char_p, C.POINTER(C.c_double)]
    L.ft8o_read_c2.restype = C.c_int32
    L.ft8o_init()
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def default_params(min_score=10, max_candidates=120, ldpc_iters=20):
    return Params(min_score, max_candidates, ldpc_iters)


def selftest_signal(seed=1):
    i = np.zeros(NSAMPLES, np.float32)
    q = np.zeros(NSAMPLES, np.float32)
    assert lib().ft8o_selftest_signal(_fp(i), _fp(q), seed) == 1
    return i, q


def waterfall(i, q, f64=False):
    i = np.ascontiguousarray(i, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    m = np.zeros(MAG_ARRAY, np.uint8)
    (lib().ft8o_waterfall_f64 if f64 else lib().ft8o_waterfall)(_fp(i), _fp(q), _u8(m))
    return m


def find_sync(mag, max_candidates=120, min_score=10):
    mag = np.ascontiguousarray(mag, np.uint8)
    heap = np.zeros(max_candidates, CAND_DTYPE)
    n = lib().ft8o_find_sync(_u8(mag), max_candidates, heap.ctypes.data, min_score)
    return heap[:n].copy()


def score_map(mag):
    mag = np.ascontiguousarray(mag, np.uint8)
    s = np.zeros((2, 2, 36, 249), np.int16)
    lib().ft8o_score_map(_u8(mag), s.ctypes.data)
    return s


def decode(mag, cand, iters=20):
    """returns dict(ok, text, hash, ldpc_errors, crc_extracted, crc_calculated, unpack_status, iters, a91)"""
    mag = np.ascontiguousarray(mag, np.uint8)
    c = np.array([cand], CAND_DTYPE) if not isinstance(cand, np.ndarray) else np.ascontiguousarray(cand).reshape(1)
    msg, st, ex = Message(), DecodeStatus(), DecodeExtra()
    ok = lib().ft8o_decode(_u8(mag), c.ctypes.data, C.byref(msg), iters, C.byref(st), C.byref(ex))
    return dict(ok=bool(ok), text=msg.text.decode() if ok else "", hash=msg.hash if ok else 0,
                ldpc_errors=st.ldpc_errors, crc_extracted=st.crc_extracted, crc_calculated=st.crc_calculated,
                unpack_status=st.unpack_status, iters=ex.iters, a91=bytes(ex.a91))


def llr(mag, cand, normalise=True):
    mag = np.ascontiguousarray(mag, np.uint8)
    c = np.ascontiguousarray(cand).reshape(1)
    out = np.zeros(174, np.float32)
    lib().ft8o_extract_likelihood(_u8(mag), c.ctypes.data, _fp(out))
    if normalise:
        lib().ft8o_normalize_logl(_fp(out))
    return out


def bp_decode(codeword, iters=20):
    cw = np.ascontiguousarray(codeword, np.float32)
    plain = np.zeros(174, np.uint8)
    ok, it = C.c_int(), C.c_int()
    lib().ft8o_bp_decode(_fp(cw), iters, _u8(plain), C.byref(ok), C.byref(it))
    return plain, ok.value, it.value


def subsystem(i, q, params=None):
    i = np.ascontiguousarray(i, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    dec = np.zeros(MAX_MESSAGES, RESULT_DTYPE)
    n = C.c_int32(0)
    p = params or default_params()
    lib().ft8o_subsystem_ex(_fp(i), _fp(q), C.byref(p), dec.ctypes.data, C.byref(n))
    return dec, n.value


def subsystem_from_waterfall(mag, params=None):
    mag = np.ascontiguousarray(mag, np.uint8)
    dec = np.zeros(MAX_MESSAGES, RESULT_DTYPE)
    n = C.c_int32(0)
    p = params or default_params()
    lib().ft8o_subsystem_from_waterfall(_u8(mag), C.byref(p), dec.ctypes.data, C.byref(n))
    return dec, n.value


def subsystem_batch(iq, params=None, nthreads=1, decodes=None):
    """decodes: the caller's record array as it is BEFORE the call ([B][50], e.g. a byte pattern): the reference writes
    a slot only for a CQ message and leaves the others as they were (rtlsdr_ft8d.c:1509-1520); None = zeros"""
    iq = np.ascontiguousarray(iq, np.float32)
    B = iq.shape[0]
    assert iq.shape[1:] == (2, NSAMPLES)
    if decodes is None:
        dec = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
    else:
        dec = np.ascontiguousarray(decodes).copy()
        assert dec.dtype == RESULT_DTYPE and dec.shape == (B, MAX_MESSAGES)
    n = np.zeros(B, np.int32)
    p = params or default_params()
    lib().ft8o_subsystem_batch(_fp(iq), B, C.byref(p), dec.ctypes.data, n.ctypes.data_as(C.POINTER(C.c_int32)), nthreads)
    return dec, n


def fftw_init(explicit_path=None):
    """(bound, detail): binds libfftw3f at run time for the oracle's reference-FFT leg; detail = the library found, or
    the names searched.  Sticky per process."""
    L = lib()
    L.ft8o_fftw_init.argtypes = [C.c_char_p]
    L.ft8o_fftw_detail.restype = C.c_char_p
    ok = L.ft8o_fftw_init(None if explicit_path is None else str(explicit_path).encode())
    return bool(ok), L.ft8o_fftw_detail().decode()


def subsystem_batch_fftw(iq, params=None, nthreads=1):
    """subsystem_batch with the reference's own FFT (needs fftw_init() to have bound the library)"""
    iq = np.ascontiguousarray(iq, np.float32)
    B = iq.shape[0]
    dec = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
    n = np.zeros(B, np.int32)
    p = params or default_params()
    L = lib()
    L.ft8o_subsystem_batch_fftw.argtypes = [C.c_void_p, C.c_int, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int]
    if L.ft8o_subsystem_batch_fftw(iq.ctypes.data, B, C.byref(p), dec.ctypes.data, n.ctypes.data, nthreads) != 0:
        raise RuntimeError("FFTW is not bound: " + L.ft8o_fftw_detail().decode())
    return dec, n


def waterfall_batch(iq, f64=False, nthreads=1):
    """f64: False = the float32 R4DIF FFT, True = float64 DFT, 2 = the reference's FFTW (after fftw_init bound it)"""
    iq = np.ascontiguousarray(iq, np.float32)
    B = iq.shape[0]
    mag = np.zeros((B, MAG_ARRAY), np.uint8)
    L = lib()
    L.ft8o_waterfall_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.ft8o_waterfall_batch(iq.ctypes.data, B, mag.ctypes.data, int(f64), nthreads)
    return mag


def subsystem_from_waterfall_batch(mag, params=None, nthreads=1):
    mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
    B = mag.shape[0]
    dec = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
    n = np.zeros(B, np.int32)
    p = params or default_params()
    L = lib()
    L.ft8o_subsystem_from_waterfall_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int]
    L.ft8o_subsystem_from_waterfall_batch(mag.ctypes.data, B, C.byref(p), dec.ctypes.data, n.ctypes.data, nthreads)
    return dec, n


def decode_from_candidates_batch(mag, cands, counts, params=None, nthreads=1):
    """everything after ft8_find_sync (rtlsdr_ft8d.c:1452-1523) for B frames: cands [B][max_candidates]"""
    mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
    B = mag.shape[0]
    p = params or default_params()
    cands = np.ascontiguousarray(cands)
    counts = np.ascontiguousarray(counts, np.int32)
    assert cands.dtype == CAND_DTYPE and cands.shape == (B, p.max_candidates)
    dec = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
    n = np.zeros(B, np.int32)
    L = lib()
    L.ft8o_decode_from_candidates_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Params),
                                                    C.c_void_p, C.c_void_p, C.c_int]
    L.ft8o_decode_from_candidates_batch(mag.ctypes.data, cands.ctypes.data, counts.ctypes.data, B, C.byref(p),
                                        dec.ctypes.data, n.ctypes.data, nthreads)
    return dec, n


def find_sync_batch(mag, max_candidates=120, min_score=10, nthreads=1):
    """ft8_find_sync for B waterfalls -> (cands [B][cap] CAND_DTYPE with zeros behind each count, counts [B])"""
    mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
    B = mag.shape[0]
    cands = np.zeros((B, max_candidates), CAND_DTYPE)
    counts = np.zeros(B, np.int32)
    L = lib()
    L.ft8o_find_sync_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.ft8o_find_sync_batch(mag.ctypes.data, B, max_candidates, min_score, cands.ctypes.data, counts.ctypes.data, nthreads)
    return cands, counts


def decode_candidates_batch(mag, cands, counts, iters=20, nthreads=1):
    """ft8_decode (rtlsdr_ft8d.c:1476) for every candidate of B frames -> uint8 [B][cap][48]: the canonical status records
    ft8gpu_decode_candidates writes (fields zero unless ft8_decode would have set them)"""
    mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
    B = mag.shape[0]
    cands = np.ascontiguousarray(cands)
    counts = np.ascontiguousarray(counts, np.int32)
    assert cands.dtype == CAND_DTYPE and cands.shape[0] == B
    cap = cands.shape[1]
    out = np.zeros((B, cap, 48), np.uint8)
    L = lib()
    L.ft8o_decode_candidates_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.ft8o_decode_candidates_batch(mag.ctypes.data, cands.ctypes.data, counts.ctypes.data, B, cap, iters, out.ctypes.data, nthreads)
    return out


def synth_cpfsk(tones, f_tone0_hz, start_sample, amplitude):
    """rtlsdr_ft8d.c:946-955 generalised to S signals, no noise: (i, q) float32 [48000] each"""
    tones = np.ascontiguousarray(tones, np.uint8).reshape(-1, 79)
    S = tones.shape[0]
    f = np.ascontiguousarray(f_tone0_hz, np.float64)
    st = np.ascontiguousarray(start_sample, np.int32)
    a = np.ascontiguousarray(amplitude, np.float64)
    assert f.shape == st.shape == a.shape == (S,)
    i = np.zeros(NSAMPLES, np.float32)
    q = np.zeros(NSAMPLES, np.float32)
    L = lib()
    L.ft8o_synth_cpfsk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ft8o_synth_cpfsk(tones.ctypes.data, f.ctypes.data, st.ctypes.data, a.ctypes.data, S, i.ctypes.data, q.ctypes.data)
    return i, q


def normalise(i, q):
    i = np.ascontiguousarray(i, np.float32).copy()
    q = np.ascontiguousarray(q, np.float32).copy()
    lib().ft8o_normalise(_fp(i), _fp(q), NSAMPLES)
    return i, q


def pack77(msg):
    out = np.zeros(12, np.uint8)
    rc = lib().ft8o_pack77(msg.encode(), _u8(out))
    return rc, out


def encode(payload):
    payload = np.ascontiguousarray(payload, np.uint8)
    tones = np.zeros(79, np.uint8)
    lib().ft8o_encode(_u8(payload), _u8(tones))
    return tones


def unpack77(a77):
    a = np.zeros(12, np.uint8)
    a[:len(a77)] = np.frombuffer(bytes(a77), np.uint8)
    buf = C.create_string_buffer(64)
    rc = lib().ft8o_unpack77(_u8(a), buf)
    return rc, buf.value.decode()


def crc14(data, nbits):
    a = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    return lib().ft8o_compute_crc(_u8(a), nbits)


def rx_capture(raw, normalise=False):
    """rtlsdr_callback() over a whole raw capture from reset; returns (i, q, n_out)"""
    raw = np.ascontiguousarray(raw, np.uint8)
    L = lib()
    L.ft8o_rx_capture.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    i = np.zeros(NSAMPLES, np.float32)
    q = np.zeros(NSAMPLES, np.float32)
    n = C.c_uint32(0)
    L.ft8o_rx_capture(raw.ctypes.data, raw.size, i.ctypes.data, q.ctypes.data, C.byref(n), int(normalise))
    return i, q, n.value


class ReportInfo(C.Structure):
    _fields_ = [("rcall", C.c_char * 13), ("rloc", C.c_char * 7), ("app_version", C.c_char * 32),
                ("dial_freq", C.c_uint32), ("unixtime", C.c_uint32), ("sequence", C.c_uint32),
                ("random_id", C.c_uint32)]


def pskreporter_datagram(decodes, n_results, info):
    """postSpots() bytes (rtlsdr_ft8d.c:386-561) for one frame's spot list"""
    decodes = np.ascontiguousarray(decodes)
    assert decodes.dtype == RESULT_DTYPE
    L = lib()
    L.ft8o_pskreporter_datagram.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ReportInfo), C.c_void_p]
    out = np.zeros(1536, np.uint8)
    n = L.ft8o_pskreporter_datagram(decodes.ctypes.data, int(n_results), C.byref(info), out.ctypes.data)
    return out[:n].copy()


def format_spots(decodes, n_results, dial_freq, year, month, mday, hour, minute):
    decodes = np.ascontiguousarray(decodes)
    L = lib()
    L.ft8o_format_spots.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_size_t]
    buf = C.create_string_buffer(4096)
    L.ft8o_format_spots(decodes.ctypes.data, int(n_results), int(dial_freq), year, month, mday, hour, minute, buf, len(buf))
    return buf.value.decode()
