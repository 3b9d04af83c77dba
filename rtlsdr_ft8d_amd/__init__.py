"""rtlsdr_ft8d_amd -- Python view of libft8gpu.so (include/ft8gpu.h).

The product is the C-ABI shared library built from ``csrc/`` (hand-written HIP kernels for
gfx950 + C host side).  This module only binds it with ctypes for the tests and ``bench.py``;
there is no Python or CPU fallback: if the library is missing, import of the binding raises.
"""
import ctypes as C
import glob
import hashlib
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libft8gpu.so")
AB_LIB_PATH = os.path.join(_HERE, "libft8gpu_ab.so")      # `make -C csrc ab`: product kernels + the alternative kernel forms

NSAMPLES = 48000            # rtlsdr_ft8d.h:34-35
MAG_ARRAY = 94208           # rtlsdr_ft8d.h:56
MAX_MESSAGES = 50           # rtlsdr_ft8d.h:46
SCORES_PER_FRAME = 2 * 2 * 36 * 249
HOST_PTRS, DEVICE_PTRS = 0, 1

RESULT_DTYPE = np.dtype([("call", "S13"), ("loc", "S7"), ("freq", "<i4"), ("snr", "<i4")], align=True)
CAND_DTYPE = np.dtype([("score", "<i2"), ("time_offset", "<i2"), ("freq_offset", "<i2"),
                       ("time_sub", "u1"), ("freq_sub", "u1")])
STATUS_DTYPE = np.dtype([("ldpc_errors", "<i2"), ("iters", "<i2"), ("crc_extracted", "<u2"),
                         ("crc_calculated", "<u2"), ("unpack_status", "i1"), ("ok", "u1"),
                         ("a91", "u1", (12,)), ("text", "S25"), ("pad", "u1")])
SIGNAL_DTYPE = np.dtype([("tones", "u1", (79,)), ("pad", "u1"), ("f0_hz", "<f4"), ("t0_s", "<f4"),
                         ("amplitude", "<f4")])
assert RESULT_DTYPE.itemsize == 28 and CAND_DTYPE.itemsize == 8
assert STATUS_DTYPE.itemsize == 48 and SIGNAL_DTYPE.itemsize == 92


class Params(C.Structure):
    _fields_ = [("min_score", C.c_int32), ("max_candidates", C.c_int32), ("ldpc_iters", C.c_int32)]


class Timings(C.Structure):
    _fields_ = [("waterfall_ms", C.c_float), ("sync_ms", C.c_float), ("heap_ms", C.c_float),
                ("decode_ms", C.c_float), ("spots_ms", C.c_float), ("total_ms", C.c_float),
                ("launches_per_stage", C.c_int32)]


class ReportInfo(C.Structure):
    """ft8gpu_report_info: receiver identity and the per-datagram constants of postSpots() (rtlsdr_ft8d.c:365-590)"""
    _fields_ = [("rcall", C.c_char * 13), ("rloc", C.c_char * 7), ("app_version", C.c_char * 32),
                ("dial_freq", C.c_uint32), ("unixtime", C.c_uint32), ("sequence", C.c_uint32),
                ("random_id", C.c_uint32)]


DATAGRAM_STRIDE = 1408
STREAM_LEGACY = 1                       # FT8GPU_STREAM_LEGACY (= hipStreamLegacy): the legacy null stream, explicitly
DBG_FORCE_IEEE_DIV, DBG_PIPELINE_FORM, DBG_NO_OVERLAP = 1, 2, 4      # FT8GPU_DBG_* test hooks (per context)
# selector bits of the alternative (bit-identical) kernel forms: accepted by the A/B build only (csrc/ft8gpu_internal.h)
AB_WATERFALL_LDS, AB_HEAP_LANE_PER_FRAME, AB_HEAP_WAVE_PER_FRAME = 8, 16, 32


class Ft8GpuError(RuntimeError):
    pass


ABI_SYMBOLS = [
    "ft8gpu_create", "ft8gpu_destroy", "ft8gpu_set_stream", "ft8gpu_get_stream", "ft8gpu_set_params", "ft8gpu_enable_timing",
    "ft8gpu_get_timings", "ft8gpu_synchronize", "ft8gpu_last_error", "ft8gpu_device_count",
    "ft8gpu_decode_batch", "ft8gpu_waterfall", "ft8gpu_find_sync", "ft8gpu_score_map",
    "ft8gpu_decode_candidates", "ft8gpu_collect_spots", "ft8gpu_pack77_std", "ft8gpu_encode",
    "ft8gpu_synth_frames", "ft8gpu_synth_frames_at", "ft8gpu_rx_decimate", "ft8gpu_pskreporter_datagrams", "ft8gpu_format_spots",
    "ft8gpu_dev_alloc", "ft8gpu_dev_free", "ft8gpu_memcpy_h2d", "ft8gpu_memcpy_d2h", "ft8gpu_host_alloc", "ft8gpu_host_free",
    "ft8gpu_overlap_active", "ft8gpu_overlap_reason", "ft8gpu_build_id", "ft8gpu_pack77",
    "ft8gpu_set_debug_flags", "ft8gpu_selftest_bp_math", "ft8gpu_selftest_norm_math", "ft8gpu_gather_spots", "ft8gpu_gather_shutdown",
    "ft8gpu_shard_workers", "ft8gpu_decode_batch_multi", "ft8gpu_decode_batch_multi_dev",
    "ft8_find_sync", "ft8_decode", "ft8_encode", "pack77",            # ft8_lib level (include/ft8_lib/ft8/*.h)
    "initFFTW", "freeFFTW", "ft8_subsystem", "ft8gpu_read_raw_iq", "ft8gpu_read_c2", "ft8gpu_write_raw_iq",
]

_lib = None


def _hash16(paths):
    h = hashlib.sha256()
    for path in paths:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def source_build_id():
    """what ft8gpu_build_id() of a library built from THIS tree returns (csrc/Makefile computes the same two hashes):
    "<dev>.<all>" -- device sources (csrc/*.hip, *.h), and every source of the library (+ csrc/*.c, Makefile, the linker version script, include/)"""
    d = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    dev = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")), key=os.path.basename)
    rest = sorted(glob.glob(os.path.join(d, "*.c")), key=os.path.basename) + [os.path.join(d, "Makefile"), os.path.join(d, "libft8gpu.map"), os.path.join(inc, "ft8gpu.h")] + \
        sorted(glob.glob(os.path.join(inc, "ft8_lib", "ft8", "*.h")), key=os.path.basename)
    return f"{_hash16(dev)}.{_hash16(dev + rest)}"


def device_source_id():
    """the first half of the build id: identity of the kernel sources (what committed PMC summaries are gated on)"""
    return source_build_id().split(".")[0]


def build_id(lib=None):
    return (lib or load_library()).ft8gpu_build_id().decode()


def check_build_id(lib=None, suffix=""):
    """raises unless the loaded library was built from the sources beside it"""
    have, want = build_id(lib), source_build_id() + suffix
    if have != want:
        raise Ft8GpuError(f"libft8gpu{'_ab' if suffix else ''}.so is stale or foreign: its build id is {have}, the sources beside it give {want} "
                          "(rebuild: make -C rtlsdr_ft8d_amd/csrc" + (" ab)" if suffix else ")"))
    return have


def load_library():
    """dlopen libft8gpu.so and declare prototypes.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP/HSA runtime under the same SONAME (libamdhip64.so.7) as
    # /opt/rocm's.  A process must hold exactly ONE of them: two copies each open the GPU and the
    # second one finds no devices.  Importing torch first makes the dynamic loader resolve this
    # library's libamdhip64.so.7 dependency to the copy torch already mapped.  (A plain C caller
    # such as rtlsdr_ft8d.c simply gets /opt/rocm's runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise Ft8GpuError(
            f"{LIB_PATH} is missing: build it with `make -C rtlsdr_ft8d_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    _lib = _declare(C.CDLL(LIB_PATH))
    return _lib


def load_ab_library():
    """the A/B build (product kernels + alternative kernel forms behind extra debug-flag bits), checked against the tree"""
    if not os.path.exists(AB_LIB_PATH):
        raise Ft8GpuError(f"{AB_LIB_PATH} is missing: build it with `make -C rtlsdr_ft8d_amd/csrc ab`")
    L = load_library_at(AB_LIB_PATH)
    check_build_id(L, "+ab")
    return L


def load_library_at(path):
    """another build of libft8gpu.so beside the product's (tools/ab_libs.py: A/B of two builds in one process)"""
    load_library()                                   # the product library first: it fixes which HIP runtime is mapped
    return _declare(C.CDLL(os.path.abspath(path)))


def _declare(L):
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    L.ft8gpu_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(Params)]
    L.ft8gpu_destroy.argtypes = [vp]
    L.ft8gpu_destroy.restype = None
    L.ft8gpu_set_stream.argtypes = [vp, vp]
    if hasattr(L, "ft8gpu_get_stream"):
        L.ft8gpu_get_stream.argtypes = [vp]
        L.ft8gpu_get_stream.restype = vp
    L.ft8gpu_set_params.argtypes = [vp, C.POINTER(Params)]
    L.ft8gpu_enable_timing.argtypes = [vp, C.c_int]
    L.ft8gpu_get_timings.argtypes = [vp, C.POINTER(Timings), C.POINTER(C.c_int32)]
    L.ft8gpu_synchronize.argtypes = [vp]
    L.ft8gpu_last_error.restype = C.c_char_p
    L.ft8gpu_decode_batch.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
    L.ft8gpu_waterfall.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    L.ft8gpu_find_sync.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
    L.ft8gpu_score_map.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    L.ft8gpu_decode_candidates.argtypes = [vp, vp, vp, vp, C.c_int, vp, C.c_int]
    L.ft8gpu_collect_spots.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.ft8gpu_pack77_std.argtypes = [C.c_char_p, vp]
    if hasattr(L, "ft8gpu_pack77"):                       # absent from older builds loaded by load_library_at
        L.ft8gpu_pack77.argtypes = [C.c_char_p, vp]
        L.ft8gpu_build_id.restype = C.c_char_p
    if hasattr(L, "ft8gpu_selftest_norm_math"):
        L.ft8gpu_selftest_norm_math.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.ft8gpu_overlap_reason.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.ft8gpu_encode.argtypes = [vp, vp]
    L.ft8gpu_encode.restype = None
    L.ft8gpu_synth_frames.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint64, vp]
    L.ft8gpu_synth_frames_at.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, vp]
    L.ft8gpu_set_debug_flags.argtypes = [vp, C.c_uint]
    if hasattr(L, "ft8gpu_selftest_bp_math"):             # absent from older builds loaded by load_library_at
        L.ft8gpu_selftest_bp_math.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ft8gpu_decode_batch_multi.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, vp, vp]
    L.ft8gpu_decode_batch_multi_dev.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_int), vp, vp]
    if hasattr(L, "ft8gpu_gather_spots"):
        L.ft8gpu_gather_spots.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(vp)]
        L.ft8gpu_gather_shutdown.restype = None
    L.ft8gpu_rx_decimate.argtypes = [vp, vp, C.c_int, C.c_size_t, vp, C.c_int, C.c_int]
    L.ft8gpu_pskreporter_datagrams.argtypes = [vp, vp, vp, C.c_int, C.POINTER(ReportInfo), vp, vp, vp, C.c_int]
    L.ft8gpu_format_spots.argtypes = [vp, C.c_int32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    L.ft8gpu_dev_alloc.argtypes = [vp, C.c_size_t]
    L.ft8gpu_dev_alloc.restype = vp
    L.ft8gpu_dev_free.argtypes = [vp, vp]
    L.ft8gpu_dev_free.restype = None
    if hasattr(L, "ft8gpu_host_alloc"):                   # absent from older builds loaded by load_library_at
        L.ft8gpu_host_alloc.argtypes = [C.c_size_t]
        L.ft8gpu_host_alloc.restype = vp
        L.ft8gpu_host_free.argtypes = [vp]
        L.ft8gpu_host_free.restype = None
        L.ft8gpu_overlap_active.argtypes = [vp]
    L.ft8gpu_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.ft8gpu_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.initFFTW.restype = None
    L.freeFFTW.restype = None
    L.ft8_subsystem.argtypes = [vp, vp, C.c_uint32, vp, i32p]
    L.ft8_subsystem.restype = None
    L.ft8gpu_read_raw_iq.argtypes = [vp, vp, C.c_char_p]
    L.ft8gpu_read_raw_iq.restype = C.c_int32
    L.ft8gpu_read_c2.argtypes = [vp, vp, C.c_char_p, C.POINTER(C.c_double)]
    L.ft8gpu_read_c2.restype = C.c_int32
    L.ft8gpu_write_raw_iq.argtypes = [vp, vp, C.c_char_p]
    L.ft8gpu_write_raw_iq.restype = C.c_int32
    return L


def _check(rc, lib=None):
    if rc != 0:
        raise Ft8GpuError((lib or load_library()).ft8gpu_last_error().decode(errors="replace"))


def _ptr(a):
    """host numpy array or device pointer (int / torch tensor) -> integer address"""
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return int(a)


def pack77_std(msg):
    out = np.zeros(10, np.uint8)
    rc = load_library().ft8gpu_pack77_std(msg.encode(), out.ctypes.data)
    if rc != 0:
        raise ValueError(f"cannot pack {msg!r} as a standard FT8 message")
    return out


def pack77(msg):
    """ft8gpu_pack77: any message type the packer knows (type 1 / 2 with grid, report, RRR / RR73 / 73, /R /P, CQ modifiers,
    <hashed> calls; type 4; telemetry; free text) -> 10 bytes"""
    out = np.zeros(10, np.uint8)
    rc = load_library().ft8gpu_pack77(msg.encode(), out.ctypes.data)
    if rc != 0:
        raise ValueError(f"cannot pack {msg!r} as an FT8 message")
    return out


def encode(payload):
    payload = np.ascontiguousarray(payload, np.uint8)
    tones = np.zeros(79, np.uint8)
    load_library().ft8gpu_encode(payload.ctypes.data, tones.ctypes.data)
    return tones


class Decoder:
    """One GPU decoder context (ft8gpu_ctx).  Host arrays are numpy; device arrays are anything
    with ``data_ptr()`` (torch tensors) or raw integer addresses."""

    def __init__(self, device=0, max_frames=64, min_score=10, max_candidates=120, ldpc_iters=20, lib=None):
        self.lib = lib or load_library()
        self.params = Params(min_score, max_candidates, ldpc_iters)
        self.max_frames = max_frames
        h = C.c_void_p()
        self._ck(self.lib.ft8gpu_create(C.byref(h), device, max_frames, C.byref(self.params)))
        self.h = h

    def _ck(self, rc):
        _check(rc, self.lib)            # the error text lives in the library that failed (the A/B build is a second library)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ft8gpu_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def max_candidates(self):
        return self.params.max_candidates

    def set_params(self, min_score=None, max_candidates=None, ldpc_iters=None):
        p = Params(self.params.min_score if min_score is None else min_score,
                   self.params.max_candidates if max_candidates is None else max_candidates,
                   self.params.ldpc_iters if ldpc_iters is None else ldpc_iters)
        self._ck(self.lib.ft8gpu_set_params(self.h, C.byref(p)))
        self.params = p

    def set_stream(self, stream_handle):
        """None: the context creates its own stream.  An integer is a hipStream_t; 0 (what torch reports for its
        default stream) is passed as hipStreamLegacy, the explicit name of the null stream."""
        if stream_handle is None:
            h = None
        else:
            h = int(stream_handle) or STREAM_LEGACY
        self._ck(self.lib.ft8gpu_set_stream(self.h, C.c_void_p(h)))

    def stream_handle(self):
        """the hipStream_t of the context as an integer (e.g. for torch.cuda.ExternalStream)"""
        return int(self.lib.ft8gpu_get_stream(self.h) or 0)

    def set_debug_flags(self, flags):
        self._ck(self.lib.ft8gpu_set_debug_flags(self.h, int(flags)))

    def overlap_active(self):
        """True: the two-part pipeline with the serial kernels on side streams is in use for large batches (the
        context has SEEN its streams run kernels concurrently); False: plain pipeline (overlap_reason() says why)"""
        return bool(self.lib.ft8gpu_overlap_active(self.h))

    def overlap_reason(self):
        buf = C.create_string_buffer(256)
        self._ck(self.lib.ft8gpu_overlap_reason(self.h, buf, len(buf)))
        return buf.value.decode()

    def selftest_bp_math(self):
        """exhaustive (2^32 inputs) comparison of the BP kernel's short division chains with the IEEE quotient"""
        out = (C.c_uint64 * 7)()
        self._ck(self.lib.ft8gpu_selftest_bp_math(self.h, out))
        keys = ("tanh_inputs", "tanh_mismatch", "atanh_inputs", "atanh_mismatch", "pair_mismatch", "tanh_max_bits", "first_bad")
        d = dict(zip(keys, [int(v) for v in out]))
        d["tanh_max"] = float(np.array([d["tanh_max_bits"]], np.uint32).view(np.float32)[0])
        return d

    def selftest_norm_math(self):
        """sqrtf(24.0f / v), the LLR scale factor, against exact arithmetic for every float v in [2^-60, 2^60]"""
        out = (C.c_uint64 * 7)()
        self._ck(self.lib.ft8gpu_selftest_norm_math(self.h, out))
        return dict(zip(("inputs", "div_bad", "sqrt_bad", "compose_bad", "first_bad", "rational_inputs", "rational_div_bad"), [int(v) for v in out]))

    def enable_timing(self, on=True):
        self._ck(self.lib.ft8gpu_enable_timing(self.h, int(on)))

    def timings(self):
        """mean per-stage milliseconds over the runs recorded since enable_timing(True)"""
        t, n = Timings(), C.c_int32(0)
        self._ck(self.lib.ft8gpu_get_timings(self.h, C.byref(t), C.byref(n)))
        d = {k: getattr(t, k) for k, _ in Timings._fields_}
        d["runs"] = n.value
        return d

    def synchronize(self):
        self._ck(self.lib.ft8gpu_synchronize(self.h))

    # ---- host (numpy) API ------------------------------------------------------------------
    def decode_batch(self, iq, decodes=None):
        iq = np.ascontiguousarray(iq, np.float32)
        B = iq.shape[0]
        assert iq.shape[1:] == (2, NSAMPLES)
        if decodes is None:
            decodes = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
        n = np.zeros(B, np.int32)
        self._ck(self.lib.ft8gpu_decode_batch(self.h, iq.ctypes.data, B, decodes.ctypes.data, n.ctypes.data, HOST_PTRS))
        return decodes, n

    def waterfall(self, iq):
        iq = np.ascontiguousarray(iq, np.float32)
        B = iq.shape[0]
        assert iq.shape[1:] == (2, NSAMPLES)
        mag = np.zeros((B, MAG_ARRAY), np.uint8)
        self._ck(self.lib.ft8gpu_waterfall(self.h, iq.ctypes.data, B, mag.ctypes.data, HOST_PTRS))
        return mag

    def find_sync(self, mag):
        mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
        B = mag.shape[0]
        cands = np.zeros((B, self.max_candidates), CAND_DTYPE)
        counts = np.zeros(B, np.int32)
        self._ck(self.lib.ft8gpu_find_sync(self.h, mag.ctypes.data, B, cands.ctypes.data, counts.ctypes.data, HOST_PTRS))
        return cands, counts

    def score_map(self, mag):
        mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
        B = mag.shape[0]
        s = np.zeros((B, 2, 2, 36, 249), np.int16)
        self._ck(self.lib.ft8gpu_score_map(self.h, mag.ctypes.data, B, s.ctypes.data, HOST_PTRS))
        return s

    def decode_candidates(self, mag, cands, counts):
        mag = np.ascontiguousarray(mag, np.uint8).reshape(-1, MAG_ARRAY)
        B = mag.shape[0]
        cands = np.ascontiguousarray(cands)
        counts = np.ascontiguousarray(counts, np.int32)
        assert cands.shape == (B, self.max_candidates) and cands.dtype == CAND_DTYPE
        st = np.zeros((B, self.max_candidates), STATUS_DTYPE)
        self._ck(self.lib.ft8gpu_decode_candidates(self.h, mag.ctypes.data, cands.ctypes.data, counts.ctypes.data,
                                                 B, st.ctypes.data, HOST_PTRS))
        return st

    def collect_spots(self, cands, counts, status, decodes=None):
        cands = np.ascontiguousarray(cands)
        counts = np.ascontiguousarray(counts, np.int32)
        status = np.ascontiguousarray(status)
        B = counts.shape[0]
        if decodes is None:
            decodes = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
        n = np.zeros(B, np.int32)
        self._ck(self.lib.ft8gpu_collect_spots(self.h, cands.ctypes.data, counts.ctypes.data, status.ctypes.data, B,
                                             decodes.ctypes.data, n.ctypes.data, HOST_PTRS))
        return decodes, n

    # ---- device-pointer API (inputs and outputs resident in HBM) --------------------------------
    def decode_batch_dev(self, iq_dev, nframes, decodes_dev, n_results_dev):
        self._ck(self.lib.ft8gpu_decode_batch(self.h, _ptr(iq_dev), nframes, _ptr(decodes_dev), _ptr(n_results_dev),
                                            DEVICE_PTRS))

    def find_sync_dev(self, mag_dev, nframes, cands_dev, counts_dev):
        """cands_dev: [nframes][max_candidates] 8-byte records, counts_dev: [nframes] int32 (all in HBM)"""
        self._ck(self.lib.ft8gpu_find_sync(self.h, _ptr(mag_dev), nframes, _ptr(cands_dev), _ptr(counts_dev), DEVICE_PTRS))

    def decode_candidates_dev(self, mag_dev, cands_dev, counts_dev, nframes, status_dev):
        """status_dev: [nframes][max_candidates] 48-byte records in HBM; only records below counts are written"""
        self._ck(self.lib.ft8gpu_decode_candidates(self.h, _ptr(mag_dev), _ptr(cands_dev), _ptr(counts_dev), nframes, _ptr(status_dev), DEVICE_PTRS))

    def waterfall_dev(self, iq_dev, nframes, mag_dev):
        self._ck(self.lib.ft8gpu_waterfall(self.h, _ptr(iq_dev), nframes, _ptr(mag_dev), DEVICE_PTRS))

    def rx_decimate(self, raw, normalise=True):
        """raw: uint8 [ncaptures][2*npairs] host array -> float32 [ncaptures][2][48000]"""
        raw = np.ascontiguousarray(raw, np.uint8)
        ncap, nbytes = raw.shape
        iq = np.zeros((ncap, 2, NSAMPLES), np.float32)
        self._ck(self.lib.ft8gpu_rx_decimate(self.h, raw.ctypes.data, ncap, nbytes // 2, iq.ctypes.data, int(normalise), HOST_PTRS))
        return iq

    def rx_decimate_dev(self, raw_dev, ncaptures, npairs, iq_dev, normalise=True):
        self._ck(self.lib.ft8gpu_rx_decimate(self.h, _ptr(raw_dev), ncaptures, npairs, _ptr(iq_dev), int(normalise), DEVICE_PTRS))

    def pskreporter_datagrams(self, decodes, n_results, info, unixtimes=None):
        """decodes: [n][50] RESULT_DTYPE, n_results: [n] -> (uint8 [n][DATAGRAM_STRIDE], int32 [n] lengths)"""
        decodes = np.ascontiguousarray(decodes)
        n_results = np.ascontiguousarray(n_results, np.int32)
        n = n_results.shape[0]
        assert decodes.dtype == RESULT_DTYPE and decodes.size == n * MAX_MESSAGES
        out = np.zeros((n, DATAGRAM_STRIDE), np.uint8)
        lengths = np.zeros(n, np.int32)
        t = None if unixtimes is None else np.ascontiguousarray(unixtimes, np.uint32)
        self._ck(self.lib.ft8gpu_pskreporter_datagrams(self.h, decodes.ctypes.data, n_results.ctypes.data, n, C.byref(info),
                                                     None if t is None else t.ctypes.data, out.ctypes.data,
                                                     lengths.ctypes.data, HOST_PTRS))
        return out, lengths

    def pskreporter_datagrams_dev(self, decodes_dev, n_results_dev, nframes, info, unixtimes_dev, datagrams_dev, lengths_dev):
        self._ck(self.lib.ft8gpu_pskreporter_datagrams(self.h, _ptr(decodes_dev), _ptr(n_results_dev), nframes, C.byref(info),
                                                     None if unixtimes_dev is None else _ptr(unixtimes_dev),
                                                     _ptr(datagrams_dev), _ptr(lengths_dev), DEVICE_PTRS))

    def synth_frames(self, signals, nframes, nsig, noise_sigma, seed, iq_dev, first_frame=0):
        """frame k of the call is global frame first_frame + k; its noise depends on (seed, global index) only"""
        signals = np.ascontiguousarray(signals)
        assert signals.dtype == SIGNAL_DTYPE and signals.size == nframes * nsig
        self._ck(self.lib.ft8gpu_synth_frames_at(self.h, signals.ctypes.data, nframes, nsig, float(noise_sigma),
                                               int(seed), int(first_frame), _ptr(iq_dev)))

    # ---- device memory helpers of the C ABI (a plain C caller has no HIP headers) ---------------
    def dev_alloc(self, nbytes):
        p = self.lib.ft8gpu_dev_alloc(self.h, nbytes)
        if not p:
            raise Ft8GpuError(self.lib.ft8gpu_last_error().decode(errors="replace"))
        return p

    def dev_free(self, p):
        self.lib.ft8gpu_dev_free(self.h, C.c_void_p(p))

    def memcpy_h2d(self, dst_dev, src):
        src = np.ascontiguousarray(src)
        self._ck(self.lib.ft8gpu_memcpy_h2d(self.h, C.c_void_p(_ptr(dst_dev)), src.ctypes.data, src.nbytes))

    def memcpy_d2h(self, dst, src_dev):
        assert dst.flags["C_CONTIGUOUS"]
        self._ck(self.lib.ft8gpu_memcpy_d2h(self.h, dst.ctypes.data, C.c_void_p(_ptr(src_dev)), dst.nbytes))


def decode_batch_multi(decoders, iq, decodes=None):
    """ft8gpu_decode_batch_multi: host frames [B][2][48000] cut into len(decoders) contiguous shards, one host
    thread and one context (normally one GPU) per shard, records gathered in the caller's host arrays"""
    iq = np.ascontiguousarray(iq, np.float32)
    B = iq.shape[0]
    assert iq.shape[1:] == (2, NSAMPLES)
    if decodes is None:
        decodes = np.zeros((B, MAX_MESSAGES), RESULT_DTYPE)
    n = np.zeros(B, np.int32)
    hs = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
    _check(load_library().ft8gpu_decode_batch_multi(hs, len(decoders), iq.ctypes.data, B, decodes.ctypes.data, n.ctypes.data))
    return decodes, n


class PinnedArray:
    """numpy view of page-locked host memory from ft8gpu_host_alloc.  The memory is freed by close() or at garbage
    collection -- and close() REFUSES while other references to `array` (or views of it) are alive: a view that outlived
    the allocation would be a use-after-free of unmapped pinned memory."""

    def __init__(self, shape, dtype=np.float32):
        lib = load_library()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib.ft8gpu_host_alloc(self.nbytes)
        if not self.ptr:
            raise Ft8GpuError(lib.ft8gpu_last_error().decode(errors="replace"))
        self._buf = (C.c_char * self.nbytes).from_address(self.ptr)
        self._base = np.frombuffer(self._buf, dtype=dtype)          # every view of `array` keeps a reference to this object
        self.array = self._base.reshape(shape)
        self._quiet = self._refs()              # the counts with no outside reference, measured (not assumed per CPython version)

    def _refs(self):
        import sys
        return sys.getrefcount(self._base), sys.getrefcount(self.array)

    def close(self):
        """frees the pinned memory; raises -- and changes NOTHING, `array` stays usable -- while outside references to
        `array` or views of it are alive"""
        if self.ptr:
            now = self._refs()
            extra = (now[0] - self._quiet[0]) + (now[1] - self._quiet[1])
            if extra > 0:
                raise Ft8GpuError(f"PinnedArray.close(): {extra} reference(s) to the pinned buffer (`array` or views of it) are still alive; drop them first")
            self.array = None
            self._base = None
            self._buf = None
            load_library().ft8gpu_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Ft8GpuError as e:
            import warnings
            warnings.warn(f"PinnedArray collected while views of it are alive: {self.nbytes} bytes of pinned host memory are NOT freed ({e})",
                          ResourceWarning, source=self)
        except Exception:
            pass


def decode_batch_multi_dev(decoders, iq_devs, nframes, decodes=None):
    """ft8gpu_decode_batch_multi_dev: shard g's frames are resident on decoders[g]'s GPU (iq_devs[g]: device
    pointer / tensor, nframes[g] frames); records gathered on the host in shard order"""
    total = int(sum(nframes))
    if decodes is None:
        decodes = np.zeros((total, MAX_MESSAGES), RESULT_DTYPE)
    n = np.zeros(total, np.int32)
    k = len(decoders)
    hs = (C.c_void_p * k)(*[d.h for d in decoders])
    ps = (C.c_void_p * k)(*[_ptr(p) if p is not None else None for p in iq_devs])
    ns = (C.c_int * k)(*[int(x) for x in nframes])
    _check(load_library().ft8gpu_decode_batch_multi_dev(hs, k, ps, ns, decodes.ctypes.data, n.ctypes.data))
    return decodes, n


def gather_spots(decoders, decodes_devs, n_results_devs, frames_per_dev, all_decodes_devs, all_n_results_devs):
    """ft8gpu_gather_spots: single-process RCCL all-gather of every GPU's device-resident records and counts"""
    k = len(decoders)
    arr = lambda xs: (C.c_void_p * k)(*[_ptr(x) for x in xs])
    hs = (C.c_void_p * k)(*[d.h for d in decoders])
    _check(load_library().ft8gpu_gather_spots(hs, k, arr(decodes_devs), arr(n_results_devs), int(frames_per_dev),
                                              arr(all_decodes_devs), arr(all_n_results_devs)))


def format_spots(decodes, n_results, dial_freq, year, month, mday, hour, minute):
    """printSpots() (rtlsdr_ft8d.c:643-663) as a string"""
    L = load_library()
    decodes = np.ascontiguousarray(decodes)
    assert decodes.dtype == RESULT_DTYPE
    buf = C.create_string_buffer(64 + 48 * MAX_MESSAGES)
    n = L.ft8gpu_format_spots(decodes.ctypes.data, int(n_results), int(dial_freq), year, month, mday, hour, minute, buf, len(buf))
    if n < 0:
        raise Ft8GpuError("ft8gpu_format_spots failed")
    return buf.value.decode()


def ft8_subsystem(i_samples, q_samples, decodes=None):
    """The reference's own entry point (rtlsdr_ft8d.h:164) through the drop-in symbol."""
    L = load_library()
    i_samples = np.ascontiguousarray(i_samples, np.float32)
    q_samples = np.ascontiguousarray(q_samples, np.float32)
    if decodes is None:
        decodes = np.zeros(MAX_MESSAGES, RESULT_DTYPE)
    n = C.c_int32(0)
    L.ft8_subsystem(i_samples.ctypes.data, q_samples.ctypes.data, NSAMPLES, decodes.ctypes.data, C.byref(n))
    return decodes, n.value
