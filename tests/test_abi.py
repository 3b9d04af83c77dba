"""CPU tests of the C-ABI library: it loads, exports every symbol include/ft8gpu.h declares, its POD
structs have the documented layout, and its host-only tooling (pack77 / encode / file readers)
agrees with the reference KAT and the oracle.  No GPU compute is invoked here."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ft8():
    import rtlsdr_ft8d_amd as m
    if not os.path.exists(m.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc"), "-j8"])
    m.load_library()
    return m


def declared_functions():
    src = ""
    for d, _, files in os.walk(os.path.join(ROOT, "include")):       # ft8gpu.h and the ft8_lib-level headers
        for f in sorted(files):
            if f.endswith(".h"):
                src += open(os.path.join(d, f)).read() + "\n"
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)              # preprocessor lines declare nothing
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n.startswith("ft8") or n in ("initFFTW", "freeFFTW", "pack77")))


def test_exports_every_declared_symbol(ft8):
    lib = ft8.load_library()
    names = declared_functions()
    assert len(names) >= 28
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(ft8.ABI_SYMBOLS) == names
    # nothing but the C ABI is in the dynamic symbol table: EVERY defined entry of whatever type (code, weak template
    # instantiations of libstdc++, data such as hipcc's __hip_cuid_*), not only the " T " ones -- csrc/libft8gpu.map
    for lib_path in (ft8.LIB_PATH, ft8.LIB_PATH.replace("libft8gpu.so", "libft8gpu_ab.so")):
        if not os.path.exists(lib_path):
            assert lib_path != ft8.LIB_PATH
            continue
        out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
        entries = [l.split() for l in out.splitlines() if l.strip()]
        assert sorted(e[-1] for e in entries) == names, (lib_path, sorted(set(e[-1] for e in entries) ^ set(names)))
        assert {e[-2] for e in entries} == {"T"}, [e for e in entries if e[-2] != "T"]


def test_struct_layouts(ft8):
    """struct decoder_results: offsets 0/13/20/24, size 28 (rtlsdr_ft8d.h:136-141)"""
    d = ft8.RESULT_DTYPE
    assert d.itemsize == 28 and [d.fields[k][1] for k in ("call", "loc", "freq", "snr")] == [0, 13, 20, 24]
    c = ft8.CAND_DTYPE
    assert c.itemsize == 8 and [c.fields[k][1] for k in ("score", "time_offset", "freq_offset", "time_sub", "freq_sub")] == [0, 2, 4, 6, 7]
    s = ft8.STATUS_DTYPE
    assert s.itemsize == 48 and s.fields["a91"][1] == 10 and s.fields["text"][1] == 22
    # cross-check against the C compiler's view of the header
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "ft8gpu.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(struct decoder_results),
 offsetof(struct decoder_results, loc), offsetof(struct decoder_results, freq), offsetof(struct decoder_results, snr),
 sizeof(ft8gpu_candidate), sizeof(ft8gpu_decode_status), offsetof(ft8gpu_decode_status, a91),
 offsetof(ft8gpu_decode_status, text), sizeof(ft8gpu_synth_signal)); return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=gnu17", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        vals = list(map(int, subprocess.check_output([exe]).split()))
    assert vals == [28, 13, 20, 24, 8, 48, 10, 22, ft8.SIGNAL_DTYPE.itemsize]


def test_encoder_tooling_matches_reference_kat(ft8, oracle):
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    p = ft8.pack77_std(kat["message"])
    assert p.tobytes().hex() == kat["packed_hex"]
    assert "".join(map(str, ft8.encode(p))) == kat["tones"]
    rng = np.random.default_rng(8)
    import synth_util as S
    for _ in range(200):
        msg = S.random_message(rng, cq=rng.random() < 0.5)
        rc, po = oracle.pack77(msg)
        assert rc == 0
        pg = ft8.pack77_std(msg)
        assert pg.tobytes() == po[:10].tobytes(), msg
        assert np.array_equal(ft8.encode(pg), oracle.encode(po)), msg
    for bad in ["", "CQ", "CQ K1JT ZZ99", "HELLO WORLD TEST", "CQ TOOLONGCALL FN20"]:
        with pytest.raises(ValueError):
            ft8.pack77_std(bad)


def test_replay_file_formats(ft8, oracle, tmp_path):
    """.iq writer/reader and .c2 reader (rtlsdr_ft8d.c:744-856) against the oracle's restatement"""
    lib = ft8.load_library()
    i, q = oracle.selftest_signal(1)
    p = str(tmp_path / "a.iq").encode()
    assert lib.ft8gpu_write_raw_iq(i.ctypes.data, q.ctypes.data, p) == 48000
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    p2 = str(tmp_path / "b.iq").encode()
    assert oracle.lib().ft8o_write_raw_iq(fp(i), fp(q), p2) == 48000
    assert open(p, "rb").read() == open(p2, "rb").read()
    gi, gq = np.zeros(48000, np.float32), np.zeros(48000, np.float32)
    oi, oq = np.zeros(48000, np.float32), np.zeros(48000, np.float32)
    assert lib.ft8gpu_read_raw_iq(gi.ctypes.data, gq.ctypes.data, p) == 48000
    assert oracle.lib().ft8o_read_raw_iq(fp(oi), fp(oq), p) == 48000
    assert np.array_equal(gi, oi) and np.array_equal(gq, oq)
    # .c2: 14-byte name, int type, double frequency, then the same payload
    c2 = str(tmp_path / "x.c2")
    with open(c2, "wb") as f:
        f.write(b"210101_0000.c2"[:14].ljust(14, b"\0"))
        f.write(np.int32(2).tobytes())
        f.write(np.float64(14074000.0).tobytes())
        f.write(open(p, "rb").read())
    fr = C.c_double(0)
    assert lib.ft8gpu_read_c2(gi.ctypes.data, gq.ctypes.data, c2.encode(), C.byref(fr)) == 48000
    assert fr.value == 14074000.0 and np.array_equal(gi, oi) and np.array_equal(gq, oq)
    # short file -> fewer samples reported
    short = str(tmp_path / "s.iq")
    open(short, "wb").write(open(p, "rb").read()[:8 * 1000])
    assert lib.ft8gpu_read_raw_iq(gi.ctypes.data, gq.ctypes.data, short.encode()) == 1000
    # ragged files: a dangling half record is dropped (nread / 2, :756), sizes off the reader's block grid, and the
    # peak normalisation of the part that was read -- all against the oracle's restatement of the reference loop
    for nfloats in (2 * 2048 + 1, 2 * 2049, 2 * 4096 - 1, 7, 1, 0, 2 * 48000 + 6):
        ragged = str(tmp_path / f"r{nfloats}.iq")
        open(ragged, "wb").write((open(p, "rb").read() + b"\0" * 64)[:4 * nfloats])
        gi[:] = 0; gq[:] = 0; oi[:] = 0; oq[:] = 0
        got = lib.ft8gpu_read_raw_iq(gi.ctypes.data, gq.ctypes.data, ragged.encode())
        assert got == oracle.lib().ft8o_read_raw_iq(fp(oi), fp(oq), ragged.encode()) == min(nfloats // 2, 48000)
        assert np.array_equal(gi, oi) and np.array_equal(gq, oq), nfloats
    assert lib.ft8gpu_read_raw_iq(gi.ctypes.data, gq.ctypes.data, b"/nonexistent/file.iq") == 0


def test_no_gpu_reports_error_not_fallback(ft8):
    """without a GPU the context cannot be created and says why; nothing silently falls back"""
    lib = ft8.load_library()
    if lib.ft8gpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(ft8.Ft8GpuError):
        ft8.Decoder(device=0, max_frames=1)
    dec, n = ft8.ft8_subsystem(np.zeros(48000, np.float32), np.zeros(48000, np.float32))
    assert n == 0


def test_missing_library_fails_loudly(monkeypatch):
    import importlib
    import rtlsdr_ft8d_amd as m
    monkeypatch.setattr(m, "_lib", None)
    monkeypatch.setattr(m, "LIB_PATH", "/nonexistent/libft8gpu.so")
    with pytest.raises(m.Ft8GpuError):
        m.load_library()


def test_product_never_touches_the_oracle():
    """the shipped package must not include, import, load or link anything under oracle/"""
    pkg = os.path.join(ROOT, "rtlsdr_ft8d_amd")
    pat = re.compile(r"oracle/|oracle_lib|ft8o_|libft8oracle|ft8_oracle")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="replace").read()
                assert not pat.search(txt), (dirpath, fn)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libft8gpu.so")]).decode()
    assert "oracle" not in out
    sym = subprocess.check_output(["nm", "-D", os.path.join(pkg, "libft8gpu.so")]).decode()
    assert "ft8o_" not in sym


def test_gather_entry_argument_errors_and_no_rccl_link_dependency(ft8):
    """ft8gpu_gather_spots checks its arguments before touching a GPU or RCCL, and libft8gpu.so itself must not
    depend on librccl (it is bound with dlopen at the first gather, so a plain C caller links without it)"""
    lib = ft8.load_library()
    assert lib.ft8gpu_gather_spots(None, 0, None, None, 4, None, None) == -1
    assert b"no contexts" in lib.ft8gpu_last_error()
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "rtlsdr_ft8d_amd", "libft8gpu.so")]).decode()
    assert "rccl" not in out and "nccl" not in out
    lib.ft8gpu_gather_shutdown()                       # harmless without communicators
    assert lib.ft8gpu_shard_workers() == 0             # no worker thread until a multi-GPU call needs one


def test_build_id_ties_the_library_to_the_sources(ft8, tmp_path):
    """ft8gpu_build_id() is the hash the Makefile took over the sources at link time; the binding recomputes it from the
    tree, so a stale or foreign .so is refused (smoke(), bench.py and the GPU tests call check_build_id)."""
    lib = ft8.load_library()
    have = lib.ft8gpu_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}\.[0-9a-f]{16}", have), have
    assert have == ft8.source_build_id() == ft8.check_build_id()
    assert have.split(".")[0] == ft8.device_source_id()
    # the shipped library has no alternative kernel forms: those live in the A/B build, whose id says so
    if os.path.exists(ft8.AB_LIB_PATH):
        assert ft8.build_id(ft8.load_ab_library()) == have + "+ab"
    # a library from other sources is refused: same tree, one byte more in a kernel source
    import shutil
    fake = tmp_path / "pkg"
    shutil.copytree(os.path.join(ROOT, "rtlsdr_ft8d_amd"), fake / "rtlsdr_ft8d_amd", ignore=shutil.ignore_patterns("build*", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), fake / "include")
    with open(fake / "rtlsdr_ft8d_amd" / "csrc" / "spots.hip", "a") as f:
        f.write("\n// edited after the library was built\n")
    code = ("import sys; sys.path.insert(0, %r); import rtlsdr_ft8d_amd as m\n"
            "try:\n    m.check_build_id()\nexcept m.Ft8GpuError as e:\n    print('REFUSED', e)\n" % str(fake))
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in out.stdout and "stale or foreign" in out.stdout, out.stdout + out.stderr


def test_pure_queries_leave_the_error_state_alone(ft8):
    """ft8gpu_overlap_active / ft8gpu_overlap_reason report through their own channel (round 4's getter overwrote
    ft8gpu_last_error() on success)"""
    lib = ft8.load_library()
    assert lib.ft8gpu_gather_spots(None, 0, None, None, 4, None, None) == -1
    before = lib.ft8gpu_last_error()
    buf = C.create_string_buffer(64)
    assert lib.ft8gpu_overlap_reason(None, buf, 64) == -1 and b"ctx is NULL" in lib.ft8gpu_last_error()
    assert before != lib.ft8gpu_last_error()


def test_pinned_array_close_refuses_while_views_live_and_keeps_the_buffer(ft8, monkeypatch):
    """PinnedArray.close(): with an outside reference to `array` or to a view of it alive it raises and changes nothing
    (the array stays usable, the memory stays allocated); with none it frees exactly once.  A collected object whose
    views are alive warns (ResourceWarning) instead of failing silently.  The allocator is stubbed with malloc: the
    real one needs a GPU (tests/test_gpu_multi.py runs it)."""
    import gc
    import warnings
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    freed = []

    class Stub:
        @staticmethod
        def ft8gpu_host_alloc(n):
            return libc.malloc(n)

        @staticmethod
        def ft8gpu_host_free(p):
            freed.append(p.value)

    monkeypatch.setattr(ft8, "load_library", lambda: Stub)
    pa = ft8.PinnedArray((4, 8), np.float32)
    ptr = pa.ptr
    pa.array[...] = 3.0
    view = pa.array[1:3]
    with pytest.raises(ft8.Ft8GpuError, match="still alive"):
        pa.close()
    assert pa.array is not None and pa.ptr == ptr and not freed and float(pa.array.sum()) == 96.0
    del view
    whole = pa.array                                     # a second name for the array itself is a live reference too
    with pytest.raises(ft8.Ft8GpuError, match="still alive"):
        pa.close()
    del whole
    pa.close()
    assert freed == [ptr] and pa.array is None and pa.ptr is None
    pa.close()                                            # idempotent
    assert freed == [ptr]
    # 1-D shape (reshape returns another view object as well) and collection with a live view
    pb = ft8.PinnedArray((16,), np.uint8)
    keep = pb.array[:4]
    ptr_b = pb.ptr
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        del pb
        gc.collect()
    assert any(issubclass(x.category, ResourceWarning) and "NOT freed" in str(x.message) for x in w), [str(x.message) for x in w]
    assert freed == [ptr]
    del keep
    libc.free(ptr); libc.free(ptr_b)
