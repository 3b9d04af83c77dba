"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the host C code (CPU box only: GPU ASan / XNACK are
not available on the GPU pool), and ThreadSanitizer runs of its host concurrency.  SURVEY.md section 4 notes the reference
only hints at ASan (Makefile:7).
  * the oracle (oracle/ft8_oracle.c, `make -C oracle asan`) under its own CPU test file;
  * csrc/ft8_compat.c + csrc/ft8_pack.c (pack77 / encode / file readers / formatter / drop-in shim) under tests/host_asan;
  * csrc/shard_pool.h and the global-context / candidate-cache logic of csrc/ft8_compat.c under tests/host_tsan."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _libasan():
    p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("gcc has no libasan here")
    return os.path.realpath(p)


def test_oracle_under_asan_ubsan():
    """the whole CPU oracle test file against the sanitizer build of the oracle (python preloads libasan)"""
    asan = _libasan()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, **SAN_ENV, LD_PRELOAD=asan, FT8O_LIB=os.path.join(ROOT, "oracle", "libft8oracle_asan.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q",
                          "-p", "no:cacheprovider"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_compat_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "compat_asan")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-ffp-contract=off", "-Wall", "-Wextra",
                           "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host_asan", "compat_asan_main.c"),
                           os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_compat.c"),
                           os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_pack.c"),
                           "-lpthread", "-lm", "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], env=dict(os.environ, **SAN_ENV), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "compat_asan ok" in out.stdout, (out.stdout + out.stderr)[-3000:]


# ---- ThreadSanitizer: the host concurrency of the library (CPU only) -----------------------------------------------------
TSAN_ENV = {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0:exitcode=66"}


def _tsan_usable(tmp_path):
    """gcc's libtsan refuses some kernels' address-space layouts: probe with a trivial program, skip with the reason"""
    src = tmp_path / "probe.c"
    src.write_text("int main(void) { return 0; }\n")
    exe = str(tmp_path / "probe")
    if subprocess.run(["gcc", "-fsanitize=thread", str(src), "-o", exe], capture_output=True).returncode != 0:
        pytest.skip("gcc has no libtsan here")
    out = subprocess.run([exe], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("ThreadSanitizer cannot run in this container: " + (out.stderr or "")[-200:])


def test_shard_pool_under_tsan(tmp_path):
    """csrc/shard_pool.h (the persistent host workers of ft8gpu_decode_batch_multi[_dev]) under run_shards' calling
    pattern: N posters x M shards with stack latches, failing and empty shards"""
    _tsan_usable(tmp_path)
    exe = str(tmp_path / "shard_pool_stress")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-Wall", "-Wextra",
                           "-I" + os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_tsan", "shard_pool_stress.cpp"), "-lpthread", "-o", exe])
    for argv in (["6", "400", "8"], ["12", "150", "3"], ["2", "600", "16"]):
        out = subprocess.run([exe] + argv, env=dict(os.environ, **TSAN_ENV), capture_output=True, text=True, timeout=600)
        tail = (out.stdout + out.stderr)[-3000:]
        assert out.returncode == 0 and "shard_pool_stress ok" in out.stdout and "ThreadSanitizer" not in tail, tail


def _build_compat_tsan(tmp_path, compat_source, name):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu17", "-fsanitize=thread", "-fno-omit-frame-pointer", "-ffp-contract=off",
                           "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host_tsan", "compat_tsan_main.c"), compat_source,
                           os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_pack.c"), "-lpthread", "-lm", "-o", exe])
    return exe


def test_compat_global_context_and_candidate_cache_under_tsan(tmp_path):
    """csrc/ft8_compat.c: six threads through initFFTW / freeFFTW / ft8_subsystem / ft8_find_sync / ft8_decode at once
    against an unsynchronised stand-in of the GPU half -- no race, and every answer the pure function of its inputs
    (a stale remembered candidate list would be a wrong value).  A mutant of the same file with ft8_decode's lock
    removed must be CAUGHT: the detector sees what it claims to see."""
    _tsan_usable(tmp_path)
    compat = os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_compat.c")
    exe = _build_compat_tsan(tmp_path, compat, "compat_tsan")
    out = subprocess.run([exe], env=dict(os.environ, **TSAN_ENV), capture_output=True, text=True, timeout=600)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0 and "compat_tsan ok" in out.stdout and "ThreadSanitizer" not in tail, tail
    src = open(compat).read()
    a = src.index("bool ft8_decode(")
    body = src[a:].replace("    pthread_mutex_lock(&g_lock);", "    /* mutant: lock removed */", 1)
    body = body.replace("    pthread_mutex_unlock(&g_lock);\n    return ok;", "    return ok;", 1)
    assert body != src[a:]
    mutant = tmp_path / "ft8_compat_mutant.c"
    mutant.write_text((src[:a] + body).replace('"../../include/', '"' + os.path.join(ROOT, "include") + "/"))
    exe = _build_compat_tsan(tmp_path, str(mutant), "compat_tsan_mutant")
    out = subprocess.run([exe], env=dict(os.environ, **TSAN_ENV), capture_output=True, text=True, timeout=600)
    assert "WARNING: ThreadSanitizer: data race" in out.stderr, "the unlocked mutant went unnoticed"


def test_candidate_cache_sees_rewrites_an_additive_checksum_misses(tmp_path):
    """csrc/ft8_compat.c, ft8_lib level (rtlsdr_ft8d.c:1450 -> :1476): between ft8_find_sync and ft8_decode the caller
    rewrites the waterfall in place so that a sum (same-lane byte swap) or a sum AND sum of sums (+d, -2d, +d over three
    words) of the 64-bit words is unchanged.  The remembered list must not answer: one-candidate launch, the new bytes'
    status.  A mutant with the round-5 additive checksum in place of the multiply-mix hash must FAIL the same harness."""
    compat = os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_compat.c")

    def build(source, name):
        exe = str(tmp_path / name)
        subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu17", "-ffp-contract=off", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "host_tsan", "compat_tsan_main.c"), source,
                               os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_pack.c"), "-lpthread", "-lm", "-o", exe])
        return exe

    out = subprocess.run([build(compat, "compat_cache"), "cache"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "compat_cache ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    src = open(compat).read()
    a = src.index("static uint64_t waterfall_checksum(const uint8_t *mag) {")
    b = src.index("static int waterfall_supported(")
    additive = ("static uint64_t waterfall_checksum(const uint8_t *mag) {\n    uint64_t a = 0x9E3779B97F4A7C15ull, b = 0;\n"
                "    for (int i = 0; i < FT8GPU_MAG_ARRAY / 8; i++) { uint64_t w; memcpy(&w, mag + 8 * i, 8); a += w; b += a; }\n"
                "    (void)fmix64; (void)rotl64;\n    return a ^ (b << 1);\n}\n\n")
    mutant = tmp_path / "ft8_compat_additive.c"
    mutant.write_text((src[:a] + additive + src[b:]).replace('"../../include/', '"' + os.path.join(ROOT, "include") + "/"))
    out = subprocess.run([build(str(mutant), "compat_cache_additive"), "cache"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "answered from the remembered list" in out.stdout, "the additive-checksum mutant went unnoticed"
