"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the host C code (CPU box only: GPU ASan / XNACK are
not available on the GPU pool).  SURVEY.md section 4 notes the reference only hints at ASan (Makefile:7).
  * the oracle (oracle/ft8_oracle.c, `make -C oracle asan`) under its own CPU test file;
  * csrc/ft8_compat.c (pack77 / encode / file readers / formatter / drop-in shim) under tests/host_asan."""
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
                           "-lpthread", "-lm", "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], env=dict(os.environ, **SAN_ENV), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "compat_asan ok" in out.stdout, (out.stdout + out.stderr)[-3000:]
