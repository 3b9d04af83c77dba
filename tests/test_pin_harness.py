"""tools/pin_ft8_lib: the harness that will pin the oracle to kgoba/ft8_lib's own sources the day a checkout is at
hand (tools/pin_ft8_lib.sh <path>; the submodule is empty in the reference snapshot, .gitmodules:1-3).  It cannot run
against upstream here, so the harness itself is proven two ways:
  CPU : linked against the oracle under ft8_lib's names, with the internals hooks: the dump is what the committed
        per-frame digests say (tests/golden/pin_dump.json) and carries the reference's own pass condition;
  GPU : linked against libft8gpu.so's ft8_lib-level symbols (the interface the unmodified rtlsdr_ft8d.c would call):
        its dump must be the oracle's, record for record."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "tools", "pin_ft8_lib")
CFLAGS = ["-O2", "-std=gnu17", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra"]


def _inputs(tmp_path):
    subprocess.check_call([sys.executable, os.path.join(P, "make_inputs.py"), str(tmp_path)])
    return str(tmp_path / "waterfalls.bin"), str(tmp_path / "messages.txt")


def _oracle_harness(tmp_path, internals):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    exe = str(tmp_path / ("pin_oracle_int" if internals else "pin_oracle"))
    subprocess.check_call(["gcc"] + CFLAGS + (["-DPIN_INTERNALS"] if internals else []) +
                          ["-I" + os.path.join(ROOT, "include", "ft8_lib"), "-I" + os.path.join(ROOT, "oracle"), "-I" + P,
                           os.path.join(P, "pin_harness.c"), os.path.join(P, "oracle_as_ft8_lib.c"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lft8oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm", "-o", exe])
    return exe


def _frames(dump):
    """{"frame k": [lines], "tail": [pack / end lines]}"""
    out, cur = {}, None
    for ln in dump.splitlines():
        if ln.startswith("frame "):
            cur = " ".join(ln.split()[:2])
            out[cur] = []
        elif ln.startswith(("pack ", "end ")):
            cur = "tail"
            out.setdefault(cur, [])
        out[cur].append(ln)
    return out


def test_harness_on_the_oracle_backend_matches_the_committed_digests(tmp_path):
    wf, msgs = _inputs(tmp_path)
    dump = subprocess.check_output([_oracle_harness(tmp_path, True), wf, msgs], timeout=600).decode()
    fr = _frames(dump)
    # the reference's own expectations are in the dump: the self-test decodes to CQ K1JT FN20 (rtlsdr_ft8d.c:966-971),
    # pack77 / ft8_encode give the known answer (:919-923)
    assert any('text "CQ K1JT FN20"' in ln and " ok 1 " in ln for ln in fr["frame 0"])
    assert 'pack "CQ K1JT FN20QI" rc 0 payload 000000204dfcdc8a1408 tones 3140652000000001005477547106035036373140652547441342116056460065174427143140652' in fr["tail"]
    assert fr["tail"][-1] == "end frames 10"
    assert sum(ln.startswith("decode ") and " ok 1 " in ln for ln in dump.splitlines()) > 150
    # mixed-traffic frames bring the message types the CQ-only frames never show
    mixed = "\n".join(fr["frame 6"] + fr["frame 7"])
    texts = {ln.split(' text "')[1].rstrip('"') for ln in mixed.splitlines() if ' text "' in ln}
    assert sum(not t.startswith("CQ") for t in texts) >= 8 and any(" R-" in t or " R+" in t for t in texts), texts
    digests = {k: hashlib.sha256("\n".join(v).encode()).hexdigest()[:24] for k, v in fr.items()}
    path = os.path.join(ROOT, "tests", "golden", "pin_dump.json")
    if os.environ.get("FT8_REGENERATE_GOLDEN"):
        json.dump({"what": "sha256 (24 hex) of each frame's records in the dump of tools/pin_ft8_lib/pin_harness.c -DPIN_INTERNALS on the oracle "
                           "backend (tools/pin_ft8_lib/make_inputs.py inputs); regenerate with FT8_REGENERATE_GOLDEN=1", "digests": digests},
                  open(path, "w"), indent=1)
    want = json.load(open(path))["digests"]
    assert digests == want, {k: (digests.get(k), want.get(k)) for k in set(digests) | set(want) if digests.get(k) != want.get(k)}


def test_recipe_refuses_a_directory_that_is_not_ft8_lib(tmp_path):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_ft8_lib.sh"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 2 and "not an ft8_lib checkout" in out.stdout


@pytest.mark.gpu
def test_harness_on_libft8gpu_equals_the_oracle_dump(tmp_path):
    """the same harness file, public interface only, against the product's ft8_lib-level symbols: every candidate list
    (caps 7 / 120 / 480), every ft8_decode outcome and the encoder lines must be the oracle's"""
    import rtlsdr_ft8d_amd as ft8
    ft8.check_build_id()
    wf, msgs = _inputs(tmp_path)
    want = subprocess.check_output([_oracle_harness(tmp_path, False), wf, msgs], timeout=600).decode()
    exe = str(tmp_path / "pin_gpu")
    pkg = os.path.join(ROOT, "rtlsdr_ft8d_amd")
    subprocess.check_call(["gcc"] + CFLAGS + ["-I" + os.path.join(ROOT, "include", "ft8_lib"), "-I" + P, os.path.join(P, "pin_harness.c"),
                                              "-L" + pkg, "-lft8gpu", "-Wl,-rpath," + pkg, "-lm", "-o", exe])
    got = subprocess.run([exe, wf, msgs], capture_output=True, text=True, timeout=900)
    assert got.returncode == 0, got.stderr[-2000:]
    a, b = got.stdout.splitlines(), want.splitlines()
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert len(a) == len(b) and not diff, (len(a), len(b), diff[:5])
    assert sum(" ok 1 " in ln for ln in a) > 150
