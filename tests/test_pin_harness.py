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


# ---- the recipe must land on the right upstream: API-era detection of tools/pin_ft8_lib.sh -----------------------------------------
# Declaration-only fake trees written by this test (the three shapes ft8/decode.h has had upstream, as far as the reference's call
# sites and the later public API tell): nothing of upstream's text, no bodies.
_OLD_H = "typedef struct { int num_blocks; int num_bins; int time_osr; int freq_osr; uint8_t* mag; } waterfall_t;\n" \
         "int find_sync(const waterfall_t* power, int num_candidates, candidate_t heap[], int min_score);\n"
_ERA_H = "typedef struct { int max_blocks; int num_blocks; int num_bins; int time_osr; int freq_osr; uint8_t* mag; int block_stride;\n" \
         "  ftx_protocol_t   protocol; } waterfall_t;\n" \
         "int ft8_find_sync(const waterfall_t* power, int num_candidates, candidate_t heap[], int min_score);\n" \
         "bool ft8_decode(const waterfall_t* power, const candidate_t* cand, message_t* message, int max_iterations, decode_status_t* status);\n"
_NEW_H = "typedef struct { int max_blocks; int num_blocks; uint8_t* mag; int block_stride; ftx_protocol_t protocol; } ftx_waterfall_t;\n" \
         "int ftx_find_candidates(const ftx_waterfall_t* power, int num_candidates, ftx_candidate_t heap[], int min_score);\n" \
         "bool ftx_decode_candidate(const ftx_waterfall_t* power, const ftx_candidate_t* cand, int max_iterations, ftx_message_t* message, ftx_decode_status_t* status);\n"


def _fake_tree(d, decode_h):
    os.makedirs(os.path.join(d, "ft8"), exist_ok=True)
    for name, text in (("decode.h", decode_h), ("decode.c", "/* empty */\n"), ("pack.h", "int pack77(const char* msg, uint8_t* c77);\n"),
                       ("encode.h", "void ft8_encode(const uint8_t* payload, uint8_t* tones);\n"),
                       ("unpack.h", "int unpack77(const uint8_t* a77, char* message);\n")):
        with open(os.path.join(d, "ft8", name), "w") as f:
            f.write(text)


def _pin(up, **env):
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_ft8_lib.sh"), up], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PIN_CHECK_ERA_ONLY="1", **env))
    return p.returncode, p.stdout + p.stderr


def test_pin_recipe_refuses_a_checkout_of_the_wrong_api_era(tmp_path):
    """A maintainer with ANY ft8_lib clone gets PINNED / DIFFERENT or an instruction which revision to check out -- never a wall
    of compiler errors.  HEAD of upstream today has the ftx_* interface; clones from before FT4 support have no waterfall_t.protocol."""
    new, old, era = str(tmp_path / "new"), str(tmp_path / "old"), str(tmp_path / "era")
    _fake_tree(new, _NEW_H)
    _fake_tree(old, _OLD_H)
    _fake_tree(era, _ERA_H)
    rc, out = _pin(new)
    assert rc == 3 and "REFUSED" in out and "too-new" in out and "ftx_find_candidates" in out and "rtlsdr_ft8d.c:1450" in out
    assert "git log --reverse" in out and "not a git work tree" in out
    rc, out = _pin(old)
    assert rc == 3 and "too-old" in out and "predates FT4 support" in out and "rtlsdr_ft8d.c:1440-1448" in out
    rc, out = _pin(era)
    assert rc == 0 and "API era: ft8_find_sync / ft8_decode with waterfall_t.protocol" in out
    rc, out = _pin(str(tmp_path / "nothing"))
    assert rc == 2 and "not an ft8_lib checkout" in out


def test_pin_recipe_names_the_revision_to_check_out_from_the_clones_own_history(tmp_path):
    """with history at hand the refusal computes the window from THIS clone (pickaxe over ft8/decode.h) and prints the checkout
    command: the parent of the commit that brought the ftx_* names"""
    up = str(tmp_path / "clone")
    os.makedirs(up)
    git = lambda *a: subprocess.check_output(["git", "-C", up, "-c", "user.name=t", "-c", "user.email=t@t", *a], text=True).strip()
    git("init", "-q")
    shas = []
    for h, msg in ((_OLD_H, "C port"), (_ERA_H, "FT4 support: protocol field"), (_ERA_H + "/* later fix */\n", "fix"), (_NEW_H, "rename to ftx_*"),
                   (_NEW_H + "/* more */\n", "more")):
        _fake_tree(up, h)
        git("add", "-A")
        git("commit", "-q", "-m", msg)
        shas.append(git("rev-parse", "HEAD"))
    rc, out = _pin(up)                                       # HEAD: the later API
    assert rc == 3 and "too-new" in out
    assert f"SUGGESTED: git -C {up} checkout {shas[2]}" in out, out
    assert "FT4 support: protocol field" in out and "fix" in out
    git("checkout", "-q", shas[0])                           # an old revision checked out: same suggestion from the full history
    rc, out = _pin(up)
    assert rc == 3 and "too-old" in out and f"checkout {shas[2]}" in out
    git("checkout", "-q", shas[2])
    rc, out = _pin(up)
    assert rc == 0


def test_pin_recipe_falls_back_to_the_public_interface_when_internal_names_differ(tmp_path):
    """the three stages behind the public calls are reached through upstream's file-local functions; a revision of the right era
    that spells one differently gets the public-interface comparison plus the -D overrides to name its equivalents -- not a
    compiler error (declaration-only fake trees; PIN_PLAN_ONLY stops before anything is compiled)"""
    full, bare = str(tmp_path / "full"), str(tmp_path / "bare")
    for d in (full, bare):
        _fake_tree(d, _ERA_H)
        with open(os.path.join(d, "ft8", "ldpc.h"), "w") as f:
            f.write("void bp_decode(float codeword[], int max_iters, uint8_t plain[], int* ok);\n")
    with open(os.path.join(full, "ft8", "decode.c"), "w") as f:
        f.write("static int ft8_sync_score(const waterfall_t* wf, const candidate_t* c);\n"
                "static void ft8_extract_likelihood(const waterfall_t* wf, const candidate_t* c, float* log174);\n"
                "static void ftx_normalize_logl(float* log174);\n")
    with open(os.path.join(bare, "ft8", "decode.c"), "w") as f:
        f.write("static int get_sync_score(const waterfall_t* wf, const candidate_t* c);\n"
                "static void ft8_extract_likelihood(const waterfall_t* wf, const candidate_t* c, float* log174);\n")

    def plan(up, **env):
        p = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_ft8_lib.sh"), up], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PIN_PLAN_ONLY="1", **env))
        return p.returncode, p.stdout + p.stderr

    rc, out = plan(full)
    assert rc == 0 and "comparison depth: public interface + internals" in out and "NOTE" not in out
    rc, out = plan(bare)
    assert rc == 0 and "comparison depth: public interface only" in out
    assert "does not define: ft8_sync_score ftx_normalize_logl" in out and "-DPIN_UPSTREAM_SYNC_SCORE=<fn>" in out
    rc, out = plan(bare, PIN_EXTRA_CFLAGS="-DPIN_UPSTREAM_SYNC_SCORE=get_sync_score -DPIN_UPSTREAM_NORMALIZE=normalize")
    assert rc == 0 and "comparison depth: public interface + internals" in out
