#!/usr/bin/env python3
"""Register / LDS / scratch table of every kernel of libft8gpu.so, from the compiler's own assembly (no GPU needed).

  python tools/kernel_resources.py [--json profiles/r04_kernel_resources.json]

Compiles every csrc/*.hip for gfx950 with the product's flags (--cuda-device-only -S) and reports, per kernel,
VGPRs, SGPRs, static LDS bytes, the private (scratch) segment size, the number of scratch_* instructions and the
resident waves per SIMD those numbers allow.  Exit status 1 if
  * any kernel has a scratch segment or a scratch instruction (scratch lines that are written travel to HBM: the
    round-3 LDPC kernel wrote 5.8 x the bytes of its own records that way), or
  * a DPP instruction is reachable from a v_cmpx (a VALU write of EXEC) within 5 wait states, along ANY path of the
    control flow (branches are followed): bp_math.h narrows EXEC inside inline assembly, which the compiler's hazard
    recogniser cannot see into.
tests/test_kernel_resources.py runs this on the CPU box, so neither can come back unnoticed."""
import argparse
import concurrent.futures as cf
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math"]
SKIP = {"probe_writelane.hip"}          # build probe (Makefile), not part of the library


def assemble(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", src, "-o", out],
                              stderr=subprocess.DEVNULL, cwd=CSRC)
        with open(out) as f:
            return f.read()


def demangle(names):
    if not names:
        return []
    try:
        out = subprocess.run(["c++filt", *names], capture_output=True, text=True, check=True, stdin=subprocess.DEVNULL).stdout.split("\n")
        return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0].replace("void ", "") for o in out[:len(names)]]
    except Exception:
        return names


def wait_states(line):
    """wait states an instruction line provides to what follows it (s_nop N = N + 1, anything else 1)"""
    m = re.match(r"s_nop\s+(\d+)", line)
    return int(m.group(1)) + 1 if m else 1


def cmpx_dpp_hazards(text):
    """[(line number, text)] of DPP instructions reachable from a v_cmpx within 5 wait states.  The walk follows the
    control flow, not the text: labels are transparent, s_branch continues at its target, s_cbranch_* continues both at
    its target and behind it (the EXEC-narrowing assembly of bp_math.h sits inside the BP loop, whose back edge and
    exit lead to code that is nowhere near the next lines of text); s_endpgm / s_setpc end a path."""
    items, label_at = [], {}
    for i, raw in enumerate(text.splitlines()):
        t = raw.split(";")[0].strip()
        if t.endswith(":"):
            label_at[t[:-1]] = len(items)              # a label names the next instruction
            continue
        if not t or t.startswith((".", ";")):
            continue
        if t.startswith("_"):
            continue
        items.append((i + 1, t))
    bad = set()
    for k, (_, t) in enumerate(items):
        if not t.startswith("v_cmpx"):
            continue
        stack, seen = [(k + 1, 0)], set()
        while stack:
            pos, waited = stack.pop()
            while pos < len(items) and waited < 5 and (pos, waited) not in seen:
                seen.add((pos, waited))
                ln2, t2 = items[pos]
                if "dpp" in t2 or "row_" in t2 or "quad_perm" in t2:
                    bad.add((ln2, t2))
                waited += wait_states(t2)
                m = re.match(r"(s_branch|s_cbranch_\w+)\s+(\S+)", t2)
                if m and m.group(2) in label_at:
                    if m.group(1) == "s_branch":
                        pos = label_at[m.group(2)]
                        continue
                    stack.append((label_at[m.group(2)], waited))
                elif t2.startswith(("s_endpgm", "s_setpc", "s_swappc")):
                    break
                pos += 1
    return sorted(bad)


def kernels_of(text):
    """metadata of the .amdhsa kernels of one assembly file"""
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", text)[1:]:
        g = lambda key: re.search(r"\.%s:\s*(\S+)" % key, blk)
        name = g("name").group(1)
        out.append({"symbol": name, "vgprs": int(g("vgpr_count").group(1)), "sgprs": int(g("sgpr_count").group(1)),
                    "lds_bytes": int(g("group_segment_fixed_size").group(1)),
                    "scratch_bytes_per_lane": int(g("private_segment_fixed_size").group(1)),
                    "max_workgroup": int(g("max_flat_workgroup_size").group(1))})
    return out


def waves_per_simd(k):
    """resident waves per SIMD allowed by VGPRs (512 per lane, granule 8, at most 8 waves), gfx950"""
    v = max(8, (k["vgprs"] + 7) // 8 * 8)
    return min(8, 512 // v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    srcs = sorted(s for s in glob.glob(os.path.join(CSRC, "*.hip")) if os.path.basename(s) not in SKIP)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        texts = list(ex.map(assemble, srcs))
    rows, problems = [], []
    for src, text in zip(srcs, texts):
        ks = kernels_of(text)
        for k, nice in zip(ks, demangle([k["symbol"] for k in ks])):
            k["kernel"] = nice
            k["file"] = os.path.basename(src)
            k["waves_per_simd_by_vgprs"] = waves_per_simd(k)
            rows.append(k)
            if k["scratch_bytes_per_lane"]:
                problems.append(f"{k['file']}: {nice} has a {k['scratch_bytes_per_lane']}-byte scratch segment")
        n_scratch = len(re.findall(r"^\s*scratch_(load|store)", text, re.M))
        if n_scratch:
            problems.append(f"{os.path.basename(src)}: {n_scratch} scratch_* instructions")
        for ln, t in cmpx_dpp_hazards(text):
            problems.append(f"{os.path.basename(src)}: DPP instruction within 5 wait states of a v_cmpx (asm line {ln}): {t}")
    out = {"what": "per-kernel resources of libft8gpu.so from hipcc -S (gfx950, product flags); scratch must be 0 everywhere",
           "command": "python tools/kernel_resources.py", "kernels": rows, "problems": problems}
    print(f"{'file':16s} {'kernel':44s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>8s} {'waves/SIMD':>10s}")
    for k in rows:
        print(f"{k['file']:16s} {k['kernel'][:44]:44s} {k['vgprs']:5d} {k['sgprs']:5d} {k['lds_bytes']:7d} {k['scratch_bytes_per_lane']:8d} {k['waves_per_simd_by_vgprs']:10d}")
    for p in problems:
        print("PROBLEM:", p)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
