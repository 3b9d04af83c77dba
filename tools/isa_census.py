#!/usr/bin/env python3
"""Instruction census of the LDPC kernel's BP loop, from the compiler's own assembly (no GPU needed).

  python tools/isa_census.py [--kernel ft8_decode_kernelILb0ELi3] [--json profiles/r03_isa_census.json] [-v]

Compiles csrc/decode.hip for gfx950 with the product's flags (--cuda-device-only -S), takes the pipeline form of
the kernel (template <false, 3>), splits its BP loop (the depth-1 loop with the most instructions) into basic
blocks and classifies every instruction:
  valu      single-rate VALU (v_add/mul/fma_f32, integer, v_cndmask, v_cmp, DPP moves ...)
  valu_pk   packed f32 (v_pk_add/mul/fma_f32): two results, but about 1.7 issue slots on gfx950
  trans     transcendental pipe (v_rcp_f32 ...)
  salu      scalar ALU / branches / waits (s_*), of which s_nop and s_waitcnt are listed separately
  lds       ds_read / ds_write
  vmem      global / buffer / flat / scalar memory
The loop contains two complete instruction streams behind one wave-uniform branch per iteration (decode.hip,
first_half): the fast stream (packed rcp/fma divisions, no v_div_scale) and the IEEE stream (the guard's fallback,
v_div_scale/v_div_fmas/v_div_fixup).  Blocks are attributed by that marker; blocks with neither division belong to
both (hard decision, parity screen, row products, guard).  The census of an iteration on the fast path is the sum of
the shared blocks and the fast blocks that every iteration executes; rarely executed blocks (exact parity check,
exact guard key, exits) are listed but left out of the per-iteration sum (flag `rare`, decided by name below).
The cycle model at the end turns the census into a predicted time per launch, to be compared with the measured one."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math"]

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_exp_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "s_load", "s_buffer", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def assemble(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", src, "-o", out],
                              stderr=subprocess.DEVNULL, cwd=CSRC)
        with open(out) as f:
            return f.read().splitlines()


def function_body(lines, needle):
    start = None
    for i, ln in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(needle), ln):
            start = i
        elif start is not None and ln.strip().startswith(".Lfunc_end"):
            return lines[start:i]
    raise SystemExit(f"kernel {needle} not found in the assembly")


def blocks_of(body):
    """[(label, loop_header or None, [ops], [full instruction text])]"""
    out, cur = [], ["entry", None, [], []]
    for ln in body[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln) or re.match(r"^; (%bb\.\d+):\s*(;.*)?$", ln)   # labels and fall-through blocks
        if m:
            out.append(tuple(cur))
            hdr = re.search(r"Header=(BB\d+_\d+)", ln)
            me = m.group(1)[2:] if m.group(1).startswith(".L") else None
            inner = "This Inner Loop Header" in ln or "This Loop Header" in ln
            cur = [m.group(1), me if inner else (hdr.group(1) if hdr else None), [], []]
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        cur[2].append(t.split()[0])
        cur[3].append(t)
    out.append(tuple(cur))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="ft8_decode_kernelILb0ELi3")
    ap.add_argument("--src", default=os.path.join(CSRC, "decode.hip"))
    ap.add_argument("--json")
    ap.add_argument("--pk-cost", type=float, default=1.75, help="issue slots of a packed f32 instruction: v_mul_f32 54.6 vs v_pk_mul_f32 31.3 G wave-instr/s (tools/ubench/valu_rate.hip on the box)")
    ap.add_argument("--trans-cost", type=float, default=2.86, help="issue slots of v_rcp_f32: 19.1 G wave-instr/s on the same run")
    ap.add_argument("--clock-ghz", type=float, default=2.1, help="measured shader clock during the kernel (bench.py gpu_clock)")
    ap.add_argument("--iters-per-frame", type=float, default=None,
                    help="BP iterations entered per frame (sum over candidates; tools/iters_probe.py); enables the time prediction")
    ap.add_argument("-v", action="store_true")
    args = ap.parse_args()

    body = function_body(assemble(args.src), args.kernel)
    blocks = blocks_of(body)
    # the BP loop: the loop header owning the most instructions
    per_loop = {}
    for lab, hdr, ops, _ in blocks:
        if hdr:
            per_loop[hdr] = per_loop.get(hdr, 0) + len(ops)
    loop = max(per_loop, key=per_loop.get)
    # straight-line cost outside the loop: everything before the loop header (candidate fetch, LLR, normalisation,
    # table loads) and the part of the epilogue every candidate runs (status record; CRC / unpack77 only for codewords)
    first_loop = next(i for i, b in enumerate(blocks) if b[1] == loop)
    last_loop = max(i for i, b in enumerate(blocks) if b[1] == loop)
    def count(bs):
        c = {}
        for _, _, ops, _ in bs:
            for op in ops:
                k = classify(op)
                c[k] = c.get(k, 0) + 1
        return c
    prologue = count(blocks[:first_loop])
    epilogue_all = count(blocks[last_loop + 1:])
    rows = []
    for lab, hdr, ops, text in blocks:
        if hdr != loop:
            continue
        c = {}
        for op in ops:
            k = classify(op)
            c[k] = c.get(k, 0) + 1
        ieee = any(op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")) for op in ops)
        fast = (not ieee) and any(op == "v_rcp_f32_e32" or op.startswith("v_rcp_f32") for op in ops)
        hist = {}
        for op in ops:
            if op.startswith("v_"):
                hist[op] = hist.get(op, 0) + 1
        rows.append({"block": lab, "n": len(ops), "stream": "ieee" if ieee else ("fast" if fast else "shared"), **c, "valu_opcodes": hist,
                     "popcnt": sum(op.startswith(("v_bcnt", "s_bcnt1")) for op in ops),
                     "ballot_cmp": sum(op.startswith("v_cmp") for op in ops)})
    keys = ("valu", "valu_pk", "trans", "salu", "s_nop", "s_waitcnt", "lds", "vmem")
    # The compiler lays the loop out as [ieee atanh][fast atanh][hard decision + screen, copy 1][ieee tanh]
    # [hard decision + screen, copy 2][fast tanh][row products, guard].  A block without a division belongs to the
    # stream of the next division block in layout order; the blocks behind the last one are common to both streams.
    # (Since round 4 the compiler rotates the loop and puts the row products and the guard at its TOP: the blocks in front
    # of the first division block are common too -- whichever rotation is chosen, an iteration is
    # [atanh][hard decision + screen][tanh] per stream plus one common part at either end.)
    first_div = next((i for i, r in enumerate(rows) if r["stream"] in ("fast", "ieee")), len(rows))
    nxt = None
    for i in range(len(rows) - 1, -1, -1):
        r = rows[i]
        if r["stream"] in ("fast", "ieee"):
            nxt = r["stream"]
        elif nxt is not None and r["n"] > 0 and i > first_div:
            r["stream"] = nxt + "*"                       # attributed copy
        else:
            r["stream"] = "common"
    # rarely executed: the exact per-row parity check (loads the row masks: the only vmem in the loop; taken when
    # the scalar group screen passes, about one iteration in eight) and the exact guard key (integer minima; taken
    # when a row product is zero or below 2^-59: the first two iterations)
    for r, (lab, hdr, ops, text) in zip(rows, [b for b in blocks if b[1] == loop]):
        only_moves = len(ops) > 0 and all(op.startswith("v_mov") for op in ops)          # zero-initialisation of the messages: iteration 0 only
        r["rare"] = bool(r.get("vmem", 0) > 0 or sum(op.startswith("v_min_u32") for op in ops) >= 4 or only_moves)
    if args.v:
        print(f"{'block':12s} {'stream':8s} " + " ".join(f"{k:>9s}" for k in keys) + "  rare")
        for r in rows:
            print(f"{r['block']:12s} {r['stream']:8s} " + " ".join(f"{r.get(k, 0):9d}" for k in keys) + ("  rare" if r["rare"] else ""))

    def total(streams, rare=False):
        t = {k: 0 for k in keys}
        for r in rows:
            if r["stream"] in streams and r["rare"] == rare:
                for k in keys:
                    t[k] += r.get(k, 0)
        return t

    fast = total(("fast", "fast*"))
    common = total(("common",))
    rare = total(("fast", "fast*", "common"), rare=True)
    shared_all = common
    per_iter = {k: fast[k] + common[k] for k in keys}
    opcodes = {}
    for r in rows:
        if r["stream"] in ("fast", "fast*", "common") and not r["rare"]:
            for op, k in r["valu_opcodes"].items():
                opcodes[op] = opcodes.get(op, 0) + k
    costs = {"valu": 1.0, "valu_pk": args.pk_cost, "trans": args.trans_cost}
    slots = sum(per_iter[k] * c for k, c in costs.items())
    out = {"kernel": args.kernel, "loop_header": loop, "blocks": rows,
           "fast_stream": fast, "ieee_stream": total(("ieee", "ieee*")), "common_blocks": common, "rarely_executed_on_the_fast_path": rare,
           "per_iteration_fast_path": per_iter,
           "per_iteration_valu_opcodes": dict(sorted(opcodes.items(), key=lambda kv: -kv[1])),
           "prologue_straight_line": prologue,
           "epilogue_all_paths_static": epilogue_all,
           "prologue_valu_issue_slots": round(prologue.get("valu", 0) + args.pk_cost * prologue.get("valu_pk", 0) + args.trans_cost * prologue.get("trans", 0), 1),
           "valu_issue_slots_per_iteration": round(slots, 1),
           "slot_costs": {"valu": 1.0, "valu_pk": args.pk_cost, "trans": round(args.trans_cost, 3),
                          "note": "one slot = one v_mul_f32 (wave64); SALU, LDS and waits issue from other ports; three-operand VOP3 "
                                  "forms (v_fma_f32 1.10, v_bfi_b32 1.45 on the box) are counted as 1"}}
    if args.iters_per_frame:
        # cycle model: VALU issue is the binding port.  One slot occupies a SIMD's VALU for `cycles_per_slot` cycles
        # (wave64 on a SIMD-32 pipe: 2 passes; measured ~4 on gfx950 for dependent f32 chains interleaved over 8 waves:
        # tools/ubench/valu_rate.hip); 1024 SIMDs.
        for cps in (2.0, 4.0):
            ms = args.iters_per_frame * slots * cps / (1024 * args.clock_ghz * 1e9) * 1e3
            out[f"predicted_ms_per_frame_at_{cps:g}_cycles_per_slot"] = ms
    print(json.dumps({k: v for k, v in out.items() if k != "blocks"}, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
