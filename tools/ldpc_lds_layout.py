#!/usr/bin/env python3
"""Search for the LDS layout of the LDPC kernel's check-row tile that minimises bank conflicts (CPU only; simulated annealing).

  python tools/ldpc_lds_layout.py [seed] [iterations] [--fixed-variables] [--write]
                                                             # --write: emit rtlsdr_ft8d_amd/csrc/ldpc_lds_layout.h

The BP kernel (csrc/decode.hip) keeps the 83 check rows of the (174,91) code in a per-wave LDS tile: the row at position P holds
its members 0..3 in float4 P of a LOW plane and members 4..6 in float4 P of a HIGH plane.  Variable lanes scatter / gather one
float per edge (nine ds_write_b32 + nine ds_read_b32 per iteration, lanes 0-31 and 32-63 served separately, bank = dword index mod
32); row owners read and write whole float4s.  Three things are free, because any choice computes the same values:
  * which tile position a check row gets,
  * which lane multiplies which row (one 6-member and one 7-member row per lane at most),
  * which three variable nodes a lane holds, and in which of its three slots (the kernel reads the map from a table; the hard
    decision is brought back to codeword order once, after the loop).  Slots 0 and 1 are full (128 variables), slot 2 holds 46.
With rows in matrix order and variable n on lane n mod 64 the eighteen scattered accesses lose 100 LDS cycles per iteration to bank
conflicts and the wide accesses 39, on top of 68 conflict-free cycles -- and the round-4 counters put the LDS index unit at 83 % busy in
this kernel (profiles/pmc_counters.json of round 3's layout: SQ_LDS_IDX_ACTIVE per CU-cycle = 211 cycles per wave-iteration, which
this model reproduces), 62 % of it conflicts.  Cost function: extra LDS cycles per iteration by the bank rules of
MI355X_MICROARCH.md (section LDS).  Moves: swap two row positions, two lanes' 6-member rows, two lanes' 7-member rows, two variable
slots.  With the variable map fixed the scattered accesses cannot go below 56 extra cycles (the member index of an edge is given by
the code, and an access with more than eight lanes on one member index mod 4 conflicts whatever the positions)."""
import math
import os
import random
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ft8_tables.h")).read()


def table(name):
    m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", src, re.S)
    return [int(x) for x in re.findall(r"-?\d+", m.group(1))]


Nm = np.array(table("kFT8_Nm")).reshape(83, 7)
Mn = np.array(table("kFT8_Mn")).reshape(174, 3)
NR = np.array(table("kFT8_Num_rows"))
R = 84
# edge e of variable n: check row and member index
E_ROW = np.zeros((174, 3), np.int64)
E_POS = np.zeros((174, 3), np.int64)
for n in range(174):
    for e in range(3):
        m = Mn[n][e] - 1
        E_ROW[n, e] = m
        E_POS[n, e] = list(Nm[m][:NR[m]] - 1).index(n)
E_OFF = np.where(E_POS < 4, E_POS, 4 * R + E_POS - 4)          # float index = 4 * position + E_OFF
rows6 = [m for m in range(83) if NR[m] == 6]
rows7 = [m for m in range(83) if NR[m] == 7]
B128_READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
                    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
B128_WRITE_GROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
B64_WRITE_GROUPS = [list(range(16 * i, 16 * i + 16)) for i in range(4)]


def scattered_cost(rho, slots):
    """extra cycles of the nine ds_write_b32 (the nine ds_read_b32 cost the same again)"""
    c = 0
    rho = np.asarray(rho)
    for r in range(3):
        v = slots[r]
        ok = v >= 0
        vv = np.where(ok, v, 0)
        for e in range(3):
            bank = (4 * rho[E_ROW[vv, e]] + E_OFF[vv, e]) % 32
            for lo in (0, 32):
                b = bank[lo:lo + 32][ok[lo:lo + 32]]
                if b.size:
                    c += int(np.bincount(b, minlength=32).max()) - 1
    return c


def wide_cost(rho, lanes6, lanes7):
    def grp(groups, f4_of_lane, mod):
        t = 0
        for g in groups:
            seen = {}
            for l in g:
                f = f4_of_lane.get(l)
                if f is not None:
                    seen.setdefault(f % mod, set()).add(f)
            t += max((len(s) for s in seen.values()), default=1) - 1
        return t
    lo6 = {lanes6[i]: rho[m] for i, m in enumerate(rows6)}
    hi6 = {l: R + p for l, p in lo6.items()}
    lo7 = {lanes7[i]: rho[m] for i, m in enumerate(rows7)}
    hi7 = {l: R + p for l, p in lo7.items()}
    c = grp(B128_READ_GROUPS, lo6, 16) + grp(B128_READ_GROUPS, hi6, 16) + grp(B128_WRITE_GROUPS, lo6, 8)
    c += grp(B64_WRITE_GROUPS, {l: 2 * f for l, f in hi6.items()}, 16)      # ds_write_b64: 16 lanes x 8 B = 32 banks
    c += grp(B128_READ_GROUPS, lo7, 16) + grp(B128_READ_GROUPS, hi7, 16) + grp(B128_WRITE_GROUPS, lo7, 8) + grp(B128_WRITE_GROUPS, hi7, 8)
    return c


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    seed = int(args[0]) if args else 11
    iters = int(args[1]) if len(args) > 1 else 300000
    move_vars = "--fixed-variables" not in sys.argv
    random.seed(seed)
    rho0 = list(range(84))
    slots0 = np.full((3, 64), -1, np.int64)
    for n in range(174):
        slots0[n // 64, n % 64] = n
    ident = list(range(64))
    print("matrix order, variable n on lane n mod 64: extra cycles per iteration (scattered x 2, wide) =",
          (2 * scattered_cost(rho0, slots0), wide_cost(rho0, ident, ident)), "on top of 68 conflict-free")
    head = rho0[:83]
    random.shuffle(head)
    rho = head + [83]                                           # position 83 stays the spare row
    slots = slots0.copy()
    lanes6, lanes7 = ident[:], ident[:]
    cs, cw = scattered_cost(rho, slots), wide_cost(rho, lanes6, lanes7)
    cur = 2 * cs + cw
    best = (cur, rho[:], slots.copy(), lanes6[:], lanes7[:], cs, cw)
    T = 3.0
    decay = (0.05 / 3.0) ** (1.0 / max(1, int(0.9 * iters)))
    for it in range(iters):
        k = random.random()
        if k < 0.35:
            i, j = random.sample(range(83), 2)
            rho[i], rho[j] = rho[j], rho[i]
            ncs, ncw = scattered_cost(rho, slots), wide_cost(rho, lanes6, lanes7)
            undo = ("rho", i, j)
        elif k < 0.45:
            i, j = random.sample(range(64), 2)
            lanes6[i], lanes6[j] = lanes6[j], lanes6[i]
            ncs, ncw = cs, wide_cost(rho, lanes6, lanes7)
            undo = ("l6", i, j)
        elif k < 0.55 or not move_vars:
            i, j = random.sample(range(64), 2)
            lanes7[i], lanes7[j] = lanes7[j], lanes7[i]
            ncs, ncw = cs, wide_cost(rho, lanes6, lanes7)
            undo = ("l7", i, j)
        else:
            while True:                                          # two slots; an empty one may only trade within slot row 2
                a = (random.randrange(3), random.randrange(64))
                b = (random.randrange(3), random.randrange(64))
                va, vb = slots[a], slots[b]
                if a == b or (va < 0 and vb < 0):
                    continue
                if (va < 0 and b[0] != 2) or (vb < 0 and a[0] != 2):
                    continue
                break
            slots[a], slots[b] = vb, va
            ncs, ncw = scattered_cost(rho, slots), cw
            undo = ("var", a, b)
        new = 2 * ncs + ncw
        if new <= cur or random.random() < math.exp((cur - new) / T):
            cur, cs, cw = new, ncs, ncw
            if cur < best[0]:
                best = (cur, rho[:], slots.copy(), lanes6[:], lanes7[:], cs, cw)
        else:
            kind, i, j = undo
            if kind == "rho":
                rho[i], rho[j] = rho[j], rho[i]
            elif kind == "l6":
                lanes6[i], lanes6[j] = lanes6[j], lanes6[i]
            elif kind == "l7":
                lanes7[i], lanes7[j] = lanes7[j], lanes7[i]
            else:
                slots[i], slots[j] = slots[j], slots[i]
        T = max(0.05, T * decay)
        if it % 20000 == 0:
            print(it, cur, best[0], round(T, 3), flush=True)
    cur, rho, slots, lanes6, lanes7, cs, cw = best
    print("best: extra cycles per iteration", cur, "= scattered x 2:", 2 * cs, "+ wide:", cw)
    assert sorted(int(v) for v in slots.ravel() if v >= 0) == list(range(174)) and (slots[:2] >= 0).all()
    if "--write" in sys.argv:
        own6_row, own7_row = [255] * 64, [255] * 64
        for i, m in enumerate(rows6):
            own6_row[lanes6[i]] = m
        for i, m in enumerate(rows7):
            own7_row[lanes7[i]] = m
        out = os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ldpc_lds_layout.h")
        with open(out, "w") as f:
            f.write("// GENERATED by tools/ldpc_lds_layout.py (seed %d, %d iterations%s) -- do not edit.\n" % (seed, iters, "" if move_vars else ", variables fixed"))
            f.write("// LDS layout of the BP kernel's check-row tile: tile position of every check row, the row each lane owns, and the variable nodes\n")
            f.write("// each lane holds, chosen to minimise bank conflicts: %d + %d extra LDS cycles per iteration (scattered 4-byte accesses + float4\n" % (2 * cs, cw))
            f.write("// accesses) against 100 + 39 with rows in matrix order and variable n on lane n mod 64; 68 cycles are conflict-free work.\n")
            f.write("// Any assignment computes the same values.\n#pragma once\n#include <stdint.h>\n")
            f.write("static const uint8_t kLdsRowPos[84] = { %s };   // check row m -> tile position (83 = the spare row)\n" % ", ".join(map(str, rho)))
            f.write("static const uint8_t kOwn6Row[64] = { %s };   // lane -> the 6-member check row it multiplies (255: none)\n" % ", ".join(map(str, own6_row)))
            f.write("static const uint8_t kOwn7Row[64] = { %s };   // lane -> the 7-member check row it multiplies (255: none)\n" % ", ".join(map(str, own7_row)))
            f.write("static const uint8_t kVarOf[3][64] = {   // [slot][lane] -> variable node (codeword bit index; 255: none, slot 2 only)\n")
            for r in range(3):
                f.write("    { %s },\n" % ", ".join(str(int(v)) if v >= 0 else "255" for v in slots[r]))
            f.write("};\n")
        print("wrote", out)


if __name__ == "__main__":
    main()
