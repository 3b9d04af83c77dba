#!/usr/bin/env python3
"""Search for the LDS layout of the LDPC kernel's check-row tile that minimises bank conflicts (CPU only; simulated annealing).

  python tools/ldpc_lds_layout.py [seed] [iterations] [--write]      # --write: emit rtlsdr_ft8d_amd/csrc/ldpc_lds_layout.h

The BP kernel (csrc/decode.hip) keeps the 83 check rows of the (174,91) code in a per-wave LDS tile: row at position P holds its
members 0..3 in float4 P of a LOW plane and members 4..6 in float4 P of a HIGH plane.  Variable lanes scatter / gather one float
per edge (nine ds_write_b32 + nine ds_read_b32 per iteration, lanes 0-31 and 32-63 served separately, bank = dword index mod 32);
row owners read and write whole float4s.  Which position a check row gets, and which lane owns which row, is free: any assignment
computes the same values.  With rows stored in matrix order the eighteen scattered accesses lose 100 LDS cycles per iteration to bank
conflicts and the wide accesses 39, on top of 68 conflict-free cycles -- and the round-4 counters put the LDS at 83 % busy in this kernel
(profiles/pmc_counters.json: SQ_LDS_IDX_ACTIVE per CU-cycle), 62 % of it conflicts.  Cost function: extra LDS cycles per iteration by the
bank rules of MI355X_MICROARCH.md (section LDS); moves: swap two row positions, swap two lanes' 6-member rows, swap two lanes' 7-member rows."""
import re, sys, random, json
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, 'rtlsdr_ft8d_amd', 'csrc', 'ft8_tables.h')).read()
def table(name):
    m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", src, re.S)
    return [int(x) for x in re.findall(r"-?\d+", m.group(1))]
Nm = np.array(table("kFT8_Nm")).reshape(83, 7)
Mn = np.array(table("kFT8_Mn")).reshape(174, 3)
NR = np.array(table("kFT8_Num_rows"))
assert NR.shape[0] == 83
R = 84
# edges per (r, e): lane -> (m, pos) or None
edge = {}
for r in range(3):
    for e in range(3):
        lst = []
        for l in range(64):
            n = l + 64 * r
            if n >= 174: lst.append(None); continue
            m = Mn[n][e] - 1
            pos = list(Nm[m][:NR[m]] - 1).index(n)
            lst.append((m, pos))
        edge[(r, e)] = lst
order = [(0,1),(0,2),(1,1),(1,2),(0,0),(1,0),(2,1),(2,2),(2,0)]
rows6 = [m for m in range(83) if NR[m] == 6]
rows7 = [m for m in range(83) if NR[m] == 7]
B128_READ_GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
                    list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
B128_WRITE_GROUPS = [list(range(8*i, 8*i+8)) for i in range(8)]
B64_WRITE_GROUPS = [list(range(16*i, 16*i+16)) for i in range(4)]

def fidx(rho, m, pos):
    return 4 * rho[m] + pos if pos < 4 else 4 * R + 4 * rho[m] + pos - 4

def cost(rho, own6, own7, detail=False):
    c32 = 0
    for (r, e) in order:
        lst = edge[(r, e)]
        for g in (range(0, 32), range(32, 64)):
            cnt = {}
            for l in g:
                if lst[l] is None: continue
                a = fidx(rho, *lst[l])
                cnt.setdefault(a % 32, set()).add(a)
            mx = max((len(v) for v in cnt.values()), default=1)
            c32 += mx - 1
    c32 *= 2                      # scatter + gather
    c128 = 0
    def grp(groups, lanes_to_f4, mod):
        t = 0
        for g in groups:
            cnt = {}
            for l in g:
                f = lanes_to_f4.get(l)
                if f is None: continue
                cnt.setdefault(f % mod, set()).add(f)
            t += max((len(v) for v in cnt.values()), default=1) - 1
        return t
    lo6 = {l: rho[m] for l, m in own6.items()}; hi6 = {l: R + rho[m] for l, m in own6.items()}
    lo7 = {l: rho[m] for l, m in own7.items()}; hi7 = {l: R + rho[m] for l, m in own7.items()}
    c128 += grp(B128_READ_GROUPS, lo6, 16) + grp(B128_READ_GROUPS, hi6, 16) + grp(B128_WRITE_GROUPS, lo6, 8)
    c128 += grp(B64_WRITE_GROUPS, {l: 2 * f for l, f in hi6.items()}, 16)          # b64: 8-byte units, 16 lanes x 8 B = 32 banks -> distinct mod 16
    c128 += grp(B128_READ_GROUPS, lo7, 16) + grp(B128_READ_GROUPS, hi7, 16) + grp(B128_WRITE_GROUPS, lo7, 8) + grp(B128_WRITE_GROUPS, hi7, 8)
    return (c32, c128) if detail else c32 + c128

rho0 = list(range(84))
own6_0 = {l: rows6[l] for l in range(len(rows6))}
own7_0 = {l: rows7[l] for l in range(len(rows7))}
print("current layout: extra cycles per iteration (b32 scatter+gather, wide)", cost(rho0, own6_0, own7_0, True), "base cycles", 18 * 2 + 8 * 4)
args = [a for a in sys.argv[1:] if not a.startswith('--')]
random.seed(int(args[0]) if len(args) > 0 else 11)
best = None
rho = rho0[:]; lanes6 = list(range(64)); lanes7 = list(range(64))
head = rho[:83]; random.shuffle(head); rho = head + [83]
def owners():
    return {lanes6[i]: rows6[i] for i in range(len(rows6))}, {lanes7[i]: rows7[i] for i in range(len(rows7))}
o6, o7 = owners()
cur = cost(rho, o6, o7)
T = 3.0
iters = int(args[1]) if len(args) > 1 else 160000
import math
for it in range(iters):
    kind = random.random()
    if kind < 0.6:
        i, j = random.sample(range(83), 2); rho[i], rho[j] = rho[j], rho[i]
        undo = lambda: rho.__setitem__(i, rho[j]) or rho.__setitem__(j, tmp)
        tmp = None
    elif kind < 0.8:
        i, j = random.sample(range(64), 2); lanes6[i], lanes6[j] = lanes6[j], lanes6[i]
    else:
        i, j = random.sample(range(64), 2); lanes7[i], lanes7[j] = lanes7[j], lanes7[i]
    o6, o7 = owners()
    new = cost(rho, o6, o7)
    if new <= cur or random.random() < math.exp((cur - new) / T):
        cur = new
        if best is None or cur < best[0]:
            best = (cur, rho[:], lanes6[:], lanes7[:])
    else:
        if kind < 0.6: rho[i], rho[j] = rho[j], rho[i]
        elif kind < 0.8: lanes6[i], lanes6[j] = lanes6[j], lanes6[i]
        else: lanes7[i], lanes7[j] = lanes7[j], lanes7[i]
    T = max(0.05, T * 0.99993)
    if it % 10000 == 0: print(it, cur, best[0], round(T, 3), flush=True)
o6 = {best[2][i]: rows6[i] for i in range(len(rows6))}; o7 = {best[3][i]: rows7[i] for i in range(len(rows7))}
print("best", best[0], cost(best[1], o6, o7, True))
print("check rows by position:", best[1])
if "--write" in sys.argv:
    own6_row = [255] * 64
    own7_row = [255] * 64
    for l, m in o6.items(): own6_row[l] = m
    for l, m in o7.items(): own7_row[l] = m
    c32, c128 = cost(best[1], o6, o7, True)
    out = os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ldpc_lds_layout.h")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/ldpc_lds_layout.py (seed %s, %d iterations) -- do not edit.\n" % (args[0] if args else "11", iters))
        f.write("// LDS layout of the BP kernel's check-row tile: position of every check row and the row each lane owns, chosen to minimise\n")
        f.write("// bank conflicts: %d + %d extra LDS cycles per iteration (scattered 4-byte accesses + float4 accesses) against 100 + 39 with the\n" % (c32, c128))
        f.write("// rows in matrix order; 68 cycles are conflict-free work.  Any assignment computes the same values.\n#pragma once\n#include <stdint.h>\n")
        f.write("static const uint8_t kLdsRowPos[84] = { %s };   // check row m -> row position (83 = the spare row)\n" % ", ".join(map(str, best[1])))
        f.write("static const uint8_t kOwn6Row[64] = { %s };   // lane -> the 6-member check row it multiplies (255: none)\n" % ", ".join(map(str, own6_row)))
        f.write("static const uint8_t kOwn7Row[64] = { %s };   // lane -> the 7-member check row it multiplies (255: none)\n" % ", ".join(map(str, own7_row)))
    print("wrote", out)
