"""The generated LDS layout of the LDPC kernel (csrc/ldpc_lds_layout.h) against the bank-conflict model that produced it
(tools/ldpc_lds_layout.py, CPU only): the header is a valid assignment, it costs what its own comment says, and it beats the
matrix-order layout by the margin DESIGN.md quotes (139 -> 66 extra LDS cycles per BP iteration)."""
import importlib.util
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("ldpc_lds_layout", os.path.join(ROOT, "tools", "ldpc_lds_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _array(text, name):
    m = re.search(name + r"(?:\[\d+\])+\s*=\s*\{(.*?)\};", text, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))                     # the arrays carry comments with numbers in them
    return [int(x) for x in re.findall(r"\d+", body)]


def test_generated_layout_is_valid_and_costs_what_it_claims():
    T = _tool()
    text = open(os.path.join(ROOT, "rtlsdr_ft8d_amd", "csrc", "ldpc_lds_layout.h")).read()
    rho = _array(text, "kLdsRowPos")
    own6, own7 = _array(text, "kOwn6Row"), _array(text, "kOwn7Row")
    var_of = np.array(_array(text, "kVarOf")).reshape(3, 64)
    assert sorted(rho) == list(range(84)) and rho[83] == 83
    rows6 = [m for m in own6 if m != 255]
    rows7 = [m for m in own7 if m != 255]
    assert sorted(rows6) == T.rows6 and sorted(rows7) == T.rows7          # every row owned exactly once, by a lane of the right kind
    # the kernel keeps variable n on lane n mod 64, slot n / 64 (asserted in decode_tables_init as well)
    expect = np.array([[l + 64 * r if l + 64 * r < 174 else 255 for l in range(64)] for r in range(3)])
    assert np.array_equal(var_of, expect)
    slots = np.where(var_of == 255, -1, var_of).astype(np.int64)
    lanes6 = [own6.index(m) for m in T.rows6]
    lanes7 = [own7.index(m) for m in T.rows7]
    scattered, wide = 2 * T.scattered_cost(rho, slots), T.wide_cost(rho, lanes6, lanes7)
    claimed = re.search(r"conflicts: (\d+) \+ (\d+) extra LDS cycles", text)
    assert (scattered, wide) == (int(claimed.group(1)), int(claimed.group(2)))
    # matrix order, rows owned in matrix order: what round 3 shipped
    ident = list(range(64))
    base = (2 * T.scattered_cost(list(range(84)), slots), T.wide_cost(list(range(84)), ident, ident))
    assert base == (100, 39)
    assert scattered + wide <= 70 < sum(base)
