#!/usr/bin/env python3
"""How many non-converging BP candidates reach an EXACT state recurrence before iteration 20?  (CPU only.)

  python tools/bp_recurrence.py [--frames 384] [--config 2|4] > profiles/r04_bp_recurrence.json

VERDICT r03 item 1: 71 % of the bench batch's candidates run all 20 iterations of ft8_lib's bp_decode (call site
rtlsdr_ft8d.c:1476) and fail; if the 522-float message state of such a candidate ever equals, bit for bit, the state
of an earlier iteration, nothing new can happen afterwards and the pipeline form of the LDPC kernel could leave the
loop there without changing a single output byte.  Kill criterion: recurrences must remove >= 15 % of all BP
iterations, otherwise the histogram is recorded and the kernel stays as it is.

Frames: the bench workloads' recipe on the host (tests/synth_util.make_frame: complex AWGN + plain-CPFSK signals,
peak-normalised), waterfall / ft8_find_sync / LLR extraction by the CPU oracle, then tools/bp_recurrence.c (the
oracle's float32 operation order with the state history kept).  Iteration count and success flag of every
candidate are cross-checked against the oracle's own ft8o_bp_decode."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TRACK_DTYPE = np.dtype([(k, "<i4") for k in ("iters", "min_errors", "state_rec_k", "state_rec_p", "hard_rec_k", "hard_rec_p",
                                              "errors_last", "sat_edges_last", "last_flip_k", "errors_min_k")])


def build_helper():
    so = os.path.join(ROOT, "tools", "ubench", "bp_recurrence.so")
    src = os.path.join(ROOT, "tools", "bp_recurrence.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-std=gnu17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                               "-o", so, src, "-lm"])
    L = C.CDLL(so)
    L.bp_track_many.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=384)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20260402)
    args = ap.parse_args()
    import oracle_lib
    import synth_util
    cfg = {2: dict(nsig=20, snr=(-18.0, 0.0), maxc=120), 4: dict(nsig=60, snr=(-24.0, -14.0), maxc=480)}[args.config]
    L = build_helper()
    enc = synth_util.oracle_encode_fn(oracle_lib)
    t0 = time.time()
    llrs = []
    for f in range(args.frames):
        iq, _ = synth_util.make_frame(args.seed + f, cfg["nsig"], enc, snr_range=cfg["snr"])
        mag = oracle_lib.waterfall(iq[0], iq[1])
        for c in oracle_lib.find_sync(mag, max_candidates=cfg["maxc"], min_score=10):
            llrs.append(oracle_lib.llr(mag, c))
    llr = np.ascontiguousarray(np.stack(llrs), np.float32)
    n = llr.shape[0]
    tr = np.zeros(n, TRACK_DTYPE)
    L.bp_track_many(llr.ctypes.data, n, args.iters, tr.ctypes.data)
    # cross-check against the oracle's own loop (every candidate)
    mismatch = 0
    for i in range(n):
        _, ok, it = oracle_lib.bp_decode(llr[i], args.iters)
        mismatch += int(ok != tr["min_errors"][i] or it != tr["iters"][i])
    conv = tr["min_errors"] == 0
    full = (~conv) & (tr["iters"] == args.iters)
    # iterations whose message update the product kernel runs: a candidate that ends at hard decision k (codeword, or
    # all-zero word) has run k updates; one that never converges runs max_iters - 1 (the last update is dead code)
    updates = np.where(tr["iters"] >= args.iters, args.iters - 1, tr["iters"]).astype(np.int64)
    rec = full & (tr["state_rec_k"] >= 0)
    saved = np.where(rec, np.maximum(args.iters - 1 - tr["state_rec_k"], 0), 0).astype(np.int64)
    hard_rec = full & (tr["hard_rec_k"] >= 0)

    def hist(a, lo, hi):
        return {str(k): int((a == k).sum()) for k in range(lo, hi + 1) if (a == k).any()}

    out = {
        "what": "exact recurrence of the 522-float BP message state (period <= 8) among candidates of host-synthesised bench-recipe frames; CPU, float32, -ffp-contract=off",
        "command": "python tools/bp_recurrence.py " + " ".join(sys.argv[1:]),
        "config": args.config, "frames": args.frames, "signals_per_frame": cfg["nsig"], "snr_db": cfg["snr"], "max_candidates": cfg["maxc"],
        "ldpc_iters": args.iters, "candidates": int(n), "oracle_cross_check_mismatches": int(mismatch),
        "converged": int(conv.sum()), "converged_frac": round(float(conv.mean()), 4),
        "ran_all_iterations_and_failed": int(full.sum()), "ran_all_iterations_and_failed_frac": round(float(full.mean()), 4),
        "exit_iteration_hist_of_converged": hist(tr["iters"][conv], 0, args.iters),
        "message_updates_total": int(updates.sum()),
        "message_updates_of_non_converging_frac": round(float(updates[full].sum() / max(1, updates.sum())), 4),
        "state_recurrences": int(rec.sum()),
        "state_recurrence_first_k_hist": hist(tr["state_rec_k"][rec], 0, args.iters),
        "state_recurrence_period_hist": hist(tr["state_rec_p"][rec], 1, 8),
        "message_updates_an_exact_exit_would_save": int(saved.sum()),
        "message_updates_saved_frac": round(float(saved.sum() / max(1, updates.sum())), 6),
        "kill_criterion_frac": 0.15,
        "orientation_only": {
            "hard_decision_repeats_k-1_or_k-2_among_non_converging": int(hard_rec.sum()),
            "hard_decision_first_repeat_k_hist": hist(tr["hard_rec_k"][hard_rec], 0, args.iters),
            "last_iteration_whose_hard_decision_changed_hist": hist(tr["last_flip_k"][full], 0, args.iters),
            "failed_checks_of_last_hard_decision_quartiles": [int(x) for x in np.percentile(tr["errors_last"][full], [0, 25, 50, 75, 100])] if full.any() else None,
            "saturated_toc_edges_of_522_in_last_update_quartiles": [int(x) for x in np.percentile(tr["sat_edges_last"][full], [0, 25, 50, 75, 100])] if full.any() else None,
            "iteration_of_minimum_error_count_hist": hist(tr["errors_min_k"][full], 0, args.iters),
        },
        "seconds": round(time.time() - t0, 1),
    }
    out["verdict"] = ("exact early exit pays: implement" if out["message_updates_saved_frac"] >= 0.15
                      else "below the kill criterion: the kernel keeps running every iteration")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
