#!/usr/bin/env python3
"""The independent numpy restatement of the ft8_lib stages (tests/ft8_spec_decode.py, written from SURVEY.md Appendix A only)
against the C oracle, at a scale the unit test cannot afford: per frame all 35 856 sync scores, the ordered candidate list at
caps 120 / 480 / 7, and for EVERY candidate the 174 normalised LLRs bit for bit, the minimum parity-error count, the iterations
entered and the 91 packed bits at 20 iterations and at one other iteration cap.  CPU only (no GPU): the oracle is what every
GPU parity test compares with, so a slip shared by the oracle and the kernels would only show here.
usage: tools/spec_vs_oracle.py [--frames 2000] [--procs 8] [--seed 1]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

_state = {}


def _init():
    import oracle_lib as O
    import ft8_spec_decode as spec
    import synth_util as S
    from rtlsdr_ft8d_amd import workload
    O.lib()
    _state.update(O=O, spec=spec, S=S, bp=spec.BP(), mixed=workload.message_pool(traffic="mixed"), enc=S.oracle_encode_fn(O))


def one_frame(job):
    seed, kind = job
    O, spec, S, bp = _state["O"], _state["spec"], _state["S"], _state["bp"]
    rng = np.random.default_rng(seed)
    nsig = int(rng.integers(0, 50))
    lo = float(rng.uniform(-24, -8))
    snr = (lo, lo + float(rng.uniform(2, 18)))
    if kind == "cq":
        iq, _ = S.make_frame(seed, nsig, _state["enc"], snr_range=snr, cq_fraction=float(rng.uniform(0.3, 1.0)))
    elif kind == "edges":
        iq, _ = S.make_frame(seed, nsig, _state["enc"], snr_range=snr, f_range=(-20.0, 1620.0), dt_range=(-1.5, 3.0))
    else:
        iq, _ = S.make_mixed_frame(seed, nsig, snr, *_state["mixed"])
    mag = O.waterfall(iq[0], iq[1])
    r = {"frames": 1, "scores_bad": 0, "lists_bad": 0, "cands": 0, "llr_bad": 0, "bp_bad": 0, "converged": 0, "iterated": 0}
    sc = spec.score_map(mag)
    r["scores_bad"] += int(not np.array_equal(sc.astype(np.int16), O.score_map(mag)))
    for cap, ms in ((120, 10), (480, 10), (7, 10)):
        mine = [list(c) for c in spec.find_sync(mag, cap, ms, scores=sc)]
        theirs = [[int(x["score"]), int(x["time_offset"]), int(x["freq_offset"]), int(x["time_sub"]), int(x["freq_sub"])] for x in O.find_sync(mag, cap, ms)]
        r["lists_bad"] += int(mine != theirs)
    cands = O.find_sync(mag)
    other = int(rng.integers(1, 20))
    for k, c in enumerate(spec.find_sync(mag, 120, 10, scores=sc)):
        r["cands"] += 1
        ll = spec.normalize_logl(spec.extract_likelihood(mag, c))
        r["llr_bad"] += int(ll.tobytes() != O.llr(mag, cands[k:k + 1]).tobytes())
        for iters in (20, other):
            errors, entered, a91 = spec.decode_candidate(bp, mag, c, iters)
            s = O.decode(mag, cands[k:k + 1], iters)
            r["bp_bad"] += int((errors, entered, a91) != (s["ldpc_errors"], s["iters"], s["a91"]))
            if iters == 20:
                r["converged"] += int(errors == 0)
                r["iterated"] += int(entered > 0)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import multiprocessing as mp
    jobs = [(args.seed * 1_000_003 + k, ("mixed", "cq", "edges")[k % 3]) for k in range(args.frames)]
    t0 = time.time()
    total = {}
    with mp.get_context("fork").Pool(args.procs, initializer=_init) as pool:
        for n, r in enumerate(pool.imap_unordered(one_frame, jobs, chunksize=4)):
            for k, v in r.items():
                total[k] = total.get(k, 0) + v
            if (n + 1) % 200 == 0:
                print(f"{n + 1} frames: {total}", flush=True)
    total.update(seconds=round(time.time() - t0, 1), seed=args.seed, kinds="mixed traffic / CQ recipe with call-call-grid / band and window edges, a third each",
                 what="numpy restatement (SURVEY Appendix A) vs C oracle: score maps, ordered candidate lists at caps 120/480/7, per candidate LLR bits and BP outcome at 20 and one other iteration cap")
    print(json.dumps(total))


if __name__ == "__main__":
    main()
