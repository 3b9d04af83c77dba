#!/usr/bin/env python3
"""Mass parity run of the RX front end (rtlsdr_callback(), rtlsdr_ft8d.c:76-202 + the decoder thread's
tail zeroing and peak normalisation, :243-263): full-length raw captures (15 s at 2.4 Msps, 72 MB each) of several byte
statistics through ft8gpu_rx_decimate, every output float compared bit for bit with the sequential oracle.
usage: tools/soak_rx.py [--batches 10] [--captures 8] [--seed 1]"""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def capture(rng, npairs, kind):
    n = 2 * npairs
    if kind == "uniform":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "extremes":                                   # a saturating ADC: 0 and 255 only
        return (rng.integers(0, 2, n, dtype=np.uint8) * 255).astype(np.uint8)
    if kind == "quiet":                                      # a few LSBs around mid-scale
        return (127 + rng.integers(0, 3, n, dtype=np.uint8)).astype(np.uint8)
    if kind == "gauss":                                      # noise of a random level, clipped
        sigma = float(rng.uniform(2.0, 90.0))
        return np.clip(np.rint(127.5 + sigma * rng.standard_normal(n, dtype=np.float32)), 0, 255).astype(np.uint8)
    if kind == "dc":                                         # constant bytes: the integrators run away and wrap
        return np.full(n, int(rng.integers(0, 256)), np.uint8)
    if kind == "steps":                                      # long runs of one value, then another: wrap-around of both integrators
        runs = rng.integers(0, 256, n // 65536 + 1, dtype=np.uint8)
        return np.repeat(runs, 65536)[:n].copy()
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--captures", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import oracle_lib as O
    import rtlsdr_ft8d_amd as ft8
    build_id = ft8.check_build_id()
    O.lib()
    rng = np.random.default_rng(args.seed)
    kinds = ("uniform", "extremes", "quiet", "gauss", "gauss", "dc", "steps", "uniform")
    dec = ft8.Decoder(device=0, max_frames=args.captures)
    bad = total = 0
    floats = 0
    t0 = time.time()
    for b in range(args.batches):
        # full 15 s captures, and in every other batch a ragged length (short last block, 8-pair granularity)
        npairs = 751 * 48000 + 8 * int(rng.integers(0, 200)) if b % 2 == 0 else 8 * int(rng.integers(751 * 1000 // 8, 751 * 48000 // 8))
        raws = np.stack([capture(rng, npairs, kinds[(b + k) % len(kinds)]) for k in range(args.captures)])
        for normalise in (False, True):
            got = dec.rx_decimate(raws, normalise=normalise)
            with ThreadPoolExecutor(max_workers=min(args.captures, 8)) as ex:            # the oracle releases the GIL inside ctypes
                ref = list(ex.map(lambda r: O.rx_capture(r, normalise=normalise), raws))
            for k in range(args.captures):
                i, q, n = ref[k]
                same = np.array_equal(got[k, 0].view(np.uint32), i.view(np.uint32)) and np.array_equal(got[k, 1].view(np.uint32), q.view(np.uint32))
                bad += not same
                total += 1
                floats += 2 * 48000
        print(f"batch {b}: {args.captures} captures of {npairs} pairs: differing so far {bad}", flush=True)
    print(json.dumps({"captures_compared": total, "output_floats_compared": floats, "raw_bytes": int(args.batches * args.captures * 2 * 751 * 48000),
                      "captures_differing": bad, "seconds": round(time.time() - t0, 1), "seed": args.seed, "build_id": build_id}))


if __name__ == "__main__":
    main()
