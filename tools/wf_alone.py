#!/usr/bin/env python3
"""The waterfall kernel alone (stage entry on device pointers, nothing beside it), every last-stage form, interleaved in one process.
  python tools/wf_alone.py [--frames 4096] [--reps 20] [--rounds 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--arms", type=int, nargs="+", default=[0, 8])
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B = args.frames
    lib = ft8.load_ab_library() if any(a & ~7 for a in args.arms) else None     # form 8 (LDS) lives in the A/B build
    dec = ft8.Decoder(device=0, max_frames=B, lib=lib)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, 20, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
    mag = torch.empty((B, ft8.MAG_ARRAY), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    res = {a: [] for a in args.arms}
    for _ in range(args.rounds):
        for a in args.arms:
            dec.set_debug_flags(a)
            for _ in range(5):
                dec.waterfall_dev(iq, B, mag)
            dec.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                dec.waterfall_dev(iq, B, mag)
            dec.synchronize()
            res[a].append(round(1e3 * (time.perf_counter() - t0) / args.reps, 4))
    dec.close()
    print(json.dumps({"frames": B, "reps": args.reps, "ms_per_launch": {str(a): v for a, v in res.items()},
                      "GBps_in": {str(a): round(B * 384000 / (min(v) * 1e-3) / 1e9, 1) for a, v in res.items()}}))


if __name__ == "__main__":
    main()
