#!/usr/bin/env python3
"""A/B of pipeline variants in ONE process and GPU session (boxes differ by several percent, so arms of an
experiment must share a box): the bench batch decoded under different FT8GPU_DBG_* flag sets, interleaved.
  python tools/ab_probe.py [--frames 4096] [--steps 20] [--rounds 3] --arms 0 4
The arms are sums of the FT8GPU_DBG_* bits of include/ft8gpu.h (1 IEEE division, 2 pipeline form of the stage entry,
4 one launch per stage); unknown bits are refused by ft8gpu_set_debug_flags.  Arms with the kernel-form selectors of
csrc/ft8gpu_internal.h (8 LDS waterfall, 16 / 32 forced heap forms) run on the A/B build libft8gpu_ab.so (`make ab`), all
arms of such a run on that one library.  (The arms 8 / 16 / 24 quoted in profiles/r02_ab_kernels.json selected experimental
kernels of builds that no longer exist; two BUILDS are compared with tools/ab_libs.py.)"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--nsig", type=int, default=20)
    ap.add_argument("--max-candidates", type=int, default=120)
    ap.add_argument("--arms", type=int, nargs="+", default=[0])
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B = args.frames
    # arms with the kernel-form selector bits (8 LDS waterfall, 16 / 32 forced heap forms) exist in the A/B build only
    lib = ft8.load_ab_library() if any(a & ~7 for a in args.arms) else None
    dec = ft8.Decoder(device=0, max_frames=B, max_candidates=args.max_candidates, lib=lib)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, args.nsig, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    dec.synth_frames(sig, B, args.nsig, 1.0, workload.SEED_BASE, iq)
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    res = {a: {"ms": [], "stages": None, "digest": None} for a in args.arms}
    for _ in range(args.rounds):
        for a in args.arms:
            dec.set_debug_flags(a)
            for _ in range(3):
                dec.decode_batch_dev(iq, B, spots, nres)
            dec.synchronize()
            dec.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dec.decode_batch_dev(iq, B, spots, nres)
            dec.synchronize()
            res[a]["ms"].append(round(1e3 * (time.perf_counter() - t0) / args.steps, 4))
            st = dec.timings()
            dec.enable_timing(False)
            res[a]["stages"] = {k: round(v, 4) for k, v in st.items() if k.endswith("_ms")}
            res[a]["digest"] = hashlib.sha256(spots.cpu().numpy().tobytes() + nres.cpu().numpy().tobytes()).hexdigest()[:16]
    out = {"frames": B, "steps": args.steps, "arms": {str(a): {"best_ms": min(r["ms"]), **r} for a, r in res.items()}}
    out["all_digests_equal"] = len({r["digest"] for r in res.values()}) == 1
    dec.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
