#!/usr/bin/env python3
"""A/B of two (or more) BUILDS of libft8gpu.so in ONE process and GPU session -- boxes of the pool differ by several
percent, so only arms that share a box and a session are comparable.  Every arm decodes the same batch; arms are
interleaved round by round; the records of all arms must be byte-identical.
  python tools/ab_libs.py --libs tools/ab/libft8gpu_prev.so rtlsdr_ft8d_amd/libft8gpu.so [--config 2|4] [--rounds 3]"""
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
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--torch-stream", action="store_true", help="run every arm on a torch.cuda.Stream (as bench.py does) instead of the context's own stream")
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    cfg = {2: dict(frames=4096, nsig=20, snr=(-18.0, 0.0), maxc=120), 4: dict(frames=1024, nsig=60, snr=(-24.0, -14.0), maxc=480)}[args.config]
    B = args.frames or cfg["frames"]
    libs = [ft8.load_library() if os.path.abspath(p) == ft8.LIB_PATH else ft8.load_library_at(p) for p in args.libs]
    decs = [ft8.Decoder(device=0, max_frames=B, max_candidates=cfg["maxc"], lib=L) for L in libs]
    if args.torch_stream:
        ts = torch.cuda.Stream()
        torch.cuda.set_stream(ts)
        for d in decs:
            d.set_stream(ts.cuda_stream)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, cfg["nsig"], tones, snr_range=cfg["snr"])
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    decs[0].synth_frames(sig, B, cfg["nsig"], 1.0, workload.SEED_BASE, iq)
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    res = [{"lib": p, "ms": [], "stages": None, "digest": None} for p in args.libs]
    for _ in range(args.rounds):
        for dec, r in zip(decs, res):
            for _ in range(3):
                dec.decode_batch_dev(iq, B, spots, nres)
            dec.synchronize()
            dec.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dec.decode_batch_dev(iq, B, spots, nres)
            r.setdefault("host_enqueue_ms", []).append(round(1e3 * (time.perf_counter() - t0) / args.steps, 4))
            dec.synchronize()
            r["ms"].append(round(1e3 * (time.perf_counter() - t0) / args.steps, 4))
            st = dec.timings()
            dec.enable_timing(False)
            r["stages"] = {k: round(v, 4) for k, v in st.items() if k.endswith("_ms")}
            r["digest"] = hashlib.sha256(spots.cpu().numpy().tobytes() + nres.cpu().numpy().tobytes()).hexdigest()[:16]
    for r in res:
        r["best_ms"] = min(r["ms"])
    out = {"frames": B, "config": args.config, "steps": args.steps, "arms": res,
           "all_digests_equal": len({r["digest"] for r in res}) == 1}
    for d in decs:
        d.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
