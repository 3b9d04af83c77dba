#!/usr/bin/env python3
"""A/B of builds of libft8gpu.so on the RX front end (ft8gpu_rx_decimate, device pointers): interleaved rounds in one
process, outputs must be bit-identical.   python tools/ab_rx.py --libs a.so b.so [--captures 16] [--rounds 4]"""
import argparse, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--captures", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    npairs = 36_000_000
    libs = [ft8.load_library() if os.path.abspath(p) == ft8.LIB_PATH else ft8.load_library_at(p) for p in args.libs]
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    decs = [ft8.Decoder(device=0, max_frames=args.captures, lib=L) for L in libs]
    for d in decs:
        d.set_stream(stream.cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(1)
    raw = torch.randint(0, 256, (args.captures, 2 * npairs), dtype=torch.uint8, device="cuda", generator=g)
    iq = torch.empty((args.captures, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    res = [{"lib": p, "ms": [], "digest": None} for p in args.libs]
    for _ in range(args.rounds):
        for dec, r in zip(decs, res):
            for _ in range(3):
                dec.rx_decimate_dev(raw, args.captures, npairs, iq, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.steps):
                dec.rx_decimate_dev(raw, args.captures, npairs, iq, True)
            e1.record(stream)
            torch.cuda.synchronize()
            r["ms"].append(round(e0.elapsed_time(e1) / args.steps, 4))
            r["digest"] = hashlib.sha256(iq.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"arms": res, "all_digests_equal": len({r["digest"] for r in res}) == 1}, indent=1))


if __name__ == "__main__":
    main()
