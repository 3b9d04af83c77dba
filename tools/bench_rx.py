#!/usr/bin/env python3
"""Throughput of the RX front end (rtlsdr_callback(), rtlsdr_ft8d.c:76-202) on
one MI355X: raw 2.4 Msps unsigned 8-bit I/Q captures resident in HBM -> decimated, normalised 15 s frames.
Not the BASELINE metric (bench.py reports that); this is the HBM-bound neighbour of the hot path."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captures", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-captures", type=int, default=1)
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    npairs = 36_000_000                       # 15 s at 2.4 Msps
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dec = ft8.Decoder(device=0, max_frames=args.captures)
    dec.set_stream(stream.cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    raw = torch.randint(0, 256, (args.captures, 2 * npairs), dtype=torch.uint8, device=dev, generator=g)
    iq = torch.empty((args.captures, 2, ft8.NSAMPLES), dtype=torch.float32, device=dev)
    for _ in range(2):
        dec.rx_decimate_dev(raw, args.captures, npairs, iq, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.steps):
        dec.rx_decimate_dev(raw, args.captures, npairs, iq, True)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    bytes_in = args.captures * 2 * npairs
    bytes_out = args.captures * 2 * ft8.NSAMPLES * 4
    out = {"stage": "rx front end (rtlsdr_callback)", "captures_per_launch": args.captures, "ms_per_launch": round(ms, 3),
           "captures_per_s": round(args.captures / ms * 1e3, 1),
           "roofline": {"bound": "hbm", "achieved": round((bytes_in + bytes_out) / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round((bytes_in + bytes_out) / ms / 1e6 / 8000.0, 4),
                        "algorithmic_bytes_per_launch": bytes_in + bytes_out, "traffic": None}}
    try:        # HBM bytes of the block kernel from the committed counter passes (16 captures per launch), scaled to this launch
        with open(os.path.join(ROOT, "profiles", "r04_rx_pmc_traffic.json")) as f:
            pm = json.load(f)["block_kernel"]
        out["roofline"]["traffic"] = int((pm["fetch_bytes_per_launch_corrected"] + pm["write_bytes_per_launch_raw"]) * args.captures / 16)
        out["roofline"]["traffic_from"] = "profiles/r04_rx_pmc_traffic.json (block kernel: FETCH_SIZE x 2 + WRITE_SIZE, per 16 captures)"
    except (OSError, KeyError, ValueError):
        pass
    if args.cpu_captures > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        h = raw[:args.cpu_captures].cpu().numpy()
        t0 = time.perf_counter()
        ok = 0
        g_iq = iq[:args.cpu_captures].cpu().numpy()
        for k in range(args.cpu_captures):
            i, q, _ = oracle_lib.rx_capture(h[k], normalise=True)
            ok += int(np.array_equal(i, g_iq[k, 0]) and np.array_equal(q, g_iq[k, 1]))
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(args.cpu_captures / dt, 3), "unit": "captures/s", "cores": 1, "kind": "port",
                               "sample": f"{args.cpu_captures} capture(s) through oracle ft8o_rx_capture", "identical": f"{ok}/{args.cpu_captures}"}
    dec.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
