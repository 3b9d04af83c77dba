#!/usr/bin/env python3
"""Throughput of the spot-report stage (postSpots(), rtlsdr_ft8d.c:365-590) on one
MI355X: the spot lists of a decoded bench batch, resident in HBM -> one PSKreporter datagram per frame.
Not the BASELINE metric (bench.py reports that)."""
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
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cpu-frames", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B = args.frames
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dec = ft8.Decoder(device=0, max_frames=B)
    dec.set_stream(stream.cuda_stream)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, 20, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device=dev)
    dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
    spots = torch.zeros((B, ft8.MAX_MESSAGES * 28), dtype=torch.uint8, device=dev)
    nres = torch.zeros((B,), dtype=torch.int32, device=dev)
    dec.decode_batch_dev(iq, B, spots, nres)
    out = torch.empty((B, ft8.DATAGRAM_STRIDE), dtype=torch.uint8, device=dev)
    lens = torch.empty((B,), dtype=torch.int32, device=dev)
    times = torch.arange(1700000000, 1700000000 + 15 * B, 15, dtype=torch.int64, device=dev).to(torch.int32)
    info = ft8.ReportInfo(rcall=b"N0CALL", rloc=b"FN20", app_version=b"rtlsdr-ft8d_v0.3.6", dial_freq=14074000,
                          unixtime=0, sequence=1, random_id=7)
    for _ in range(3):
        dec.pskreporter_datagrams_dev(spots, nres, B, info, times, out, lens)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.steps):
        dec.pskreporter_datagrams_dev(spots, nres, B, info, times, out, lens)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    n_host = nres.cpu().numpy()
    l_host = lens.cpu().numpy()
    # algorithmic bytes: the records below n_results and the count in; the datagram and its length out
    alg = int(n_host.clip(0, 50).sum()) * 28 + 4 * B + int(l_host.sum()) + 4 * B
    moved = int(n_host.clip(0, 50).sum()) * 28 + 8 * B + B * ft8.DATAGRAM_STRIDE + 4 * B          # fixed-stride rows are stored whole
    res = {"stage": "spot report (postSpots datagrams)", "frames_per_launch": B, "ms_per_launch": round(ms, 4),
           "frames_per_s": round(B / ms * 1e3, 1), "spots_per_frame": round(float(n_host.mean()), 2),
           "mean_datagram_bytes": round(float(l_host.mean()), 1),
           "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(alg / ms / 1e6 / 8000.0, 4), "algorithmic_bytes_per_launch": alg,
                        "bytes_moved_per_launch": moved, "moved_GBps": round(moved / ms / 1e6, 1)}}
    if args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        m = min(args.cpu_frames, B)
        d = spots[:m].cpu().numpy().view(oracle_lib.RESULT_DTYPE).reshape(m, 50)
        g = out[:m].cpu().numpy()
        th = times[:m].cpu().numpy().view(np.uint32)
        oi = oracle_lib.ReportInfo(rcall=b"N0CALL", rloc=b"FN20", app_version=b"rtlsdr-ft8d_v0.3.6", dial_freq=14074000,
                                   unixtime=0, sequence=1, random_id=7)
        ok = 0
        t0 = time.perf_counter()
        for f in range(m):
            oi.unixtime = int(th[f])
            want = oracle_lib.pskreporter_datagram(d[f], int(n_host[f]), oi)
            ok += int(want.size == l_host[f] and want.tobytes() == g[f, :want.size].tobytes() and not g[f, want.size:].any())
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(m / dt, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{m} frames through oracle ft8o_pskreporter_datagram via ctypes (includes the comparison)",
                               "gpu_vs_oracle_identical_frames": f"{ok}/{m}"}
    dec.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
