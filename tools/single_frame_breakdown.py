#!/usr/bin/env python3
"""Where the 0.24 ms of one frame go (the daemon's operating point: one 15 s frame per call): per-stage hipEvent times of the
plain pipeline for 1, 2, 8, 64 and 256 frames from device memory, and the host-buffer call (H2D + kernels + D2H + sync)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    out = {}
    B = 256
    dec = ft8.Decoder(device=0, max_frames=B)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, 20, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    host = iq.cpu().numpy()
    for n in (1, 2, 8, 64, 256):
        for _ in range(5):
            dec.decode_batch_dev(iq, n, spots, nres)
        dec.synchronize()
        dec.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(30):
            dec.decode_batch_dev(iq, n, spots, nres)
            dec.synchronize()
        wall = (time.perf_counter() - t0) / 30
        st = dec.timings()
        dec.enable_timing(False)
        lat = []
        for _ in range(30):
            t0 = time.perf_counter()
            dec.decode_batch(host[:n])
            lat.append(time.perf_counter() - t0)
        out[str(n)] = {"device_call_plus_sync_ms": round(1e3 * wall, 4), "host_call_ms_median": round(1e3 * float(np.median(lat)), 4),
                       **{k: round(v, 4) for k, v in st.items() if k.endswith("_ms")}}
    dec.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
