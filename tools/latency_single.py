#!/usr/bin/env python3
"""Latency of the drop-in single-frame path: ft8_subsystem() through libft8gpu.so, host buffers in, spots out."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, synth_util as S, rtlsdr_ft8d_amd as ft8
enc = S.oracle_encode_fn(O)
iq, _ = S.make_frame(13, 20, enc, snr_range=(-18, 0))
lib = ft8.load_library(); lib.initFFTW()
for _ in range(5): ft8.ft8_subsystem(iq[0], iq[1])
t = []
for _ in range(50):
    t0 = time.perf_counter(); dec, n = ft8.ft8_subsystem(iq[0], iq[1]); t.append(time.perf_counter() - t0)
t0 = time.perf_counter(); rdec, rn = O.subsystem(iq[0], iq[1]); tc = time.perf_counter() - t0
print({"single_frame_ms_median": round(1e3 * float(np.median(t)), 3), "min": round(1e3 * min(t), 3), "n_results": int(n),
       "oracle_ms": round(1e3 * tc, 2), "identical": bool(n == rn and dec.tobytes() == rdec.tobytes())})
lib.freeFFTW()
