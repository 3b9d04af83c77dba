#!/usr/bin/env python3
"""What the context's co-execution probe sees, and what the overlap is worth (GPU box).
  python tools/stream_probe.py
Creates decoder contexts in three situations -- first thing in the process, behind 9 busy framework streams, and as the
8th context of a process -- and reports ft8gpu_overlap_active() plus the step time of the bench batch with and without the
overlap (FT8GPU_DBG_NO_OVERLAP), interleaved."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def step_ms(dec, iq, B, spots, nres, steps=15):
    for _ in range(4):
        dec.decode_batch_dev(iq, B, spots, nres)
    dec.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dec.decode_batch_dev(iq, B, spots, nres)
    dec.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / steps, 4)


def main():
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B = 4096
    out = {}
    t0 = time.perf_counter()
    d0 = ft8.Decoder(device=0, max_frames=B)
    out["create_ms_first_context"] = round(1e3 * (time.perf_counter() - t0), 1)
    out["first_context_overlap"] = d0.overlap_active()
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, 20, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    d0.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    crowd = [torch.cuda.Stream() for _ in range(9)]
    for st in crowd:
        with torch.cuda.stream(st):
            torch.zeros(1024, device="cuda").sum()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d1 = ft8.Decoder(device=0, max_frames=B)
    out["create_ms_behind_9_streams"] = round(1e3 * (time.perf_counter() - t0), 1)
    out["context_behind_9_streams_overlap"] = d1.overlap_active()
    more = [ft8.Decoder(device=0, max_frames=64) for _ in range(6)]
    out["contexts_3_to_8_overlap"] = [d.overlap_active() for d in more]
    for d in more:
        d.close()
    rounds = []
    for _ in range(3):
        r = {}
        for name, dec in (("first", d0), ("behind_streams", d1)):
            dec.set_debug_flags(0)
            r[name + "_overlapped_ms"] = step_ms(dec, iq, B, spots, nres)
            dec.set_debug_flags(ft8.DBG_NO_OVERLAP)
            r[name + "_plain_ms"] = step_ms(dec, iq, B, spots, nres)
            dec.set_debug_flags(0)
        rounds.append(r)
    out["step_ms_rounds"] = rounds
    d1.set_stream(crowd[0].cuda_stream)
    out["on_a_borrowed_torch_stream_overlap"] = d1.overlap_active()
    out["on_a_borrowed_torch_stream_ms"] = step_ms(d1, iq, B, spots, nres)
    d0.close()
    d1.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
