#!/usr/bin/env python3
"""How far can the spot list move when the FFT is not OUR float32 FFT?

The reference transforms with fftw3f (rtlsdr_ft8d.c:326, :1411), whose codelets are machine dependent, so no
FFT can be bit-identical to it; the oracle and the kernel fix one float32 order ("R4DIF-1024").  The only
estimate of the divergence from the real reference available here is to run everything after the waterfall
(ft8_find_sync -> ft8_decode -> dedup -> spots, rtlsdr_ft8d.c:1438-1523) twice per frame -- from the R4DIF
waterfall and from a float64-DFT waterfall (the value any correct float32 FFT scatters around) -- and compare.

  python tools/fft_parity.py --source gpu  --frames 4096     # the bench batch (device synthesis, GPU box)
  python tools/fft_parity.py --source host --frames 64       # numpy-synthesised frames (CPU only)

Prints one JSON object: waterfall cells that differ (all +-1), frames whose candidate list differs, frames
whose spot records differ and how (a message missing / extra, or the same messages with another freq / snr /
order, i.e. another duplicate won the dedup).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def compare(oracle, iq, nthreads, max_candidates=120):
    p = oracle.default_params(10, max_candidates, 20)
    m32 = oracle.waterfall_batch(iq, f64=False, nthreads=nthreads)
    m64 = oracle.waterfall_batch(iq, f64=True, nthreads=nthreads)
    d = m32.astype(np.int16) - m64.astype(np.int16)
    d32, n32 = oracle.subsystem_from_waterfall_batch(m32, p, nthreads)
    d64, n64 = oracle.subsystem_from_waterfall_batch(m64, p, nthreads)
    B = iq.shape[0]
    frames_cells = int(np.count_nonzero(np.abs(d).max(axis=1)))
    cand_diff = same = reorder = content = 0
    msgs_total = msgs_missing = msgs_extra = 0
    examples = []
    for f in range(B):
        if not np.array_equal(m32[f], m64[f]):
            c32 = oracle.find_sync(m32[f], max_candidates, 10)
            c64 = oracle.find_sync(m64[f], max_candidates, 10)
            cand_diff += int(len(c32) != len(c64) or not np.array_equal(c32, c64))
        a = [(x["call"], x["loc"]) for x in d32[f][:n32[f]]]
        b = [(x["call"], x["loc"]) for x in d64[f][:n64[f]]]
        msgs_total += len(b)
        if n32[f] == n64[f] and d32[f].tobytes() == d64[f].tobytes():
            same += 1
            continue
        sa, sb = set(a), set(b)
        if sa == sb and n32[f] == n64[f]:
            reorder += 1                      # same CQ messages; freq / snr / slot of some record differs
        else:
            content += 1
            msgs_missing += len(sb - sa)
            msgs_extra += len(sa - sb)
        if len(examples) < 5:
            examples.append({"frame": f, "n_r4dif": int(n32[f]), "n_f64": int(n64[f]),
                             "only_r4dif": sorted(str(x) for x in sa - sb), "only_f64": sorted(str(x) for x in sb - sa)})
    return {
        "frames": B,
        "waterfall_cells_differing": int(np.count_nonzero(d)), "waterfall_cells_total": int(d.size),
        "waterfall_max_abs_diff": int(np.abs(d).max()), "frames_with_a_differing_cell": frames_cells,
        "frames_candidate_list_differs": cand_diff,
        "frames_spot_records_identical": same,
        "frames_same_messages_other_freq_snr_or_slot": reorder,
        "frames_message_set_differs": content,
        "cq_messages_f64": msgs_total, "cq_messages_missing_in_r4dif": msgs_missing, "cq_messages_extra_in_r4dif": msgs_extra,
        "examples": examples,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--source", choices=("gpu", "host"), default="gpu")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--nsig", type=int, default=20)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    import oracle_lib as oracle
    oracle.build()
    oracle.lib()
    nthreads = args.threads or len(os.sched_getaffinity(0))
    if args.source == "gpu":
        import torch
        import rtlsdr_ft8d_amd as ft8
        from rtlsdr_ft8d_amd import workload
        _, tones = workload.message_pool()
        sig, _ = workload.frame_signals(0, args.frames, args.nsig, tones)
        with ft8.Decoder(device=0, max_frames=args.frames) as dec:
            t = torch.empty((args.frames, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
            dec.synth_frames(sig, args.frames, args.nsig, 1.0, workload.SEED_BASE, t)
            iq = t.cpu().numpy()
        what = f"bench batch: {args.frames} device-synthesised frames, {args.nsig} signals/frame, SNR U[-18,0] dB"
    else:
        import synth_util as S
        enc = S.oracle_encode_fn(oracle)
        iq = np.stack([S.make_frame(1000 + k, args.nsig, enc, snr_range=(-18, 0))[0] for k in range(args.frames)])
        what = f"{args.frames} numpy-synthesised frames (tests/synth_util.make_frame seeds 1000..), {args.nsig} signals/frame"
    out = compare(oracle, iq, nthreads)
    out["input"] = what
    print(json.dumps(out))


if __name__ == "__main__":
    main()
