#!/usr/bin/env python3
"""Mass parity run: many batches of on-device synthetic frames with varying density and SNR; every frame's spot
records from the GPU are compared byte for byte with the CPU oracle (all host cores).
usage: tools/soak_parity.py [--batches 40] [--frames 4096] [--seed 123] [--traffic cq|mixed]
--traffic mixed: the message pool of workload.mixed_message_pool (about a quarter CQ calls, the rest QSO traffic of every
message type, one frame in four with a message heard twice), and every record slot starts as the byte 0xA5 on both sides
so that the slots the reference leaves untouched (rtlsdr_ft8d.c:1509-1520) are part of the comparison."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=123, help="draws the batches' densities, SNR windows and caps, and offsets the frame seeds (123: the run of rounds 2-4)")
    ap.add_argument("--traffic", choices=("cq", "mixed"), default="cq")
    ap.add_argument("--edges", action="store_true",
                    help="signals all over the edges of the search window: tone 0 between -20 and 1620 Hz (bins 0 and 255, aliasing above "
                         "1600 Hz) and start times between -1.5 and +3.0 s (frames that begin before the window or run out of it): real decodes "
                         "where the sync score drops terms and the LLR extraction reads blocks that do not exist")
    ap.add_argument("--vary-min-score", action="store_true",
                    help="draw K_MIN_SCORE per batch from 10, 10, 5, 0, -3, 20, 30 (the reference fixes 10, rtlsdr_ft8d.h:43; the run-time form must "
                         "follow the same rules at any threshold: at 0 and below every position of the scan survives the gate)")
    ap.add_argument("--vary-frames", action="store_true",
                    help="draw the number of frames of each batch log-uniformly from 1 ... --frames (and once in eight exactly at a threshold of the "
                         "pipeline: 511, 512, 513, 1024, 1025, 2048, 2049): the plain and the two-part pipeline, the part split, the heap kernel forms "
                         "chosen by launch size and the 1024-frame chunks of the stage comparison all depend on it")
    ap.add_argument("--vary-gain", action="store_true",
                    help="scale the frames of two batches in three by 10^U(-4, 2.5) (the same float32 product on both sides): the log-magnitude "
                         "quantiser (rtlsdr_ft8d.c:1413-1433) then works all over its range and into both clamps -- 0 below -120 dB, 255 above "
                         "+7.5 dB -- where the device's threshold table stands in for log10f; partly or wholly saturated waterfalls also give "
                         "candidates whose LLRs are all zero (variance 0: the scale factor is inf, the LLRs NaN, as in the reference)")
    ap.add_argument("--gain-decades", type=float, nargs=2, default=(-4.0, 2.5), metavar=("LO", "HI"),
                    help="exponent range of --vary-gain (default -4 2.5; -18 17 reaches denormal products at one end and |X|^2 = +inf at the other: "
                         "the quantiser's fence for +inf is 255 on both sides, DESIGN.md section 2)")
    ap.add_argument("--wide-caps", action="store_true",
                    help="draw K_MAX_CANDIDATES per batch from the whole accepted range instead of 60 ... 480: 1, 2, 7, 33, 120, 481, 777, 1024 "
                         "(FT8GPU_ABS_MAX_CANDIDATES; the reference fixes 120, rtlsdr_ft8d.h:44): a heap of one entry, caps off the 4-candidate "
                         "block grid of the LDPC launch, and the largest list the context accepts")
    ap.add_argument("--vary-iters", action="store_true",
                    help="draw K_LDPC_ITERS per batch from 20, 20, 1, 5, 13, 50 on both sides (the reference fixes 20, rtlsdr_ft8d.h:45, and passes it "
                         "at rtlsdr_ft8d.c:1476; the kernel's iteration loop, its skipped dead last update and the iteration field of the status "
                         "record must follow upstream's bp_decode at any cap)")
    ap.add_argument("--debug-flags", type=int, default=0,
                    help="FT8GPU_DBG_* bits for the context: 1 = every BP division through the compiler's IEEE expansion and the reference-domain "
                         "sums (the complete second instruction stream of the LDPC kernel, which a normal run enters only for the iterations whose "
                         "guard fails); 4 = the plain pipeline (one launch per stage) whatever the batch size.  Results must not depend on them")
    ap.add_argument("--wide-iters", action="store_true",
                    help="draw K_LDPC_ITERS from 1, 2, 3, 19, 20, 21, 50, 137, 1000 (the accepted range is 1 ... 1000; use small batches: a "
                         "candidate that never converges costs the oracle fifty times the usual at 1000)")
    ap.add_argument("--records", "--stages", action="store_true", dest="records",
                    help="also compare every stage boundary of every frame through the stage entries: all 94 208 waterfall bytes, the ordered "
                         "candidate list, and the 48-byte status record of EVERY candidate (parity errors, iterations, packed bits, CRCs, unpack "
                         "status, text) -- the text of a message that is not a CQ call never reaches the spot records, and a deviation inside "
                         "a candidate that does not decode reaches nothing.  The records are taken from BOTH forms of the LDPC kernel: the stage "
                         "entry's own (ft8_decode_kernel<true,1>, every byte) and the one the batch pipeline runs (<false,3>, FT8GPU_DBG_PIPELINE_FORM: "
                         "every byte but ldpc_errors, which it reports as 0 / 83 and must agree on zero) -- tests/stage_check.py")
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import stage_check
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    sys.path.insert(0, ROOT)
    from bench import usable_cores
    cores = usable_cores()
    build_id = ft8.check_build_id()               # a stale or foreign library is refused before anything is measured
    B = BMAX = args.frames
    sizes = []
    gains = []
    mixed = args.traffic == "mixed"
    _, tones = workload.message_pool(traffic=args.traffic)
    fill = 0xA5 if mixed else 0
    stale_rec = np.full(28, fill, np.uint8).tobytes()
    start_all = np.full((B, 1400), fill, np.uint8).view(O.RESULT_DTYPE).reshape(B, 50) if mixed else None
    rng = np.random.default_rng(args.seed)
    dec = ft8.Decoder(device=0, max_frames=B)
    if args.debug_flags:
        dec.set_debug_flags(args.debug_flags)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    bad = total = msgs = written = 0
    stages = stage_check.new_counters()
    first_bad = []
    iters_hist = {}
    t0 = time.time()
    for b in range(args.batches):
        nsig = int(rng.integers(0, 61))
        lo_snr = float(rng.uniform(-26, -10)); hi_snr = lo_snr + float(rng.uniform(2, 20))
        cap = int(rng.choice([1, 2, 7, 33, 120, 481, 777, 1024])) if args.wide_caps else int(rng.choice([120, 120, 120, 60, 240, 480]))
        min_score = int(rng.choice([10, 10, 5, 0, -3, 20, 30])) if args.vary_min_score else 10
        iters = int(rng.choice([20, 20, 1, 5, 13, 50])) if args.vary_iters else 20
        if args.wide_iters:
            iters = int(rng.choice([1, 2, 3, 19, 20, 21, 50, 137, 1000]))
        if args.vary_frames:
            B = int(rng.choice([511, 512, 513, 1024, 1025, 2048, 2049])) if rng.integers(0, 8) == 0 else int(round(float(np.exp(rng.uniform(0.0, np.log(BMAX))))))
            B = max(1, min(B, BMAX))
            sizes.append(B)
        iters_hist[iters] = iters_hist.get(iters, 0) + B
        dec.set_params(min_score=min_score, max_candidates=cap, ldpc_iters=iters)
        sig, _ = workload.frame_signals(1_000_000 + (args.seed - 123) * 10_000_000 + b * BMAX, B, nsig, tones, snr_range=(lo_snr, hi_snr),
                                        dup_fraction=workload.MIXED_DUP_FRACTION if mixed else 0.0,
                                        **(dict(f_range=(-20.0, 1620.0), dt_range=(-1.5, 3.0)) if args.edges else {}))
        dec.synth_frames(sig, B, nsig, 1.0, 777 + b + (args.seed - 123) * 100_003, iq)
        gain = 1.0
        if args.vary_gain and rng.integers(0, 3) != 0:
            gain = float(10.0 ** rng.uniform(args.gain_decades[0], args.gain_decades[1]))
            dec.synchronize()                            # the synthesis runs on the decoder's stream, the scaling on torch's
            iq.mul_(gain)
            gains.append(gain)
        spots.fill_(fill)
        torch.cuda.synchronize()                     # the fill runs on torch's stream, the decoder on its own
        dec.decode_batch_dev(iq, B, spots, nres)
        dec.synchronize()
        g = spots[:B].cpu().numpy().view(ft8.RESULT_DTYPE).reshape(B, 50)
        gn = nres[:B].cpu().numpy()
        start = start_all[:B] if start_all is not None else None
        rdec, rn = O.subsystem_batch(iq[:B].cpu().numpy(), O.default_params(min_score, cap, iters), cores, decodes=start)
        mism = [k for k in range(B) if gn[k] != rn[k] or g[k].tobytes() != rdec[k].tobytes()]
        if args.records:
            before = stage_check.differing(stages)
            stage_check.stage_boundaries_vs_oracle(ft8, O, dec, iq, B, cap, min_score, iters, cores, counters=stages, first_bad=first_bad, base_flags=args.debug_flags)
            if stage_check.differing(stages) != before:
                print(f"batch {b}: stage boundaries differ: {stages} first {first_bad}", flush=True)
        w = int(sum(1 for k in range(B) for j in range(min(int(gn[k]), 50)) if g[k, j].tobytes() != stale_rec)) if mixed else int(np.minimum(gn, 50).sum())
        bad += len(mism); total += B; msgs += int(gn.sum()); written += w
        print(f"batch {b}: {B} frames, nsig {nsig} snr [{lo_snr:.0f},{hi_snr:.0f}] cap {cap} min_score {min_score} iters {iters} gain {gain:.3g}: {int(gn.sum())} messages, {w} CQ spots, mismatching frames {len(mism)}", flush=True)
    print(json.dumps({"frames": total, "messages": msgs, "cq_spots_written": written, "mismatching_frames": bad, "seconds": round(time.time() - t0, 1), "seed": args.seed,
                      "batches": args.batches, "traffic": args.traffic, "edges": bool(args.edges), "vary_min_score": bool(args.vary_min_score), "initial_record_byte": fill, "messages_per_frame": round(msgs / max(total, 1), 3),
                      "cq_spots_per_frame": round(written / max(total, 1), 3), "build_id": build_id,
                      "vary_iters": bool(args.vary_iters), "vary_frames": bool(args.vary_frames), "debug_flags": args.debug_flags, "wide_caps": bool(args.wide_caps), "wide_iters": bool(args.wide_iters), "vary_gain": bool(args.vary_gain), **({"gains_min_max": [min(gains), max(gains)], "batches_scaled": len(gains)} if gains else {}), **({"batch_sizes_min_median_max": [int(min(sizes)), int(np.median(sizes)), int(max(sizes))], "batches_below_512_frames": int(sum(x < 512 for x in sizes))} if sizes else {}), "frames_by_ldpc_iters": {str(k): v for k, v in sorted(iters_hist.items())},
                      **({"stages": stages, "stage_differences_total": stage_check.differing(stages), "first_differences": first_bad} if args.records else {})}))
    return 1 if bad or (args.records and stage_check.differing(stages)) else 0


if __name__ == "__main__":
    sys.exit(main())
