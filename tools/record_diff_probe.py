#!/usr/bin/env python3
"""Where do per-candidate status records of the GPU stage entry differ from the oracle's ft8_decode?  Replays the batch sequence
of tools/soak_parity.py (same seed, same recipe) through waterfall -> find_sync -> decode_candidates and prints every differing
record field by field, plus the same candidate decoded again with the IEEE-division path forced (FT8GPU_DBG_FORCE_IEEE_DIV) and
with the pipeline form of the kernel.
usage: tools/record_diff_probe.py [--batches 8] [--seed 2034] [--traffic mixed] [--edges]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def fields(rec, ft8):
    r = rec.view(ft8.STATUS_DTYPE)[0]
    return {"ldpc_errors": int(r["ldpc_errors"]), "iters": int(r["iters"]), "crc_x": int(r["crc_extracted"]), "crc_c": int(r["crc_calculated"]),
            "unpack": int(r["unpack_status"]), "ok": int(r["ok"]), "a91": bytes(r["a91"]).hex(), "text": bytes(r["text"]).split(b"\0")[0].decode("latin-1")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=2034)
    ap.add_argument("--traffic", choices=("cq", "mixed"), default="mixed")
    ap.add_argument("--edges", action="store_true")
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    from bench import usable_cores
    cores = usable_cores()
    ft8.check_build_id()
    B = args.frames
    mixed = args.traffic == "mixed"
    _, tones = workload.message_pool(traffic=args.traffic)
    rng = np.random.default_rng(args.seed)
    dec = ft8.Decoder(device=0, max_frames=B)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    mag = torch.empty((B, ft8.MAG_ARRAY), dtype=torch.uint8, device="cuda")
    counts = torch.zeros((B,), dtype=torch.int32, device="cuda")
    out = []
    saved = {"mag": [], "cand": [], "gpu_by_iters": []}
    for b in range(args.batches):
        nsig = int(rng.integers(0, 61))
        lo_snr = float(rng.uniform(-26, -10)); hi_snr = lo_snr + float(rng.uniform(2, 20))
        cap = int(rng.choice([120, 120, 120, 60, 240, 480]))
        dec.set_params(max_candidates=cap)
        sig, _ = workload.frame_signals(1_000_000 + (args.seed - 123) * 10_000_000 + b * B, B, nsig, tones, snr_range=(lo_snr, hi_snr),
                                        dup_fraction=workload.MIXED_DUP_FRACTION if mixed else 0.0,
                                        **(dict(f_range=(-20.0, 1620.0), dt_range=(-1.5, 3.0)) if args.edges else {}))
        dec.synth_frames(sig, B, nsig, 1.0, 777 + b + (args.seed - 123) * 100_003, iq)
        st = torch.zeros((B, cap, 48), dtype=torch.uint8, device="cuda")
        cd = torch.zeros((B, cap, 8), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        dec.waterfall_dev(iq, B, mag)
        dec.find_sync_dev(mag, B, cd, counts)
        dec.decode_candidates_dev(mag, cd, counts, B, st)
        dec.synchronize()
        got = st.cpu().numpy()
        variants = {}
        for name, flag in (("ieee", ft8.DBG_FORCE_IEEE_DIV), ("pipeline", ft8.DBG_PIPELINE_FORM)):
            dec.set_debug_flags(flag)
            st.zero_(); torch.cuda.synchronize()
            dec.decode_candidates_dev(mag, cd, counts, B, st)
            dec.synchronize()
            variants[name] = st.cpu().numpy()
        dec.set_debug_flags(0)
        h_mag, h_counts = mag.cpu().numpy(), counts.cpu().numpy()
        h_cands = cd.cpu().numpy().view(O.CAND_DTYPE).reshape(B, cap)
        want = O.decode_candidates_batch(h_mag, h_cands, h_counts, 20, cores)
        bad = np.argwhere((got != want).any(axis=2))
        print(f"batch {b}: nsig {nsig} cap {cap}: {len(bad)} of {int(h_counts.sum())} records differ", flush=True)
        for f, c in bad[:12]:
            cand = h_cands[f, c]
            llr = O.llr(h_mag[f], h_cands[f, c:c + 1])
            raw = O.llr(h_mag[f], h_cands[f, c:c + 1], normalise=False)
            e = {"batch": b, "frame": int(f), "cand": int(c), "candidate": [int(x) for x in cand.tolist()],
                 "gpu": fields(got[f, c], ft8), "oracle": fields(want[f, c], ft8),
                 "gpu_ieee": fields(variants["ieee"][f, c], ft8), "gpu_pipeline": fields(variants["pipeline"][f, c], ft8),
                 "llr_abs_min": float(np.abs(llr).min()), "llr_abs_max": float(np.abs(llr).max()), "llr_zeros": int((llr == 0).sum()),
                 "raw_llr_zeros": int((raw == 0).sum()), "llr_nan": int(np.isnan(llr).sum())}
            e["differing_fields"] = [k for k in e["gpu"] if e["gpu"][k] != e["oracle"][k]]
            # the same candidate with max_iterations = 1 .. 20: where does the GPU leave the oracle?
            one = np.zeros((1, cap), O.CAND_DTYPE); one[0, 0] = cand
            per = []
            for it in range(1, 21):
                dec.set_params(ldpc_iters=it)
                g1 = dec.decode_candidates(h_mag[f:f + 1], one.view(ft8.CAND_DTYPE), np.array([1], np.int32))[0, 0]
                o1 = O.decode(h_mag[f], h_cands[f, c:c + 1], it)
                per.append([int(g1["ldpc_errors"]), bytes(g1["a91"]).hex(), o1["ldpc_errors"], o1["a91"].hex()])
            dec.set_params(ldpc_iters=20)
            e["first_iters_differing"] = next((i + 1 for i, r in enumerate(per) if r[0] != r[2] or r[1] != r[3]), None)
            saved["mag"].append(h_mag[f].copy()); saved["cand"].append(np.array(cand.tolist(), np.int32)); saved["gpu_by_iters"].append(per)
            out.append(e)
            print(json.dumps(e), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "record_diff_cases.npz"), mag=np.array(saved["mag"], np.uint8), cand=np.array(saved["cand"], np.int32))
    json.dump(saved["gpu_by_iters"], open(os.path.join(ROOT, "gpurun_out", "record_diff_by_iters.json"), "w"))
    print("SUMMARY", json.dumps({"diffs_listed": len(out), "fields": sorted({k for e in out for k in e["differing_fields"]}),
                                 "ieee_equals_oracle": sum(e["gpu_ieee"] == e["oracle"] for e in out)}))


if __name__ == "__main__":
    main()
