#!/usr/bin/env python3
"""Rebuilds tests/golden/sqrt_ulp_cases.{npz,json}: a few of the candidates listed in
profiles/r05_record_diff_probe_before_sqrt_fix.jsonl (status records that differed from the oracle while the LLR scale factor's
root was HIP's 1-ulp __fsqrt_rn), with their waterfalls regenerated from the soak's seeds on the GPU and, per max_iterations
1 ... 20, what a library WITH the old root says (round 4's shipped library, tools/ab/libft8gpu_r04.so) next to the oracle.
usage (GPU box): tools/make_sqrt_fixture.py [--old-lib tools/ab/libft8gpu_r04.so] [--cases 4]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--old-lib", default=os.path.join(ROOT, "tools", "ab", "libft8gpu_r04.so"))
    ap.add_argument("--cases", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2034)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    listed = [json.loads(ln) for ln in open(os.path.join(ROOT, "profiles", "r05_record_diff_probe_before_sqrt_fix.jsonl"))]
    # a mix: records whose packed bits differed and records whose error count differed
    by_kind = {"a91": [e for e in listed if "a91" in e["differing_fields"]], "ldpc_errors": [e for e in listed if e["differing_fields"] == ["ldpc_errors"]]}
    chosen = (by_kind["a91"][:(args.cases + 1) // 2] + by_kind["ldpc_errors"][:args.cases // 2])[:args.cases]
    want_batches = sorted({e["batch"] for e in chosen})
    B = 4096
    _, tones = workload.message_pool(traffic="mixed")
    rng = np.random.default_rng(args.seed)
    dec = ft8.Decoder(device=0, max_frames=B)
    old = ft8.Decoder(device=0, max_frames=1, lib=ft8.load_library_at(args.old_lib))
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    mag = torch.empty((B, ft8.MAG_ARRAY), dtype=torch.uint8, device="cuda")
    mags, cands, cases = [], [], []
    for b in range(max(want_batches) + 1):
        nsig = int(rng.integers(0, 61))
        lo_snr = float(rng.uniform(-26, -10)); hi_snr = lo_snr + float(rng.uniform(2, 20))
        cap = int(rng.choice([120, 120, 120, 60, 240, 480]))
        if b not in want_batches:
            continue
        sig, _ = workload.frame_signals(1_000_000 + (args.seed - 123) * 10_000_000 + b * B, B, nsig, tones, snr_range=(lo_snr, hi_snr),
                                        dup_fraction=workload.MIXED_DUP_FRACTION)
        dec.synth_frames(sig, B, nsig, 1.0, 777 + b + (args.seed - 123) * 100_003, iq)
        dec.waterfall_dev(iq, B, mag)
        dec.synchronize()
        for e in [e for e in chosen if e["batch"] == b]:
            m = mag[e["frame"]].cpu().numpy()
            c = np.zeros((1, 120), ft8.CAND_DTYPE)
            c[0, 0] = tuple(e["candidate"])
            per = []
            for it in range(1, 21):
                old.set_params(ldpc_iters=it)
                g = old.decode_candidates(m[None], c, np.array([1], np.int32))[0, 0]
                o = O.decode(m, c[0, :1].view(O.CAND_DTYPE), it)
                per.append([int(g["ldpc_errors"]), bytes(g["a91"]).hex(), o["ldpc_errors"], o["a91"].hex()])
            assert any(r[0] != r[2] or r[1] != r[3] for r in per), "the old library agrees with the oracle here: wrong library?"
            mags.append(m); cands.append(np.array(e["candidate"], np.int32))
            cases.append({"batch": b, "frame": e["frame"], "candidate": e["candidate"], "by_max_iterations": per})
    np.savez_compressed(os.path.join(args.out, "sqrt_ulp_cases.npz"), mag=np.array(mags, np.uint8), cand=np.array(cands, np.int32))
    json.dump({"what": "candidates of the soak sequence (seed %d, mixed traffic) whose status record differed from the oracle while the LLR scale factor's "
                       "root was HIP's 1-ulp __fsqrt_rn; per max_iterations 1..20: [old-library ldpc_errors, old-library a91, oracle ldpc_errors, oracle "
                       "a91], the old library being round 4's shipped libft8gpu.so; regenerate with tools/make_sqrt_fixture.py on a GPU box" % args.seed,
               "cases": cases}, open(os.path.join(args.out, "sqrt_ulp_cases.json"), "w"))
    print("fixture written:", len(cases), "cases", [(c["batch"], c["frame"]) for c in cases])


if __name__ == "__main__":
    main()
