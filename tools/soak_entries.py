#!/usr/bin/env python3
"""Mass parity run of the OTHER ways into the library than the device-pointer batch entry tools/soak_parity.py walks:

  --entry host    ft8gpu_decode_batch with host buffers (512-frame upload chunks on a copy stream, records staged back): ragged batch
                  sizes around the chunk and pipeline thresholds, candidate caps over the whole accepted range, the caller's record
                  array pre-filled with 0xA5 -- every frame against the oracle
  --entry dropin  the reference's own symbol, ft8_subsystem (rtlsdr_ft8d.h:164), one frame per call on the process-global
                  single-frame context, caller's records pre-filled -- every frame against the oracle
  --entry ft8lib  the ft8_lib-level symbols as rtlsdr_ft8d.c:1450 / :1476 call them, but with call patterns the reference never
                  produces: ft8_find_sync at random caps / thresholds, then ft8_decode for candidates of the list in random order
                  and at changing iteration counts, for candidates that are NOT in the list, after the waterfall was rewritten in
                  place, with ft8_subsystem calls (same global context, other parameters) in between.  ft8_decode must stay the
                  pure function of (waterfall bytes, candidate, max_iterations) that upstream's is: every answer against the oracle.

  --entry multi   the multi-GPU entries with several contexts on the one GPU of the box: ft8gpu_decode_batch_multi (one host array cut
                  into contiguous shards) and ft8gpu_decode_batch_multi_dev (device-resident shards of ANY sizes, empty ones included),
                  1 ... 6 contexts, ragged totals, patterned caller records -- every frame against the oracle

  --entry report  ft8gpu_pskreporter_datagrams (postSpots' datagram bytes, rtlsdr_ft8d.c:365-590) on random record lists -- stale caller
                  bytes, unterminated fields, 0 ... 50 spots, the 1200-byte cut, per-frame times -- every datagram against the oracle

usage: tools/soak_entries.py --entry host|dropin|ft8lib|multi|report [--frames N] [--seed S]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def synth(ft8, workload, dec, torch, first, n, nsig, snr, tones, seed_off, edges):
    sig, _ = workload.frame_signals(first, n, nsig, tones, snr_range=snr, dup_fraction=workload.MIXED_DUP_FRACTION,
                                    **(dict(f_range=(-20.0, 1620.0), dt_range=(-1.5, 3.0)) if edges else {}))
    iq = torch.empty((n, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    dec.synth_frames(sig, n, nsig, 1.0, workload.SEED_BASE + seed_off, iq, first_frame=first)
    dec.synchronize()
    return iq.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entry", choices=("host", "dropin", "ft8lib", "multi", "report"), required=True)
    ap.add_argument("--frames", type=int, default=20000, help="frames in total (ft8lib: waterfalls)")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    from bench import usable_cores
    cores = usable_cores()
    build_id = ft8.check_build_id()
    rng = np.random.default_rng(args.seed)
    _, tones = workload.message_pool(traffic="mixed")
    FILL = 0xA5
    out = {"entry": args.entry, "seed": args.seed, "build_id": build_id}
    t0 = time.time()
    gen = ft8.Decoder(device=0, max_frames=4096)              # synthesises the frames (and, for `host`, decodes them)
    done = bad = msgs = 0

    if args.entry == "host":
        sizes, caps = [], {}
        b = 0
        while done < args.frames:
            n = int(rng.choice([511, 512, 513, 1023, 1024, 1025, 1536, 2048, 2049])) if rng.integers(0, 4) == 0 else int(round(float(np.exp(rng.uniform(0.0, np.log(3000.0))))))
            n = max(1, min(n, 4096))
            cap = int(rng.choice([1, 3, 4, 5, 33, 120, 120, 120, 480, 1024]))
            min_score = int(rng.choice([10, 10, 0, 20]))
            iters = int(rng.choice([20, 20, 5, 50]))
            nsig = int(rng.integers(0, 45))
            iq = synth(ft8, workload, gen, torch, 3_000_000 + args.seed * 10_000_000 + b * 4096, n, nsig, (-20.0, 0.0), tones, 200 + b, bool(rng.integers(0, 2)))
            gen.set_params(min_score=min_score, max_candidates=cap, ldpc_iters=iters)
            start = np.full((n, 1400), FILL, np.uint8).view(ft8.RESULT_DTYPE).reshape(n, 50)
            d, k = gen.decode_batch(iq, decodes=start.copy())
            rdec, rn = O.subsystem_batch(iq, O.default_params(min_score, cap, iters), cores, decodes=start.copy().view(O.RESULT_DTYPE).reshape(n, 50))
            mism = int(sum(1 for f in range(n) if k[f] != rn[f] or d[f].tobytes() != rdec[f].tobytes()))
            bad += mism; done += n; msgs += int(k.sum()); sizes.append(n); caps[cap] = caps.get(cap, 0) + n; b += 1
            if mism:
                print(f"batch {b}: {n} frames cap {cap} min_score {min_score} iters {iters}: {mism} frames differ", flush=True)
        out.update({"frames": done, "batches": b, "messages": msgs, "mismatching_frames": bad, "batch_sizes_min_median_max": [min(sizes), int(np.median(sizes)), max(sizes)],
                    "frames_by_cap": {str(c): v for c, v in sorted(caps.items())}})

    elif args.entry == "dropin":
        b = 0
        while done < args.frames:
            n = min(2048, args.frames - done)
            nsig = int(rng.integers(0, 45))
            iq = synth(ft8, workload, gen, torch, 5_000_000 + args.seed * 10_000_000 + b * 4096, n, nsig, (-20.0, 0.0), tones, 400 + b, bool(rng.integers(0, 2)))
            start = np.full((n, 1400), FILL, np.uint8).view(ft8.RESULT_DTYPE).reshape(n, 50)
            rdec, rn = O.subsystem_batch(iq, O.default_params(10, 120, 20), cores, decodes=start.copy().view(O.RESULT_DTYPE).reshape(n, 50))
            for f in range(n):
                d, k = ft8.ft8_subsystem(np.ascontiguousarray(iq[f, 0]), np.ascontiguousarray(iq[f, 1]), decodes=start[f].copy())
                if int(k) != int(rn[f]) or d.tobytes() != rdec[f].tobytes():
                    bad += 1
                msgs += int(k)
            done += n; b += 1
        out.update({"frames": done, "messages": msgs, "mismatching_frames": bad})

    elif args.entry == "report":
        import test_report as TR
        longest = 0
        while done < args.frames:
            F = int(rng.choice([1, 63, 64, 65, 1000, 4096]))
            d, n = TR._random_lists(O, rng, max(F, 4))
            d, n = d[:F], n[:F]
            times = rng.integers(0, 2**32, F, dtype=np.uint64).astype(np.uint32)
            info = TR._info(ft8.ReportInfo, rcall=b"VE2XYZ/QRP12", rloc=b"FN35ab", app=b"rtlsdr-ft8d_v0.3.6", dial=int(rng.integers(0, 2**32)))
            dial = info.dial_freq
            outb, lens = gen.pskreporter_datagrams(d.view(ft8.RESULT_DTYPE), n, info, times)
            for f in range(F):
                oi = TR._info(O.ReportInfo, rcall=b"VE2XYZ/QRP12", rloc=b"FN35ab", app=b"rtlsdr-ft8d_v0.3.6", dial=dial, now=int(times[f]))
                want = O.pskreporter_datagram(d[f], max(0, int(n[f])), oi)
                if lens[f] != want.size or outb[f, :lens[f]].tobytes() != want.tobytes() or outb[f, lens[f]:].any():
                    bad += 1
                longest = max(longest, int(want.size))
            done += F
        out.update({"frames": done, "mismatching_frames": bad, "longest_datagram_bytes": longest})

    elif args.entry == "multi":
        P = 1024
        pool = [ft8.Decoder(device=0, max_frames=P) for _ in range(6)]
        b = 0
        shapes = {}
        while done < args.frames:
            ndev = int(rng.integers(1, 7))
            decs = pool[:ndev]
            cap = int(rng.choice([2, 33, 120, 120, 480]))
            iters = int(rng.choice([20, 20, 7]))
            for d_ in decs:
                d_.set_params(min_score=10, max_candidates=cap, ldpc_iters=iters)
            nsig = int(rng.integers(0, 40))
            prm = O.default_params(10, cap, iters)
            if rng.integers(0, 2) == 0:                                      # one host array, shards cut by the library
                n = int(rng.integers(0, ndev * P + 1)) if rng.integers(0, 6) else int(rng.integers(0, ndev + 2))
                n = max(n, 1)
                iq = np.concatenate([synth(ft8, workload, gen, torch, 9_000_000 + args.seed * 10_000_000 + b * 8192 + lo, min(4096, n - lo), nsig, (-19.0, 0.0), tones, 800 + b, False)
                                     for lo in range(0, n, 4096)])
                start = np.full((n, 1400), FILL, np.uint8).view(ft8.RESULT_DTYPE).reshape(n, 50)
                d, k = ft8.decode_batch_multi(decs, iq, decodes=start.copy())
                form = "host array"
            else:                                                            # device-resident shards of any sizes
                counts = [int(rng.integers(0, P + 1)) if rng.integers(0, 5) else 0 for _ in range(ndev)]
                if sum(counts) == 0:
                    counts[int(rng.integers(0, ndev))] = int(rng.integers(1, P + 1))
                n = sum(counts)
                iq = np.concatenate([synth(ft8, workload, gen, torch, 9_000_000 + args.seed * 10_000_000 + b * 8192 + lo, min(4096, n - lo), nsig, (-19.0, 0.0), tones, 800 + b, False)
                                     for lo in range(0, n, 4096)])
                devs, lo = [], 0
                for c_ in counts:
                    devs.append(torch.from_numpy(iq[lo:lo + c_]).cuda() if c_ else None)
                    lo += c_
                torch.cuda.synchronize()
                start = np.full((n, 1400), FILL, np.uint8).view(ft8.RESULT_DTYPE).reshape(n, 50)
                d, k = ft8.decode_batch_multi_dev(decs, devs, counts, decodes=start.copy())
                form = "device shards"
            rdec, rn = O.subsystem_batch(iq, prm, cores, decodes=start.copy().view(O.RESULT_DTYPE).reshape(n, 50))
            mism = int(sum(1 for f in range(n) if k[f] != rn[f] or d[f].tobytes() != rdec[f].tobytes()))
            if mism:
                print(f"call {b}: {form}, {ndev} contexts, {n} frames, cap {cap}: {mism} frames differ", flush=True)
            bad += mism; done += n; msgs += int(k.sum()); b += 1
            shapes[f"{form} x{ndev}"] = shapes.get(f"{form} x{ndev}", 0) + 1
        for d_ in pool:
            d_.close()
        out.update({"frames": done, "calls": b, "messages": msgs, "mismatching_frames": bad, "calls_by_form_and_contexts": dict(sorted(shapes.items()))})

    else:
        class Waterfall(C.Structure):
            _fields_ = [("max_blocks", C.c_int), ("num_blocks", C.c_int), ("num_bins", C.c_int), ("time_osr", C.c_int),
                        ("freq_osr", C.c_int), ("mag", C.c_void_p), ("block_stride", C.c_int), ("protocol", C.c_int)]

        class Message(C.Structure):
            _fields_ = [("text", C.c_char * 25), ("hash", C.c_uint16)]

        class Status(C.Structure):
            _fields_ = [("ldpc_errors", C.c_int), ("crc_extracted", C.c_uint16), ("crc_calculated", C.c_uint16), ("unpack_status", C.c_int)]

        L = ft8.load_library()
        L.ft8_find_sync.argtypes = [C.POINTER(Waterfall), C.c_int, C.c_void_p, C.c_int]
        L.ft8_decode.argtypes = [C.POINTER(Waterfall), C.c_void_p, C.POINTER(Message), C.c_int, C.POINTER(Status)]
        L.ft8_decode.restype = C.c_bool
        calls = lists_bad = ok_count = not_in_list = after_rewrite = interleaved = 0
        kinds = {"list": 0, "other_iters": 0, "not_in_list": 0, "after_rewrite": 0, "after_subsystem": 0}

        def check(wf, mag, cand, iters):
            nonlocal calls, bad, ok_count
            m, st = Message(), Status()
            ok = bool(L.ft8_decode(C.byref(wf), cand.ctypes.data, C.byref(m), iters, C.byref(st)))
            r = O.decode(mag, cand, iters)
            good = ok == r["ok"] and st.ldpc_errors == r["ldpc_errors"]
            if good and r["ldpc_errors"] == 0:
                good = (st.crc_extracted, st.crc_calculated) == (r["crc_extracted"], r["crc_calculated"])
            if good and r["ok"]:
                good = m.text.decode() == r["text"] and m.hash == r["hash"] and st.unpack_status == r["unpack_status"]
            calls += 1
            ok_count += int(ok)
            bad += 0 if good else 1
            return good

        b = 0
        while done < args.frames:
            n = min(512, args.frames - done)
            nsig = int(rng.integers(3, 40))
            iq = synth(ft8, workload, gen, torch, 7_000_000 + args.seed * 10_000_000 + b * 4096, n, nsig, (-19.0, 0.0), tones, 600 + b, bool(rng.integers(0, 2)))
            mags = O.waterfall_batch(iq, False, cores)
            for f in range(n):
                mag = mags[f].copy()
                wf = Waterfall(0, 92, 256, 2, 2, mag.ctypes.data, 1024, 1)
                cap = int(rng.choice([1, 2, 7, 33, 120, 120, 120, 480]))
                min_score = int(rng.choice([10, 10, 10, 0, 25]))
                heap = np.zeros(cap, ft8.CAND_DTYPE)
                k = L.ft8_find_sync(C.byref(wf), cap, heap.ctypes.data, min_score)
                ref = O.find_sync(mag, cap, min_score)
                if k != len(ref) or not np.array_equal(heap[:k], ref):
                    lists_bad += 1
                    continue
                if k == 0:
                    continue
                iters = 20
                order = rng.permutation(k)[:int(rng.integers(1, min(k, 24) + 1))]          # some of the list, in any order
                for j, c in enumerate(order):
                    roll = rng.integers(0, 40)
                    if roll == 0:                                                       # another iteration count from here on
                        iters = int(rng.choice([1, 5, 20, 50])); kinds["other_iters"] += 1
                    elif roll == 1:                                                     # a candidate the list does not hold
                        other = heap[c:c + 1].copy()
                        other["freq_offset"] = (int(other["freq_offset"][0]) + 1) % 249
                        other["time_sub"] ^= 1
                        check(wf, mag, other, iters); kinds["not_in_list"] += 1
                    elif roll == 2:                                                     # the caller rewrites the waterfall in place
                        lo = int(rng.integers(0, mag.size - 4096))
                        mag[lo:lo + 4096] = rng.integers(0, 256, 4096, dtype=np.uint8)
                        kinds["after_rewrite"] += 1
                    elif roll == 3:                                                     # the drop-in symbol in between (same global context)
                        g = int(rng.integers(0, n))
                        ft8.ft8_subsystem(np.ascontiguousarray(iq[g, 0]), np.ascontiguousarray(iq[g, 1]))
                        kinds["after_subsystem"] += 1
                    check(wf, mag, heap[c:c + 1], iters); kinds["list"] += 1
            done += n; b += 1
        out.update({"waterfalls": done, "ft8_decode_calls": calls, "ft8_decode_true": ok_count, "answers_differing_from_the_oracle": bad, "candidate_lists_differing": lists_bad,
                    "call_kinds": kinds, "mismatching_frames": bad + lists_bad})
    gen.close()
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out))
    return 1 if out.get("mismatching_frames") else 0


if __name__ == "__main__":
    sys.exit(main())
