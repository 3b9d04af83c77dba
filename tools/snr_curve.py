#!/usr/bin/env python3
"""Closed-loop sanity of the restated decoder: decode probability of one isolated FT8 signal
per frame versus SNR (2500 Hz reference bandwidth, AWGN), generated and decoded on the GPU."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B = 2048
    msgs, tones = workload.message_pool()
    dec = ft8.Decoder(device=0, max_frames=B)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    out = {}
    for snr in range(-26, -9):
        sig, picks = workload.frame_signals(5_000_000 + (snr + 30) * B, B, 1, tones, snr_range=(snr, snr))
        dec.synth_frames(sig, B, 1, 1.0, 4242 + snr, iq)
        spots.zero_()
        torch.cuda.synchronize()                     # the fill runs on torch's stream, the decoder on its own
        dec.decode_batch_dev(iq, B, spots, nres)
        dec.synchronize()
        g = spots.cpu().numpy().view(ft8.RESULT_DTYPE).reshape(B, 50)
        gn = nres.cpu().numpy()
        ok = false = 0
        for f in range(B):
            want = msgs[picks[f, 0]].split()[1]
            calls = [x["call"].decode() for x in g[f][:gn[f]] if x["call"]]
            ok += want in calls
            false += sum(1 for c in calls if c != want)
        out[snr] = (round(ok / B, 4), false)
        print(snr, out[snr], flush=True)
    print(json.dumps({"frames_per_point": B, "decode_probability_and_false_decodes_by_snr_db": out}))


if __name__ == "__main__":
    main()
