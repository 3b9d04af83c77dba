#!/usr/bin/env python3
"""Inputs of tools/pin_ft8_lib/pin_harness.c: the waterfalls (as the reference's ft8_subsystem computes them,
rtlsdr_ft8d.c:1395-1435, here by the CPU oracle) of
  * the reference's self-test frame (decoderSelfTest, :913-972),
  * the seeded multi-signal frames of tests/golden/frames.json (same seeds, same generator),
  * two frames of mixed on-air style traffic (workload.mixed_message_pool) incl. a message heard twice,
  * an all-zero frame and a frame of uniform random bytes (degenerate waterfalls),
and the message texts for the encoder side.
usage: make_inputs.py <out_dir>   ->  <out_dir>/waterfalls.bin, <out_dir>/messages.txt"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def waterfalls(light=False):
    """light: no import of the product package (mixed frames need its encoder) -- the self-test and golden frames only"""
    import oracle_lib as O
    import synth_util as S
    enc = S.oracle_encode_fn(O)
    mags = []
    i, q = O.selftest_signal(1)
    mags.append(O.waterfall(i, q))
    with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as f:
        golden = json.load(f)
    for fr in golden["frames"]:
        iq, _ = S.make_frame(fr["seed"], fr["nsig"], enc, snr_range=tuple(fr.get("snr_range", (-18.0, 0.0))))
        mags.append(O.waterfall(iq[0], iq[1]))
    if not light:
        from rtlsdr_ft8d_amd import workload
        texts, tones = workload.message_pool(traffic="mixed")
        for seed, nsig, snr in ((17, 20, (-16, 0)), (18, 45, (-20, -4))):          # the mixed frames of tests/golden/mixed.json
            iq, _ = S.make_mixed_frame(seed, nsig, snr, texts, tones)
            mags.append(O.waterfall(iq[0], iq[1]))
    mags.append(np.zeros(94208, np.uint8))
    mags.append(np.random.default_rng(5).integers(0, 256, 94208, dtype=np.uint8))
    return np.stack(mags)


MESSAGES = ["CQ K1JT FN20QI", "CQ K1JT FN20", "K1ABC W9XYZ EN37", "K1ABC W9XYZ -11", "K1ABC W9XYZ R-09", "K1ABC W9XYZ +05", "K1ABC W9XYZ RRR",
            "K1ABC W9XYZ RR73", "K1ABC W9XYZ 73", "K1ABC W9XYZ", "QRZ DL1ABC JO62", "DE 9A1A JN75", "CQ 3DA0XY KG53"]
# (free text is left out on purpose: upstream's pack77 falls back to it for anything it cannot pack as a standard message,
#  the oracle's restatement stops at the standard message the self-test needs, rtlsdr_ft8d.c:927)


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    waterfalls("--light" in sys.argv).tofile(os.path.join(out, "waterfalls.bin"))
    with open(os.path.join(out, "messages.txt"), "w") as f:
        f.write("\n".join(MESSAGES) + "\n")


if __name__ == "__main__":
    main()
