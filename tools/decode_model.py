#!/usr/bin/env python3
"""Cycle model of the LDPC kernel against its measured time (run on the GPU box).

  python tools/decode_model.py [--census profiles/r03_isa_census.json] > profiles/r03_decode_model.json

Inputs, all from this run except the census (compiler output, tools/isa_census.py):
  * BP iterations entered by every candidate of the bench batch (ft8gpu_decode_candidates in the pipeline form of the
    kernel: the `iters` field) and how many candidates end as codewords;
  * VALU issue slots per iteration / per prologue from the ISA census (fast stream; the IEEE stream runs on about
    2 % of the iterations and is priced with its own count);
  * the time of one issue slot on THIS box at THIS moment: tools/ubench/valu_rate (v_mul_f32 wave-instructions per
    second over all 1024 SIMDs), run right before the decode so that both see the same DVFS state;
  * the measured time of the LDPC kernel in the batch pipeline (hipEvents, 20 runs) and the shader clock / socket
    power while it runs (sysfs, 5 ms period).
Model: every SIMD issues one slot per t_slot; the kernel's work is sum over candidates of
(prologue + iterations x per-iteration slots [+ epilogue for codewords]); time = work / 1024 SIMDs x t_slot.
The ratio measured / model is the VALU-issue utilisation the kernel would need for the model to be exact; it is
compared with the PMC figure (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles) of profiles/pmc_traffic.json."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def slot_rate():
    """(wave64 v_mul_f32 per second per SIMD, relative costs, median shader clock while it ran) from the micro-benchmark"""
    from bench import ClockSampler
    with ClockSampler(0) as clk:
        out = subprocess.run([os.path.join(ROOT, "tools", "ubench", "valu_rate")], capture_output=True, text=True, timeout=120).stdout
    slot_rate.sclk.append(clk.summary().get("sclk_mhz_median"))
    rate = lambda name: float(re.search(name + r"\s*:\s*[\d.]+ ms\s+([\d.]+) Gop/s", out).group(1))
    mul = rate("v_mul_f32")
    g = mul * 1e9 / 64 / 1024
    return g, {"pk": mul / rate("v_pk_mul_f32"), "rcp": mul / rate("v_rcp_f32"), "fma": mul / rate("v_fma_f32"), "vop3": mul / rate("v_bfi_b32")}


slot_rate.sclk = []

FMA_OPS = ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32")
VOP3_OPS = ("v_bfi_b32", "v_min3_", "v_max3_", "v_med3_", "v_perm_b32", "v_alignbit", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_lshl_or_b32")


def price(opcodes, cost):
    """VALU issue slots of an opcode histogram (one slot = one v_mul_f32)"""
    total = 0.0
    for op, k in opcodes.items():
        if op.startswith("v_pk_"):
            c = cost["pk"]
        elif op.startswith("v_rcp_"):
            c = cost["rcp"]
        elif op.startswith(FMA_OPS):
            c = cost["fma"]
        elif op.startswith(VOP3_OPS):
            c = cost["vop3"]
        else:
            c = 1.0
        total += c * k
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--census", default=os.path.join(ROOT, "profiles", "r03_isa_census.json"))
    ap.add_argument("--frames", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    from bench import ClockSampler
    cen = json.load(open(args.census))
    B = args.frames
    dec = ft8.Decoder(device=0, max_frames=B)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, 20, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
    # ---- per-candidate iteration counts (pipeline form of the kernel, device-resident stage entries)
    mag = torch.empty((B, ft8.MAG_ARRAY), dtype=torch.uint8, device="cuda")
    cands = torch.zeros((B, 120, 8), dtype=torch.uint8, device="cuda")
    counts = torch.zeros((B,), dtype=torch.int32, device="cuda")
    status = torch.zeros((B, 120, 48), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    L = dec.lib
    D = ft8.DEVICE_PTRS
    ft8._check(L.ft8gpu_waterfall(dec.h, iq.data_ptr(), B, mag.data_ptr(), D))
    ft8._check(L.ft8gpu_find_sync(dec.h, mag.data_ptr(), B, cands.data_ptr(), counts.data_ptr(), D))
    dec.set_debug_flags(ft8.DBG_PIPELINE_FORM)
    ft8._check(L.ft8gpu_decode_candidates(dec.h, mag.data_ptr(), cands.data_ptr(), counts.data_ptr(), B, status.data_ptr(), D))
    dec.synchronize()
    dec.set_debug_flags(0)
    st = status.cpu().numpy().view(ft8.STATUS_DTYPE).reshape(B, 120)
    n = counts.cpu().numpy()
    valid = np.arange(120)[None, :] < n[:, None]
    iters = st["iters"][valid].astype(np.int64)
    codeword = (st["ldpc_errors"][valid] == 0)
    # an iteration "entered" runs the hard decision; the message update (the expensive part) runs unless the loop
    # ends there: iterations with a full update = iters for candidates that stop on a codeword at `iters`
    # (the stop happens before the update), iters - 1 ... see decode.hip: the last iteration skips the update
    full_updates = np.where(codeword, iters, np.maximum(iters - 1, 0))
    hist = np.bincount(iters, minlength=21)[:21]

    # ---- slot time, then the measured kernel, back to back
    spots = torch.zeros((B, 1400), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(5):
        dec.decode_batch_dev(iq, B, spots, nres)
    dec.synchronize()
    rate, cost = slot_rate()
    dec.enable_timing(True)
    with ClockSampler(0) as clk:
        clk_period = 0.005
        t0 = time.perf_counter()
        for _ in range(100):
            dec.decode_batch_dev(iq, B, spots, nres)
        dec.synchronize()
        wall = time.perf_counter() - t0
    t = dec.timings()
    dec.enable_timing(False)
    rate2, cost2 = slot_rate()
    cost = {k: 0.5 * (cost[k] + cost2[k]) for k in cost}
    pk_cost, rcp_cost = cost["pk"], cost["rcp"]
    t_slot = 2.0 / (rate + rate2)

    per_iter = cen["per_iteration_fast_path"]
    s_iter = price(cen["per_iteration_valu_opcodes"], cost)
    pro = cen["prologue_straight_line"]
    s_pro = pro.get("valu", 0) + pk_cost * pro.get("valu_pk", 0) + rcp_cost * pro.get("trans", 0)
    # epilogue: status record for everybody (about 60 VALU: the 48-byte record is assembled by lane 0); codewords add the
    # CRC reduction and unpack77, which runs on the scalar unit (2 000 static SALU instructions, outside this model)
    s_epi = 60.0
    ncand = int(valid.sum())
    work = ncand * (s_pro + s_epi) + float(full_updates.sum()) * s_iter + float((iters - full_updates).sum()) * 45.0   # hard decision + screen only
    model_ms = work / 1024.0 * t_slot * 1e3
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("decode", {})
    except (OSError, ValueError):
        pass
    # the slot time was measured on a lighter kernel that clocks higher (the LDPC kernel is power-limited): scale it by
    # the ratio of the shader clocks sampled during the two (sysfs), when both are available
    f_dec = clk.summary().get("sclk_mhz_median")
    f_ub = [f for f in slot_rate.sclk if f]
    clock_ratio = (sum(f_ub) / len(f_ub)) / f_dec if (f_dec and f_ub) else None
    out = {
        "batch": {"frames": B, "candidates": ncand, "codewords": int(codeword.sum()),
                  "bp_iterations_entered": int(iters.sum()), "iterations_with_message_update": int(full_updates.sum()),
                  "mean_iterations_per_candidate": round(float(iters.mean()), 3),
                  "share_running_all_20": round(float((iters >= 20).mean()), 4), "iterations_histogram_0_20": hist.tolist()},
        "slot": {"v_mul_f32_wave_instr_per_s_per_simd": round((rate + rate2) / 2, 1), "ns": round(t_slot * 1e9, 4),
                 "cost_in_slots_measured": {k: round(v, 3) for k, v in cost.items()},
                 "source": "tools/ubench/valu_rate run before and after the timed decode (mean)"},
        "census": {"file": os.path.relpath(args.census, ROOT), "slots_per_iteration": round(s_iter, 1), "slots_prologue": round(s_pro, 1),
                   "slots_epilogue_assumed": s_epi, "per_iteration": per_iter, "per_iteration_valu_opcodes": cen["per_iteration_valu_opcodes"]},
        "model_ms_at_100pct_valu_issue": round(model_ms, 4),
        "measured_decode_ms": round(t["decode_ms"], 4), "measured_step_ms": round(1e3 * wall / 100, 4),
        "implied_valu_issue_utilisation": round(model_ms / t["decode_ms"], 4),
        "pmc_valu_busy_frac": pmc.get("valu_busy_frac"),
        "model_with_pmc_busy_ms": round(model_ms / pmc["valu_busy_frac"], 4) if pmc.get("valu_busy_frac") else None,
        "model_error_vs_measured": round(model_ms / pmc["valu_busy_frac"] / t["decode_ms"] - 1.0, 4) if pmc.get("valu_busy_frac") else None,
        "clock_ratio_slot_benchmark_over_ldpc_kernel": round(clock_ratio, 4) if clock_ratio else None,
        "model_clock_corrected_ms": round(model_ms * clock_ratio / pmc["valu_busy_frac"], 4) if (clock_ratio and pmc.get("valu_busy_frac")) else None,
        "model_clock_corrected_error_vs_measured": round(model_ms * clock_ratio / pmc["valu_busy_frac"] / t["decode_ms"] - 1.0, 4) if (clock_ratio and pmc.get("valu_busy_frac")) else None,
        "gpu_during_the_timed_loop": clk.summary(),
        "sclk_mhz_during_the_slot_benchmark": slot_rate.sclk,
    }
    dec.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
