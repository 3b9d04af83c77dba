#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch) and derive the HBM
traffic file bench.py reads.

usage: tools/pmc_summary.py [--traffic profiles/pmc_traffic.json --frames 4096] gpurun_out/pmc/*/*_counter_collection.csv

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE in separate passes, units
of KB; on gfx950 FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced stream, so it is doubled for
the kernels whose reads are such streams (waterfall, sync, rx_block) -- checked on a known byte count: the
waterfall kernel requests 23 x 2816 x 8 B x 4096 frames = 2.12 GB per launch and the doubled counter reads
2.12 GB.  Byte/dword gathers (decode, heap, spots) are an uncalibrated pattern: the raw counter is kept."""
import argparse
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    """identity of the kernel sources the counters were collected on: the device half of the library's build id
    (bench.py recomputes it and reports the PMC figures only while it matches)"""
    import sys
    sys.path.insert(0, ROOT)
    import rtlsdr_ft8d_amd as ft8
    return ft8.device_source_id()

KERNELS = ("ft8_decode_kernel", "ft8_waterfall_kernel", "ft8_sync_kernel", "ft8_heap_simt_kernel", "ft8_heap_kernel", "ft8_spots_kernel",
           "ft8_synth_kernel", "ft8_rx_block_kernel")
WIDE = {"waterfall", "sync", "rx_block"}
PER_FRAME = {"waterfall", "sync", "heap", "heap_simt", "decode", "spots"}       # kernels whose launches cover a number of frames (read off the sync kernel's grid)
SYNC_THREADS_PER_FRAME = 4 * 2 * 512                                            # ft8_sync_kernel: 4 segments x 2 halves x 512 threads per frame


def short(name):
    for k in KERNELS:
        if k in name:
            return k.replace("ft8_", "").replace("_kernel", "")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--traffic")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--no-overlap", action="store_true", help="(ignored since round 4: the frames a launch covers are read off the sync kernel's grid size)")
    ap.add_argument("--config-key", default=None, help="store the figures under this key of an EXISTING traffic file (e.g. config4) instead of at its top level")
    ap.add_argument("--command", default="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs")
    args = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    sync_grids = []
    for p in args.paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if k == "sync":
                    sync_grids.append(int(row["Grid_Size"]))
    # Frames per launch.  rocprofv3's counter passes serialise kernels, so the context's co-execution probe
    # (ft8gpu_overlap_active) reports that its streams do not run side by side and the PLAIN pipeline runs: one launch per
    # stage for the whole batch.  Without that (older builds, other tools) the two-part pipeline gives two launches per
    # batch.  Either way the mean number of frames per launch is what the sync kernel's grid says.
    frames_per_launch = int(round(sum(sync_grids) / len(sync_grids) / SYNC_THREADS_PER_FRAME)) if sync_grids else args.frames
    out = {}
    for k, d in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in d.items()}
        out[k]["dispatches"] = max(len(v) for v in d.values())
    print(json.dumps(out, indent=1))
    if args.traffic:
        t = {"csrc_sha": csrc_hash(),
             # which pipeline the counted run took: "plain" = one launch per stage and batch (the profiler serialises
             # kernels, so the co-execution probe falls back), "two-part" = the product's overlapped form
             "pipeline_form": "plain" if frames_per_launch >= args.frames else "two-part",
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                       f"`{args.command}`, mean per dispatch; tools/gpu_round.sh + tools/pmc_summary.py"}
        for k, v in out.items():
            if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or k == "synth":
                continue
            f, w = v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
            frames = frames_per_launch if k in PER_FRAME else args.frames
            hbm = int((2 * f if k in WIDE else f) + w)
            t[k] = {"fetch_size_kb_raw": round(v["FETCH_SIZE"], 1), "write_size_kb_raw": round(v["WRITE_SIZE"], 1),
                    "fetch_correction": "x2 (16 B/lane coalesced stream)" if k in WIDE else "x1 (byte/dword gathers: uncalibrated pattern, raw counter)",
                    "hbm_bytes_per_launch": hbm, "frames_per_launch": frames, "hbm_bytes_per_frame": round(hbm / frames, 1),
                    "write_bytes_per_frame": round(w / frames, 1)}
            if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                # The two counters come from different passes (= different runs of the command), so on a saturated pipe the quotient
                # can read a hair above 1: the fraction is stored clamped, the raw quotient beside it.
                raw = round(v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * v["GRBM_GUI_ACTIVE"] / 8), 3)
                t[k]["valu_busy_frac"] = min(1.0, raw)
                t[k]["valu_busy_frac_raw"] = raw
            for cn in ("SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
                if cn in v:
                    t[k][cn] = v[cn]
            if "SQ_INSTS_VALU" in v:
                t[k]["valu_instructions_per_frame"] = round(v["SQ_INSTS_VALU"] / frames, 1)
        if args.config_key:
            whole = json.load(open(args.traffic))
            if whole.get("csrc_sha") != t["csrc_sha"]:
                raise SystemExit("the traffic file was collected on other kernel sources")
            whole[args.config_key] = {k: v for k, v in t.items() if k != "csrc_sha"}
            t = whole
        json.dump(t, open(args.traffic, "w"), indent=1)


if __name__ == "__main__":
    main()
