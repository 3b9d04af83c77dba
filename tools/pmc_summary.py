#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch).
usage: tools/pmc_summary.py gpurun_out/pmc/*/ *_counter_collection.csv"""
import csv
import collections
import json
import sys


def short(name):
    for k in ("ft8_decode_kernel", "ft8_waterfall_kernel", "ft8_sync_kernel", "ft8_heap_kernel", "ft8_spots_kernel", "ft8_synth_kernel"):
        if k in name:
            return k.replace("ft8_", "").replace("_kernel", "")
    return None


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, d in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in d.items()}
        out[k]["dispatches"] = max(len(v) for v in d.values())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
