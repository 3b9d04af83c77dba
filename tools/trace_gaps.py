#!/usr/bin/env python3
"""Timeline of the last bench steps from a rocprofv3 --kernel-trace CSV: kernels in start order with HSA queue / stream
ids (used to see which of the context's streams share a hardware queue, and where the RCCL exchange of --force-dist
costs time).
  rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 3 --warmup 2 [--force-dist] ...
  python tools/trace_gaps.py out/t_kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])


def nm(n):
    m = re.search(r"(ft8_\w+|rccl\w*|nccl\w*|copyBuffer|elementwise\w*)", n)
    return m.group(1) if m else n[:40]


wf = [i for i, r in enumerate(rows) if "waterfall" in r["Kernel_Name"]]
start = wf[-4] if len(wf) >= 4 else 0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id')} s{r.get('Stream_Id')} {nm(r['Kernel_Name'])}")
