#!/usr/bin/env python3
"""print value and stage_ms of the bench JSON line found in gpurun_out/.last_call.json"""
import json, re, sys
d = json.load(open("gpurun_out/.last_call.json"))
t = json.dumps(d)
m = re.search(r'\\"value\\": ([0-9.]+)', t); s = re.search(r'\\"stage_ms\\": (\{[^}]*\})', t)
print(m.group(1) if m else None, s.group(1).replace('\\"', '') if s else None)
