#!/usr/bin/env python3
"""How much kernels overlap in a rocprofv3 kernel trace: python tools/overlap_check.py <kernel_trace.csv> [name substring]
Prints, for the selected kernels, the total busy time (union of intervals), the summed durations and their ratio."""
import csv
import sys

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
iv = []
with open(path) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
total = sum(e - s for s, e in iv)
print(f"{len(iv)} launches, summed durations {total / 1e6:.3f} ms, union {union / 1e6:.3f} ms, overlap factor {total / max(union, 1):.3f}")
