#!/usr/bin/env python3
"""Per-launch rows of selected kernels from a rocprofv3 --kernel-trace CSV:
    python tools/trace_rows.py <..._kernel_trace.csv> <name substring> [<name substring> ...] [--last N]
Prints CSV (kernel, launch index, start offset in us, duration in us) and, per kernel, the average over all launches and
over the last N (the timed steps of a bench run whose warm-up comes first)."""
import csv
import sys

args = sys.argv[1:]
last = None
if "--last" in args:
    i = args.index("--last")
    last = int(args[i + 1])
    del args[i:i + 2]
path, subs = args[0], args[1:]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0 = min(r[1] for r in rows)
print("kernel,launch,start_us,duration_us")
for sub in subs:
    sel = sorted((r for r in rows if sub in r[0]), key=lambda r: r[1])
    d = [(e - s) / 1e3 for _, s, e in sel]
    for k, (name, s, e) in enumerate(sel):
        print(f"{sub},{k},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.2f}")
    if d:
        tail = d[-last:] if last else d
        print(f"# {sub}: {len(d)} launches, avg {sum(d) / len(d):.2f} us; last {len(tail)}: avg {sum(tail) / len(tail):.2f} us, "
              f"min {min(tail):.2f}, max {max(tail):.2f}")
