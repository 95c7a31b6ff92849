#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel: python tools/pmc_summary.py <counter_collection.csv> [...]
Prints KiB per launch for FETCH_SIZE / WRITE_SIZE (rocprofv3 reports them in KiB... see the header of the output)."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            agg[r["Counter_Name"]][name].append(float(r["Counter_Value"]))
    for counter, kernels in sorted(agg.items()):
        for name, vals in sorted(kernels.items()):
            if "dvq::" in name:
                print(f"{counter:11s} {name[-52:]:52s} launches={len(vals):4d} avg={sum(vals) / len(vals):14.1f}")
