"""Per kernel and grid size: launches, average duration and average counter values (and the value per GPU clock,
from GRBM_GUI_ACTIVE / 8 XCDs, when that counter was collected) from a rocprofv3 --pmc counter_collection.csv:
    python tools/pmc_sum.py <counter_collection.csv>"""
import collections, csv, sys
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "dvq::" not in name: continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"name": name.split("(")[0].replace("void dvq::", ""), "grid": r["Grid_Size"], "t": (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
for d in rows.values():
    a = agg.setdefault((d["name"], d["grid"]), collections.Counter())
    a["n"] += 1; a["dur"] += (d["t"][1] - d["t"][0]) / 1e3
    for k, v in d.items():
        if k not in ("name", "grid", "t"): a[k] += v
for (name, grid), a in agg.items():
    n = a["n"]
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8 / n
    print(f"{name[:40]:40s} grid {grid:>9s} n={n:4d} dur {a['dur']/n:8.1f} us  " + "  ".join(f"{k}={v/n:.3g}" + (f"({v/n/max(cyc,1):.2f}/clk)" if cyc else "") for k, v in a.items() if k not in ("n", "dur")))
