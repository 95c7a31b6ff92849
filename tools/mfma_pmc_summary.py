#!/usr/bin/env python3
"""One line per launch of the dvq matrix-core kernels from a rocprofv3 --pmc counter_collection.csv:
    python tools/mfma_pmc_summary.py <counter_collection.csv> [kernel-name substring, default "dvq::"]
duration (us) from the dispatch timestamps; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; matrix-pipe busy share =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); SQ_* wave counters in quad-cycles, shown as shares of
SQ_WAVE_CYCLES."""
import collections
import csv
import sys

want = sys.argv[2] if len(sys.argv) > 2 else "dvq::"
rows = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if want not in name or "pack" in name:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": name.split("(")[0].replace("void ", ""), "grid": r["Grid_Size"],
                                                     "t": (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("dispatch kernel grid dur_us clock_GHz mfma_busy wave_active wave_wait_any wave_wait_inst valu_insts")
for k, d in rows.items():
    dur = (d["t"][1] - d["t"][0]) / 1e3
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    wc = max(d.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{k} {d['name'][-44:]} {d['grid']} {dur:.1f} {cyc / max(dur, 1e-9) / 1e3:.3f} "
          f"{d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cyc * 1024, 1):.3f} {d.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} "
          f"{d.get('SQ_WAIT_ANY', 0) / wc:.3f} {d.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} {d.get('SQ_INSTS_VALU', 0):.0f}")
