#!/usr/bin/env python3
"""effective shader clock and MFMA-busy share per kernel from a rocprofv3 --pmc pass
   (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES): tools/clock_probe.py <counter_collection.csv>
   clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
   (4 SIMDs x 256 CUs x clock x duration)."""
import collections
import csv
import sys

rows = collections.defaultdict(dict)
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Dispatch_Id"]
    rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[k] = (r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for k, c in rows.items():
    name, grid, wg, dur = meta[k]
    if "igemm_kernel" not in name and "wgrad_kernel" not in name:
        continue
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / dur      # GHz (cycles per ns)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * c.get("GRBM_GUI_ACTIVE", 1) / 8) if c.get("GRBM_GUI_ACTIVE") else 0
    agg[(name[name.find("<"):name.find(">") + 1], grid // wg)].append((dur / 1e3, clk, mf))
for (name, wgs), v in sorted(agg.items()):
    v.sort()
    m = v[len(v) // 2]
    print(f"{name:48s} wgs {wgs:5d} n {len(v):3d}  median {m[0]:8.1f} us  clock {m[1]:5.2f} GHz  mfma-busy {100 * m[2]:5.1f} %")
