#!/usr/bin/env python
"""Effective shader clock of individual long dispatches from a `rocprofv3 --pmc GRBM_GUI_ACTIVE [SQ_VALU_MFMA_BUSY_CYCLES]
--kernel-trace --output-format csv` run:  clock = GRBM_GUI_ACTIVE / 8 XCDs / (end - start).
   python tools/pmc_clock.py <dir> [min_us]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, min_us=500.0):
    per = defaultdict(dict)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = per[(f, r["Dispatch_Id"])]
            d["name"] = r["Kernel_Name"][:70]
            d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            d["grid"] = r.get("Grid_Size", "")
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print("| kernel | grid | us | effective GHz | MFMA busy % |")
    print("|---|---|---|---|---|")
    for (f, did), d in sorted(per.items(), key=lambda kv: int(kv[0][1])):
        if d["us"] < min_us or "GRBM_GUI_ACTIVE" not in d:
            continue
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        busy = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024) if cyc else 0
        print("| `%s` | %s | %.1f | %.3f | %.1f |" % (d["name"], d["grid"], d["us"], cyc / d["us"] / 1e3, busy))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 500.0)
