#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family.
   python tools/pmc_summary.py gpurun_out/pmc > profiles/rNN_pmc.md
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); on gfx950 FETCH_SIZE under-reports wide coalesced
streams by exactly 2x (MI355X_MICROARCH.md, HBM section), so the corrected read bytes = 2 x FETCH_SIZE."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"vc::(\w+)", name)
    base = m.group(1) if m else name.split("(")[0][:40]
    if "conv_kernel" in name:
        k = re.search(r">,\s*(\d)\s*>\(", name)
        kinds = {"0": "fwd", "1": "dgrad", "2": "wgrad"}
        return "conv_kernel<%s>" % kinds.get(k.group(1) if k else "?", "?")
    return base


def main(root):
    data = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    dur = defaultdict(float)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            data[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen and r["Counter_Name"] in ("FETCH_SIZE",):
                seen.add(key)
                calls[fam] += 1
                dur[fam] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    print("| kernel family | launches | FETCH_SIZE GB (raw) | read GB (x2 gfx950 correction) | WRITE_SIZE GB | HBM GB / launch | MFMA busy % of GUI_ACTIVE x SIMDs |")
    print("|---|---|---|---|---|---|---|")
    rows = sorted(data.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))
    for fam, c in rows[:25]:
        fetch = c.get("FETCH_SIZE", 0) * 1024 / 1e9
        write = c.get("WRITE_SIZE", 0) * 1024 / 1e9
        n = max(calls[fam], 1)
        mfma = ""
        if c.get("GRBM_GUI_ACTIVE"):
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over SEs/SIMDs; normalise by active cycles x 1024 SIMDs
            mfma = "%.1f" % (100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] * 1024 / 8))
        print("| `%s` | %d | %.2f | %.2f | %.2f | %.3f | %s |" % (fam, n, fetch, 2 * fetch, write, (2 * fetch + write) / n, mfma))


if __name__ == "__main__":
    main(sys.argv[1])
