#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family.
   python tools/pmc_summary.py gpurun_out/pmc > profiles/rNN_pmc.md
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); on gfx950 FETCH_SIZE under-reports wide coalesced
streams by exactly 2x (MI355X_MICROARCH.md, HBM section), so the corrected read bytes = 2 x FETCH_SIZE.
Effective clock: the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS give-back); GRBM_GUI_ACTIVE counts
active cycles per XCD, so cycles / 8 / kernel wall time is the clock the kernel actually ran at."""
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
    gdur = defaultdict(float)  # durations of the launches of the pass that carried GRBM_GUI_ACTIVE
    gseen = set()
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
            if key not in gseen and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gseen.add(key)
                gdur[fam] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    print("| kernel family | launches | FETCH_SIZE GB (raw) | read GB (x2 gfx950 correction) | WRITE_SIZE GB | HBM GB / launch | MFMA busy % of GUI_ACTIVE x SIMDs | effective clock GHz (GUI_ACTIVE / 8 XCDs / kernel time) |")
    print("|---|---|---|---|---|---|---|---|")
    rows = sorted(data.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))
    for fam, c in rows[:25]:
        fetch = c.get("FETCH_SIZE", 0) * 1024 / 1e9
        write = c.get("WRITE_SIZE", 0) * 1024 / 1e9
        n = max(calls[fam], 1)
        mfma = clk = ""
        if c.get("GRBM_GUI_ACTIVE") and gdur[fam] > 0:
            clk = "%.2f" % (c["GRBM_GUI_ACTIVE"] / 8 / gdur[fam] / 1e9)
        if c.get("GRBM_GUI_ACTIVE"):
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over SEs/SIMDs; normalise by active cycles x 1024 SIMDs
            mfma = "%.1f" % (100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] * 1024 / 8))
        print("| `%s` | %d | %.2f | %.2f | %.2f | %.3f | %s | %s |" % (fam, n, fetch, 2 * fetch, write, (2 * fetch + write) / n, mfma, clk))


if __name__ == "__main__":
    main(sys.argv[1])
