#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counters:  python tools/pmc_kernels.py <dir with *counter_collection.csv> [kernel substring]
One row per kernel (template arguments kept), one column per counter found in any pass: mean value per dispatch, and the dispatch
count / mean duration of the pass that carried the counter."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"vc::(\w+(?:<[^>]*>)?)", name) or re.search(r"_ZN2vc\d+(\w+?)(?:I|E)", name)
    return m.group(1) if m else name[:60]


def main(root, pat=""):
    val = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(list)
    gdur = defaultdict(float)   # sum of the durations (us) of the dispatches that carried GRBM_GUI_ACTIVE: the effective clock
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat and pat not in k:
                continue
            k = short(k)
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
                gdur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            d = (f, r["Dispatch_Id"])
            if d not in seen and "Start_Timestamp" in r:
                seen.add(d)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = sorted({c for k in val for c in val[k]})
    print("| kernel | dispatches / pass | mean us | clock GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration) | " + " | ".join(names) + " |")
    print("|---|---|---|---|" + "---|" * len(names))
    for k in sorted(val):
        n = max(cnt[k].values())
        row = ["%.4g" % (val[k][c] / cnt[k][c]) if cnt[k][c] else "" for c in names]
        ghz = "%.2f" % (val[k]["GRBM_GUI_ACTIVE"] / 8 / (gdur[k] * 1e3)) if gdur[k] else ""
        print("| `%s` | %d | %.1f | %s | %s |" % (k, n, sum(dur[k]) / max(len(dur[k]), 1), ghz, " | ".join(row)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
