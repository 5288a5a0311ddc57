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
    if "conv_wino4_kernel" in name:  # conv_wino4_kernel<KIND, POOL> (round 3: Winograd F(4x4,3x3))
        k = re.search(r"conv_wino4_kernel<\s*(\d)", name)
        return "conv_wino4_kernel<%s>" % {"0": "fwd", "1": "dgrad"}.get(k.group(1) if k else "?", "?")
    if "conv_wino2_kernel" in name:  # conv_wino2_kernel<KIND, POOL> (round 3: 16x16x4 tiles, two workgroups per CU)
        k = re.search(r"conv_wino2_kernel<\s*(\d)", name)
        return "conv_wino2_kernel<%s>" % {"0": "fwd", "1": "dgrad"}.get(k.group(1) if k else "?", "?")
    if "conv_wino_kernel" in name:  # conv_wino_kernel<KIND, POOL>
        k = re.search(r"conv_wino_kernel<\s*(\d)", name)
        return "conv_wino_kernel<%s>" % {"0": "fwd", "1": "dgrad"}.get(k.group(1) if k else "?", "?")
    if "wino_wgrad_kernel" in name:
        return "wino_wgrad_kernel<4x8 | 4x7 | 2x14>"
    return base


CONV_FAMILY = ("conv_kernel", "conv1_", "conv_tail_reduce", "wgrad_reduce",
               "conv_wino_kernel", "conv_wino2_kernel", "conv_wino4_kernel", "wino_wgrad")


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
    # the convolution family as a whole, per traced step (one preprocess_kernel launch per fine-tune step)
    steps = max(calls.get("preprocess_kernel", 0), 1)
    conv = [(fam, c) for fam, c in data.items() if any(k in fam for k in CONV_FAMILY)]
    rd = sum(2 * c.get("FETCH_SIZE", 0) * 1024 for _, c in conv)
    wr = sum(c.get("WRITE_SIZE", 0) * 1024 for _, c in conv)
    print("\nconvolution family (all 3x3 convolution kernels incl. tails and split reduces): %.2f GB read (2 x FETCH_SIZE) + %.2f GB written "
          "= %.2f GB per step over %d traced steps" % (rd / 1e9 / steps, wr / 1e9 / steps, (rd + wr) / 1e9 / steps, steps))
    if len(sys.argv) > 2:
        import json
        json.dump({"bytes_per_step": (rd + wr) / steps, "read_bytes_per_step": rd / steps, "write_bytes_per_step": wr / steps, "steps_traced": steps,
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 4 --warmup 1 --no-cpu-baseline`; "
                             "(2 x FETCH_SIZE [gfx950 16-B/lane stream correction, MI355X_MICROARCH.md] + WRITE_SIZE) KiB summed over every 3x3 "
                             "convolution kernel (main, K-split tail, split reduce) / traced steps; L2-miss-side traffic (Infinity-Cache hits are "
                             "counted); tools/pmc_summary.py"}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
