"""python tools/pmc_probe_summary.py <dir>: per-kernel-kind averages of every counter collected over tools/pmc_probe.py."""
import collections
import csv
import glob
import re
import sys

for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = None
        m = re.search(r"conv_kernel<.*>,\s*(\d)\s*>\(", n)
        if m:
            k = "conv_" + {"0": "fwd", "1": "dgrad", "2": "wgrad"}[m.group(1)]
        elif "conv_patch_kernel" in n:
            m2 = re.search(r"conv_patch_kernel<\s*\d+,\s*\d+,\s*(\d)\s*(?:,\s*\w+\s*)*>", n)
            if "tail" not in n and m2:
                k = "patch_" + {"0": "fwd", "1": "dgrad"}[m2.group(1)]
        elif "wgrad_patch_kernel" in n:
            m3 = re.search(r"wgrad_patch_kernel<\s*(\d+),\s*(\d+)\s*>", n)
            k = "patch_wgrad(%sx%s)" % (m3.group(1), m3.group(2)) if m3 else "patch_wgrad"
        elif "conv_wino_kernel" in n:
            m4 = re.search(r"conv_wino_kernel<\s*(\d)", n)
            k = "wino_" + {"0": "fwd", "1": "dgrad"}.get(m4.group(1) if m4 else "?", "?")
        elif "wino_wgrad_kernel" in n:
            m5 = re.search(r"wino_wgrad_kernel<\s*(\d+),\s*(\d+)\s*>", n)
            k = "wino_wgrad(%sx%s)" % (m5.group(1), m5.group(2)) if m5 else "wino_wgrad"
        elif "gemm_kernel" in n:
            k = "gemm"
        if k:
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(d):
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d[k].items()})
