import collections
import csv
import glob
import sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = "conv" if "conv_kernel" in r["Kernel_Name"] else ("gemm" if "gemm_kernel" in r["Kernel_Name"] else None)
        if k:
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in d:
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d[k].items()})
