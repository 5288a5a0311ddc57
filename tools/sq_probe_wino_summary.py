"""Per-dispatch summary of tools/sq_probe_wino.sh: the launches of one pass are in program order (4 forwards + 1 data gradient per round,
3 rounds); counters of the three rounds are averaged per position in the round."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
labels = ["conv3_2 fwd", "conv3_2 dgrad", "conv1_2 fwd", "conv4_2 fwd", "conv5_2 fwd"]
vals = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "wino" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"] or "xform" in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        rows[d][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[d]["__dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 if "End_Timestamp" in r else 0.0
    for k, d in enumerate(sorted(rows)):
        lab = labels[k % len(labels)]
        for c, v in rows[d].items():
            if c == "__dur":
                dur[lab].append(v)
            else:
                vals[lab][c].append(v)
mean = lambda a: sum(a) / max(len(a), 1)
for lab in labels:
    v = {c: mean(a) for c, a in vals[lab].items()}
    if not v:
        continue
    mf = v.get("SQ_INSTS_MFMA", 0.0)
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    print("== %s" % lab)
    if mf:
        print("  MFMA insts %.4g; per MFMA: VALU %.2f (incl. the MFMA)  LDS %.2f  SALU %.2f  VMEM_RD %.3f" % (
            mf, v.get("SQ_INSTS_VALU", 0) / mf, v.get("SQ_INSTS_LDS", 0) / mf, v.get("SQ_INSTS_SALU", 0) / mf, v.get("SQ_INSTS_VMEM_RD", 0) / mf))
    if wc:
        print("  of SQ_WAVE_CYCLES: WAIT_ANY %.1f %%  WAIT_INST_ANY %.1f %%  WAIT_INST_LDS %.1f %%  ACTIVE_INST_LDS %.1f %%  LDS_BANK_CONFLICT %.1f %%" % (
            100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc,
            100 * v.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
    if "GRBM_GUI_ACTIVE" in v:
        print("  MFMA busy %.1f %% (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))" % (
            100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    if "SQ_BUSY_CYCLES" in v:
        print("  raw: " + "  ".join("%s %.4g" % (c, v[c]) for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES") if c in v))
    print("  mean duration %.3f ms" % (1e3 * mean(dur[lab])))
