#!/usr/bin/env python3
"""Per-launch means of SQ counters for every surrogate kernel (tools/prof_sq_tbptt.sh): one row per kernel."""
import collections
import csv
import glob
import os
import re
import sys

out_dir = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "anonymous namespace" not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
        vals[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
        "SQ_INSTS_SALU", "SQ_BUSY_CYCLES"]
print("per-launch means; fractions are of SQ_WAVE_CYCLES (quad-cycles summed over waves)")
print(f"{'kernel':34s} {'launches':>8s} {'waves':>7s} {'wavecyc':>10s} {'wait':>6s} {'stall':>6s} {'active':>6s} {'valu':>6s} "
      f"{'valu/wave':>9s} {'lds/wave':>8s} {'salu/wave':>9s} {'ldsact':>6s} {'ldsidx':>10s} {'bankconf':>10s} {'mfma_busy':>10s}")
for name, c in sorted(vals.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc, w = m.get("SQ_WAVE_CYCLES", 0.0), m.get("SQ_WAVES", 0.0)
    fr = lambda k: f"{m[k] / wc:6.3f}" if k in m and wc else "     -"
    per = lambda k: f"{m[k] / w:9.0f}" if k in m and w else "        -"
    print(f"{name[:34]:34s} {len(next(iter(c.values()))):8d} {w:7.0f} {wc:10.0f} {fr('SQ_WAIT_ANY')} {fr('SQ_WAIT_INST_ANY')} "
          f"{fr('SQ_ACTIVE_INST_ANY')} {fr('SQ_ACTIVE_INST_VALU')} {per('SQ_INSTS_VALU')} {per('SQ_INSTS_LDS')[1:]} {per('SQ_INSTS_SALU')} "
          f"{fr('SQ_ACTIVE_INST_LDS')} {m.get('SQ_LDS_IDX_ACTIVE', 0):10.0f} {m.get('SQ_LDS_BANK_CONFLICT', 0):10.0f} "
          f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):10.0f}")
