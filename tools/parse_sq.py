#!/usr/bin/env python3
"""Per-launch means of the SQ counters of ks_rk4_fused (or the kernels matching argv[3]) from a rocprofv3 --pmc CSV
(tools/prof_sq.sh, tools/prof_sq_burgers.sh)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
pattern = sys.argv[3] if len(sys.argv) > 3 else "ks_rk4_fused"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "sq", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if pattern in name:
            vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {"tag": tag, "kernels": {}}
for name, ctrs in vals.items():
    m = {c: sum(v) / len(v) for c, v in ctrs.items()}
    d = {"launches": len(next(iter(ctrs.values()))), "mean": m}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        d["fractions_of_wave_cycles"] = {k: m[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                                 "SQ_ACTIVE_INST_VALU") if k in m}
    if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
        d["valu_instructions_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    summary["kernels"][name] = d
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
