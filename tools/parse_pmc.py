#!/usr/bin/env python3
"""Per-launch FETCH_SIZE / WRITE_SIZE of ks_rk4_fused from rocprofv3 --pmc CSVs."""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
res = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals = []
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "ks_rk4_fused" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    # the first launches are the 250-sub-step attractor warm-up (phi = 0), then the timed ones: all same shape
    res[counter] = {"launches": len(vals), "mean": sum(vals) / len(vals) if vals else None,
                    "min": min(vals) if vals else None, "max": max(vals) if vals else None}
fetch_kb, write_kb = res["FETCH_SIZE"]["mean"], res["WRITE_SIZE"]["mean"]
summary = {"tag": tag, "raw_counters_KB": res}
if fetch_kb is not None and write_kb is not None:
    # guide: counters are in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> double it
    summary["hbm_bytes_per_launch"] = (2.0 * fetch_kb + write_kb) * 1024.0
    summary["fetch_bytes_corrected"] = 2.0 * fetch_kb * 1024.0
    summary["write_bytes"] = write_kb * 1024.0
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
