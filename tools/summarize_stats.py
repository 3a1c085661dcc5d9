#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a fixed-width table (top 25 kernels + totals)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
tot_calls = sum(int(r["Calls"]) for r in rows)
print(f"# {sys.argv[1].split('/')[-1]}: {len(rows)} distinct kernels, {tot_calls} launches, {tot_ns/1e6:.3f} ms GPU time")
print("%-70s %8s %14s %12s %7s %10s %10s" % ("Name", "Calls", "TotalNs", "AvgNs", "Pct", "MinNs", "MaxNs"))
def line(r):
    print("%-70s %8s %14s %12.0f %7.2f %10s %10s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"],
                                                  float(r["AverageNs"]), float(r["Percentage"]), r["MinNs"], r["MaxNs"]))


for r in rows[:25]:
    line(r)
# every kernel of THIS repo (libkspde.so: namespace ks; libsurrogate_hip.so: anonymous namespace), wherever it ranks
own = [r for r in rows[25:] if "ks::" in r["Name"] or "(anonymous namespace)::" in r["Name"]]
if own:
    print("# this repo's kernels below the top 25:")
    for r in own:
        line(r)
