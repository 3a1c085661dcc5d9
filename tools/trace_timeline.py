#!/usr/bin/env python3
"""Timeline of the last replayed TBPTT step from a rocprofv3 kernel trace CSV: start (us, relative), duration, stream/queue, name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the step boundary: the bucket zero fill is the first kernel of a replay; find the last Adam multi_tensor kernel and walk back
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]
last = ends[-1]
# previous step's last multi_tensor kernel
prev = max(i for i in ends if i < last - 20)
step = rows[prev + 1:last + 1]
t0 = int(step[0]["Start_Timestamp"])
print(f"{len(step)} kernels, span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
last_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    st = r.get("Stream_Id", "?")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3} s{st:>3}  {r['Kernel_Name'][:90]}")
