#!/usr/bin/env python3
"""Timeline of the last replayed TBPTT step from a rocprofv3 kernel trace CSV: start (us, relative), duration, stream/queue, name.
A step ends with its gradient-reduction (flush) launches; the next kernel after them starts the next step."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_flush = [("flush_grads_kernel" in r["Kernel_Name"] or "flush_all_kernel" in r["Kernel_Name"]) for r in rows]
starts = [i for i in range(1, len(rows)) if is_flush[i - 1] and not is_flush[i]]
ends = [i for i in range(len(rows) - 1) if is_flush[i] and not is_flush[i + 1]] + ([len(rows) - 1] if is_flush[-1] else [])
# the last complete step: from the last start that has a flush group after it
last_end = max(e for e in ends)
begin = max(b for b in starts if b < last_end - 5)
step = rows[begin:last_end + 1]
# torch's Adam kernels (unfused-optimizer runs) follow the flushes: include them
k = last_end + 1
while k < len(rows) and "multi_tensor_apply" in rows[k]["Kernel_Name"]:
    step.append(rows[k])
    k += 1
t0 = int(step[0]["Start_Timestamp"])
print(f"{len(step)} kernels, span {(max(int(r['End_Timestamp']) for r in step) - t0) / 1e3:.1f} us")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    st = r.get("Stream_Id", "?")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3} s{st:>3}  {r['Kernel_Name'][:90]}")
