#!/bin/bash
# rocprofv3 kernel stats of tools/tbptt_profile_run.py (fused hipGraph TBPTT steps); keeps the summary and the
# timeline of the last step only.
# usage (on the GPU box, from the repo root): tools/prof_tbptt.sh <tag>
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/tools/tbptt_profile_run.py" "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log" || true
cd "$R"
for f in $(find "$OUT" -name '*_kernel_trace.csv'); do
  python3 "$R/tools/trace_timeline.py" "$f" > "$OUT/timeline.txt" 2>&1 || true
  rm -f "$f"
done
find "$OUT" -name '*_kernel_stats.csv' -exec python3 "$R/tools/summarize_stats.py" {} \; > "$OUT/summary.txt"
head -12 "$OUT/summary.txt"
