#!/bin/bash
# rocprofv3 kernel stats of eager FNO TBPTT steps.  usage (GPU box, repo root): tools/prof_fno.sh <tag>
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/tools/fno_profile_run.py" "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log" || true
cd "$R"
find "$OUT" -name '*_kernel_trace.csv' -delete
find "$OUT" -name '*_kernel_stats.csv' -exec python3 "$R/tools/summarize_stats.py" {} \; > "$OUT/summary.txt"
head -24 "$OUT/summary.txt"
