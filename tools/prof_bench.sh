#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py invocation; keeps the small summaries only.
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench.py args...]
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/bench.py" "$@" > "$OUT/bench_stdout.log" 2> "$OUT/bench_stderr.log" || true
cd "$R"
for f in $(find "$OUT" -name '*_kernel_trace.csv'); do
  head -n 400 "$f" > "$f.head"; rm -f "$f"
done
find "$OUT" -name '*_kernel_stats.csv' -exec python3 "$R/tools/summarize_stats.py" {} \; > "$OUT/summary.txt"
cat "$OUT/summary.txt" | head -40
