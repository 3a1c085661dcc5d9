#!/bin/bash
# Issue-slot accounting of the KS kernel from SQ counters (one rocprofv3 --pmc pass, no tracing flags).
# usage (GPU box, repo root): tools/prof_sq.sh <tag> [bench.py args...]
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d "$OUT/sq" -- python3 "$R/bench.py" --no-cpu-baseline --no-tbptt --no-secondary --no-extras "$@" > "$OUT/sq.log" 2>&1 || true
cd "$R"
python3 tools/parse_sq.py "$OUT" "$TAG"
find "$OUT" -name '*counter_collection.csv' -delete
