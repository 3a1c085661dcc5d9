#!/bin/bash
# Issue-slot / LDS accounting of the fused surrogate kernels from SQ counters: two rocprofv3 --pmc passes (no tracing
# flags) over tools/tbptt_profile_run.py.   usage (GPU box, repo root): tools/prof_sq_tbptt.sh <tag> [n256]
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/sqt_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d "$OUT/p1" -- python3 "$R/tools/tbptt_profile_run.py" "$@" > "$OUT/p1.log" 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU \
  --output-format csv -d "$OUT/p2" -- python3 "$R/tools/tbptt_profile_run.py" "$@" > "$OUT/p2.log" 2>&1 || true
cd "$R"
python3 tools/parse_sq_tbptt.py "$OUT" > "$OUT/summary.txt"
find "$OUT" -name '*counter_collection.csv' -delete
cat "$OUT/summary.txt"
