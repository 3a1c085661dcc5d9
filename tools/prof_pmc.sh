#!/bin/bash
# HBM traffic of the KS kernel from PMC counters: two separate rocprofv3 passes (FETCH_SIZE needs 3 TCC
# slots, WRITE_SIZE 2 -- they do not fit one pass), no tracing flags next to --pmc.
# usage (GPU box, repo root): tools/prof_pmc.sh <tag> [bench.py args...]
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" --no-cpu-baseline --no-tbptt --no-secondary --no-extras "$@" > "$OUT/fetch.log" 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" --no-cpu-baseline --no-tbptt --no-secondary --no-extras "$@" > "$OUT/write.log" 2>&1 || true
cd "$R"
python3 tools/parse_pmc.py "$OUT" "$TAG"
