#!/bin/bash
# Round-end validation on the GPU box (repo root): smoke, the WHOLE GPU test suite, default bench line.  Stops at the
# first failure and exits non-zero; records the identity of the tree it ran on (tools/tree_sha.py --verify checks, in
# the build container, that nothing has changed since).
set -eo pipefail
mkdir -p gpurun_out
SHA=$(python tools/tree_sha.py)
echo "{\"tree_sha\": \"$SHA\", \"gpu_tests_ok\": false, \"stage\": \"started\"}" > gpurun_out/final_check.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1 || { tail -5 gpurun_out/final_smoke.log; echo "SMOKE FAILED"; exit 1; }
tail -2 gpurun_out/final_smoke.log
if ! timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/t_all.log 2>&1; then
  grep -E "^(FAILED|ERROR)|Fatal Python error|Aborted" gpurun_out/t_all.log | head -5 || true
  tail -5 gpurun_out/t_all.log | cut -c1-300
  echo "GPU SUITE FAILED"
  exit 1
fi
tail -2 gpurun_out/t_all.log | cut -c1-200
SUMMARY=$(tail -1 gpurun_out/t_all.log | tr -d '"')
echo "{\"tree_sha\": \"$SHA\", \"gpu_tests_ok\": true, \"summary\": \"$SUMMARY\"}" > gpurun_out/final_check.json
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print(d["config"]["workload"])
print("value", d["value"], "frac", d["roofline"]["frac"], "c2 frac", d.get("workload_c2", {}).get("roofline", {}).get("frac"))
t = d.get("tbptt", {})
print("tbptt N256", t.get("value"), "eager fused", t.get("eager_fused", {}).get("value"), "n64", t.get("n64", {}).get("value"))
PY
