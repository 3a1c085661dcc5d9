#!/bin/bash
# Round-end validation on the GPU box (repo root): smoke, GPU test suite, default bench line, kernel stats of the bench
# command, TBPTT step timeline, 2-rank rehearsal of the N > 1 code path (two ranks sharing the one GPU, gloo).
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 && python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -2 gpurun_out/t_all.log | cut -c1-200
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tools/prof_tbptt.sh vfinal > /dev/null
tools/prof_bench.sh bfinal --no-cpu-baseline > /dev/null
BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29516 \
  bench.py --gpus 2 --steps 20 --warmup 3 2> gpurun_out/bench_2rank_gloo.err | grep "^{" > gpurun_out/bench_2rank_gloo.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["workload_c3"]["roofline"]["frac"], d["tbptt"]["value"], d["tbptt"]["ensemble"]["value"],
      d["tbptt"].get("n256", {}).get("value"), d["tbptt"]["first_loss"]["rel_diff_fused"])
e = json.load(open("gpurun_out/bench_2rank_gloo.json"))
print(e["n_gpus"], e["value"], e["tbptt"]["value"], e["tbptt"]["ranks_in_sync"])
PY
