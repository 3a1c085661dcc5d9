#!/bin/bash
# Issue-slot accounting of the Burgers stepper (bg_step_kernel) from SQ counters: one rocprofv3 --pmc pass, no tracing flags.
# usage (GPU box, repo root): tools/prof_sq_burgers.sh <tag>   -> gpurun_out/sq_<tag>/summary.json -> profiles/burgers_sq_counters.json
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d "$OUT/sq" -- python3 "$R/tools/burgers_profile_run.py" > "$OUT/sq.log" 2>&1 || true
cd "$R"
python3 tools/parse_sq.py "$OUT" "$TAG" bg_step_kernel > /dev/null
find "$OUT" -name '*counter_collection.csv' -delete
python3 - "$OUT" <<'PY'
import hashlib, json, os, sys
out = sys.argv[1]
root = os.getcwd()
s = json.load(open(os.path.join(out, "summary.json")))
sha = hashlib.sha256(open(os.path.join(root, "model-based-pde-control_amd", "csrc", "burgers.hip"), "rb").read()).hexdigest()[:16]
E, N, cfg = 8192, 512, 50
res = {"how": "tools/prof_sq_burgers.sh: one rocprofv3 --pmc pass (8 SQ counters) of tools/burgers_profile_run.py (8192 x 512, 50 sub-steps "
              "per launch); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves"}
for name, d in s.get("kernels", {}).items():
    m = d["mean"]
    res["c4"] = {"kernel": name, "launches": d["launches"], "mean_per_launch": m,
                 "fractions_of_wave_cycles": d.get("fractions_of_wave_cycles", {}),
                 "valu_instructions_per_wave": d.get("valu_instructions_per_wave"),
                 "valu_wave_instructions_per_launch": m.get("SQ_INSTS_VALU"),
                 "valu_instructions_per_point_substep": m["SQ_INSTS_VALU"] * 64.0 / (E * N * cfg) if "SQ_INSTS_VALU" in m else None,
                 "kernel_source_sha": sha}
json.dump(res, open(os.path.join(root, "profiles", "burgers_sq_counters.json"), "w"), indent=1)
# gpurun merges only gpurun_out/ back: leave a copy there to be carried into profiles/ by hand
os.makedirs(os.path.join(root, "gpurun_out", "profiles_out"), exist_ok=True)
json.dump(res, open(os.path.join(root, "gpurun_out", "profiles_out", "burgers_sq_counters.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
