#!/usr/bin/env python3
"""Fold the summaries of tools/prof_pmc.sh / tools/prof_sq.sh (gpurun_out/pmc_<tag>/summary.json,
gpurun_out/sq_<tag>/summary.json) into profiles/ks_pmc_traffic.json / profiles/ks_sq_counters.json, stamped with the
identity of the kernel sources they were measured on (bench.py quotes them only while that identity is current).

usage: tools/update_ks_profiles.py <workload c2|c3> <tag> [key] [extra bench args, e.g. "--mode exact"]
(key: the entry's name in the json files, default = the workload; "c3_exact" for the exact-mode kernel)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

workload, tag = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else workload
E, N, L, _ = bench.WORKLOADS[workload]
sha = bench.kernel_source_sha()
args = "--no-cpu-baseline --no-tbptt --no-secondary --no-extras --workload " + workload + "".join(" " + a for a in sys.argv[4:])


def merge(path, entry):
    try:
        data = json.load(open(path))
    except (OSError, ValueError):
        data = {}
    data[key] = entry
    json.dump(data, open(path, "w"), indent=1)
    # gpurun merges only gpurun_out/ back: leave a copy there to be carried into profiles/ by hand
    out = os.path.join(ROOT, "gpurun_out", "profiles_out")
    os.makedirs(out, exist_ok=True)
    json.dump(data, open(os.path.join(out, os.path.basename(path)), "w"), indent=1)


pmc = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "summary.json")
if os.path.exists(pmc):
    s = json.load(open(pmc))
    if s.get("hbm_bytes_per_launch") is not None:
        merge(os.path.join(ROOT, "profiles", "ks_pmc_traffic.json"), {
            "hbm_bytes_per_launch": s["hbm_bytes_per_launch"], "fetch_bytes_corrected": s["fetch_bytes_corrected"],
            "write_bytes": s["write_bytes"], "raw_counters_KB": s["raw_counters_KB"], "kernel_source_sha": sha,
            "command": f"rocprofv3 --pmc FETCH_SIZE (then a second pass --pmc WRITE_SIZE) --output-format csv -- python3 bench.py {args}",
            "note": "per-launch mean over the ks_rk4_fused dispatches; FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B; "
                    "calibrated in the guide for 16 B/lane reads, this kernel reads 8 B/lane so the read side is an upper "
                    "estimate), WRITE_SIZE as reported"})
        print("profiles/ks_pmc_traffic.json <-", workload, s["hbm_bytes_per_launch"])
sq = os.path.join(ROOT, "gpurun_out", f"sq_{tag}", "summary.json")
if os.path.exists(sq):
    s = json.load(open(sq))
    for name, d in s.get("kernels", {}).items():
        m = d["mean"]
        entry = {"kernel": name, "launches": d["launches"], "mean_per_launch": m,
                 "fractions_of_wave_cycles": d.get("fractions_of_wave_cycles", {}),
                 "valu_instructions_per_wave": d.get("valu_instructions_per_wave"),
                 "valu_instructions_per_point_substep": m["SQ_INSTS_VALU"] * 64.0 / (E * N * bench.CFG_STEPS) if "SQ_INSTS_VALU" in m else None,
                 "kernel_source_sha": sha,
                 "command": f"rocprofv3 --pmc <8 SQ counters> --output-format csv -- python3 bench.py {args}"}
        merge(os.path.join(ROOT, "profiles", "ks_sq_counters.json"), entry)
        print("profiles/ks_sq_counters.json <-", workload, name)
