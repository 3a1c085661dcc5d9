#!/bin/bash
# One GPU-box call that refreshes every profile-derived number bench.py quotes, for the CURRENT kernel sources:
#   PMC traffic (two passes) and SQ issue counters of the KS kernel at both single-GPU workloads, folded into profiles/*.json
#   with the kernel-source identity; rocprofv3 --kernel-trace --stats of the default bench command.
# usage (GPU box, repo root): tools/prof_round.sh <round tag, e.g. r02>
set -e
TAG=$1
for W in c3 c2; do
  tools/prof_pmc.sh ${TAG}_$W --workload $W > /dev/null
  tools/prof_sq.sh ${TAG}_$W --workload $W > /dev/null
  python3 tools/update_ks_profiles.py $W ${TAG}_$W
done
tools/prof_bench.sh ${TAG}_bench --no-cpu-baseline > /dev/null
cp gpurun_out/prof_${TAG}_bench/summary.txt gpurun_out/${TAG}_bench_default_kernel_stats.txt
head -12 gpurun_out/${TAG}_bench_default_kernel_stats.txt
