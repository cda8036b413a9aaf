#!/bin/bash
# One round's measurement set for a workload, on the GPU box: PMC passes (each its own rocprofv3 run) + their summary,
# rocprofv3 kernel stats of a 4-step bench run, and the bench line itself.  Results under gpurun_out/<tag>/ — copy what is to
# be kept into profiles/.   usage: tools/collect_profiles.sh <tag> <workload> <max_per_leaf> [bench steps]
set -u
TAG=$1; WL=${2:-cornellbox}; LEAF=${3:-2}; STEPS=${4:-32}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RT_PMC_ARGS="--workload $WL" bash tools/profile_pmc.sh "$OUT/pmc" &&
python3 tools/summarize_pmc.py "$OUT/pmc" "$OUT/pmc_summary_$WL.json" "$WL" "$LEAF" > /dev/null &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --workload "$WL" --steps 4 --warmup 1 --no-cpu-baseline --no-side > "$OUT/prof.log" 2>&1 &&
cp "$OUT"/prof/*/*_kernel_stats.csv "$OUT/kernel_stats_${WL}_steps4.csv" &&
cp "$OUT/pmc_summary_$WL.json" profiles/r03_pmc_summary_$WL.json &&
timeout -k 10 600 python3 bench.py --workload "$WL" --steps "$STEPS" > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err" &&
tail -c 600 "$OUT/bench_$WL.json"
