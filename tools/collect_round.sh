#!/bin/bash
# The round's whole measurement set in one GPU call: counters + kernel stats + bench line for the headline workload and the two
# side workloads, lane statistics.  usage (GPU box): bash tools/collect_round.sh <tag>; results under gpurun_out/<tag>/
TAG=${1:-round}
mkdir -p gpurun_out/$TAG
bash tools/collect_profiles.sh $TAG/cornellbox cornellbox 2 32 > gpurun_out/$TAG/collect_cornellbox.log 2>&1 || { echo "cornellbox failed"; tail -5 gpurun_out/$TAG/collect_cornellbox.log; exit 1; }
bash tools/collect_profiles.sh $TAG/spheres10m spheres10m 2 4 > gpurun_out/$TAG/collect_spheres10m.log 2>&1 || { echo "spheres10m failed"; exit 1; }
bash tools/collect_profiles.sh $TAG/instanced5m instanced5m 2 8 > gpurun_out/$TAG/collect_instanced5m.log 2>&1 || { echo "instanced5m failed"; exit 1; }
find gpurun_out/$TAG -name "*.db" -delete
# the headline line again, now that all three counter files of this source are in profiles/ (the side legs' rooflines need theirs)
timeout -k 10 600 python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err || { echo "default bench failed"; exit 1; }
bash tools/lane_stats.sh > gpurun_out/$TAG/lane_stats.txt 2>&1
cp gpurun_out/lane_stats_cornellbox.json gpurun_out/$TAG/lane_stats_cornellbox.json
tail -c 400 gpurun_out/$TAG/bench_default.json
