#!/bin/bash
# Copies what tools/collect_round.sh <tag> left under gpurun_out/<tag>/ into profiles/ under this round's names and checks that the
# counter files were measured on the kernel sources in the tree.  usage (CPU container, after the GPU call): tools/adopt_round.sh <tag> [round]
set -e
F=gpurun_out/${1:?tag}; R=${2:-r03}
cp $F/cornellbox/pmc_summary_cornellbox.json profiles/${R}_pmc_summary_cornellbox.json
cp $F/spheres10m/pmc_summary_spheres10m.json profiles/${R}_pmc_summary_spheres10m.json
cp $F/instanced5m/pmc_summary_instanced5m.json profiles/${R}_pmc_summary_instanced5m.json
cp $F/cornellbox/kernel_stats_cornellbox_steps4.csv profiles/${R}_kernel_stats_bench_steps4.csv
cp $F/spheres10m/kernel_stats_spheres10m_steps4.csv profiles/${R}_kernel_stats_spheres10m_steps4.csv
cp $F/instanced5m/kernel_stats_instanced5m_steps4.csv profiles/${R}_kernel_stats_instanced5m_two_level_steps4.csv
cp $F/bench_default.json profiles/${R}_bench_full.json
cp $F/spheres10m/bench_spheres10m.json profiles/${R}_bench_spheres10m.json
cp $F/instanced5m/bench_instanced5m.json profiles/${R}_bench_instanced5m_two_level.json
cp $F/lane_stats.txt profiles/${R}_lane_stats_cornellbox.txt
cp $F/lane_stats_cornellbox.json profiles/${R}_lane_stats_cornellbox.json
python3 tools/kernel_resources.py > profiles/${R}_kernel_resources.txt 2>/dev/null
python3 - "$R" <<'PY'
import json, sys
import bench
r = sys.argv[1]
sha = bench.kernel_source_sha16()
for w in ("cornellbox", "spheres10m", "instanced5m"):
    got = json.load(open(f"profiles/{r}_pmc_summary_{w}.json"))["_meta"]["source_sha16"]
    assert got == sha, (w, got, sha)
b = json.load(open(f"profiles/{r}_bench_full.json"))
print("sources", sha, "| headline", round(b["value"]), b["unit"], "%.1f ms per step" % b["ms_per_step"], "| roofline frac %.3f" % b["roofline"]["frac"],
      "| side", {k: (round(v["value"]), round(v["roofline"]["frac"], 3)) for k, v in b["side"].items()})
PY
