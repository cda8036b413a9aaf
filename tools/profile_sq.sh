#!/bin/bash
# One SQ pass (VALU instruction count, lane utilisation, wait shares) over a one-step bench run.
OUT=${1:-gpurun_out/sq}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq1" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side > "$OUT/sq1.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/sq1.log"; exit 1; }
python3 tools/summarize_pmc.py "$OUT" "$OUT/summary.json" > /dev/null
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("k_closest", "k_shadow", "k_shade"):
    c = d[k]
    print(k, "ms %.1f" % c["ms_profiled_sq1"], "VALU %.2fG" % (c["SQ_INSTS_VALU"] / 1e9), "lanes %.1f" % c["valu_active_lanes_of_64"],
          "wait %.2f" % c["SQ_WAIT_ANY_frac_of_wave_cycles"], "stall %.2f" % c["SQ_WAIT_INST_ANY_frac_of_wave_cycles"],
          "valu_act %.3f" % c["SQ_ACTIVE_INST_VALU_frac_of_wave_cycles"], "waves/busy %.1f" % c["mean_waves_per_busy_cycle"])
PY
