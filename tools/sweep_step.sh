#!/bin/bash
# sweeps step size and pass size of bench.py: tools/sweep_step.sh spp_step:spp_per_pass[:steps] ...
for cfg in "$@"; do
  IFS=: read -r s p k <<< "$cfg"; k=${k:-4}
  timeout -k 10 200 python3 -u bench.py --steps $k --warmup 1 --spp-step $s --spp-per-pass $p --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('spp_step $s per_pass $p steps $k:', round(d['value']), 'Mrays/s  ms/step %.1f' % d['ms_per_step'])" || exit 1
done
