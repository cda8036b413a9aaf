#!/bin/bash
# RT_SWEEP_ARGS='--workload instanced5m' adds bench.py arguments
# short bench per "so:ENV=VAL,ENV=VAL" item, e.g. tools/sweep_env.sh build/sw5.so:RT_TUNE_SHADE_MULT=5 (GPU box only)
cp raytracing_amd/librt_hip.so /tmp/librt_hip_keep.so
for item in "$@"; do
  so=${item%%:*}; envs=${item#*:}; [ "$envs" = "$item" ] && envs=""
  cp "$so" raytracing_amd/librt_hip.so
  env $(echo $envs | tr ',' ' ') timeout -k 10 120 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side ${RT_SWEEP_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$item:', round(d['value']), 'closest %.1f shade %.1f shadow %.1f' % (k['ms_closest'], k['ms_shade'], k['ms_shadow']))" || { cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so; exit 1; }
done
cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so
