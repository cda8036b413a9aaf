#!/bin/bash
# runs a short bench with each given build of librt_hip.so swapped in (GPU box only; the tree there is a scratch copy):
# tools/sweep_so.sh build/a.so build/b.so ...   (env for RT_TUNE_* knobs passes through)
cp raytracing_amd/librt_hip.so /tmp/librt_hip_keep.so
for so in "$@"; do
  cp "$so" raytracing_amd/librt_hip.so
  timeout -k 10 120 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$so:', round(d['value']), 'closest %.1f shade %.1f shadow %.1f' % (k['ms_closest'], k['ms_shade'], k['ms_shadow']))" || { cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so; exit 1; }
done
cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so
