#!/bin/bash
# sweeps the traversal tuning knobs with a 2-step bench each: tools/tune.sh refill:leaf[:wavemin[:chunk]] ...
for cfg in "$@"; do
  IFS=: read -r r l w c <<< "$cfg"; w=${w:-0}; c=${c:-128}
  RT_TUNE_REFILL_MIN=$r RT_TUNE_LEAF_MIN=$l RT_TUNE_WAVE_MIN=$w RT_TUNE_FETCH_CHUNK=$c timeout -k 10 120 python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('refill $r leaf $l wavemin $w chunk $c:', round(d['value']), 'closest %.1f shade %.1f shadow %.1f' % (k['ms_closest'], k['ms_shade'], k['ms_shadow']))" || exit 1
done
