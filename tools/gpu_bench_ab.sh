#!/bin/bash
# A/B of bench.py flags on one box, interleaved runs.  Usage: bash tools/gpu_bench_ab.sh "<flags A>" "<flags B>" [repeats]
A="$1"; B="$2"; N=${3:-2}
for i in $(seq 1 $N); do
  for F in "$A" "$B"; do
    python bench.py --no-cpu-baseline --no-shipped-ratio --no-side-configs --no-roofline --steps 10 --warmup 3 $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$F', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms/epoch', d['config'].get('rollout_launches_per_step'))"
  done
done
