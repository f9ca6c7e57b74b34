#!/bin/bash
# Row-tile size of the split GEMMs (128 rows x 2 workgroups per CU vs 256 rows x 1): interleaved bench.py runs on the experiments build
for i in 1 2; do for WM in 4 2; do
  AIRGYM_EXPERIMENTS=1 AIRGYM_SPLIT_WM=$WM python bench.py --no-cpu-baseline --no-shipped-ratio --no-side-configs --no-roofline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WM=$WM', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms/epoch')"
done; done
