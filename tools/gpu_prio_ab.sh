#!/bin/bash
# s_setprio 1 for waves 4..7 of the split GEMM / wgrad workgroups: interleaved bench.py runs, experiments build vs its -DAG_SPLIT_SETPRIO variant
for i in 1 2; do for V in base prio; do
  if [ $V = prio ]; then export AIRGYM_EXP_LIB=$(pwd)/airgym_amd/_native/libairgym_hip_exp_prio.so; else unset AIRGYM_EXP_LIB; fi
  AIRGYM_EXPERIMENTS=1 python bench.py --no-cpu-baseline --no-shipped-ratio --no-side-configs --no-roofline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms/epoch')"
done; done
