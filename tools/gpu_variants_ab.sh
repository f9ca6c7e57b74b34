#!/bin/bash
# Interleaved bench.py runs of the experiments build ("base", optional extra flags in BASE_FLAGS) and of several tagged variants.
# Usage: bash tools/gpu_variants_ab.sh "<tag> <tag> ..." [repeats]
TAGS="$1"; N=${2:-2}
run() {
  python bench.py --no-cpu-baseline --no-shipped-ratio --no-side-configs --no-roofline --steps 10 --warmup 3 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms/epoch')"
}
for i in $(seq 1 $N); do
  unset AIRGYM_EXP_LIB; AIRGYM_EXPERIMENTS=1 run "base" "$BASE_FLAGS"
  for T in $TAGS; do AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=$(pwd)/airgym_amd/_native/libairgym_hip_exp_$T.so run $T ""; done
done
