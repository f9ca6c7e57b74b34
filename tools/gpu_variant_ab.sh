#!/bin/bash
# A/B of a tagged experiments-library variant (python airgym_amd/csrc/build.py --experiments --tag <tag> [-- -D...]) against the
# current experiments build: interleaved bench.py runs on one box.  Usage: bash tools/gpu_variant_ab.sh <tag> [repeats] [pytest-k]
TAG=${1:-old}; N=${2:-2}
for i in $(seq 1 $N); do for V in base $TAG; do
  if [ $V = base ]; then unset AIRGYM_EXP_LIB; else export AIRGYM_EXP_LIB=$(pwd)/airgym_amd/_native/libairgym_hip_exp_$TAG.so; fi
  AIRGYM_EXPERIMENTS=1 python bench.py --no-cpu-baseline --no-shipped-ratio --no-side-configs --no-roofline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms/epoch')"
done; done
