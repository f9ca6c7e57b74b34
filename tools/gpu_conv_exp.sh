#!/bin/bash
# conv_probe.py against variants of the experiments library (build.py --experiments --tag T -- -D...).  Usage: bash tools/gpu_conv_exp.sh tag [tag ...]
set -u
OUT=gpurun_out; mkdir -p $OUT
for T in "$@"; do
  echo "== $T"
  if [ "$T" = "shipped" ]; then
    timeout 300 python tools/conv_probe.py --reps 10 --layers conv2,conv3 2> $OUT/conv_exp_$T.err | python -c "import sys,json; [print(' ', d['layer'], d['pass'][:14].ljust(14), d['hip_us']) for d in map(json.loads, sys.stdin)]"
  else
    AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=airgym_amd/_native/libairgym_hip_exp_$T.so timeout 300 python tools/conv_probe.py --reps 10 --layers conv2,conv3 2> $OUT/conv_exp_$T.err | python -c "import sys,json; [print(' ', d['layer'], d['pass'][:14].ljust(14), d['hip_us']) for d in map(json.loads, sys.stdin)]"
  fi
done
