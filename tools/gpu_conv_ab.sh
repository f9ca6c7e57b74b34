#!/bin/bash
# tools/conv_probe.py on the shipped library and on variants of the experiments build (build.py --experiments --tag T -- -D...).
# Usage (repo root, under gpurun): bash tools/gpu_conv_ab.sh shipped tagA shipped tagA
set -u
for T in "$@"; do
  echo "== $T"
  if [ "$T" = "shipped" ]; then E=""; else E="AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=airgym_amd/_native/libairgym_hip_exp_$T.so"; fi
  env $E timeout 300 python tools/conv_probe.py --reps 10 --layers conv2,conv3 2>/dev/null | python -c "import sys,json; [print(' ', d['layer'], d['pass'][:14].ljust(14), d['hip_us'], '  maxdiff', '%.1e' % d['max_rel_diff']) for d in map(json.loads, sys.stdin)]"
done
