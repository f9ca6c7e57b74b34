#!/bin/bash
# tools/update_seq_probe.py (recompute path) on the experiments build and on tagged -D variants of it, interleaved.
# Usage (repo root, under gpurun): bash tools/gpu_variant_probe.sh "<tag> <tag> ..." [rounds]
TAGS="$1"; N=${2:-2}
run() {
  python tools/update_seq_probe.py --recompute 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', [k['us'] for k in d['kernels']], d['sum_us'])"
}
for i in $(seq 1 $N); do
  unset AIRGYM_EXP_LIB; AIRGYM_EXPERIMENTS=1 run base
  for T in $TAGS; do AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=$(pwd)/airgym_amd/_native/libairgym_hip_exp_$T.so run $T; done
done
