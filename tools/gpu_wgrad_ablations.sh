#!/bin/bash
# Timing ablations of split_wgrad_fin_kernel (results wrong by construction): the launch sequence of tools/update_seq_probe.py with
# the experiments build and its -DAG_WF_ABL_* variants (python airgym_amd/csrc/build.py --experiments --tag wf_<V> -- -DAG_WF_ABL_<V>).
# Usage (repo root, under gpurun): bash tools/gpu_wgrad_ablations.sh
run() {
  python tools/update_seq_probe.py --recompute 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', [k['us'] for k in d['kernels']])"
}
for i in 1 2; do
  unset AIRGYM_EXP_LIB; AIRGYM_EXPERIMENTS=1 run base
  for V in NO_BARRIER NO_MFMA NO_LOADS NO_STAGE NO_PROD; do
    AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=$(pwd)/airgym_amd/_native/libairgym_hip_exp_wf_$V.so run $V
  done
done
