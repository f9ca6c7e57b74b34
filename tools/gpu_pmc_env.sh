#!/bin/bash
# HBM traffic of the SHIPPED env-step kernel from PMC counters: separate passes per counter and per entry point
# (ag_step_rollout = what the PPO rollout launches; ag_step = the drop-in API form).  bash tools/gpu_pmc_env.sh <tag>
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -f $OUT/${TAG}_pmc_summary.txt
for FORM in rollout api; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${TAG}_${FORM}_$C; rm -rf $D
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o f -- python $REPO/tools/sweep_env_kernel.py --blocks 0 --forms $FORM --replays 6 --nograph > $OUT/${TAG}_pmc_${FORM}_$C.sweep 2> $OUT/${TAG}_pmc_${FORM}_$C.err
    echo "form=$FORM $(python $REPO/tools/pmc_summary.py $D $C step_kernel)" >> $OUT/${TAG}_pmc_summary.txt
  done
done
cat $OUT/${TAG}_pmc_summary.txt
