#!/bin/bash
# HBM traffic of the SHIPPED env-step kernels from PMC counters: separate rocprofv3 passes per counter and per entry point
# (api = ag_step, rollout = ag_step_rollout, fused = ag_step_rollout_fused), kernel-trace only.  Writes
# gpurun_out/<tag>_env_kernel_pmc.json with the kernel-source hash bench.py checks before quoting `roofline.traffic`.
#   bash tools/gpu_pmc_env.sh <tag>
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -f $OUT/${TAG}_pmc_summary.txt
for FORM in rollout api fused; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${TAG}_${FORM}_$C; rm -rf $D
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o f -- python $REPO/tools/env_kernel_probe.py --forms $FORM --replays 8 > $OUT/${TAG}_pmc_${FORM}_$C.probe 2> $OUT/${TAG}_pmc_${FORM}_$C.err
    echo "form=$FORM $(python $REPO/tools/pmc_summary.py $D $C step_kernel_ws2)" >> $OUT/${TAG}_pmc_summary.txt
  done
done
cat $OUT/${TAG}_pmc_summary.txt
python $REPO/tools/pmc_env_json.py $OUT/${TAG}_pmc_summary.txt $TAG > $OUT/${TAG}_env_kernel_pmc.json
cat $OUT/${TAG}_env_kernel_pmc.json
