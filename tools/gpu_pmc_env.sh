#!/bin/bash
# HBM traffic of the SHIPPED env-step kernels from PMC counters: separate rocprofv3 passes per counter and per entry point
# (multi = ag_step_multi with 24 steps per launch, api = ag_step, rollout = ag_step_rollout, fused = ag_step_rollout_fused),
# kernel-trace only.  A third argument "sq" adds the SQ wave-state counters of the multi form (where the waves wait).  Writes
# gpurun_out/<tag>_env_kernel_pmc.json with the kernel-source hash bench.py checks before quoting `roofline.traffic`.
#   bash tools/gpu_pmc_env.sh <tag>
set -u
TAG=${1:-r03}
TASK=${TASK:-hovering}; CTL=${CTL:-rate}      # env: TASK=tracking CTL=vel bash tools/gpu_pmc_env.sh r06_tracking
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -f $OUT/${TAG}_pmc_summary.txt
FORMS=${2:-"multi rollout api fused"}
for FORM in $FORMS; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${TAG}_${FORM}_$C; rm -rf $D
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o f -- python $REPO/tools/env_kernel_probe.py --task $TASK --ctl $CTL --forms $FORM --replays 8 --nograph > $OUT/${TAG}_pmc_${FORM}_$C.probe 2> $OUT/${TAG}_pmc_${FORM}_$C.err
    echo "form=$FORM $(python $REPO/tools/pmc_summary.py $D $C step_kernel_ | head -1)" >> $OUT/${TAG}_pmc_summary.txt
  done
done
cat $OUT/${TAG}_pmc_summary.txt
if [ "${3:-}" = "sq" ]; then
  rm -f $OUT/${TAG}_env_multi_sq.txt
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
    D=/tmp/pmcsq_${TAG}_$(echo $SET | tr ' ' '_' | cut -c1-20); rm -rf $D
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $REPO/tools/env_kernel_probe.py --task $TASK --ctl $CTL --forms multi --replays 8 --nograph > /dev/null 2>> $OUT/${TAG}_env_multi_sq.err
    for C in $SET; do python $REPO/tools/pmc_summary.py $D $C step_kernel_multi >> $OUT/${TAG}_env_multi_sq.txt 2>&1; done
  done
  cat $OUT/${TAG}_env_multi_sq.txt
fi
python $REPO/tools/pmc_env_json.py $OUT/${TAG}_pmc_summary.txt $TAG $TASK $CTL $OUT/${TAG}_env_multi_sq.txt > $OUT/${TAG}_env_kernel_pmc.json
cat $OUT/${TAG}_env_kernel_pmc.json
