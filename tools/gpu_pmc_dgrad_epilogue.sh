#!/bin/bash
# PMC counters of conv3's input gradient without / with the ReLU + BatchNorm backward in its epilogue (tools/dgrad_epilogue_probe.py at
# 4 750 images).  Usage (repo root, under gpurun): bash tools/gpu_pmc_dgrad_epilogue.sh <tag>
set -u
TAG=${1:-r03_dgrad_epi}
R=$(pwd); OUT=$R/gpurun_out/${TAG}_pmc.txt; rm -f $OUT
cd /tmp; export TMPDIR=/tmp
KERN=("30, 7, 0>" "30, 7, 1>" "30, 7, 2>" "relu_bn_bwd_dx_kernel")
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmcd_$(echo $SET | tr ' ' '_' | cut -c1-24); rm -rf $D
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $R/tools/dgrad_epilogue_probe.py --reps 4 > /dev/null 2>> $R/gpurun_out/${TAG}_pmc.err
  for C in $SET; do python $R/tools/pmc_summary.py $D $C "${KERN[@]}" >> $OUT 2>&1; done
done
cat $OUT
