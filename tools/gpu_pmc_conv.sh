#!/bin/bash
# PMC counters of the convolution kernels (tools/conv_probe.py at 4 750 images).  Usage (repo root, under gpurun): bash tools/gpu_pmc_conv.sh <tag>
set -u
TAG=${1:-r03_conv}
R=$(pwd); OUT=$R/gpurun_out/${TAG}_pmc.txt; rm -f $OUT
cd /tmp; export TMPDIR=/tmp
KERN="conv_s2_fwd_kernel<16 conv_s2_fwd_kernel<32 conv_s2_dgrad_kernel<16 conv_s2_dgrad_kernel<32 conv_s2_wgrad_kernel<16 conv_s2_wgrad_kernel<32 conv1_fwd_kernel conv1_wgrad_kernel"
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmcc_$(echo $SET | tr ' ' '_' | cut -c1-24); rm -rf $D
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $R/tools/conv_probe.py --reps 2 > /dev/null 2>> $R/gpurun_out/${TAG}_pmc.err
  for C in $SET; do python $R/tools/pmc_summary.py $D $C $KERN >> $OUT 2>&1; done
done
cat $OUT
