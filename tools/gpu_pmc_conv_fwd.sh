#!/bin/bash
# SQ counters of the two forward convolution kernels (f32-MFMA and bf16-split) under tools/conv_fwd_probe.py.
#   bash tools/gpu_pmc_conv_fwd.sh <tag>
set -u
TAG=${1:-r06_conv_fwd}
R=$(pwd); OUT=$R/gpurun_out/${TAG}_pmc.txt; mkdir -p $R/gpurun_out; rm -f $OUT
cd /tmp; export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmccf_$(echo $SET | tr ' ' '_' | cut -c1-24); rm -rf $D
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $R/tools/conv_fwd_probe.py --images 4750 --reps 3 > /dev/null 2>> $R/gpurun_out/${TAG}_pmc.err
  for C in $SET; do python $R/tools/pmc_summary.py $D $C "conv_s2_fwd_split_kernel<16" "conv_s2_fwd_split_kernel<32" "conv_s2_fwd_kernel<16" "conv_s2_fwd_kernel<32" >> $OUT 2>&1; done
done
cat $OUT
