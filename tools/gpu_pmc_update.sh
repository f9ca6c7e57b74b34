#!/bin/bash
# PMC counters of the update's kernels as bench.py launches them (eager minibatches at the headline configuration):
# HBM traffic (FETCH_SIZE, WRITE_SIZE; separate passes), matrix-pipe occupancy and where the waves wait.  Usage (repo root, under gpurun):
#   bash tools/gpu_pmc_update.sh <tag>
set -u
TAG=${1:-r03_update}
R=$(pwd); OUT=$R/gpurun_out/${TAG}_pmc.txt; rm -f $OUT
cd /tmp; export TMPDIR=/tmp
KERN="split_gemm_kernel<true split_gemm_kernel<false split_wgrad_kernel split_wgrad_fin_kernel mlp_chain_fwd_kernel input_layer_reg sum_rows_stage1 step_kernel_ws2"
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  D=/tmp/pmcu_$(echo $SET | tr ' ' '_' | cut -c1-24); rm -rf $D
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-shipped-ratio --no-roofline --no-side-configs > /dev/null 2>> $R/gpurun_out/${TAG}_pmc.err
  for C in $SET; do python $R/tools/pmc_summary.py $D $C $KERN >> $OUT 2>&1; done
done
cat $OUT
