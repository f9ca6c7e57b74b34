#!/bin/bash
# SQ / LDS counters of the MLP chain kernel (tools/chain_probe.py workload).  Usage (repo root, under gpurun): bash tools/gpu_pmc_chain.sh <tag>
set -u
TAG=${1:-r04_chain}
R=$(pwd); OUT=$R/gpurun_out/${TAG}_pmc.txt; rm -f $OUT
cd /tmp; export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC"; do
  D=/tmp/pmcc_$(echo $SET | tr ' ' '_' | cut -c1-24); rm -rf $D
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $R/tools/chain_probe.py --rows 196608 > /dev/null 2>> $R/gpurun_out/${TAG}_pmc.err
  for C in $SET; do python $R/tools/pmc_summary.py $D $C mlp_chain_fwd_kernel split_gemm_kernel input_layer_reg >> $OUT 2>&1; done
done
cat $OUT
