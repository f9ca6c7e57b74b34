#!/bin/bash
# SQ counters of ag_split_gemm on the GPU box (separate rocprofv3 --pmc passes, kernel-trace only).  bash tools/gpu_pmc_split_gemm.sh <tag>
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -f $OUT/${TAG}_split_pmc.txt
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmc_split_$(echo $SET | tr ' ' '_' | cut -c1-40); rm -rf $D
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $REPO/tools/split_gemm_probe.py > /dev/null 2> $OUT/${TAG}_split_pmc.err
  for C in $SET; do python $REPO/tools/pmc_summary.py $D $C split_gemm_kernel Cijk >> $OUT/${TAG}_split_pmc.txt 2>&1; done
done
cat $OUT/${TAG}_split_pmc.txt
