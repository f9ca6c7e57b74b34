cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/r02_split_pmc_v6.txt
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
  D=/tmp/pmc_v6_$(echo $SET | tr ' ' '_' | cut -c1-30); rm -rf $D
  timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $R/tools/split_gemm_probe.py --M 196608 --variants 6 --skip-lib > /dev/null 2>> $R/gpurun_out/r02_split_pmc_v6.err
  for C in $SET; do python $R/tools/pmc_summary.py $D $C split_gemm_kernel >> $R/gpurun_out/r02_split_pmc_v6.txt 2>&1; done
done
cat $R/gpurun_out/r02_split_pmc_v6.txt
