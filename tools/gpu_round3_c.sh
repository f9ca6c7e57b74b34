#!/bin/bash
# round-3 GPU pass C: GPU suite on the shipped build (256-row split-GEMM tiles), kernel trace of the bench, seed study, cascade sweep
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/r3c_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c_pytest.log; tail -4 $OUT/r3c_pytest.log
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/r3c_trace; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r3c_trace -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio > $OUT/r3c_trace_bench.json 2> $OUT/r3c_trace.err
DB=$(find /tmp/r3c_trace -name "*.db" | head -1); echo "db=$DB"
python $REPO/tools/rocprof_summary.py "$DB" $OUT/r03_bench_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio" | head -32
cd $REPO
timeout 900 python tools/seed_study.py --seeds 0 1 2 3 4 --epochs 200 > $OUT/r03_seed_study.jsonl 2> $OUT/r03_seed_study.err; echo "seed study rc=$?"; tail -1 $OUT/r03_seed_study.jsonl | cut -c1-3000
timeout 500 python tools/cascade_sweep.py --checkpoint runs/ref_ckpt/planning_cnn_rate.pth > $OUT/r03_cascade_sweep.jsonl 2> $OUT/r03_cascade_sweep.err; echo "cascade rc=$?"
AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=airgym_amd/_native/libairgym_hip_exp_worldw.so timeout 300 python tools/cascade_sweep.py --checkpoint runs/ref_ckpt/planning_cnn_rate.pth --tag world_frame_omega --signs +++ +-- --- >> $OUT/r03_cascade_sweep.jsonl 2>> $OUT/r03_cascade_sweep.err
cut -c1-330 $OUT/r03_cascade_sweep.jsonl
