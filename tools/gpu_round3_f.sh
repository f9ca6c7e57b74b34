#!/bin/bash
# round-3 GPU pass F: kernel trace of the Planning PPO epoch (trainable CNN, frame de-duplication) + the dedup test
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_planning.py -m gpu -q -k "dedup or weighted or drop_in" > $OUT/r3f_pytest.log 2>&1; tail -3 $OUT/r3f_pytest.log
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/r3f_trace; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r3f_trace -o t -- python $REPO/tools/bench_planning_ppo.py --envs 16384 --steps 1 --warmup 1 > $OUT/r3f_trace_bench.json 2> $OUT/r3f_trace.err
DB=$(find /tmp/r3f_trace -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py "$DB" $OUT/r03_planning_cnn_dedup_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python tools/bench_planning_ppo.py --envs 16384 --steps 1 --warmup 1" | head -45 | cut -c1-230
