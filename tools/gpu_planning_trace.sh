#!/bin/bash
# kernel trace of the Planning PPO epoch (trainable CNN).  Usage (repo root, under gpurun): bash tools/gpu_planning_trace.sh <tag>
set -u
TAG=${1:-r03_planning_cnn}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pl_trace; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pl_trace -o t -- python $REPO/tools/bench_planning_ppo.py --envs 16384 --steps 1 --warmup 1 > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
DB=$(find /tmp/pl_trace -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py "$DB" $OUT/${TAG}_kernel_trace.md "rocprofv3 --kernel-trace --stats -- python tools/bench_planning_ppo.py --envs 16384 --steps 1 --warmup 1" | head -60 | cut -c1-200
python $REPO/tools/gap_report.py "$DB" $OUT/${TAG}_gaps.md --window 0.55 0.98 | tail -64 | cut -c1-160
