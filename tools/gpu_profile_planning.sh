#!/bin/bash
# Kernel trace of one Planning PPO epoch (CNN policy) at a reduced env count with the full-scale minibatch size.
# Usage (repo root, under gpurun): bash tools/gpu_profile_planning.sh <tag> [envs] [minibatches]
set -u
TAG=${1:-r02_planning}; ENVS=${2:-4096}; MB=${3:-6}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pp_$TAG; mkdir -p /tmp/pp_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pp_$TAG -o kt -- python $REPO/tools/bench_planning_ppo.py --envs $ENVS --minibatches $MB --steps 1 --warmup 1 ${4:-} > $OUT/${TAG}.json 2> $OUT/${TAG}.err
DB=$(find /tmp/pp_$TAG -name '*_results.db' | head -1)
python $REPO/tools/rocprof_summary.py "$DB" $OUT/${TAG}_kernel_trace.md "python tools/bench_planning_ppo.py --envs $ENVS --minibatches $MB --steps 1 --warmup 1 ${4:-}" > /dev/null 2>> $OUT/${TAG}.err
cat $OUT/${TAG}.json
head -40 $OUT/${TAG}_kernel_trace.md | cut -c1-200
