#!/bin/bash
# Profiles of the env-step kernel on the GPU box: kernel trace of bench.py + separate PMC passes (FETCH_SIZE, WRITE_SIZE) on the
# SHIPPED kernel in its rollout form.  Usage (from the repo root, under gpurun):  bash tools/gpu_profile_env.sh <tag>
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shipped-ratio > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rocprof_kt.err
DB=$(find /tmp/prof_$TAG/kt -name '*_results.db' | head -1)
python $REPO/tools/rocprof_summary.py "$DB" $OUT/${TAG}_bench_kernel_trace.md "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shipped-ratio" > /dev/null 2>> $OUT/${TAG}_rocprof_kt.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$TAG/pmc_$C -o f -- python $REPO/tools/sweep_env_kernel.py --blocks 0 --forms rollout api --replays 6 --nograph > $OUT/${TAG}_pmc_$C.sweep 2> $OUT/${TAG}_pmc_$C.err
  python $REPO/tools/pmc_summary.py /tmp/prof_$TAG/pmc_$C $C step_kernel >> $OUT/${TAG}_pmc_summary.txt 2>&1
done
cat $OUT/${TAG}_pmc_summary.txt
head -30 $OUT/${TAG}_bench_kernel_trace.md
