#!/bin/bash
# Kernel trace of bench.py for one task; prints the split-GEMM rows and the epoch time.  Usage: bash tools/gpu_trace_task.sh <tag> [bench args]
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tt_$TAG; mkdir -p /tmp/tt_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tt_$TAG -o kt -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shipped-ratio --no-roofline "$@" > $OUT/tt_$TAG.json 2> $OUT/tt_$TAG.err
DB=$(find /tmp/tt_$TAG -name '*_results.db' | head -1)
python $REPO/tools/rocprof_summary.py "$DB" $OUT/tt_$TAG.md "bench.py $*" > /dev/null 2>> $OUT/tt_$TAG.err
echo "== $TAG: $(python -c "import json;d=json.load(open('$OUT/tt_$TAG.json'));print(d['ms_per_step'])") ms"
head -24 $OUT/tt_$TAG.md | tail -15 | awk -F'|' '{print substr($2,1,70), $3, $5}'
