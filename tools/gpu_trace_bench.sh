#!/bin/bash
# rocprofv3 kernel trace of the headline bench command -> gpurun_out/<tag>_bench_kernel_trace.md.  Usage: bash tools/gpu_trace_bench.sh <tag>
set -u
TAG=${1:-r04}
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio --no-side-configs"
rm -rf /tmp/trace_$TAG
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d /tmp/trace_$TAG -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio --no-side-configs > $R/gpurun_out/${TAG}_trace_bench.json 2> $R/gpurun_out/${TAG}_trace_bench.err
DB=$(find /tmp/trace_$TAG -name "*results.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_bench_kernel_trace.md "rocprofv3 --kernel-trace --stats -- $CMD"
head -40 $R/gpurun_out/${TAG}_bench_kernel_trace.md | cut -c1-200
python $R/tools/gap_report.py $DB $R/gpurun_out/${TAG}_epoch_gaps.md > /dev/null 2> $R/gpurun_out/${TAG}_epoch_gaps.err; head -70 $R/gpurun_out/${TAG}_epoch_gaps.md; tail -3 $R/gpurun_out/${TAG}_epoch_gaps.err
