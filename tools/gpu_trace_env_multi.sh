#!/bin/bash
# rocprofv3 kernel trace of the multi-step env launch ALONE (ag_step_multi, 24 steps per launch): the average duration that
# bench.py's `roofline` object must agree with (in the bench's own trace the kernel's row mixes 24-step and 1-step launches).
# Usage: bash tools/gpu_trace_env_multi.sh <tag>
set -u
TAG=${1:-r04}
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/trace_envm_$TAG
CMD="python tools/env_kernel_probe.py --forms multi --replays 52"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d /tmp/trace_envm_$TAG -o t -- python $R/tools/env_kernel_probe.py --forms multi --replays 52 > $R/gpurun_out/${TAG}_env_multi_probe.json 2> $R/gpurun_out/${TAG}_env_multi_probe.err
DB=$(find /tmp/trace_envm_$TAG -name "*results.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_env_multi_kernel_trace.md "rocprofv3 --kernel-trace --stats -- $CMD"
head -16 $R/gpurun_out/${TAG}_env_multi_kernel_trace.md | cut -c1-220; cat $R/gpurun_out/${TAG}_env_multi_probe.json | cut -c1-600
