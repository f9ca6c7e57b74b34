#!/bin/bash
# In-situ A/B of ag_split_gemm variants: kernel trace of bench.py with the variant pinned; prints the GEMM rows.
# Usage (repo root, under gpurun): bash tools/gpu_trace_variant.sh <variant> [<variant> ...]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  rm -rf /tmp/tv_$V; mkdir -p /tmp/tv_$V
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tv_$V -o kt -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shipped-ratio --no-roofline --split-variant $V > $OUT/tv_$V.json 2> $OUT/tv_$V.err
  DB=$(find /tmp/tv_$V -name '*_results.db' | head -1)
  python $REPO/tools/rocprof_summary.py "$DB" $OUT/tv_$V.md "bench.py --split-variant $V" > /dev/null 2>> $OUT/tv_$V.err
  echo "== variant $V: $(python -c "import json;d=json.load(open('$OUT/tv_$V.json'));print(d['ms_per_step'])")"
  grep "split_gemm_kernel" $OUT/tv_$V.md | cut -c1-60,150-230
done
