#!/bin/bash
# Round-end measurement set on one box: default bench line, kernel trace, PMC counters of the update kernels and of the MLP chain
# kernel.  Usage (repo root, under gpurun): bash tools/gpu_round_end.sh <round tag, e.g. r04>
TAG=${1:-r04}
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.json
bash tools/gpu_trace_bench.sh $TAG > /dev/null 2>&1
bash tools/gpu_pmc_update.sh ${TAG}_update > /dev/null 2>&1
bash tools/gpu_pmc_chain.sh ${TAG}_chain > /dev/null 2>&1
ls -la gpurun_out/ | tail -12
