#!/bin/bash
# round-3 GPU pass B: full GPU suite, split-GEMM row-tile A/B (4 vs 8 waves), kernel trace of the bench, PMC traffic of the env kernels
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/r3b_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3b_pytest.log; tail -4 $OUT/r3b_pytest.log
AIRGYM_EXPERIMENTS=1 AIRGYM_SPLIT_WM=4 timeout 600 python -m pytest tests/test_gpu_split_gemm.py tests/test_gpu_split_wgrad.py tests/test_gpu_rollout_kernels.py tests/test_gpu_golden.py -m gpu -q --maxfail=8 > $OUT/r3b_pytest_wm4.log 2>&1; echo "pytest wm4 rc=$?" >> $OUT/r3b_pytest_wm4.log; tail -4 $OUT/r3b_pytest_wm4.log
for rep in 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shipped-ratio > $OUT/r3b_bench_wm2_$rep.json 2> $OUT/r3b_bench_wm2_$rep.err
  AIRGYM_EXPERIMENTS=1 AIRGYM_SPLIT_WM=4 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shipped-ratio > $OUT/r3b_bench_wm4_$rep.json 2> $OUT/r3b_bench_wm4_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3b_bench_wm*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, round(d['value'] / 1e6, 2), round(d['ms_per_step'], 2), [(round(u['us_per_launch'], 1), u['kernel'][:28]) for u in d.get('update_kernels', []) if 'split' in u['kernel']])
    except Exception as e:
        print(f, 'unparsed', e)
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/r3b_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r3b_trace -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio > $OUT/r3b_trace_bench.json 2> $OUT/r3b_trace.err
python $REPO/tools/rocprof_summary.py /tmp/r3b_trace > $OUT/r3b_bench_kernel_trace.md 2>&1; head -30 $OUT/r3b_bench_kernel_trace.md
cd $REPO; bash tools/gpu_pmc_env.sh r03 2>&1 | tail -12
