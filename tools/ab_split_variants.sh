#!/bin/bash
# In-situ A/B of the update path's options: bench.py (no CPU baseline, no roofline), one line per setting.
run() { timeout 150 python bench.py --no-cpu-baseline --no-shipped-ratio --no-roofline --steps 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run
run --fuse-gemm-input-wgrad 0
run
run --fuse-gemm-input-wgrad 0
run --fuse-gemm-input-wgrad 0 --fuse-gemm-heads 0
