#!/bin/bash
# What the driver runs at round end, plus the side measurements: every -m gpu test, smoke(), the headline bench, the Planning PPO
# epoch.  Usage (repo root, under gpurun): bash tools/gpu_full_check.sh <tag>
set -u
TAG=${1:-check}
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $OUT/${TAG}_pytest.log; cat $OUT/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; cat $OUT/${TAG}_bench.json | cut -c1-900
# (the Planning PPO epoch is part of the bench line since round 4: side_configs.planning_cnn_16384)
