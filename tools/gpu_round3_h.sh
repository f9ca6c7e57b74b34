#!/bin/bash
# round 3, convolution kernels in the Planning PPO loop.  Usage (repo root, under gpurun): bash tools/gpu_round3_h.sh
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv_kernels.py tests/test_gpu_planning.py tests/test_gpu_cnn_kernels.py -q -m gpu 2>&1 | tail -30 > $OUT/r3h_pytest.log
cat $OUT/r3h_pytest.log
timeout 900 python tools/bench_planning_ppo.py --envs 16384 --steps 2 --warmup 1 > $OUT/r3h_planning_cnn_16384.json 2> $OUT/r3h_planning.err
cat $OUT/r3h_planning_cnn_16384.json; tail -3 $OUT/r3h_planning.err
