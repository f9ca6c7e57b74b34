#!/bin/bash
# round-3 GPU pass E: frame de-duplication for the trainable-CNN Planning policy (tests + the 16 384-env PPO epoch)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_planning.py tests/test_gpu_tasks.py tests/test_gpu_cnn_kernels.py -m gpu -q --maxfail=8 > $OUT/r3e_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3e_pytest.log; tail -6 $OUT/r3e_pytest.log
timeout 600 python tools/bench_planning_ppo.py --envs 16384 --steps 2 --warmup 1 > $OUT/r03_planning_cnn_16384.json 2> $OUT/r3e_planning.err; echo "planning rc=$?"; cat $OUT/r03_planning_cnn_16384.json
timeout 600 python tools/bench_planning_ppo.py --envs 4096 --steps 2 --warmup 1 > $OUT/r03_planning_cnn_4096.json 2>> $OUT/r3e_planning.err; cat $OUT/r03_planning_cnn_4096.json
tail -3 $OUT/r3e_planning.err
