#!/bin/bash
# round 3, convolution kernels: parity tests, then the per-layer timing probe.  Usage (repo root, under gpurun): bash tools/gpu_round3_g.sh [images]
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv_kernels.py -q -m gpu 2>&1 | tail -40 > $OUT/r3g_conv_pytest.log
cat $OUT/r3g_conv_pytest.log
timeout 600 python tools/conv_probe.py --images ${1:-4750} > $OUT/r3g_conv_probe.jsonl 2> $OUT/r3g_conv_probe.err
cat $OUT/r3g_conv_probe.jsonl; tail -5 $OUT/r3g_conv_probe.err
