#!/bin/bash
# round-3 GPU pass D: Tracking's fused first-layer backward (D = 48), full suite, Tracking / Hovering bench lines, world-frame-omega cascade variant
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/r3d_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3d_pytest.log; tail -4 $OUT/r3d_pytest.log
timeout 300 python bench.py --task tracking --ctl vel --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio > $OUT/r03_bench_tracking.json 2> $OUT/r3d_bench_tracking.err
timeout 300 python bench.py --task tracking --ctl vel --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio --no-roofline --fuse-gemm-input-wgrad 0 > $OUT/r3d_bench_tracking_unfused.json 2>> $OUT/r3d_bench_tracking.err
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_default.json 2> $OUT/r3d_bench.err
python - <<'PY'
import json, glob
for f in ['gpurun_out/r03_bench_tracking.json', 'gpurun_out/r3d_bench_tracking_unfused.json', 'gpurun_out/r03_bench_default.json']:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, round(d['value'] / 1e6, 2), round(d['ms_per_step'], 2), (d.get('roofline') or {}).get('frac'), ((d.get('roofline') or {}).get('rollout_fused') or {}).get('frac'), (d.get('shipped_ratio') or {}).get('value'))
    except Exception as e:
        print(f, 'unparsed', e)
PY
AIRGYM_EXPERIMENTS=1 AIRGYM_EXP_LIB=airgym_amd/_native/libairgym_hip_exp_worldw.so timeout 300 python tools/cascade_sweep.py --checkpoint runs/ref_ckpt/planning_cnn_rate.pth --tag world_frame_omega --signs ppp ppn > $OUT/r03_cascade_worldw.jsonl 2> $OUT/r03_cascade_worldw.err
cut -c1-330 $OUT/r03_cascade_worldw.jsonl
