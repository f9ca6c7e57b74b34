#!/bin/bash
# round-3 GPU pass A: GPU test-suite, headline bench with A/B legs, wgrad / env-kernel probes.   bash tools/gpu_round3_a.sh
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --maxfail=5 > $OUT/r3a_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3a_pytest.log
tail -5 $OUT/r3a_pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $OUT/r3a_bench.json 2> $OUT/r3a_bench.err; echo "bench rc=$?"
for leg in "--fuse-rollout-tail 0" "--split-wgrad 0" "--fuse-rollout-tail 0 --split-wgrad 0"; do
  tag=$(echo $leg | tr -d ' -'); timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shipped-ratio --no-roofline $leg > $OUT/r3a_bench_$tag.json 2> $OUT/r3a_bench_$tag.err
done
AIRGYM_EXPERIMENTS=1 timeout 200 python tools/wgrad_probe.py > $OUT/r3a_wgrad_probe.jsonl 2> $OUT/r3a_wgrad_probe.err
timeout 200 python tools/env_kernel_probe.py > $OUT/r3a_env_probe.jsonl 2> $OUT/r3a_env_probe.err
timeout 100 python bench.py --gpus 2 --steps 1 > $OUT/r3a_bench_gpus2.json 2>&1; echo "gpus2 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3a_bench*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), ((d.get('roofline') or {}).get('rollout_fused') or {}).get('frac'), d.get('error'))
    except Exception as e:
        print(f, 'unparsed', e)
PY
cat $OUT/r3a_wgrad_probe.jsonl; cat $OUT/r3a_env_probe.jsonl | cut -c1-400
