timeout 300 python tools/sweep_env_kernel.py --blocks 0 --forms rollout --stagger 0 1 2 3 4 6 8 0 --replays 30 2>/dev/null > gpurun_out/stagger.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/stagger.jsonl'):
    d=json.loads(l); print(d["stagger"], round(d["us_per_step"],3), round(d["frac"],4))
PY
