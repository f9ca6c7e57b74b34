timeout 500 python tools/learning_curves.py --epochs 160 --only "reference ratio" --seeds 0 1 2 2>/dev/null > gpurun_out/lc_ratio_fused.jsonl
timeout 300 python tools/learning_curves.py --epochs 160 --only "reference ratio" --seeds 0 1 --fused-epilogues 0 2>/dev/null > gpurun_out/lc_ratio_unfused.jsonl
python - <<'PY'
import json
for f in ("gpurun_out/lc_ratio_fused.jsonl","gpurun_out/lc_ratio_unfused.jsonl"):
    for l in open(f):
        d=json.loads(l); print(d["run"][-45:], d["wall_s"], [(p["epoch"], p["reward"], p["c_loss"]) for p in d["curve"] if p["epoch"]%20==0])
PY
