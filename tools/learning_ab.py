#!/usr/bin/env python3
"""Same-seed A/B of the hand-written path against the plain torch path: Hovering / CTBR, 65 536 envs,
MLP(256,256), 196 608-sample minibatches, the SAME seeds in both arms:

    fused     every hand-written kernel of the headline (chain forward + fused rollout step, split-bf16 GEMMs, fused loss, HIP Adam)
    autograd  model forward / loss / backward / Adam in torch (autograd, library f32 GEMMs), env.step per rollout step

each with the default LR bounds and with the opt-in `max_lr: 1e-3`.  Per run: the reference's episode meter every 10 epochs, the
whole-population reward per env-step at epochs 120 and 200, and (at the end) every env's first episode from a fresh reset under the
final policy.  Two trajectories from one seed diverge within a few epochs (float32 rounding; chaotic), so what is compared is the
DISTRIBUTION over seeds: the summary prints min / median / max per arm and whether the fused arm's values lie inside the autograd
arm's spread.

    python tools/learning_ab.py --seeds 0 1 2 3 4 --epochs 200 > profiles/r06_learning_ab.jsonl
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.learning_curves import run  # noqa: E402

AUTOGRAD = {"use_fused_update": False, "use_fused_rollout": False, "use_fused_loss": False, "use_fused_adam": False}
ARMS = {
    "fused": {},
    "autograd": dict(AUTOGRAD),
    "fused_max_lr_1e-3": {"max_lr": 1e-3},
    "autograd_max_lr_1e-3": dict(AUTOGRAD, max_lr=1e-3),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--arms", nargs="+", default=list(ARMS))
    a = ap.parse_args()
    rows = {}
    for arm in a.arms:
        for seed in a.seeds:
            out = run(f"{arm} / seed {seed}", 65536, 8, a.epochs, 10, seed=seed, extra=ARMS[arm], evaluate=True)
            out.update(arm=arm, seed=seed)
            print(json.dumps(out), flush=True)
            by = {c["epoch"]: c for c in out["curve"]}
            rows.setdefault(arm, []).append({
                "seed": seed, "meter_best_by_120": max((c["reward"] or 0.0) for c in out["curve"] if c["epoch"] <= 120),
                "step_reward_120": by.get(120, {}).get("step_reward"), "step_reward_final": out["final_step_reward"],
                "eval_return": out["eval"]["eval_return"], "env_steps_per_s": out["env_steps_per_s"]})
    summary = {}
    for arm, rs in rows.items():
        summary[arm] = {}
        for k in ("meter_best_by_120", "step_reward_120", "step_reward_final", "eval_return", "env_steps_per_s"):
            v = sorted(r[k] for r in rs if r[k] is not None)
            if v:
                summary[arm][k] = {"min": round(v[0], 3), "median": round(statistics.median(v), 3), "max": round(v[-1], 3)}
    for pair in (("fused", "autograd"), ("fused_max_lr_1e-3", "autograd_max_lr_1e-3")):
        if all(p in summary for p in pair):
            f, g = summary[pair[0]], summary[pair[1]]
            summary[f"{pair[0]}_vs_{pair[1]}"] = {
                k: {"fused_median_minus_autograd_median": round(f[k]["median"] - g[k]["median"], 3),
                    "fused_median_inside_autograd_spread": bool(g[k]["min"] <= f[k]["median"] <= g[k]["max"]),
                    "fused_median_at_least_autograd_min": bool(f[k]["median"] >= g[k]["min"])}
                for k in ("meter_best_by_120", "step_reward_120", "step_reward_final", "eval_return") if k in f and k in g}
    print(json.dumps({"summary": summary}), flush=True)


if __name__ == "__main__":
    main()
