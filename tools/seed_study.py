#!/usr/bin/env python3
"""Seed study of the fast path: Hovering / CTBR at 65 536 envs, MLP(256,256),
>= 5 seeds x {headline 196 608-sample minibatches, reference-ratio 32 768-sample minibatches} x
{default, fused GEMM epilogues off, split GEMMs off (library f32), time-out fix on}.  One JSON line per run (the learning curve),
then a summary table: median / IQR of the mean episode reward at the listed epochs and the number of seeds below 500.

    python tools/seed_study.py --seeds 0 1 2 3 4 --epochs 200 > profiles/r03_seed_study.jsonl
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.learning_curves import run  # noqa: E402

ARMS = {
    "default": {},
    "epilogues_off": {"fuse_gemm_heads": False, "fuse_gemm_input_wgrad": False},
    "split_gemm_off": {"use_split_gemm": False},
    "fix_time_outs": {"_env": {"fix_time_outs": True}},
    # opt-in: a full reset starts every env at its own progress, so the 2 400-step time limit does not end (nearly) all
    # episodes of the shard inside the same rollouts (AG_FLAG_STAGGER_PHASE)
    "stagger": {"_env": {"stagger_episode_phase": True}},
    "stagger_fix_time_outs": {"_env": {"stagger_episode_phase": True, "fix_time_outs": True}},
}
CONFIGS = {"headline_196608": 8, "ratio_32768": 48}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--arms", nargs="+", default=list(ARMS))
    ap.add_argument("--configs", nargs="+", default=list(CONFIGS))
    ap.add_argument("--report", type=int, nargs="+", default=[120, 160, 200])
    a = ap.parse_args()
    table = {}
    for cfg in a.configs:
        for arm in a.arms:
            for seed in a.seeds:
                extra = dict(ARMS[arm])
                env_extra = extra.pop("_env", None)
                out = run(f"{cfg} / {arm} / seed {seed}", 65536, CONFIGS[cfg], a.epochs, 10, seed=seed, extra=extra,
                          env_extra=env_extra)
                out.update(config=cfg, arm=arm, seed=seed)
                print(json.dumps(out), flush=True)
                by_epoch = {c["epoch"]: c for c in out["curve"]}
                table.setdefault((cfg, arm), []).append({e: by_epoch[e]["reward"] for e in a.report if e in by_epoch})
    summary = []
    for (cfg, arm), rows in table.items():
        row = {"config": cfg, "arm": arm, "seeds": len(rows)}
        for e in a.report:
            vals = sorted(r[e] for r in rows if r.get(e) is not None)
            if not vals:
                continue
            q = statistics.quantiles(vals, n=4) if len(vals) >= 4 else [vals[0], statistics.median(vals), vals[-1]]
            row[f"epoch_{e}"] = {"median": round(statistics.median(vals), 1), "q1": round(q[0], 1), "q3": round(q[2], 1),
                                 "min": round(vals[0], 1), "max": round(vals[-1], 1), "below_500": sum(v < 500 for v in vals)}
        summary.append(row)
    print(json.dumps({"summary": summary}), flush=True)


if __name__ == "__main__":
    main()
