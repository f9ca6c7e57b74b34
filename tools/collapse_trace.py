#!/usr/bin/env python3
"""Why does the headline job un-learn after episodes first reach the time limit?  

Hovering / CTBR, 65 536 envs, MLP(256,256), 8 minibatches of 196 608 samples per mini-epoch (the bench's configuration), per
seed and arm ONE JSON line per EPOCH with what the reference's dashboards would show around the collapse:

    lr (after the epoch's last optimizer step), KL (mean over the epoch),
    logstd mean, explained variance of the value function, clip fraction (mean over the mini-epochs), gradient norm BEFORE
    clipping (last optimizer step of the epoch: sqrt of the sum ag_adam_clip_step's first phase leaves in the optimizer state),
    value-normaliser variance, mean episode reward / length (the reference's meter: the last `games_to_track` = 100 episodes that
    ENDED), and - what that meter cannot show - statistics of the WHOLE population: `step_reward` = mean raw reward per env-step over
    the epoch's 24 x 65 536 samples, `ended` / `ended_len` / `ended_reward` = number, mean length and mean reward of ALL episodes
    that ended in the epoch, `mean_age` = mean progress counter over the envs at the end of the epoch.

Arms (none changes a default of the shipped configuration; all are YAML keys):
    default            lr_schedule adaptive, schedule_type legacy (per-minibatch rule), bounds [1e-6, 1e-2]
                       (lib/core/schedulers.py:19-32, a2c_continuous.py:104-123 of the reference)
    identity_3e-4      lr_schedule identity: the YAML's learning_rate 3e-4 throughout
    adaptive_max_1e-3  the same adaptive rule with `max_lr: 1e-3`
    standard           schedule_type standard: the rule is evaluated once per mini-epoch on the mean KL

    python tools/collapse_trace.py --seeds 0 1 2 3 4 --epochs 200 > profiles/r05_collapse_trace.jsonl
    python tools/collapse_trace.py --summarise profiles/r05_collapse_trace.jsonl > profiles/r05_collapse_trace.md
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ARMS = {
    "default": {},
    "identity_3e-4": {"lr_schedule": "identity"},
    "adaptive_max_1e-3": {"max_lr": 1e-3},
    "standard": {"schedule_type": "standard"},
    # opt-in env flags of round 4 (first-episode phase staggered; time-outs flagged so that the loop bootstraps them)
    "stagger": {"_env": {"stagger_episode_phase": True}},
    "stagger_fix_time_outs": {"_env": {"stagger_episode_phase": True, "fix_time_outs": True}},
    "fix_time_outs": {"_env": {"fix_time_outs": True}},
}


def run(arm, seed, epochs, envs=65536, minibatches=8, extra=None):
    import torch

    import bench

    class A:
        pass
    A.envs, A.minibatches, A.graph, A.task, A.ctl, A.tuned_gemms = envs, minibatches, 1, "hovering", "rate", 1
    params = bench.build_params(A, 1)
    params["seed"] = seed
    params["config"]["env_config"]["seed"] = seed
    params["config"]["use_diagnostics"] = True
    arm_cfg = dict(ARMS[arm])
    params["config"]["env_config"].update(arm_cfg.pop("_env", {}))
    params["config"].update(arm_cfg)
    params["config"].update(extra or {})
    torch.manual_seed(seed)
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    agent = A2CAgent("trace", params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    opt = agent.optimizer
    rows = []
    t0 = time.time()
    for ep in range(1, epochs + 1):
        agent.epoch_num = ep
        st = agent.train_epoch()
        have = agent.game_rewards.current_size > 0
        partial = opt.state[4:36].view(torch.float32)            # 64 block sums of g^2 of the last optimizer step (before clipping)
        dd = agent.diag_dict
        clips = [float(v) for k, v in dd.items() if k.startswith("diagnostics/clip_frac/")]
        vms = agent.model.value_mean_std if agent.normalize_value else None
        eps = agent.ep_stats.sum(0).tolist()                     # this epoch's [count, sum reward, sum shaped, sum length]
        rows.append({"arm": arm, "seed": seed, "epoch": ep,
                     "reward": round(float(agent.game_rewards.get_mean()[0]), 2) if have else None,
                     "length": round(float(agent.game_lengths.get_mean()[0]), 1) if have else None,
                     "lr": st["last_lr"], "kl": st["kl"], "a_loss": st["a_loss"], "c_loss": st["c_loss"],
                     "logstd": round(float(agent.model.logstd.detach().mean()), 5),
                     "step_reward": round(float(agent.raw_rewards_buf.mean()), 5),
                     "ended": int(eps[0]), "ended_len": round(eps[3] / max(eps[0], 1.0), 1),
                     "ended_reward": round(eps[1] / max(eps[0], 1.0), 2),
                     "mean_age": round(float(agent.vec_env.env.progress_buf.float().mean()), 1),
                     "exp_var": round(float(dd.get("diagnostics/exp_var", float("nan"))), 5),
                     "clip_frac": round(sum(clips) / len(clips), 5) if clips else None,
                     "grad_norm": round(float(partial.sum().sqrt()), 5),
                     "value_var": round(float(vms.running_var.reshape(-1)[0]), 5) if vms is not None else None})
    wall = time.time() - t0
    from tools.learning_curves import evaluate_population
    ev = evaluate_population(agent)          # every env's first episode from a fresh reset, under the final policy
    agent.vec_env.env.hip.close()
    return rows, wall, ev


def summarise(path):
    by, evals, run_no = {}, {}, {}
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        if "epoch" in r:
            if r["epoch"] == 1:                      # a new run of this (arm, seed) starts (e.g. the 120-epoch runs behind the 200-epoch ones)
                run_no[(r["arm"], r["seed"])] = run_no.get((r["arm"], r["seed"]), -1) + 1
            by.setdefault((r["arm"] + ("" if run_no[(r["arm"], r["seed"])] == 0 else f" (run {run_no[(r['arm'], r['seed'])] + 1})"),
                           r["seed"]), []).append(r)
        elif "run_done" in r and "eval_return" in r["run_done"]:
            d = r["run_done"]
            evals.setdefault((d["arm"], d["epochs"]), []).append(d)
    arms = []
    for (arm, _s) in by:
        if arm not in arms:
            arms.append(arm)
    out = ["# r05 collapse trace (tools/collapse_trace.py; Hovering CTBR, 65 536 envs, 8 x 196 608-sample minibatches, MI355X)", ""]
    out.append("Median over seeds of the mean episode reward at epochs 100 / 120 / 160 / 200; seeds below 3 000 and below 500 at 200.")
    out.append("")
    out.append("| arm | seeds | e100 | e120 | e160 | e200 | min e200 | < 3000 | < 500 |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    for arm in arms:
        runs = [by[k] for k in by if k[0] == arm]
        cells = []
        for e in (100, 120, 160, 200):
            vals = [r[e - 1]["reward"] or 0.0 for r in runs if len(r) >= e]
            cells.append(f"{statistics.median(vals):.0f}" if vals else "-")
        fin = [r[-1]["reward"] or 0.0 for r in runs]
        out.append(f"| {arm} | {len(runs)} | " + " | ".join(cells) + f" | {min(fin):.0f} | {sum(v < 3000 for v in fin)} | {sum(v < 500 for v in fin)} |")
    if evals:
        out.append("")
        out.append("Whole-population evaluation after the last epoch (tools/learning_curves.py evaluate_population: fresh reset, every env's "
                   "first episode under the training policy, 2 400 steps): mean return per seed, median, worst seed; fraction of envs "
                   "that flew the whole episode.")
        out.append("")
        out.append("| arm | epochs | eval return per seed | median | min | full-length fraction (median) | meter at the same epoch (median) |")
        out.append("|---|---|---|---|---|---|---|")
        for (arm, ep), ds in evals.items():
            rets = [d["eval_return"] for d in ds]
            out.append(f"| {arm} | {ep} | " + " / ".join(f"{v:.0f}" for v in rets) + f" | {statistics.median(rets):.0f} | {min(rets):.0f} | "
                       f"{statistics.median(d['eval_full_length_frac'] for d in ds):.3f} | "
                       f"{statistics.median((d['final_reward'] or 0.0) for d in ds):.0f} |")
    out.append("")
    out.append("Per-arm medians over seeds in 20-epoch windows (lr = end of epoch; grad = pre-clip norm of the epoch's last step; "
               "clip = PPO clip fraction; ev = explained variance):")
    for arm in arms:
        runs = [by[k] for k in by if k[0] == arm]
        out.append("")
        out.append(f"## {arm}")
        out.append("")
        out.append("| epochs | reward | length | lr | kl | logstd | ev | clip | grad | c_loss | step_reward | ended/epoch | ended_len | mean_age |")
        out.append("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
        n = min(len(r) for r in runs)
        for lo in range(0, n, 20):
            hi = min(lo + 20, n)

            def med(key):
                vals = [r[i][key] for r in runs for i in range(lo, hi) if r[i].get(key) is not None and r[i][key] == r[i][key]]
                return statistics.median(vals) if vals else float("nan")
            out.append(f"| {lo + 1}-{hi} | {med('reward'):.0f} | {med('length'):.0f} | {med('lr'):.2e} | {med('kl'):.4f} | "
                       f"{med('logstd'):.3f} | {med('exp_var'):.3f} | {med('clip_frac'):.3f} | {med('grad_norm'):.3f} | {med('c_loss'):.4f} | "
                       f"{med('step_reward'):.3f} | {med('ended'):.0f} | {med('ended_len'):.0f} | {med('mean_age'):.0f} |")
    print("\n".join(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--arms", nargs="+", default=list(ARMS))
    ap.add_argument("--summarise", default="")
    a = ap.parse_args()
    if a.summarise:
        return summarise(a.summarise)
    for arm in a.arms:
        for seed in a.seeds:
            rows, wall, ev = run(arm, seed, a.epochs)
            for r in rows:
                print(json.dumps(r))
            done = {"arm": arm, "seed": seed, "epochs": a.epochs, "wall_s": round(wall, 1), "final_reward": rows[-1]["reward"]}
            done.update(ev)
            print(json.dumps({"run_done": done}), flush=True)


if __name__ == "__main__":
    main()
