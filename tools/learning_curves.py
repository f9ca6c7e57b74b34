#!/usr/bin/env python3
"""Does the HEADLINE configuration learn?  

Trains Hovering / CTBR with the bench's configuration - 65 536 envs, 196 608-sample minibatches (8 per mini-epoch), MLP(256,256)
- and, next to it, (b) the same 65 536 envs at the reference's minibatch RATIO (48 per mini-epoch, 32 768 samples) and (c) the
shipped small configuration (4 096 envs, 2 048-sample minibatches, scripts/config/ppo_hovering.yaml:54-61), all from seed 0.
Prints one JSON object per run: mean episode reward / length (the AverageMeter over the last `games_to_track` episodes, as the
reference logs them), KL and learning rate every `--every` epochs, plus frames consumed.

    python tools/learning_curves.py --epochs 120 --small-epochs 480 > profiles/r02_learning_curves.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def evaluate_population(agent, steps=None):
    """Return of the FIRST episode of EVERY env from a fresh full reset, under the training policy (sampled actions, as in the
    rollout): what players.py reports for n_games = all envs.  The reference's training meter (AverageMeter over the last 100
    episodes that ENDED, a2c_base.py:100-102) cannot show this at 65 536 envs: once the policy stops crashing, the only episodes
    that end between two time-limit waves are the few that crashed.  Uses the agent's own rollout (hipGraph replay) and reads
    its raw-reward / done buffers; training must not be continued on this agent afterwards (the env was reset)."""
    import math
    H, n = agent.horizon_length, agent.num_actors * agent.num_agents
    max_len = int(getattr(agent.vec_env.env, "max_episode_length", 2400))
    steps = steps or max_len
    agent.obs = agent.env_reset()
    dev = agent.ppo_device
    ret = torch.zeros(n, dtype=torch.float64, device=dev)
    length = torch.zeros(n, dtype=torch.float64, device=dev)
    alive = torch.ones(n, dtype=torch.float64, device=dev)
    for _ in range(math.ceil(steps / H)):
        agent.play_steps()
        rr = agent.raw_rewards_buf.double().view(H, n)
        dn = agent.dones_buf[1:H + 1].double().view(H, n)
        for t in range(H):
            ret += alive * rr[t]
            length += alive
            alive = alive * (1.0 - dn[t])
    full = (length >= 0.99 * max_len).double().mean()      # flew (practically) to the time limit
    return {"eval_return": round(float(ret.mean()), 2), "eval_return_p10": round(float(ret.quantile(0.10)), 2),
            "eval_length": round(float(length.mean()), 1), "eval_full_length_frac": round(float(full), 4),
            "eval_still_alive_frac": round(float(alive.mean()), 4), "eval_steps": steps, "eval_envs": n}


def run(name, envs, minibatches, epochs, every, units=(256, 256), seed=0, extra=None, env_extra=None, evaluate=False):
    class A:
        pass
    A.envs, A.minibatches, A.graph, A.task, A.ctl, A.tuned_gemms = envs, minibatches, 1, "hovering", "rate", 1
    params = bench.build_params(A, 1)
    params["network"]["mlp"]["units"] = list(units)
    params["seed"] = seed
    params["config"]["env_config"]["seed"] = seed
    params["config"].update(extra or {})
    params["config"]["env_config"].update(env_extra or {})
    torch.manual_seed(seed)
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    agent = A2CAgent("curve", params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    curve = []
    t0 = time.time()
    for ep in range(1, epochs + 1):
        agent.epoch_num = ep
        st = agent.train_epoch()
        if ep % every == 0 or ep == 1:
            have = agent.game_rewards.current_size > 0
            curve.append({"epoch": ep, "frames": ep * envs * agent.horizon_length,
                          "reward": round(float(agent.game_rewards.get_mean()[0]), 2) if have else None,
                          "length": round(float(agent.game_lengths.get_mean()[0]), 1) if have else None,
                          "kl": round(st["kl"], 5), "lr": round(st["last_lr"], 7), "a_loss": round(st["a_loss"], 5),
                          "c_loss": round(st["c_loss"], 5),
                          # mean raw reward per env-step over ALL samples of this epoch's rollout (the whole population)
                          "step_reward": round(float(agent.raw_rewards_buf.mean()), 4) if hasattr(agent, "raw_rewards_buf") else None,
                          "exp_var": round(float(agent.diag_dict.get("diagnostics/exp_var", float("nan"))), 4)})
    wall = time.time() - t0
    out = {"run": name, "envs": envs, "minibatch_size": agent.minibatch_size,
           "optimizer_steps_per_epoch": agent.mini_epochs_num * agent.num_minibatches, "mlp": list(units), "epochs": epochs,
           "wall_s": round(wall, 2), "env_steps_per_s": round(epochs * envs * agent.horizon_length / wall), "curve": curve,
           # mean raw reward per env-step over ALL samples of the last rollout (the whole population, not the episodes that ended)
           "final_step_reward": round(float(agent.raw_rewards_buf.mean()), 4) if hasattr(agent, "raw_rewards_buf") else None}
    if evaluate:
        out["eval"] = evaluate_population(agent)
    agent.vec_env.env.hip.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=120)
    ap.add_argument("--small-epochs", type=int, default=480)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0])
    ap.add_argument("--only", default="", help="substring of the run name: run only those")
    ap.add_argument("--fused-epilogues", type=int, default=1, help="0: fuse_gemm_heads / fuse_gemm_input_wgrad off (A/B)")
    a = ap.parse_args()
    extra = {} if a.fused_epilogues else {"fuse_gemm_heads": False, "fuse_gemm_input_wgrad": False}
    runs = [("headline: 65536 envs, 8 minibatches/mini-epoch (196608 samples)", 65536, 8, a.epochs, a.every, (256, 256)),
            ("65536 envs, reference ratio: 48 minibatches/mini-epoch (32768 samples)", 65536, 48, a.epochs, a.every, (256, 256)),
            ("shipped small config: 4096 envs, 48 minibatches/mini-epoch (2048 samples), MLP(256,256)", 4096, 48, a.small_epochs,
             a.every * 4, (256, 256)),
            ("shipped small config with the shipped network [64,128,64]", 4096, 48, a.small_epochs, a.every * 4, (64, 128, 64))]
    for seed in a.seeds:
        for name, envs, mbs, epochs, every, units in runs:
            if a.only in name and epochs > 0:
                tag = name + (f" [seed {seed}]" if seed else "") + ("" if a.fused_epilogues else " [epilogues unfused]")
                print(json.dumps(run(tag, envs, mbs, epochs, every, units=units, seed=seed, extra=extra)), flush=True)
