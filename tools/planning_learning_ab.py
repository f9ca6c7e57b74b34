#!/usr/bin/env python3
"""Does the Planning policy (trainable CNN, the shipped YAML) still learn on the hand-written trunk?

Trains scripts/config/ppo_planning.yaml (CTBR, 24-step horizon, 2 048-sample minibatches, 5 mini-epochs) at `--envs` envs in three
arms from the same seeds and prints one JSON object per run (mean episode reward / length over the last `games_to_track` episodes,
KL, losses every `--every` epochs):
  hip_trunk         the default: frame de-duplication + csrc/conv_kernels.hip + lib/network/fused_cnn.py
  hip_trunk_f32_forward   the same with the f32-input-MFMA forward of the 3 x 3 layers instead of the bf16-split one (round 6 A/B)
  library_convs     frame de-duplication, torch's conv2d (MIOpen) layer by layer with the ReLU + BatchNorm kernels
  reference_shape   dedup_frames: false + torch's conv2d: every sample's image stored and convolved, as the reference does

    python tools/planning_learning_ab.py --envs 2048 --epochs 60 --seeds 0 1 > profiles/rNN_planning_learning_ab.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402
import yaml  # noqa: E402


def run(arm, envs, epochs, every, seed):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = yaml.safe_load(open(os.path.join(repo, "scripts", "config", "ppo_planning.yaml")))["params"]
    c = params["config"]
    c.update(num_actors=envs, device="cuda:0", max_epochs=-1, write_summaries=False, print_stats=False, save_frequency=0,
             save_best_after=10 ** 9)
    c["env_config"] = {"use_image": True, "num_envs": envs, "ctl_mode": "rate", "seed": seed, "sim_device": "cuda:0", "headless": True}
    if arm == "reference_shape":
        c["dedup_frames"] = False
    params["seed"] = seed
    torch.manual_seed(seed)
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    from airgym_amd.lib.network import fused_cnn
    fused_cnn.SPLIT_FWD = arm != "hip_trunk_f32_forward"      # (round 6) the two 3 x 3 forwards on the bf16 matrix cores, or the f32 kernel
    agent = A2CAgent("planning_ab", params)
    if arm not in ("hip_trunk", "hip_trunk_f32_forward"):
        for mod in agent.model.modules():
            if hasattr(mod, "fused_trunk"):
                mod.fused_trunk = False
                mod.hip_convs = False
    agent.init_tensors()
    agent.obs = agent.env_reset()
    curve = []
    t0 = time.time()
    for ep in range(1, epochs + 1):
        agent.epoch_num = ep
        st = agent.train_epoch()
        if ep % every == 0 or ep == 1:
            have = agent.game_rewards.current_size > 0
            curve.append({"epoch": ep, "reward": round(float(agent.game_rewards.get_mean()[0]), 2) if have else None,
                          "length": round(float(agent.game_lengths.get_mean()[0]), 1) if have else None,
                          "kl": round(st["kl"], 5), "lr": round(st["last_lr"], 7), "a_loss": round(st["a_loss"], 5),
                          "c_loss": round(st["c_loss"], 5)})
    torch.cuda.synchronize()
    wall = time.time() - t0
    out = {"arm": arm, "seed": seed, "envs": envs, "minibatch_size": agent.minibatch_size, "epochs": epochs,
           "dedup": bool(getattr(agent, "_dedup", False)), "wall_s": round(wall, 1),
           "env_steps_per_s": round(epochs * envs * agent.horizon_length / wall), "curve": curve}
    agent.vec_env.env.hip.close()
    del agent
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--arms", nargs="+", default=["hip_trunk", "library_convs", "reference_shape"])
    a = ap.parse_args()
    for seed in a.seeds:
        for arm in a.arms:
            print(json.dumps(run(arm, a.envs, a.epochs, a.every, seed)), flush=True)
