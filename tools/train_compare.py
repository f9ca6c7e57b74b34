#!/usr/bin/env python3
"""Does the hand-scheduled / fused PPO path LEARN like the plain autograd path?  Trains Hovering (CTBR) twice from the same
seed - fused kernels on, then off - and prints the mean episode reward / length over epochs."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def run(fused, envs, epochs, every, minibatches=4, graph=None):
    class A:
        pass
    A.envs, A.minibatches, A.graph, A.task, A.ctl, A.tuned_gemms = envs, minibatches, int(fused if graph is None else graph), "hovering", "rate", 1
    params = bench.build_params(A, 1)
    c = params["config"]
    c["use_fused_update"] = c["use_fused_rollout"] = c["use_fused_loss"] = c["use_fused_adam"] = bool(fused)
    torch.manual_seed(0)
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    agent = A2CAgent("cmp", params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    out = []
    for ep in range(1, epochs + 1):
        agent.epoch_num = ep
        st = agent.train_epoch()
        if ep % every == 0:
            out.append({"epoch": ep, "reward": round(float(agent.game_rewards.get_mean()[0]), 3),
                        "length": round(float(agent.game_lengths.get_mean()[0]), 1), "kl": round(st["kl"], 5),
                        "lr": round(st["last_lr"], 6)})
    agent.vec_env.env.hip.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--minibatches", type=int, default=4)
    a = ap.parse_args()
    for fused, graph in ((1, 1), (1, 0), (0, 0)):
        print(json.dumps({"fused": bool(fused), "hip_graphs": bool(graph),
                          "curve": run(fused, a.envs, a.epochs, a.every, a.minibatches, graph)}))
