"""GPU time of each phase of one PPO epoch at the bench configuration (synchronised between phases)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    gpus = 1; envs = bench.ENVS_PER_GPU; minibatches = 8; graph = 1


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


def main():
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    agent = A2CAgent("phase", bench.build_params(A, 1))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    for _ in range(2):
        agent.epoch_num += 1
        agent.train_epoch()
    H = agent.horizon_length
    for rep in range(2):
        _, t_graph = timed(lambda: agent._graphs["rollout"].replay())
        batch, t_play = timed(agent.play_steps)
        agent.model.train()
        agent.curr_frames = batch.pop("played_frames")
        _, t_prep = timed(lambda: agent.prepare_dataset(batch))
        agent.model.running_mean_std.eval()
        agent.model.update_stats = True
        _, t_mb0 = timed(lambda: agent.train_actor_critic(0))
        agent.model.update_stats = False
        _, t_mb1 = timed(lambda: agent.train_actor_critic(1))
        _, t_mb8 = timed(lambda: [agent.train_actor_critic(i) for i in range(8)])
        print(f"rollout graph {t_graph:.2f} ms ({t_graph / H * 1e3:.0f} us/step) | play_steps total {t_play:.2f} | "
              f"prepare_dataset {t_prep:.2f} | minibatch(stats) {t_mb0:.2f} | minibatch {t_mb1:.2f} | 8 minibatches {t_mb8:.2f}")


if __name__ == "__main__":
    main()
