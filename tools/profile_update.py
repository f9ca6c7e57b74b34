"""Run warm PPO epochs then a burst of minibatch steps (for rocprofv3 --kernel-trace); with --dump DB prints the last
dispatches of a trace in time order."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(db_path, n):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    rows = list(cur.execute(f"select name, start, end, duration from kernels order by start desc limit {n}"))[::-1]
    t0 = rows[0][1]
    for name, s, e, d in rows:
        print(f"{(s - t0) / 1e3:9.1f} us  +{d / 1e3:7.1f} us  {name[:110]}")
    print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, busy {sum(r[3] for r in rows) / 1e3:.1f} us")


def main():
    import torch
    import bench
    from tools.phase_times import A
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    A.task = os.getenv("PROFILE_TASK", "hovering")
    A.ctl = os.getenv("PROFILE_CTL", "rate")
    agent = A2CAgent("phase", bench.build_params(A, 1))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    for _ in range(2):
        agent.epoch_num += 1
        agent.train_epoch()
    batch = agent.play_steps()
    agent.model.train()
    agent.curr_frames = batch.pop("played_frames")
    agent.prepare_dataset(batch)
    agent.model.running_mean_std.eval()
    agent.model.update_stats = False
    for i in range(16):
        agent.train_actor_critic(i % 8)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        dump(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    else:
        main()
