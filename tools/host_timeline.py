#!/usr/bin/env python3
"""Where the host thread is during a PPO epoch of the headline job (AIRGYM_HOST_TRACE stamps of A2CAgent.train_epoch):
mean microseconds between consecutive stamps over the timed epochs, incl. the stretch from one epoch's end to the next one's begin
(the caller's loop).  The GPU has nothing queued from `stats_on_host` until the next `rollout_enqueued`.

    python tools/host_timeline.py [--envs 65536] [--epochs 12]
"""
import argparse
import collections
import json
import os
import sys

os.environ["AIRGYM_HOST_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from airgym_amd.lib.agent.a2c_continuous import A2CAgent  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=12)
    a = ap.parse_args()

    class Args:
        envs = a.envs; minibatches = 8; task = "hovering"; ctl = "rate"; tuned_gemms = 1; graph = 1
    agent = A2CAgent("timeline", bench.build_params(Args, 1))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    for _ in range(4):
        agent.epoch_num += 1
        agent.train_epoch()
    torch.cuda.synchronize()
    agent._host_trace.clear()
    for _ in range(a.epochs):
        agent.epoch_num += 1
        agent.train_epoch()
    tr = agent._host_trace
    seg = collections.OrderedDict()
    for (l0, t0), (l1, t1) in zip(tr, tr[1:]):
        seg.setdefault(f"{l0} -> {l1}", []).append((t1 - t0) * 1e6)
    out = {k: round(sum(v) / len(v), 1) for k, v in seg.items()}
    idle = out.get("stats_on_host -> epoch_end", 0) + out.get("epoch_end -> epoch_begin", 0) + out.get("epoch_begin -> rollout_enqueued", 0)
    print(json.dumps({"envs": a.envs, "epochs": a.epochs, "host_us_between_stamps": out,
                      "gpu_queue_empty_us_per_epoch": round(idle, 1)}))


if __name__ == "__main__":
    main()
