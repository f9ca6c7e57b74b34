#!/usr/bin/env python3
"""A/B the launch geometry / variants of the env-step kernel on the GPU box (dev tool; one JSON line per config).

    python tools/sweep_env_kernel.py --blocks 0 1 2 3 4 64 --forms rollout api --envs 65536 131072
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd.hip_env import HipEnvHandle  # noqa: E402
from airgym_amd.utils.kernel_bench import kernel_name, measure_env_kernel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--task", default="hovering")
ap.add_argument("--ctl", default="rate")
ap.add_argument("--envs", type=int, nargs="+", default=[65536])
ap.add_argument("--blocks", type=int, nargs="+", default=[0, 1, 2, 3, 4, 64])
ap.add_argument("--forms", nargs="+", default=["rollout", "api"])
ap.add_argument("--replays", type=int, default=20)
ap.add_argument("--nograph", action="store_true", help="eager launches (rocprofv3 counter passes)")
ap.add_argument("--stagger", type=int, nargs="+", default=[0], help="block 0 only: de-phase every second workgroup of a CU by k x 0.5 us")
ap.add_argument("--no-noise", action="store_true", help="observation noise off (diagnostic: what the noise wave costs)")
a = ap.parse_args()
print(torch.cuda.get_device_name(0), file=sys.stderr)
for n in a.envs:
    env = HipEnvHandle(a.task, a.ctl, n, seed=0, reward_terms=True, obs_noise=not a.no_noise)
    for block, stag in [(b, s) for b in a.blocks for s in (a.stagger if b == 0 else [0])]:
        env.set_launch_params(block, 1 + stag)
        for form in a.forms:
            if form == "rollout" and (block == 1 or block >= 64):
                continue        # the rollout form exists for the ws2 family only
            r = measure_env_kernel(env, replays=a.replays, rollout_form=(form == "rollout"), use_graph=not a.nograph)
            r.update(task=a.task, ctl=a.ctl, envs=n, block=block, stagger=stag, obs_noise=not a.no_noise, kernel=kernel_name(a.task, a.ctl, block),
                     frac=r["gbps_algorithmic"] / 8000.0)
            print(json.dumps(r), flush=True)
    env.close()
