#!/usr/bin/env python3
"""Sweep launch geometry of the env-step kernel on the GPU box (dev tool; prints one JSON line per config)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd.hip_env import HipEnvHandle  # noqa: E402
from airgym_amd.utils.kernel_bench import measure_env_kernel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--task", default="hovering")
ap.add_argument("--ctl", default="rate")
ap.add_argument("--envs", type=int, nargs="+", default=[65536])
ap.add_argument("--blocks", type=int, nargs="+", default=[0, 64, 256])
ap.add_argument("--terms", type=int, default=1)
ap.add_argument("--nograph", action="store_true")
a = ap.parse_args()
print(torch.cuda.get_device_name(0), file=sys.stderr)
for n in a.envs:
    for terms in ([a.terms] if a.terms in (0, 1) else [0, 1]):
        env = HipEnvHandle(a.task, a.ctl, n, seed=0, reward_terms=bool(terms))
        for block in a.blocks:
            for lds in (1, 0):
                env.set_launch_params(block, lds)
                for graph in ([False] if a.nograph else [True, False]):
                    r = measure_env_kernel(env, use_graph=graph)
                    r.update(task=a.task, ctl=a.ctl, envs=n, block=block, obs_via_lds=lds, reward_terms=terms)
                    print(json.dumps(r))
        env.close()
