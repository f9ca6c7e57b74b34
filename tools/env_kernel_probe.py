#!/usr/bin/env python3
"""Time the env-step kernel in its three forms on the GPU box (one JSON line per form); also the workload of the PMC passes
(tools/gpu_pmc_env.sh runs it under rocprofv3 with --nograph).

    python tools/env_kernel_probe.py --forms api rollout fused --envs 65536
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from airgym_amd.lib.agent.a2c_continuous import A2CAgent  # noqa: E402
from airgym_amd.utils.kernel_bench import (env_kernel_source_sha, kernel_name, measure_env_kernel,  # noqa: E402
                                           measure_env_multi, measure_fused_rollout_kernel)

ap = argparse.ArgumentParser()
ap.add_argument("--task", default="hovering")
ap.add_argument("--ctl", default="rate")
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--forms", nargs="+", default=["api", "rollout", "fused"], help="api | rollout | fused | multi (ag_step_multi)")
ap.add_argument("--multi-steps", type=int, default=24, help="env steps per ag_step_multi launch")
ap.add_argument("--replays", type=int, default=20)
ap.add_argument("--nograph", action="store_true", help="eager launches (rocprofv3 counter passes)")
a = ap.parse_args()
print(torch.cuda.get_device_name(0), file=sys.stderr)


class Args:
    envs = a.envs; minibatches = 8; graph = 0; task = a.task; ctl = a.ctl


agent = A2CAgent("probe", bench.build_params(Args, 1))
agent.init_tensors()
agent.obs = agent.env_reset()
env = agent._hip_env
for form in a.forms:
    if form == "fused":
        if a.nograph:       # counter passes: plain launches, enough of them for a steady state
            import ctypes
            r0 = measure_fused_rollout_kernel(agent, steps_per_graph=48, replays=max(2, a.replays // 4))
            r = dict(r0)
        else:
            r = measure_fused_rollout_kernel(agent, replays=a.replays)
    elif form == "multi":
        r = measure_env_multi(env, K=a.multi_steps, launches_per_graph=2, replays=a.replays, use_graph=not a.nograph)
        r["kernel"] = kernel_name(a.task, a.ctl, False)
    else:
        r = measure_env_kernel(env, replays=a.replays, rollout_form=(form == "rollout"), use_graph=not a.nograph)
        r["kernel"] = kernel_name(a.task, a.ctl, False)
    r.update(task=a.task, ctl=a.ctl, envs=a.envs, form=form, frac=r["gbps_algorithmic"] / 8000.0, source_sha=env_kernel_source_sha())
    print(json.dumps(r), flush=True)
