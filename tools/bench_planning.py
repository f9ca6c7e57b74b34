#!/usr/bin/env python3
"""Planning task timings on the GPU box: env-only env-steps/s (incl. the camera render every 4th step) and the
three kernels' durations; optional PPO epoch with the CNN policy.  (Dev tool; BASELINE config 4 is 16 384 envs/GPU.)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airgym_amd.hip_env import HipEnvHandle

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=16384)
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--ppo", action="store_true")
a = ap.parse_args()
n = a.envs
env = HipEnvHandle("planning", "rate", n, seed=0)
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randn(8, n, 4, generator=g, device="cuda").clamp_(-1, 1) * 0.3
acts[..., 3] = -0.69
for t in range(8):
    env.step(acts[t % 8])
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for t in range(a.steps):
    env.step(acts[t % 8])
e.record(); e.synchronize()
ms = s.elapsed_time(e)
out = {"task": "planning", "envs": n, "steps": a.steps, "ms_per_env_step": ms / a.steps,
       "env_steps_per_s": n * a.steps / (ms * 1e-3), "note": "env only, render every 4th step (212x120 depth + post-processing)"}
# render-only timing
env.planning_render_next_step(); env.step(acts[0]); torch.cuda.synchronize()
ts = []
for _ in range(5):
    env.planning_render_next_step()
    s.record(); env.step(acts[1]); e.record(); e.synchronize()
    ts.append(s.elapsed_time(e))
out["ms_step_with_render"] = sorted(ts)[2]
ts = []
for _ in range(5):
    while (env.lib.ag_get_tick(env.h) + 0) and False:
        pass
    s.record(); env.step(acts[2]); e.record(); e.synchronize()
    ts.append(s.elapsed_time(e))
out["ms_step_samples"] = [round(x, 3) for x in ts]
out["image_GB"] = n * 212 * 120 * 4 / 1e9
print(json.dumps(out))
