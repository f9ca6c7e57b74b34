#!/usr/bin/env python3
"""Where does the Planning render kernel spend its time? (diagnostic skip masks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airgym_amd.hip_env import HipEnvHandle
n = 16384
env = HipEnvHandle("planning", "rate", n, seed=0)
a = torch.zeros(n, 4, device="cuda"); a[:, 3] = -0.69
for _ in range(6):
    env.step(a)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mask, name in [(0, "full"), (1, "no raycast"), (2, "no noise"), (4, "no 5x5"), (6, "raycast only"), (7, "skeleton")]:
    ts = []
    for _ in range(3):
        env.planning_render_next_step(mask)
        torch.cuda.synchronize(); s.record(); env.step(a); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    print(f"{name:14s} {sorted(ts)[1]:.2f} ms")
