import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.hovering_ref import HoveringRef
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
n = 65536
g = torch.Generator().manual_seed(1)
a = torch.randn(n, 4, generator=g).clamp_(-1, 1)
for th in (1, 4, 8, 16, 32, 64):
    torch.set_num_threads(th)
    env = HoveringRef(n, "rate", seed=0)
    env.step(a)
    t0 = time.time(); k = 0
    while time.time() - t0 < 3.0 and k < 50:
        env.step(a); k += 1
    dt = (time.time() - t0) / k
    print(f"threads {th:3d}: {dt*1e3:8.1f} ms/step  {n/dt/1e6:6.2f} M env-steps/s", flush=True)
