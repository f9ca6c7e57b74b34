#!/usr/bin/env python3
"""Launch + memory-latency floor of one env-step launch: same bytes as the step kernel, no arithmetic."""
import ctypes, os, sys
os.environ.setdefault("AIRGYM_EXPERIMENTS", "1")      # ag_debug_* live in the experiments build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airgym_amd.hip_env import HipEnvHandle
from airgym_amd import _native as N
lib = N.load()
import itertools
for n, mode in itertools.product((65536, 262144), (0, 1, 2, 3)):
    env = HipEnvHandle("hovering", "rate", n, seed=0, reward_terms=False)
    def launch(sp):
        if mode == 0:
            lib.ag_debug_touch(env.h, a.data_ptr(), sp)
        else:
            lib.ag_debug_touch_variant(env.h, a.data_ptr(), mode, sp)
    a = torch.zeros(n, 4, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sp = ctypes.c_void_p(st.cuda_stream)
        for _ in range(5):
            launch(sp)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(48):
                launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3):
            g.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(st)
        for _ in range(20):
            g.replay()
        e.record(st); e.synchronize()
    print(f"touch kernel mode {mode} (0 plain, 1 nt stores, 2 nt loads+stores, 3 empty) {n} envs: {s.elapsed_time(e) * 1e3 / 960:.2f} us per launch")
    env.close()
