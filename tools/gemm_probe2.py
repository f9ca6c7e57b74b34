#!/usr/bin/env python3
"""Probe split-K weight-gradient formulations (dev tool, GPU box)."""
import torch

dev = "cuda"
M = 196608


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3


for K, N in [(256, 256), (18, 256), (256, 5), (256, 16)]:
    x = torch.randn(M, K, device=dev)
    go = torch.randn(M, N, device=dev)
    ref = go.t() @ x
    print(f"wgrad {K}->{N}: plain {timeit(lambda: go.t() @ x):.1f} us")
    for S in (8, 16, 32, 48, 64, 96, 128, 256, 512):
        if M % S:
            continue
        def f():
            return torch.bmm(go.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0)
        err = (f() - ref).abs().max().item() / ref.abs().max().item()
        print(f"   bmm split S={S:4d}: {timeit(f):8.1f} us  relerr {err:.1e}")
    # x^T go orientation
    print(f"   (x.t() @ go).t(): {timeit(lambda: (x.t() @ go).t()):.1f} us")
