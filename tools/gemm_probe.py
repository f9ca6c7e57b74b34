#!/usr/bin/env python3
"""Probe hipBLASLt fp32 on the PPO MLP's GEMM shapes (dev tool, GPU box)."""
import torch
import torch.nn.functional as F

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


print("shape(K->N)  fwd_us  dgrad_us  wgrad_us")
for K, N in [(18, 256), (32, 256), (64, 256), (256, 256), (256, 1), (256, 4), (256, 5), (256, 8), (256, 16), (256, 32), (256, 64)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    go = torch.randn(M, N, device=dev)
    f = timeit(lambda: F.linear(x, w, b))
    dg = timeit(lambda: go @ w)
    wg = timeit(lambda: go.t() @ x)
    print(f"{K:4d}->{N:4d}  {f:8.1f} {dg:8.1f} {wg:8.1f}")
x = torch.randn(M, 256, device=dev)
print("elu fwd", timeit(lambda: F.elu(x)), "copy", timeit(lambda: x.clone()), "sum0", timeit(lambda: x.sum(0)))
# M = 65536 inference shapes
M2 = 65536
for K, N in [(18, 256), (256, 256), (256, 4), (256, 1), (256, 5)]:
    x = torch.randn(M2, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    print(f"infer {K}->{N}: {timeit(lambda: F.linear(x, w, b)):.1f} us")
