"""Time the three 256x256 GEMMs of the PPO minibatch (fwd, dgrad, split-K wgrad) with the default hipBLASLt heuristic;
run with PYTORCH_TUNABLEOP_ENABLED=1 to see what TunableOp finds."""
import os
import sys
import time

import torch

M, C, S = 196608, 256, 64
dev = "cuda"
x = torch.randn(M, C, device=dev)
w = torch.randn(C, C, device=dev) * 0.05
b = torch.randn(C, device=dev)
out = torch.empty(M, C, device=dev)
part = torch.empty(S, C, C, device=dev)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


flop = 2 * M * C * C
for name, fn in [("fwd addmm", lambda: torch.addmm(b, x, w.t(), out=out)),
                 ("dgrad mm", lambda: torch.mm(x, w, out=out)),
                 ("wgrad bmm", lambda: torch.bmm(x.view(S, M // S, C).transpose(1, 2), out.view(S, M // S, C), out=part))]:
    us = timeit(fn)
    print(f"{name:10s} {us:7.1f} us  {flop / us / 1e6:6.1f} TFLOP/s", flush=True)
print("tunableop:", os.getenv("PYTORCH_TUNABLEOP_ENABLED"))
