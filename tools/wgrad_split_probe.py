"""Best library time of the split-K weight-gradient bmm for different slice counts S (run with PYTORCH_TUNABLEOP_ENABLED=1)."""
import time

import torch

M, C = 196608, 256
g = torch.Generator(device="cuda").manual_seed(0)
dz = torch.randn(M, C, device="cuda", generator=g) * 0.01
h = torch.randn(M, C, device="cuda", generator=g)
for S in (16, 32, 64, 128, 256):
    part = torch.empty(S, C, C, device="cuda")
    out = torch.empty(C, C, device="cuda")

    def f():
        torch.bmm(dz.view(S, M // S, C).transpose(1, 2), h.view(S, M // S, C), out=part)
        torch.sum(part, 0, out=out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    # interleave with a streaming op so the GEMM is not timed at sustained-MFMA clocks
    t0 = time.perf_counter()
    for _ in range(10):
        f()
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    print(f"S={S:4d}: bmm + sum {us:7.1f} us (incl. one sync per call)")
