"""Does a streaming (HBM-bound) kernel overlap with a hipBLASLt fp32 GEMM issued on another stream?"""
import time

import torch
import torch.nn.functional as F

M, C = 196608, 256
x = torch.randn(M, C, device="cuda") * 0.5
w = torch.randn(C, C, device="cuda") * 0.05
out = torch.empty(M, C, device="cuda")
y = torch.randn(M, C, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def gemm():
    torch.mm(x, w, out=out)


def stream_op():
    F.elu_(y)


def serial():
    gemm(); stream_op()


def overlapped():
    with torch.cuda.stream(s1):
        gemm()
    with torch.cuda.stream(s2):
        stream_op()


def overlapped_half():
    # two half-size GEMMs on s1, two streaming ops on s2
    with torch.cuda.stream(s1):
        torch.mm(x[:M // 2], w, out=out[:M // 2]); torch.mm(x[M // 2:], w, out=out[M // 2:])
    with torch.cuda.stream(s2):
        F.elu_(y[:M // 2]); F.elu_(y[M // 2:])


print(f"gemm {t(gemm):.1f} us | elu {t(stream_op):.1f} us | serial {t(serial):.1f} us | two streams {t(overlapped):.1f} us | "
      f"two streams, halves {t(overlapped_half):.1f} us")
