"""Correctness + timing of ag_mlp_hidden_heads (airgym_amd/csrc/experimental/mlp_gemm.hip) against addmm + ag_elu_heads.
The kernel is not in the default build: add it to csrc/build.py units() and bind it in _native/__init__.py to re-run."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from airgym_amd import _native as N

lib = N.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(0)


def run(M, K=256, A1=5, check=True):
    X = F.elu(torch.randn(M, K, device="cuda", generator=g))
    W = torch.randn(256, K, device="cuda", generator=g) * 0.06
    b = torch.randn(256, device="cuda", generator=g) * 0.1
    Wh = torch.randn(A1, 256, device="cuda", generator=g) * 0.1
    bh = torch.randn(A1, device="cuda", generator=g)
    Z = torch.empty(M, 256, device="cuda")
    heads = torch.empty(M, A1, device="cuda")
    N.check(lib.ag_mlp_hidden_heads(X.data_ptr(), W.data_ptr(), b.data_ptr(), Wh.data_ptr(), bh.data_ptr(), Z.data_ptr(),
                                    heads.data_ptr(), M, K, 256, A1, st), "ag_mlp_hidden_heads")
    torch.cuda.synchronize()
    if check:
        Zr = torch.addmm(b, X, W.t())
        hr = F.elu(Zr) @ Wh.t() + bh
        print(f"M={M} K={K} A1={A1}: max|Z-Zref| {(Z - Zr).abs().max().item():.2e}  max|heads-ref| {(heads - hr).abs().max().item():.2e}")
    return X, W, b, Wh, bh, Z, heads


for M, K, A1 in [(1, 256, 5), (127, 256, 5), (129, 64, 6), (5000, 256, 5), (65536, 256, 5)]:
    run(M, K, A1)

for M in (65536, 196608):
    X, W, b, Wh, bh, Z, heads = run(M, check=False)

    def mine():
        lib.ag_mlp_hidden_heads(X.data_ptr(), W.data_ptr(), b.data_ptr(), Wh.data_ptr(), bh.data_ptr(), Z.data_ptr(),
                                heads.data_ptr(), M, 256, 256, 5, st)

    def ref():
        torch.addmm(b, X, W.t(), out=Z)
        lib.ag_elu_heads(Z.data_ptr(), Wh.data_ptr(), bh.data_ptr(), heads.data_ptr(), M, 256, 5, 0, st)

    for name, fn in (("fused MFMA kernel", mine), ("addmm + ag_elu_heads", ref)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        print(f"M={M}: {name:22s} {us:7.1f} us  ({2 * M * 256 * 256 / us / 1e6:.1f} TFLOP/s on the GEMM flops)")
