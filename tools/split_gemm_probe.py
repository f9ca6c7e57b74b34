#!/usr/bin/env python3
"""Time ag_split_gemm (bf16 x 6 float32-accurate GEMM) against the library's float32 GEMM at the update's shapes (GPU box)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd import _native as N  # noqa: E402
from airgym_amd.utils.gemm_tuning import enable_tuned_gemms  # noqa: E402

lib = N.load()
enable_tuned_gemms()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def time_us(fn, iters=30, warmup=5):
    s = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warmup):
        fn()
    a.record(s)
    for _ in range(iters):
        fn()
    b.record(s)
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


import argparse
_ap = argparse.ArgumentParser()
_ap.add_argument("--M", type=int, nargs="+", default=[65536, 196608])
_ap.add_argument("--variants", type=int, nargs="+", default=[0, 2, 3, 4, 6, 7])
_ap.add_argument("--skip-lib", action="store_true")
_a = _ap.parse_args()
for M in _a.M:
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, 256, device="cuda", generator=g)
    W = torch.randn(256, 256, device="cuda", generator=g) / 16
    Wt = W.t().contiguous()
    C = torch.empty(M, 256, device="cuda")
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, 0, st), "prep")
    us_prep = time_us(lambda: lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, 0, st))
    us_var = {}
    for var in _a.variants:
        lib.ag_debug_split_gemm_variant(var)
        us_var[var] = time_us(lambda: lib.ag_split_gemm(A.data_ptr(), planes.data_ptr(), None, C.data_ptr(), M, 256, 256, st))
    lib.ag_debug_split_gemm_variant(-1)
    us_split = time_us(lambda: lib.ag_split_gemm(A.data_ptr(), planes.data_ptr(), None, C.data_ptr(), M, 256, 256, st))
    us_nn = us_nt = float("nan")
    if not _a.skip_lib:
        us_nn = time_us(lambda: torch.mm(A, Wt, out=C))
        us_nt = time_us(lambda: torch.mm(A, W.t(), out=C))
    fl = 2.0 * M * 256 * 256
    print(json.dumps({"M": M, "split_us": us_split, "split_f32_equiv_tflops": fl / us_split / 1e6,
                      "split_bf16_tflops": 6 * fl / us_split / 1e6, "lib_nn_us": us_nn, "lib_nt_us": us_nt,
                      "lib_tflops": fl / min(us_nn, us_nt) / 1e6, "prepare_us": us_prep, "variant_us": us_var}), flush=True)
