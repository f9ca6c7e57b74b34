#!/usr/bin/env python3
"""Time ag_split_gemm (bf16 x 6 float32-accurate GEMM) against the library's float32 GEMM at the update's shapes (GPU box)."""
import ctypes
import json
import os
import sys
os.environ.setdefault("AIRGYM_EXPERIMENTS", "1")      # ag_debug_split_gemm_variant lives in the experiments build

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
import statistics


def round_robin_us(cands, rounds=12, iters=6):
    """Median over `rounds` of each candidate's mean launch time, candidates interleaved: the clock of an MFMA-bound kernel
    drifts with temperature, so back-to-back blocks per candidate compare different clocks."""
    s = torch.cuda.current_stream()
    for fn in cands.values():
        fn()
    out = {k: [] for k in cands}
    for _ in range(rounds):
        for k, fn in cands.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(iters):
                fn()
            b.record(s)
            b.synchronize()
            out[k].append(a.elapsed_time(b) * 1e3 / iters)
    return {k: round(statistics.median(v), 2) for k, v in out.items()}


_ap = argparse.ArgumentParser()
_ap.add_argument("--M", type=int, nargs="+", default=[65536, 196608])
_ap.add_argument("--variants", type=int, nargs="+", default=[2, 3, 6, 7, 14])
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
    b = torch.randn(256, device="cuda", generator=g); Wh = torch.randn(5, 256, device="cuda", generator=g) / 16
    bh = torch.zeros(5, device="cuda"); H = torch.empty(M, 5, device="cuda")

    def plain(var):
        def f():
            lib.ag_debug_split_gemm_variant(var)
            lib.ag_split_gemm(A.data_ptr(), planes.data_ptr(), None, C.data_ptr(), M, 256, 256, st)
        return f

    def fused(var):
        def f():
            lib.ag_debug_split_gemm_variant(var)
            lib.ag_split_gemm_elu_heads(A.data_ptr(), planes.data_ptr(), b.data_ptr(), Wh.data_ptr(), bh.data_ptr(),
                                        C.data_ptr(), H.data_ptr(), M, 256, 256, 5, st)
        return f

    cands = {f"v{v}": plain(v) for v in _a.variants}
    cands["auto"] = plain(-1)
    cands["fused_auto"] = fused(-1)
    for v in (2, 3, 6, 7):
        cands[f"fused_v{v}"] = fused(v)
    h1 = torch.nn.functional.elu(torch.randn(M, 256, device="cuda", generator=g))
    xin = torch.randn(M, 18, device="cuda", generator=g)
    tiles = (M + 127) // 128
    dwp = torch.empty(tiles, 256, 18, device="cuda"); dbp = torch.empty(tiles, 256, device="cuda")

    def input_wgrad():
        lib.ag_debug_split_gemm_variant(-1)
        lib.ag_split_gemm_input_wgrad(A.data_ptr(), planes.data_ptr(), h1.data_ptr(), xin.data_ptr(), dwp.data_ptr(), dbp.data_ptr(),
                                      M, 256, 256, 18, st)
    cands["input_wgrad"] = input_wgrad
    cands["elu_heads"] = lambda: lib.ag_elu_heads(C.data_ptr(), Wh.data_ptr(), bh.data_ptr(), H.data_ptr(), M, 256, 5, 0,
                                                   b.data_ptr(), st)
    cands["prepare"] = lambda: lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, 0, st)
    if not _a.skip_lib:
        cands["lib_nn"] = lambda: torch.mm(A, Wt, out=C)
        cands["lib_nt"] = lambda: torch.mm(A, W.t(), out=C)
    us = round_robin_us(cands)
    lib.ag_debug_split_gemm_variant(-1)
    fl = 2.0 * M * 256 * 256
    rec = {"M": M, "us": us, "split_f32_equiv_tflops": round(fl / us["auto"] / 1e6, 1),
           "split_bf16_tflops": round(6 * fl / us["auto"] / 1e6, 1)}
    if not _a.skip_lib:
        rec["lib_tflops"] = round(fl / min(us["lib_nn"], us["lib_nt"]) / 1e6, 1)
    print(json.dumps(rec), flush=True)
