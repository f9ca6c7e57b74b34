#!/usr/bin/env python3
"""ag_split_wgrad against the library's split-K f32 weight gradient on the GPU box: interleaved medians (HIP events), both
issue-order variants of the kernel (experiments build: AIRGYM_EXPERIMENTS=1), the partial-sum reduction priced separately.

    AIRGYM_EXPERIMENTS=1 python tools/wgrad_probe.py [--M 196608 65536]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, nargs="+", default=[196608, 65536])
ap.add_argument("--rounds", type=int, default=12)
a = ap.parse_args()
lib = N.load()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def timed(fn, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for M in a.M:
    g = torch.Generator(device="cuda").manual_seed(0)
    dz, x = torch.randn(M, 256, device="cuda", generator=g), torch.randn(M, 256, device="cuda", generator=g)
    S = lib.ag_split_wgrad_slices(M)
    parts = torch.empty(S, 256, 256, device="cuda")
    out = torch.empty(256, 256, device="cuda")
    SK = 64
    lparts = torch.empty(SK, 256, 256, device="cuda")
    scratch = torch.empty(lib.ag_sum_rows_groups() * 65536, device="cuda")
    jobs = (N.AgSumJob * 1)(N.AgSumJob(parts.data_ptr(), out.data_ptr(), S, 65536))
    ljobs = (N.AgSumJob * 1)(N.AgSumJob(lparts.data_ptr(), out.data_ptr(), SK, 65536))

    def split(ordered):
        if N.EXPERIMENTS:
            lib.ag_debug_split_wgrad_ordered(ordered)
        N.check(lib.ag_split_wgrad(dz.data_ptr(), x.data_ptr(), parts.data_ptr(), M, 256, 256, S, st()), "ag_split_wgrad")
    cands = {
        "split_wgrad_ordered": lambda: split(1),
        "library_splitk_bmm": lambda: torch.bmm(dz.view(SK, M // SK, 256).transpose(1, 2), x.view(SK, M // SK, 256), out=lparts),
        "reduce_split_partials": lambda: N.check(lib.ag_sum_rows_multi(jobs, 1, scratch.data_ptr(), scratch.numel(), st()), "sum"),
        "reduce_library_partials": lambda: N.check(lib.ag_sum_rows_multi(ljobs, 1, scratch.data_ptr(), scratch.numel(), st()), "sum"),
    }
    if N.EXPERIMENTS:
        cands["split_wgrad_compiler_order"] = lambda: split(0)
    for fn in cands.values():
        fn()
    torch.cuda.synchronize()
    res = {k: [] for k in cands}
    for _ in range(a.rounds):
        for k, fn in cands.items():
            res[k].append(timed(fn))
    med = {k: statistics.median(v) for k, v in res.items()}
    flops = 2.0 * M * 256 * 256
    print(json.dumps({"M": M, "slices": S, "us_median": med,
                      "split_wgrad_bf16_tflops": 6 * flops / med["split_wgrad_ordered"] / 1e6,
                      "split_plus_reduce_us": med["split_wgrad_ordered"] + med["reduce_split_partials"],
                      "library_plus_reduce_us": med["library_splitk_bmm"] + med["reduce_library_partials"]}), flush=True)
