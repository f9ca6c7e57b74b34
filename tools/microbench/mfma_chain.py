#!/usr/bin/env python3
"""Issue rate of v_mfma_f32_32x32x16_bf16 in a dependent chain (same accumulator) vs round-robin over NACC accumulators, one or two
waves per SIMD.  Builds tools/microbench/libmfma_chain.so on first use (hipcc); prints ns per MFMA on one SIMD's pipe and the chip-wide bf16 rate.

    python tools/microbench/mfma_chain.py
"""
import ctypes
import json
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmfma_chain.so")
if not os.path.exists(SO):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "mfma_chain.hip")])
lib = ctypes.CDLL(SO)
lib.mfma_chain_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(256 * 512, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = 2000
res = []
for threads in (256, 512):
    for nacc in (1, 2, 3, 4, 6):
        def run():
            assert lib.mfma_chain_launch(nacc, threads, 256, ITERS, out.data_ptr(), st) == 0
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        per_wave = ITERS * 48                     # MFMAs per wave
        waves_per_simd = threads // 256
        ns = ms * 1e6 / (per_wave * waves_per_simd)     # per MFMA on one SIMD's pipe
        res.append({"waves_per_simd": waves_per_simd, "accumulators": nacc, "ns_per_mfma_per_simd": round(ns, 2),
                    "bf16_tflops_chip": round(2 * 32 * 32 * 16 * per_wave * (threads // 64) * 256 / (ms * 1e-3) / 1e12, 1)})
print(json.dumps(res))
