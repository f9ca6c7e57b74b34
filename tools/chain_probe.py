#!/usr/bin/env python3
"""Time ag_mlp_chain_forward against the two launches it replaces (ag_mlp_input_layer + ag_split_gemm_elu_heads), on the GPU box.
    python tools/chain_probe.py [--rows 65536 196608]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd import _native as N  # noqa: E402
from airgym_amd.utils.kernel_bench import _time_us  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[65536, 196608])
ap.add_argument("--D", type=int, default=18)
ap.add_argument("--ablate", action="store_true", help="AIRGYM_EXPERIMENTS=1 build: time the kernel with DMA / MFMA / barriers left out")
a = ap.parse_args()
lib = N.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
D, A1 = a.D, 5
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
W1, b1, W2, b2, Wh, bh = r(256, D) * 0.1, r(256) * 0.1, r(256, 256) * 0.06, r(256) * 0.1, r(A1, 256) * 0.05, r(A1) * 0.1
mean, var = (r(D) * 0.1).double(), (torch.rand(D, device="cuda") + 0.5).double()
image = torch.empty(lib.ag_mlp_chain_image_bytes(D), dtype=torch.uint8, device="cuda")
N.check(lib.ag_mlp_chain_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), Wh.data_ptr(), A1, image.data_ptr(), st), "prep")
planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
N.check(lib.ag_split_gemm_prepare(W2.data_ptr(), planes.data_ptr(), 256, 256, 0, st), "prep2")
for M in a.rows:
    obs = r(M, D)
    heads, xn = torch.empty(M, A1, device="cuda"), torch.empty(M, D, device="cuda")
    h1, h2 = torch.empty(M, 256, device="cuda"), torch.empty(M, 256, device="cuda")

    def chain(store):
        N.check(lib.ag_mlp_chain_forward(obs.data_ptr(), mean.data_ptr(), var.data_ptr(), 1e-5, 5.0, image.data_ptr(), b2.data_ptr(),
                                         bh.data_ptr(), heads.data_ptr(), xn.data_ptr() if store else None,
                                         h1.data_ptr() if store else None, h2.data_ptr() if store else None, M, D, A1, st), "chain")

    def two():
        N.check(lib.ag_mlp_input_layer(obs.data_ptr(), mean.data_ptr(), var.data_ptr(), W1.data_ptr(), b1.data_ptr(), xn.data_ptr(),
                                       h1.data_ptr(), M, D, 256, 1e-5, 5.0, st), "in")
        N.check(lib.ag_split_gemm_elu_heads(h1.data_ptr(), planes.data_ptr(), b2.data_ptr(), Wh.data_ptr(), bh.data_ptr(), h2.data_ptr(),
                                            heads.data_ptr(), M, 256, 256, A1, st), "gemm")
    flops = 2.0 * M * (32 * 256 + 256 * 256 + 256 * 32) * 6
    out = {"rows": M, "D": D}
    for name, fn in (("chain_heads_only", lambda: chain(False)), ("chain_store_all", lambda: chain(True)), ("two_launches", two)):
        us = _time_us(fn, iters=30, warmup=5)
        out[name + "_us"] = us
        if name.startswith("chain"):
            out[name + "_bf16_tflops"] = flops / us / 1e6
    if a.ablate:
        for mask, label in ((1, "no_dma"), (2, "no_mfma"), (4, "no_barrier"), (8, "no_l2_frag_valu"), (16, "no_frag_reads"),
                            (24, "no_l2_valu_no_reads"), (29, "mfma_and_heads_only")):
            lib.ag_debug_chain_skip(mask)
            out["ablate_" + label + "_us"] = _time_us(lambda: chain(False), iters=20, warmup=3)
        lib.ag_debug_chain_skip(0)
    print(json.dumps(out), flush=True)
