#!/usr/bin/env python3
"""Time the forward of the two 3 x 3 layers: f32-input MFMA kernel (ag_cnn_conv_fwd) against the bf16-split kernel
(ag_cnn_conv_fwd_split), with the previous layer's ReLU + BatchNorm applied and statistics on, as the trunk calls them.

    python tools/conv_fwd_probe.py [--images 4750] [--reps 10]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=4750)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
lib = N.load()
dev = torch.device("cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, cin, cout, hin, win in (("conv2", 16, 32, 106, 60), ("conv3", 32, 64, 53, 30)):
    torch.manual_seed(0)
    n = a.images
    x = torch.randn(n, cin, hin, win, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    b = torch.randn(cout, device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
    ho, wo = (hin - 1) // 2 + 1, win // 2
    y = torch.empty(n, cout, ho, wo, device=dev)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=dev)
    out = {"layer": name, "images": n, "gflop": 2.0 * n * cout * ho * wo * cin * 9 / 1e9,
           "gb": (x.numel() + y.numel()) * 4 / 1e9}
    ys = {}
    for key, fn, bands_fn in (("f32_mfma", lib.ag_cnn_conv_fwd, lib.ag_cnn_conv_fwd_bands),
                              ("bf16_split", lib.ag_cnn_conv_fwd_split, lib.ag_cnn_conv_fwd_split_bands)):
        stats = torch.empty(n, bands_fn(cin, cout, hin, win), cout, 2, device=dev)

        def run():
            N.check(fn(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), n, cin, cout,
                       hin, win, ws.data_ptr(), st), key)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        out[key + "_us"] = round(us, 1)
        out[key + "_tbps"] = round(out["gb"] / us * 1e3, 2)
        ys[key] = (y.clone(), stats.sum(1).clone())
    out["max_rel_diff_y"] = ((ys["f32_mfma"][0] - ys["bf16_split"][0]).abs().max() / ys["f32_mfma"][0].abs().max()).item()
    out["max_rel_diff_stats"] = ((ys["f32_mfma"][1] - ys["bf16_split"][1]).abs().max() / ys["f32_mfma"][1].abs().max()).item()
    print(json.dumps(out), flush=True)
    del x, y, ys
    torch.cuda.empty_cache()
