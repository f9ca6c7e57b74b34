#!/usr/bin/env python3
"""Time conv3's input gradient with and without the second layer's ReLU + BatchNorm backward in its epilogue (one MI355X; side
measurement for DESIGN.md §4.4): ag_cnn_conv_dgrad + ag_relu_bn_bwd_dx_weighted (plane + border sums in passing) against
ag_cnn_conv_dgrad_bn with and without its sums.

    python tools/dgrad_epilogue_probe.py [--images 4750] [--reps 10]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4750)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    from airgym_amd import _native as N
    lib = N.load()
    dev = torch.device("cuda:0")
    n = a.images
    torch.manual_seed(0)
    dz = torch.randn(n, 64, 27, 15, device=dev)
    w = torch.randn(64, 32, 3, 3, device=dev) * 0.1
    x = torch.randn(n, 32, 53, 30, device=dev)
    tab = torch.randn(32, 4, device=dev)
    sums2 = torch.randn(32, 2, device=dev)
    wts = torch.ones(n, device=dev)
    dx = torch.empty_like(x)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(32, 64), device=dev)
    rows = lib.ag_cnn_conv_dgrad_bn_rows(n, 32, 64, 53, 30)
    sums = torch.empty(rows, 32, 6, device=dev)
    ps, bs = torch.empty(n, 32, device=dev), torch.empty(n, 32, 5, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def plain():
        N.check(lib.ag_cnn_conv_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), n, 32, 64, 53, 30, ws.data_ptr(), stream), "dgrad")

    def separate():
        plain()
        N.check(lib.ag_relu_bn_bwd_dx_weighted(dx.data_ptr(), x.data_ptr(), tab.data_ptr(), sums2.data_ptr(), wts.data_ptr(), dx.data_ptr(),
                                               ps.data_ptr(), bs.data_ptr(), 30, n, 32, 53 * 30, stream), "dx")
        return ps.sum(0), bs.sum(0)

    def fused(with_sums):
        N.check(lib.ag_cnn_conv_dgrad_bn(dz.data_ptr(), w.data_ptr(), x.data_ptr(), tab.data_ptr(), wts.data_ptr(), dx.data_ptr(),
                                         sums.data_ptr() if with_sums else None, n, 32, 64, 53, 30, ws.data_ptr(), stream), "dgrad_bn")
        return sums.sum(0) if with_sums else None

    for name, fn in (("input gradient alone", plain), ("input gradient, then the ReLU + BatchNorm backward pass", separate),
                     ("epilogue, no sums", lambda: fused(False)), ("epilogue with sums", lambda: fused(True))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"what": name, "images": n, "us": round(e0.elapsed_time(e1) * 1e3 / a.reps, 1)}), flush=True)


if __name__ == "__main__":
    main()
