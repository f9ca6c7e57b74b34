#!/usr/bin/env python3
"""Time the convolution kernels of csrc/conv_kernels.hip against torch's (MIOpen's) for the three layers of the depth-image
feature extractor at the image count of a de-duplicated Planning minibatch (one MI355X; side measurement for DESIGN.md §4.4).

    python tools/conv_probe.py [--images 4750] [--reps 5] > profiles/rNN_conv_probe.jsonl

One JSON line per (layer, pass): microseconds of the HIP entry point, of the library call, and the largest difference between the
two float32 results relative to the result's scale."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

LAYERS = [("conv1", 1, 16, 5, 212, 120), ("conv2", 16, 32, 3, 106, 60), ("conv3", 32, 64, 3, 53, 30)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, out


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4750)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--layers", default="conv1,conv2,conv3")
    args = ap.parse_args()
    from airgym_amd.lib.network import hip_conv
    dev = torch.device("cuda:0")
    n = args.images
    for name, cin, cout, k, hin, win in LAYERS:
        if name not in args.layers.split(","):
            continue
        torch.manual_seed(0)
        conv = nn.Conv2d(cin, cout, k, stride=2, padding=k // 2).to(dev)
        x = torch.randn(n, cin, hin, win, device=dev)
        gb = x.numel() * 4 / 1e9
        # forward
        with torch.no_grad():
            t_hip, y = timed(lambda: hip_conv.conv2d(x, conv), args.reps)
            t_lib, y_lib = timed(lambda: conv(x), args.reps)
        print(json.dumps({"layer": name, "pass": "forward", "images": n, "hip_us": round(t_hip, 1), "library_us": round(t_lib, 1),
                          "max_rel_diff": rel(y, y_lib), "in_gb": round(gb, 3), "out_gb": round(y.numel() * 4 / 1e9, 3)}), flush=True)
        dy = torch.randn_like(y)
        del y_lib
        # backward pieces through aten (the library) and through the autograd node (HIP)
        mask_w = [False, True, True]
        t_lib_w, gw = timed(lambda: torch.ops.aten.convolution_backward(dy, x, conv.weight, [cout], [2, 2], [k // 2, k // 2], [1, 1],
                                                                        False, [0, 0], 1, mask_w), args.reps)
        xr = x.detach().requires_grad_(False)

        # time the weight-gradient entry alone: call the Function's backward through autograd on a prepared graph
        yy = hip_conv.conv2d(xr, conv)

        def hip_bw():
            conv.weight.grad = None
            conv.bias.grad = None
            yy.backward(dy, retain_graph=True)
            return conv.weight.grad, conv.bias.grad
        t_hip_w, (hw, hb) = timed(hip_bw, args.reps)
        print(json.dumps({"layer": name, "pass": "weight gradient (+ bias)", "images": n, "hip_us": round(t_hip_w, 1),
                          "library_us": round(t_lib_w, 1), "max_rel_diff": rel(hw, gw[1]), "max_rel_diff_bias": rel(hb, gw[2])}),
              flush=True)
        del yy
        if cin > 1:
            t_lib_d, gd = timed(lambda: torch.ops.aten.convolution_backward(dy, x, conv.weight, [cout], [2, 2], [k // 2, k // 2],
                                                                            [1, 1], False, [0, 0], 1, [True, False, False]),
                                args.reps)
            xg = x.detach().requires_grad_(True)
            yy = hip_conv.conv2d(xg, conv)

            def hip_both():
                xg.grad = None
                conv.weight.grad = None
                conv.bias.grad = None
                yy.backward(dy, retain_graph=True)
                return xg.grad
            t_hip_both, hd = timed(hip_both, args.reps)
            print(json.dumps({"layer": name, "pass": "input gradient", "images": n, "hip_us": round(t_hip_both - t_hip_w, 1),
                              "library_us": round(t_lib_d, 1), "max_rel_diff": rel(hd, gd[0]),
                              "note": "hip_us = (input + weight gradient) - weight gradient"}), flush=True)
            del yy, xg, gd, hd
        del x, dy, y, gw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
