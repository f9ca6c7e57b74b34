#!/usr/bin/env python3
"""Do MIOpen's fp32 convolutions of the Planning CNN (lib/network/cnn.py:3-33) run without the NCHW<->NHWC transposes when they
are handed channels-last tensors?  Times forward + backward of each of the three layers in both memory formats (GPU box)."""
import json
import os
import sys

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 4800
torch.backends.cudnn.benchmark = False
layers = [("conv1", 1, 16, 5, 2, (212, 120)), ("conv2", 16, 32, 3, 1, (106, 60)), ("conv3", 32, 64, 3, 1, (53, 30))]


def timed(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


for name, cin, cout, k, pad, (h, w) in layers:
    res = {"layer": name, "images": U}
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        conv = nn.Conv2d(cin, cout, k, stride=2, padding=pad).cuda().to(memory_format=fmt)
        x = torch.randn(U, cin, h, w, device="cuda").to(memory_format=fmt).requires_grad_(cin > 1)
        y = conv(x)
        gy = torch.randn_like(y)

        def fwd():
            return conv(x)

        def fwdbwd():
            conv.zero_grad(set_to_none=True)
            if x.grad is not None:
                x.grad = None
            conv(x).backward(gy)
        try:
            res[fmt_name] = {"fwd_ms": round(timed(fwd), 3), "fwd_bwd_ms": round(timed(fwdbwd), 3),
                             "out_is_channels_last": bool(y.is_contiguous(memory_format=torch.channels_last))}
        except Exception as ex:      # noqa: BLE001
            res[fmt_name] = {"error": str(ex)[:200]}
    print(json.dumps(res), flush=True)
