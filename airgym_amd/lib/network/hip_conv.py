"""The three stride-2 convolutions of the depth-image feature extractor on the HIP kernels of csrc/conv_kernels.hip
(reference modules: lib/network/cnn.py:11-13 - nn.Conv2d(1, 16, 5, 2, 2), nn.Conv2d(16, 32, 3, 2, 1), nn.Conv2d(32, 64, 3, 2, 1)
on (1, 212, 120) images).

Same function as `F.conv2d(x, w, b, stride=2, padding=k // 2)` and its autograd - exact float32 products and sums, in a
different summation order than the library's - for exactly these shapes on NCHW tensors: no NCHW<->NHWC transposes around
the kernels, weight gradients as fixed-order partial sums (deterministic).  Anything else (other shapes, CPU tensors, double
backward) is not this module's business: `supported()` says no and the caller keeps `nn.Conv2d`."""
import ctypes

import torch

from airgym_amd import _native as N

_CONV1 = (1, 16, 212, 120)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def supported(x, conv):
    """True when `conv` (an nn.Conv2d) applied to x is one of the three layers the kernels are written for."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.bias is not None and conv.groups == 1
            and conv.dilation == (1, 1) and conv.stride == (2, 2) and conv.padding_mode == "zeros"):
        return False
    shape = (conv.in_channels, conv.out_channels, x.shape[2], x.shape[3])
    if x.shape[1] != conv.in_channels:
        return False
    if conv.kernel_size == (5, 5) and conv.padding == (2, 2):
        return shape == _CONV1
    if conv.kernel_size == (3, 3) and conv.padding == (1, 1):
        return bool(N.load().ag_cnn_conv_supported(*shape))
    return False


class _Conv1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        lib = N.load()
        n = x.shape[0]
        x = x.contiguous()
        w = w.contiguous()
        y = torch.empty(n, 16, 106, 60, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.ag_cnn_conv_workspace_floats(1, 16), dtype=torch.float32, device=x.device)
        N.check(lib.ag_cnn_conv1_fwd(x.data_ptr(), None, None, None, w.data_ptr(), b.data_ptr(), y.data_ptr(), None, n, ws.data_ptr(),
                                     _stream(x)), "ag_cnn_conv1_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the first convolution's input is the image: no input gradient kernel")
        (x,) = ctx.saved_tensors
        lib = N.load()
        n = x.shape[0]
        dy = dy.contiguous()
        g = lib.ag_cnn_conv1_wgrad_partials(n)
        partials = torch.empty(g, 16, 32, dtype=torch.float32, device=x.device)
        N.check(lib.ag_cnn_conv1_wgrad(dy.data_ptr(), None, None, None, x.data_ptr(), None, None, None, partials.data_ptr(), n,
                                       _stream(x)),
                "ag_cnn_conv1_wgrad")
        s = partials.sum(0)
        return None, s[:, :25].reshape(16, 1, 5, 5), s[:, 25].clone()


class _ConvS2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        lib = N.load()
        n, cin, hin, win = x.shape
        cout = w.shape[0]
        x = x.contiguous()
        w = w.contiguous()
        y = torch.empty(n, cout, (hin - 1) // 2 + 1, win // 2, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=x.device)
        N.check(lib.ag_cnn_conv_fwd(x.data_ptr(), None, None, w.data_ptr(), b.data_ptr(), y.data_ptr(), None, n, cin, cout, hin,
                                    win, ws.data_ptr(), _stream(x)), "ag_cnn_conv_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        lib = N.load()
        n, cin, hin, win = x.shape
        cout = w.shape[0]
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=x.device)
            N.check(lib.ag_cnn_conv_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, cin, cout, hin, win, ws.data_ptr(),
                                          _stream(x)), "ag_cnn_conv_dgrad")
        g = lib.ag_cnn_conv_wgrad_partials(n, cin, cout, hin, win)
        partials = torch.empty(g, cout * cin * 9 + cout, dtype=torch.float32, device=x.device)
        N.check(lib.ag_cnn_conv_wgrad(dy.data_ptr(), x.data_ptr(), None, None, partials.data_ptr(), 1, n, cin, cout, hin, win,
                                      _stream(x)), "ag_cnn_conv_wgrad")
        s = partials.sum(0)
        return dx, s[:cout * cin * 9].reshape(cout, cin, 3, 3), s[cout * cin * 9:].clone()


def conv2d(x, conv):
    """`conv(x)` on the HIP kernels; the caller has checked `supported(x, conv)`."""
    if conv.kernel_size == (5, 5):
        return _Conv1.apply(x, conv.weight, conv.bias)
    return _ConvS2.apply(x, conv.weight, conv.bias)
