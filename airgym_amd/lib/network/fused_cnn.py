"""The convolutional trunk of the depth-image feature extractor as ONE autograd node on the HIP kernels of csrc/conv_kernels.hip and
csrc/cnn_kernels.hip (reference: lib/network/cnn.py:11-14 - three times Conv2d -> ReLU -> BatchNorm2d, then AdaptiveAvgPool2d((1, 1))).

Same function of the parameters as the module sequence (float32; summation orders differ), with three tensors that the module
sequence writes and reads never formed:
  * the ReLU + BatchNorm outputs of layers 1 and 2 - the next convolution applies `relu(x) * scale[c] + shift[c]` while it stages its
    input (forward and weight gradient both recompute it from the convolution output x);
  * the ReLU + BatchNorm output of layer 3 and its gradient - the global average pool only needs per-plane sums of relu(x3), from
    which the batch statistics AND the pooled features follow; in the backward every pixel of a plane receives the same upstream
    gradient (`ag_relu_bn_bwd_dx_plane`);
  * the normalised image and the gradient of the first convolution's output (both folded into the first layer's kernels).
The batch statistics of every ReLU + BatchNorm come out of the producing convolution's epilogue (no statistics pass).
Per-image multiplicities (`weights`, frame de-duplication) enter exactly as in fused_relu_bn.py."""
import ctypes

import torch
import torch.nn as nn

from airgym_amd import _native as N

_SHAPES = ((1, 16, 5, 212, 120), (16, 32, 3, 106, 60), (32, 64, 3, 53, 30))     # (cin, cout, k, hin, win) of the three layers
_HW = (106 * 60, 53 * 30, 27 * 15)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def usable(x, features):
    """True when `features` is the reference's layer sequence at the reference's image size and x lives on the GPU."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(x.shape[1:]) == (1, 212, 120)):
        return False
    layers = list(features)
    if len(layers) != 10 or not isinstance(layers[9], nn.AdaptiveAvgPool2d) or layers[9].output_size not in ((1, 1), 1):
        return False
    for i, (cin, cout, k, _, _) in enumerate(_SHAPES):
        conv, relu, bn = layers[3 * i:3 * i + 3]
        if not (isinstance(conv, nn.Conv2d) and isinstance(relu, nn.ReLU) and isinstance(bn, nn.BatchNorm2d)):
            return False
        if not (conv.in_channels == cin and conv.out_channels == cout and conv.kernel_size == (k, k) and conv.stride == (2, 2)
                and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None
                and conv.padding_mode == "zeros"):
            return False
        if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
            return False
    return True


# ---------------------------------------------------------------- kernel wrappers
def _norm_ptrs(norm):
    return (norm[0].data_ptr(), norm[1].data_ptr()) if norm is not None else (None, None)


def _conv1_fwd(lib, img, norm, w, b, want_stats):
    """(y, stats): stats [n, 16, 2] = per image (sum relu(y), sum relu(y)^2) per channel, or None."""
    n = img.shape[0]
    y = torch.empty(n, 16, 106, 60, dtype=torch.float32, device=img.device)
    stats = torch.empty(n, 16, 2, dtype=torch.float32, device=img.device) if want_stats else None
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(1, 16), dtype=torch.float32, device=img.device)
    N.check(lib.ag_cnn_conv1_fwd(img.data_ptr(), *_norm_ptrs(norm), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                 stats.data_ptr() if want_stats else None, n, ws.data_ptr(), _stream(img)), "ag_cnn_conv1_fwd")
    return y, stats


def _conv_fwd(lib, x, scale, shift, w, b, want_stats):
    """(y, stats): stats [n, cout, 2] = per image (sum relu(y), sum relu(y)^2) per output channel (the kernel's per-band sums
    added up), or None."""
    n, cin, hin, win = x.shape
    cout = w.shape[0]
    y = torch.empty(n, cout, (hin - 1) // 2 + 1, win // 2, dtype=torch.float32, device=x.device)
    bands = lib.ag_cnn_conv_fwd_bands(cin, cout, hin, win)
    stats = torch.empty(n, bands, cout, 2, dtype=torch.float32, device=x.device) if want_stats else None
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=x.device)
    N.check(lib.ag_cnn_conv_fwd(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                stats.data_ptr() if want_stats else None, n, cin, cout, hin, win, ws.data_ptr(), _stream(x)),
            "ag_cnn_conv_fwd")
    return y, (stats.sum(1) if want_stats else None)


def _conv_dgrad(lib, dz, w, like):
    n, cin, hin, win = like.shape
    cout = w.shape[0]
    dx = torch.empty_like(like)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=dz.device)
    N.check(lib.ag_cnn_conv_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), n, cin, cout, hin, win, ws.data_ptr(), _stream(dz)),
            "ag_cnn_conv_dgrad")
    return dx


def _conv_wgrad(lib, dz, x, scale, shift, cout):
    n, cin, hin, win = x.shape
    g = lib.ag_cnn_conv_wgrad_partials(n, cin, cout, hin, win)
    partials = torch.empty(g, cout * cin * 9 + cout, dtype=torch.float32, device=x.device)
    N.check(lib.ag_cnn_conv_wgrad(dz.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(), partials.data_ptr(), n, cin, cout,
                                  hin, win, _stream(x)), "ag_cnn_conv_wgrad")
    s = partials.sum(0)
    return s[:cout * cin * 9].reshape(cout, cin, 3, 3), s[cout * cin * 9:]


def _conv1_wgrad(lib, dy, x1, tab, weights, img, norm):
    """Weight / bias gradient of the first convolution from the gradient dy of its ReLU + BatchNorm output (the BatchNorm backward
    is folded into the kernel's staging: tab [16, 4], see ag_cnn_conv1_wgrad)."""
    n = img.shape[0]
    g = lib.ag_cnn_conv1_wgrad_partials(n)
    partials = torch.empty(g, 16, 32, dtype=torch.float32, device=img.device)
    N.check(lib.ag_cnn_conv1_wgrad(dy.data_ptr(), x1.data_ptr(), tab.data_ptr(), _wptr(weights), img.data_ptr(), *_norm_ptrs(norm),
                                   partials.data_ptr(), n, _stream(img)), "ag_cnn_conv1_wgrad")
    s = partials.sum(0)
    return s[:, :25].reshape(16, 1, 5, 5), s[:, 25]


def _blocks(lib, n, c):
    ppb = lib.ag_relu_bn_planes_per_block()
    return (n * c + ppb - 1) // ppb


def _wptr(weights):
    return weights.data_ptr() if weights is not None else None


def _channel_sums(stats, weights):
    """float64 [C, 2]: weighted sum over images of the per-image sums [n, C, 2] a forward kernel produced."""
    d = stats.double()
    return (d if weights is None else d * weights.double().view(-1, 1, 1)).sum(0)


def _coefficients(sums, m, bn, training):
    """(mean, invstd, scale, shift) float32 [C] of ReLU + BatchNorm: batch statistics from `sums` (training; running statistics
    updated as nn.BatchNorm2d does) or the running statistics (eval)."""
    if training:
        mean = sums[:, 0] / m
        var = torch.clamp(sums[:, 1] / m - mean * mean, min=0.0)
        with torch.no_grad():
            mom = float(bn.momentum)
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            bn.running_mean.mul_(1.0 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_((var * (m / max(m - 1.0, 1.0))).to(bn.running_var.dtype), alpha=mom)
    else:
        mean, var = bn.running_mean.double(), bn.running_var.double()
    invstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight.detach().double() * invstd
    shift = bn.bias.detach().double() - mean * scale
    return mean.float(), invstd.float(), scale.float().contiguous(), shift.float().contiguous()


def _bn_reduce(lib, dy, x, mean, invstd):
    """float32 [C, 2] = (dbeta, dgamma) of ReLU + BatchNorm from the gradient dy of its output and the layer's own output x."""
    n, c, h, w = x.shape
    partials = torch.empty(_blocks(lib, n, c), c, 2, dtype=torch.float32, device=x.device)
    N.check(lib.ag_relu_bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), partials.data_ptr(), n, c,
                                      h * w, _stream(x)), "ag_relu_bn_bwd_reduce")
    return partials.sum(0, dtype=torch.float64).float().contiguous()


def _bn_backward(lib, dy, x, mean, invstd, gamma, m, weights):
    """ReLU + BatchNorm backward of a layer whose output gradient dy is a tensor: returns (dx written over dy, dgamma, dbeta)."""
    n, c, h, w = x.shape
    sums = _bn_reduce(lib, dy, x, mean, invstd)
    coef = torch.stack((mean, invstd, gamma.detach() * invstd, torch.full_like(mean, 1.0 / m)), dim=1).contiguous()
    N.check(lib.ag_relu_bn_bwd_dx_weighted(dy.data_ptr(), x.data_ptr(), coef.data_ptr(), sums.data_ptr(), _wptr(weights),
                                           dy.data_ptr(), n, c, h * w, _stream(x)), "ag_relu_bn_bwd_dx")
    return dy, sums[:, 1], sums[:, 0]


class _Trunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, weights, bns, training, norm, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3):
        lib = N.load()
        img = img.contiguous()
        n = img.shape[0]
        if weights is not None:
            weights = weights.to(device=img.device, dtype=torch.float32).contiguous()
            assert weights.shape == (n,)
            wsum = float(weights.double().sum().item())
        else:
            wsum = float(n)
        w1, w2, w3 = w1.contiguous(), w2.contiguous(), w3.contiguous()
        if norm is not None:
            norm = tuple(t.to(device=img.device, dtype=torch.float32).contiguous().view(-1) for t in norm)
            assert norm[0].numel() == 212 * 120 and norm[1].numel() == 212 * 120
        x1, st1 = _conv1_fwd(lib, img, norm, w1, b1, training)
        mean1, invstd1, sc1, sh1 = _coefficients(_channel_sums(st1, weights) if training else None, wsum * _HW[0], bns[0], training)
        x2, st2 = _conv_fwd(lib, x1, sc1, sh1, w2, b2, training)
        mean2, invstd2, sc2, sh2 = _coefficients(_channel_sums(st2, weights) if training else None, wsum * _HW[1], bns[1], training)
        x3, ps = _conv_fwd(lib, x2, sc2, sh2, w3, b3, True)        # ps [n, 64, 2]: plane sums of relu(x3), relu(x3)^2
        sums3 = _channel_sums(ps, weights) if training else None
        mean3, invstd3, sc3, sh3 = _coefficients(sums3, wsum * _HW[2], bns[2], training)
        pooled = ps[:, :, 0] * (sc3 / _HW[2]) + sh3
        ctx.wsum = wsum
        ctx.has_weights = weights is not None
        ctx.norm = norm
        ctx.save_for_backward(img, x1, x2, x3, ps, weights if weights is not None else img.new_empty(0), w2, w3, g1, g2, g3,
                              mean1, invstd1, sc1, sh1, mean2, invstd2, sc2, sh2, mean3, invstd3)
        return pooled

    @staticmethod
    def backward(ctx, dpool):
        (img, x1, x2, x3, ps, weights, w2, w3, g1, g2, g3, mean1, invstd1, sc1, sh1, mean2, invstd2, sc2, sh2, mean3,
         invstd3) = ctx.saved_tensors
        lib = N.load()
        weights = weights if ctx.has_weights else None
        n = img.shape[0]
        m1, m2, m3 = (ctx.wsum * hw for hw in _HW)
        dpool = dpool.contiguous()
        # layer 3: the pool spreads dpool / HW over the plane; sum dy = sum_n dpool, sum dy xhat = sum_n dpool mean_hw(xhat)
        dbeta3 = dpool.sum(0, dtype=torch.float64)
        xhat_mean = (ps[:, :, 0].double() / _HW[2] - mean3.double()) * invstd3.double()
        dgamma3 = (dpool.double() * xhat_mean).sum(0)
        sums3 = torch.stack((dbeta3, dgamma3), dim=1).float().contiguous()
        coef3 = torch.stack((mean3, invstd3, g3.detach() * invstd3, torch.full_like(mean3, 1.0 / m3)), dim=1).contiguous()
        dyp = (dpool / _HW[2]).contiguous()
        dx3 = torch.empty_like(x3)
        N.check(lib.ag_relu_bn_bwd_dx_plane(dyp.data_ptr(), x3.data_ptr(), coef3.data_ptr(), sums3.data_ptr(), _wptr(weights),
                                            dx3.data_ptr(), n, 64, _HW[2], _stream(img)), "ag_relu_bn_bwd_dx_plane")
        dw3, db3 = _conv_wgrad(lib, dx3, x2, sc2, sh2, 64)
        dy2 = _conv_dgrad(lib, dx3, w3, x2)
        del dx3
        dx2, dgamma2, dbeta2 = _bn_backward(lib, dy2, x2, mean2, invstd2, g2, m2, weights)
        dw2, db2 = _conv_wgrad(lib, dx2, x1, sc1, sh1, 32)
        dy1 = _conv_dgrad(lib, dx2, w2, x1)
        del dx2, dy2
        # layer 1: the ReLU + BatchNorm backward is folded into the weight-gradient kernel (dx1 is never written)
        sums1 = _bn_reduce(lib, dy1, x1, mean1, invstd1)
        a = g1.detach() * invstd1
        tab = torch.stack((a, -a * invstd1 * sums1[:, 1] / m1, a * (invstd1 * mean1 * sums1[:, 1] - sums1[:, 0]) / m1,
                           torch.zeros_like(a)), dim=1).contiguous()
        dw1, db1 = _conv1_wgrad(lib, dy1, x1, tab, weights, img, ctx.norm)
        return (None, None, None, None, None, dw1, db1, sums1[:, 1], sums1[:, 0], dw2, db2, dgamma2, dbeta2, dw3, db3,
                sums3[:, 1].clone(), sums3[:, 0].clone())


def trunk(x, features, weights=None, norm=None):
    """`features(x)` flattened to [N, 64] (the caller has checked `usable(x, features)`): batch statistics when the BatchNorm
    layers are in training mode (all three must agree), running statistics otherwise.  norm = (mean, std) (optional, per-pixel
    [212 * 120]): x is the RAW image and the first convolution normalises it, clamp((x - mean) / std, -5, 5), while staging."""
    layers = list(features)
    convs, bns = (layers[0], layers[3], layers[6]), (layers[2], layers[5], layers[8])
    training = bns[0].training
    assert all(bn.training == training for bn in bns)
    if x.requires_grad:
        raise NotImplementedError("no input-gradient kernel for the first convolution (its input is the image)")
    args = []
    for conv, bn in zip(convs, bns):
        args += [conv.weight, conv.bias, bn.weight, bn.bias]
    return _Trunk.apply(x, weights if training else None, bns, training, norm, *args)
