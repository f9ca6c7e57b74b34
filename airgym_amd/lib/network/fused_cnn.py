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


def _conv1_fwd(lib, img, index, norm, w, b, want_stats):
    """(y, stats): stats [n, 1, 16, 2] = per image (sum relu(y), sum relu(y)^2) per channel, or None.  index (optional, int64
    [n]): image i is img[index[i]]."""
    n = img.shape[0] if index is None else index.shape[0]
    y = torch.empty(n, 16, 106, 60, dtype=torch.float32, device=img.device)
    stats = torch.empty(n, 1, 16, 2, dtype=torch.float32, device=img.device) if want_stats else None
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(1, 16), dtype=torch.float32, device=img.device)
    N.check(lib.ag_cnn_conv1_fwd(img.data_ptr(), _wptr(index), *_norm_ptrs(norm), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                 stats.data_ptr() if want_stats else None, n, ws.data_ptr(), _stream(img)), "ag_cnn_conv1_fwd")
    return y, stats


# forward of the two 3 x 3 layers on the bf16 matrix cores at float32 accuracy (ag_cnn_conv_fwd_split); False: the f32-input MFMA kernel
SPLIT_FWD = True


def _conv_fwd(lib, x, coef, w, b, want_stats):
    """(y, stats): the layer applied to relu(x) * coef[2] + coef[3]; stats [n, bands, cout, 2] = per (image, band of output rows)
    the sums (relu(y), relu(y)^2) per output channel, or None."""
    n, cin, hin, win = x.shape
    cout = w.shape[0]
    y = torch.empty(n, cout, (hin - 1) // 2 + 1, win // 2, dtype=torch.float32, device=x.device)
    bands = (lib.ag_cnn_conv_fwd_split_bands if SPLIT_FWD else lib.ag_cnn_conv_fwd_bands)(cin, cout, hin, win)
    stats = torch.empty(n, bands, cout, 2, dtype=torch.float32, device=x.device) if want_stats else None
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=x.device)
    fwd = lib.ag_cnn_conv_fwd_split if SPLIT_FWD else lib.ag_cnn_conv_fwd
    N.check(fwd(x.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                stats.data_ptr() if want_stats else None, n, cin, cout, hin, win, ws.data_ptr(), _stream(x)),
            "ag_cnn_conv_fwd_split" if SPLIT_FWD else "ag_cnn_conv_fwd")
    return y, stats


def _conv_dgrad(lib, dz, w, like):
    n, cin, hin, win = like.shape
    cout = w.shape[0]
    dx = torch.empty_like(like)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=dz.device)
    N.check(lib.ag_cnn_conv_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), n, cin, cout, hin, win, ws.data_ptr(), _stream(dz)),
            "ag_cnn_conv_dgrad")
    return dx


def _conv_dgrad_bn(lib, dz, w, x, tab, weights):
    """Input gradient of the third convolution with the backward of the ReLU + BatchNorm in front of it applied in the kernel's
    epilogue: (gradient of the second convolution's output x, [32, 6] sums of it: total, row 0, last row, column 0, two corners)."""
    n, cin, hin, win = x.shape
    cout = w.shape[0]
    dx = torch.empty_like(x)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(cin, cout), dtype=torch.float32, device=dz.device)
    rows = lib.ag_cnn_conv_dgrad_bn_rows(n, cin, cout, hin, win)
    sums = torch.empty(rows, cin, 6, dtype=torch.float32, device=dz.device)
    N.check(lib.ag_cnn_conv_dgrad_bn(dz.data_ptr(), w.data_ptr(), x.data_ptr(), tab.data_ptr(), _wptr(weights), dx.data_ptr(),
                                     sums.data_ptr(), n, cin, cout, hin, win, ws.data_ptr(), _stream(dz)), "ag_cnn_conv_dgrad_bn")
    return dx, sums.sum(0)


def _conv_wgrad(lib, dz, x, coef, cout, out=None):
    """Weight gradient of a 3x3 layer whose input is relu(x) * coef[2] + coef[3] (the bias gradient comes from the plane sums of dz
    that the kernel producing dz emits: with_bias = 0).  out: a contiguous [cout, cin, 3, 3] tensor the result is summed into
    directly (returns None then)."""
    n, cin, hin, win = x.shape
    g = lib.ag_cnn_conv_wgrad_partials(n, cin, cout, hin, win)
    partials = torch.empty(g, cout * cin * 9 + cout, dtype=torch.float32, device=x.device)
    N.check(lib.ag_cnn_conv_wgrad(dz.data_ptr(), x.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), partials.data_ptr(), 0, n, cin,
                                  cout, hin, win, _stream(x)), "ag_cnn_conv_wgrad")
    if out is not None:
        torch.sum(partials[:, :cout * cin * 9], 0, out=out.view(-1))
        return None
    return partials[:, :cout * cin * 9].sum(0).reshape(cout, cin, 3, 3)


def _conv1_wgrad(lib, dy, x1, tab, weights, img, index, norm, out=None):
    """Weight / bias gradient of the first convolution from the gradient dy of its ReLU + BatchNorm output (the BatchNorm backward
    is folded into the kernel's staging: tab [16, 4], see ag_cnn_conv1_wgrad)."""
    n = x1.shape[0]
    g = lib.ag_cnn_conv1_wgrad_partials(n)
    partials = torch.empty(g, 16, 32, dtype=torch.float32, device=img.device)
    N.check(lib.ag_cnn_conv1_wgrad(dy.data_ptr(), x1.data_ptr(), tab.data_ptr(), _wptr(weights), img.data_ptr(), _wptr(index),
                                   *_norm_ptrs(norm), partials.data_ptr(), n, _stream(img)), "ag_cnn_conv1_wgrad")
    if out is not None:
        torch.sum(partials[:, :, :25], 0, out=out[0].view(16, 25))
        torch.sum(partials[:, :, 25], 0, out=out[1])
        return None, None
    s = partials.sum(0)
    return s[:, :25].reshape(16, 1, 5, 5), s[:, 25]


def _conv_dgrad_conv1_wgrad(lib, dz, w, x1, tab, weights, img, index, norm, out=None):
    """conv2's input gradient, the first layer's ReLU + BatchNorm backward and the first convolution's weight / bias gradient as one
    kernel (ag_cnn_conv_dgrad_conv1_wgrad): the gradient of the first convolution's output is never written."""
    n = x1.shape[0]
    rows = lib.ag_cnn_conv_dgrad_conv1_wgrad_partials(n)
    partials = torch.empty(rows, 16, 32, dtype=torch.float32, device=img.device)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(16, 32), dtype=torch.float32, device=img.device)
    N.check(lib.ag_cnn_conv_dgrad_conv1_wgrad(dz.data_ptr(), w.data_ptr(), x1.data_ptr(), tab.data_ptr(), _wptr(weights), img.data_ptr(),
                                              _wptr(index), *_norm_ptrs(norm), partials.data_ptr(), n, ws.data_ptr(), _stream(img)),
            "ag_cnn_conv_dgrad_conv1_wgrad")
    if out is not None:
        torch.sum(partials[:, :, :25], 0, out=out[0].view(16, 25))
        torch.sum(partials[:, :, 25], 0, out=out[1])
        return None, None
    s = partials.sum(0)
    return s[:, :25].reshape(16, 1, 5, 5), s[:, 25]


def _blocks(lib, n, c):
    ppb = lib.ag_relu_bn_planes_per_block()
    return (n * c + ppb - 1) // ppb


def _wptr(weights):
    return weights.data_ptr() if weights is not None else None


def _finalize(lib, stats, weights, n, m, bn, training, hw, pool=False):
    """ag_bn_finalize: coef [4, C] = (mean, invstd, scale, shift) of the ReLU + BatchNorm that follows a convolution, from the
    convolution's `stats`; running statistics updated in training.  pool: also (plane1, pooled) [n, C] (last layer)."""
    c = bn.num_features
    dev = bn.weight.device
    coef = torch.empty(4, c, dtype=torch.float32, device=dev)
    plane1 = torch.empty(n, c, dtype=torch.float32, device=dev) if pool else None
    pooled = torch.empty(n, c, dtype=torch.float32, device=dev) if pool else None
    g = stats.shape[1] if stats is not None else 1
    nb = bn.num_batches_tracked
    scratch = torch.empty(lib.ag_bn_scratch_doubles(), dtype=torch.float64, device=dev)
    N.check(lib.ag_bn_finalize(stats.data_ptr() if stats is not None else None, _wptr(weights), n, g, c, float(m), bn.weight.data_ptr(),
                               bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                               nb.data_ptr() if nb is not None else None, float(bn.momentum), float(bn.eps), int(training),
                               coef.data_ptr(), plane1.data_ptr() if pool else None, pooled.data_ptr() if pool else None, hw,
                               scratch.data_ptr(), _stream(coef)), "ag_bn_finalize")
    return coef, plane1, pooled


def _gptr(t):
    return t.data_ptr() if t is not None else None


def _bn_reduce(lib, dy, x, coef, gamma, m, mode, gout=(None, None)):
    """ReLU + BatchNorm backward, first half: (sums [C, 2] = (dbeta, dgamma), tab [C, 4]) from the gradient dy of the layer's output
    and the layer's own output x (ag_relu_bn_bwd_reduce + ag_bn_bwd_prep; mode: see ag_bn_bwd_prep).  gout = (dgamma, dbeta) tensors
    the two parameter gradients are also written into."""
    n, c, h, w = x.shape
    blocks = _blocks(lib, n, c)
    partials = torch.empty(blocks, c, 2, dtype=torch.float32, device=x.device)
    N.check(lib.ag_relu_bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), partials.data_ptr(), n, c,
                                      h * w, _stream(x)), "ag_relu_bn_bwd_reduce")
    sums = torch.empty(c, 2, dtype=torch.float32, device=x.device)
    tab = torch.empty(c, 4, dtype=torch.float32, device=x.device)
    scratch = torch.empty(lib.ag_bn_scratch_doubles(), dtype=torch.float64, device=x.device)
    N.check(lib.ag_bn_bwd_prep(partials.data_ptr(), blocks, c, coef.data_ptr(), gamma.data_ptr(), float(m), mode, sums.data_ptr(),
                               tab.data_ptr(), _gptr(gout[0]), _gptr(gout[1]), scratch.data_ptr(), _stream(x)), "ag_bn_bwd_prep")
    return sums, tab


def bn_sums_from_conv(w, dw, total, border, gamma, beta, hin):
    """The two reductions of a ReLU + BatchNorm backward - (dbeta, dgamma) = (sum dy, sum dy xhat) over images and pixels, [C, 2] -
    WITHOUT a pass over dy, for a layer whose output y feeds a 3x3 / stride-2 / pad-1 convolution with weights w [co, C, 3, 3]:
    with dz the convolution's output gradient, dy = conv^T(dz) and dw its weight gradient,
        sum_p dy[c,p] y[c,p] = sum_{co,tap} w[co,c,tap] dw[co,c,tap]                (both are the same bilinear form in (dz, y))
        sum_p dy[c,p]        = sum_{co,tap} w[co,c,tap] S[co,tap]
    where S[co,tap] sums dz[co] over the output pixels whose tap lies inside the (unpadded) input: the plane total minus border rows /
    columns - total [co] = sum dz, border [co, 5] = (row 0, last row, column 0, dz[0][0], dz[last][0]) sums (ag_plane_border_sums).
    xhat = (y - beta) / gamma, so dgamma = (sum dy y - beta sum dy) / gamma; gamma = 0 (y constant: xhat is not recoverable) is
    clamped away from zero - callers that can meet it use the reduction kernel instead (`CNNFeatureExtractor.bn_sums_from_weights`)."""
    w64, dw64 = w.detach().double(), dw.detach().double()
    t, b = total.double(), border.double()
    last_row_out = (hin % 2 == 1)           # tap ky = 2 of the last output row reads row hin (padding) iff hin is odd
    S = t.view(-1, 1, 1).repeat(1, 3, 3)
    S[:, 0, :] -= b[:, 0:1]                  # ky = 0: output row 0 reads input row -1
    S[:, :, 0] -= b[:, 2:3]                  # kx = 0: output column 0 reads input column -1
    S[:, 0, 0] += b[:, 3]
    if last_row_out:
        S[:, 2, :] -= b[:, 1:2]
        S[:, 2, 0] += b[:, 4]
    sum_dy_y = (w64 * dw64).sum((0, 2, 3))
    sum_dy = torch.einsum("ockl,okl->c", w64, S)
    g = gamma.detach().double()
    g = torch.where(g.abs() < 1e-30, torch.full_like(g, 1e-30), g)
    return torch.stack((sum_dy, (sum_dy_y - beta.detach().double() * sum_dy) / g), dim=1)


def _bn_prep_from_conv(lib, w, dw, total, border, coef, gamma, beta, hin, m, mode, gout=(None, None)):
    """(sums [C, 2], tab [C, 4]) of the ReLU + BatchNorm in front of a convolution from that convolution's weights w, weight
    gradient dw, the total [cout] (= its bias gradient) and border sums [cout, 5] of its output gradient: `bn_sums_from_conv` as one
    launch (ag_bn_sums_from_conv), then ag_bn_bwd_prep on the single row."""
    cout, c = w.shape[0], w.shape[1]
    part = torch.empty(1, c, 2, dtype=torch.float32, device=w.device)
    N.check(lib.ag_bn_sums_from_conv(w.data_ptr(), dw.data_ptr(), total.data_ptr(), border.data_ptr(), cout, c, hin, gamma.data_ptr(),
                                     beta.data_ptr(), part.data_ptr(), _stream(w)), "ag_bn_sums_from_conv")
    sums = torch.empty(c, 2, dtype=torch.float32, device=w.device)
    tab = torch.empty(c, 4, dtype=torch.float32, device=w.device)
    scratch = torch.empty(lib.ag_bn_scratch_doubles(), dtype=torch.float64, device=w.device)
    N.check(lib.ag_bn_bwd_prep(part.data_ptr(), 1, c, coef.data_ptr(), gamma.data_ptr(), float(m), mode, sums.data_ptr(), tab.data_ptr(),
                               _gptr(gout[0]), _gptr(gout[1]), scratch.data_ptr(), _stream(w)), "ag_bn_bwd_prep")
    return sums, tab


class _Trunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, weights, bns, training, norm, index, gout, from_weights, dgrad_epilogue, conv1_fused, w1, b1, g1, be1, w2, b2, g2,
                be2, w3, b3, g3, be3):
        lib = N.load()
        img = img.contiguous()
        if index is not None:
            index = index.to(device=img.device, dtype=torch.long).contiguous()
        n = img.shape[0] if index is None else index.shape[0]
        if weights is not None:
            # the sum of the multiplicities is the number of samples the distinct images stand for: a caller that knows it
            # (DedupFrames: the length of the minibatch slice) attaches it as `ag_sum`, and the step has no host synchronisation
            known = getattr(weights, "ag_sum", None)
            weights = weights.to(device=img.device, dtype=torch.float32).contiguous()
            assert weights.shape == (n,)
            wsum = float(known) if known is not None else float(weights.double().sum().item())
        else:
            wsum = float(n)
        w1, w2, w3 = w1.contiguous(), w2.contiguous(), w3.contiguous()
        if norm is not None:
            norm = tuple(t.to(device=img.device, dtype=torch.float32).contiguous().view(-1) for t in norm)
            assert norm[0].numel() == 212 * 120 and norm[1].numel() == 212 * 120
        x1, st1 = _conv1_fwd(lib, img, index, norm, w1, b1, training)
        coef1, _, _ = _finalize(lib, st1, weights, n, wsum * _HW[0], bns[0], training, _HW[0])
        x2, st2 = _conv_fwd(lib, x1, coef1, w2, b2, training)
        coef2, _, _ = _finalize(lib, st2, weights, n, wsum * _HW[1], bns[1], training, _HW[1])
        x3, st3 = _conv_fwd(lib, x2, coef2, w3, b3, True)
        coef3, plane1, pooled = _finalize(lib, st3, weights, n, wsum * _HW[2], bns[2], training, _HW[2], pool=True)
        ctx.wsum = wsum
        ctx.has_weights = weights is not None
        ctx.norm = norm
        ctx.index = index
        ctx.gout = gout
        ctx.from_weights = from_weights
        ctx.dgrad_epilogue = bool(from_weights and dgrad_epilogue)
        ctx.conv1_fused = bool(from_weights and conv1_fused)
        ctx.save_for_backward(img, x1, x2, x3, plane1, weights if weights is not None else img.new_empty(0), w2, w3, g1, g2, g3,
                              coef1, coef2, coef3, be1, be2)
        return pooled

    @staticmethod
    def backward(ctx, dpool):
        img, x1, x2, x3, plane1, weights, w2, w3, g1, g2, g3, coef1, coef2, coef3, be1, be2 = ctx.saved_tensors
        lib = N.load()
        weights = weights if ctx.has_weights else None
        n = x1.shape[0]
        m1, m2, m3 = (ctx.wsum * hw for hw in _HW)
        dev = img.device
        dpool = dpool.contiguous()
        # layer 3: the pool spreads dpool / HW over the plane; sum dy = sum_n dpool, sum dy xhat = sum_n dpool mean_hw(xhat)
        # gout: the twelve parameters' .grad tensors (views of the optimizer's flat gradient buffer): results are written there
        # directly and autograd is handed None (no accumulation launches); otherwise the gradients are returned
        go = ctx.gout
        G = (lambda i: go[i]) if go is not None else (lambda i: None)
        sums3 = torch.empty(64, 2, dtype=torch.float32, device=dev)
        tab3 = torch.empty(64, 4, dtype=torch.float32, device=dev)
        dyp = torch.empty(n, 64, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.ag_bn_scratch_doubles(), dtype=torch.float64, device=dev)
        N.check(lib.ag_bn_pool_bwd_prep(dpool.data_ptr(), plane1.data_ptr(), n, 64, coef3.data_ptr(), g3.data_ptr(), float(m3), _HW[2],
                                        sums3.data_ptr(), tab3.data_ptr(), dyp.data_ptr(), _gptr(G(10)), _gptr(G(11)),
                                        scratch.data_ptr(), _stream(img)), "ag_bn_pool_bwd_prep")
        dx3 = torch.empty_like(x3)
        fw = ctx.from_weights       # the ReLU + BatchNorm reductions from the next convolution's (w, dw) instead of a pass over dy
        ps3 = torch.empty(n, 64, dtype=torch.float32, device=dev)          # per-plane sums of dx3: db3 = their sum over images
        bs3 = torch.empty(n, 64, 5, dtype=torch.float32, device=dev) if fw else None      # and its border sums
        N.check(lib.ag_relu_bn_bwd_dx_plane(dyp.data_ptr(), x3.data_ptr(), tab3.data_ptr(), sums3.data_ptr(), _wptr(weights),
                                            dx3.data_ptr(), ps3.data_ptr(), _gptr(bs3), 15, n, 64, _HW[2], _stream(img)),
                "ag_relu_bn_bwd_dx_plane")
        border3 = bs3.sum(0) if fw else None
        dw3 = _conv_wgrad(lib, dx3, x2, coef2, 64, G(8))
        db3 = torch.sum(ps3, 0, out=G(9)) if go is not None else ps3.sum(0)
        epi = ctx.dgrad_epilogue    # layer 2's ReLU + BatchNorm backward in the epilogue of conv3's input gradient
        if fw:
            sums2, tab2 = _bn_prep_from_conv(lib, w3, (G(8) if go is not None else dw3).contiguous(), db3.contiguous(), border3, coef2,
                                             g2, be2, 53, m2, 1 if epi else 0, (G(6), G(7)))
        if epi:
            dy2, tot2 = _conv_dgrad_bn(lib, dx3, w3, x2, tab2, weights)       # dy2 is the gradient of x2 already
            del dx3
            border2 = tot2[:, 1:].contiguous()
        else:
            dy2 = _conv_dgrad(lib, dx3, w3, x2)
            del dx3
            # layer 2: dx2 written over dy2
            if not fw:
                sums2, tab2 = _bn_reduce(lib, dy2, x2, coef2, g2, m2, 0, (G(6), G(7)))
            ps2 = torch.empty(n, 32, dtype=torch.float32, device=dev)
            bs2 = torch.empty(n, 32, 5, dtype=torch.float32, device=dev) if fw else None
            N.check(lib.ag_relu_bn_bwd_dx_weighted(dy2.data_ptr(), x2.data_ptr(), tab2.data_ptr(), sums2.data_ptr(), _wptr(weights),
                                                   dy2.data_ptr(), ps2.data_ptr(), _gptr(bs2), 30, n, 32, _HW[1], _stream(img)),
                    "ag_relu_bn_bwd_dx")
            border2 = bs2.sum(0) if fw else None
        dw2 = _conv_wgrad(lib, dy2, x1, coef1, 32, G(4))
        if epi:
            db2 = G(5).copy_(tot2[:, 0]) if go is not None else tot2[:, 0].contiguous()
        else:
            db2 = torch.sum(ps2, 0, out=G(5)) if go is not None else ps2.sum(0)
        if fw:
            sums1, tab1 = _bn_prep_from_conv(lib, w2, (G(4) if go is not None else dw2).contiguous(), db2.contiguous(), border2, coef1,
                                             g1, be1, 106, m1, 1, (G(2), G(3)))
        if ctx.conv1_fused:
            # layer 1 in the same kernel as conv2's input gradient: neither that gradient nor dx1 is ever written
            dw1, db1 = _conv_dgrad_conv1_wgrad(lib, dy2, w2, x1, tab1, weights, img, ctx.index, ctx.norm,
                                               (G(0), G(1)) if go is not None else None)
            del dy2
        else:
            dy1 = _conv_dgrad(lib, dy2, w2, x1)
            del dy2
            # layer 1: the ReLU + BatchNorm backward is folded into the weight-gradient kernel (dx1 is never written)
            if not fw:
                sums1, tab1 = _bn_reduce(lib, dy1, x1, coef1, g1, m1, 1, (G(2), G(3)))
            dw1, db1 = _conv1_wgrad(lib, dy1, x1, tab1, weights, img, ctx.index, ctx.norm, (G(0), G(1)) if go is not None else None)
        if go is not None:
            return (None,) * 22
        return ((None,) * 10 + (dw1, db1, sums1[:, 1], sums1[:, 0], dw2, db2, sums2[:, 1], sums2[:, 0], dw3, db3, sums3[:, 1],
                               sums3[:, 0]))


def trunk(x, features, weights=None, norm=None, index=None, direct_grads=False, sums_from_weights=True, dgrad_epilogue=True,
          conv1_fused=True):
    """`features(x)` flattened to [N, 64] (the caller has checked `usable(x, features)`): batch statistics when the BatchNorm
    layers are in training mode (all three must agree), running statistics otherwise.  norm = (mean, std) (optional, per-pixel
    [212 * 120]): x is the RAW image and the first convolution normalises it, clamp((x - mean) / std, -5, 5), while staging.
    index (optional, int64 [N]): the batch is x[index] - read in place, e.g. out of the rollout's frame store.
    direct_grads: the backward OVERWRITES the parameters' existing .grad tensors instead of handing gradients to autograd for
    accumulation (the caller zeroes its gradient buffer before every backward and the trunk is applied once per backward).
    sums_from_weights: the backward takes the reductions of the first two ReLU + BatchNorm layers from the following convolution's
    weights and weight gradient (`bn_sums_from_conv`: no pass over the 1.9 GB / 1.0 GB gradients); False = the reduction kernel.
    dgrad_epilogue (needs sums_from_weights, which makes the coefficients known in time): the second layer's ReLU + BatchNorm
    backward runs in the epilogue of the third convolution's input gradient instead of as a pass of its own.
    conv1_fused (needs sums_from_weights too): the second convolution's input gradient, the first layer's ReLU + BatchNorm backward
    and the first convolution's weight gradient run as one kernel."""
    layers = list(features)
    convs, bns = (layers[0], layers[3], layers[6]), (layers[2], layers[5], layers[8])
    training = bns[0].training
    assert all(bn.training == training for bn in bns)
    if x.requires_grad:
        raise NotImplementedError("no input-gradient kernel for the first convolution (its input is the image)")
    args = []
    for conv, bn in zip(convs, bns):
        args += [conv.weight, conv.bias, bn.weight, bn.bias]
    # parameters with pre-allocated contiguous float32 .grad (the agent keeps them as views of one flat buffer it has just zeroed):
    # the backward writes the gradients there itself.  Valid because the trunk is the parameters' only consumer in the graph.
    gout = None
    if direct_grads and training and torch.is_grad_enabled():
        grads = [p.grad for p in args]
        if all(g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == p.device for g, p in zip(grads, args)):
            gout = grads
    return _Trunk.apply(x, weights if training else None, bns, training, norm, index, gout, bool(sums_from_weights), bool(dgrad_epilogue), bool(conv1_fused), *args)
