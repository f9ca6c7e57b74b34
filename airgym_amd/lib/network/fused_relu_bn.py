"""ReLU -> BatchNorm2d as one autograd node on the HIP kernels of csrc/cnn_kernels.hip (reference modules:
lib/network/cnn.py:3-33, `nn.ReLU()` followed by `nn.BatchNorm2d(c)`).

Same arithmetic as the two torch modules - batch statistics (biased variance) in training with the running statistics
updated by `momentum` from the unbiased variance, running statistics in eval - but the ReLU output is never written
(the backward recomputes it from the convolution output) and every pass runs over the whole chip instead of MIOpen's
16-64 workgroups.  The modules themselves stay in the network (state-dict keys, `num_batches_tracked`)."""
import ctypes

import torch

from airgym_amd import _native as N


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _blocks(lib, n, c):
    ppb = lib.ag_relu_bn_planes_per_block()
    return (n * c + ppb - 1) // ppb


def usable(x, bn):
    # momentum = None means a cumulative average (factor 1 / num_batches_tracked) in nn.BatchNorm2d: not implemented by the
    # fused node, such a layer stays on the library path
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and bn.num_features <= 64
            and bn.affine and bn.track_running_stats and bn.momentum is not None)


class _ReluBatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, weights=None):
        """weights [N] (optional): multiplicity of each image in the minibatch it stands for (see relu_batchnorm)."""
        lib = N.load()
        n, c, h, w = x.shape
        hw = h * w
        m = float(n * hw) if weights is None else float(weights.double().sum().item()) * hw
        ctx.m = m
        partials = torch.empty(_blocks(lib, n, c), c, 2, dtype=torch.float32, device=x.device)
        N.check(lib.ag_relu_bn_stats_weighted(x.data_ptr(), weights.data_ptr() if weights is not None else None,
                                              partials.data_ptr(), n, c, hw, _stream(x)), "ag_relu_bn_stats")
        sums = partials.sum(0, dtype=torch.float64)
        mean = sums[:, 0] / m
        var = torch.clamp(sums[:, 1] / m - mean * mean, min=0.0)
        invstd = torch.rsqrt(var + eps)
        scale = gamma.double() * invstd
        shift = beta.double() - mean * scale
        y = torch.empty_like(x)
        scale32, shift32 = scale.float(), shift.float()      # named: a temporary's storage is recycled before the launch reads it
        N.check(lib.ag_relu_bn_apply(x.data_ptr(), scale32.data_ptr(), shift32.data_ptr(), y.data_ptr(), n, c, hw,
                                     _stream(x)), "ag_relu_bn_apply")
        with torch.no_grad():                   # nn.BatchNorm2d: running = (1 - momentum) running + momentum stat (unbiased var)
            running_mean.mul_(1.0 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
            running_var.mul_(1.0 - momentum).add_((var * (m / max(m - 1, 1))).to(running_var.dtype), alpha=momentum)
        ctx.save_for_backward(x, gamma, mean.float(), invstd.float(), weights if weights is not None else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, weights = ctx.saved_tensors
        lib = N.load()
        n, c, h, w = x.shape
        hw, m = h * w, ctx.m
        dy = dy.contiguous()
        partials = torch.empty(_blocks(lib, n, c), c, 2, dtype=torch.float32, device=x.device)
        N.check(lib.ag_relu_bn_bwd_reduce(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), partials.data_ptr(),
                                          n, c, hw, _stream(x)), "ag_relu_bn_bwd_reduce")
        sums = partials.sum(0, dtype=torch.float64).float().contiguous()      # [C, 2] = (dbeta, dgamma)
        coef = torch.stack((mean, invstd, gamma.detach() * invstd, torch.full_like(mean, 1.0 / m)), dim=1).contiguous()
        dx = torch.empty_like(x)
        N.check(lib.ag_relu_bn_bwd_dx_weighted(dy.data_ptr(), x.data_ptr(), coef.data_ptr(), sums.data_ptr(),
                                               weights.data_ptr() if weights.numel() else None, dx.data_ptr(), None, None, 0, n, c, hw,
                                               _stream(x)), "ag_relu_bn_bwd_dx")
        return dx, sums[:, 1].clone(), sums[:, 0].clone(), None, None, None, None, None


def relu_batchnorm(x, bn, weights=None):
    """relu then `bn` (an nn.BatchNorm2d) on the HIP kernels; x is the convolution output [N, C, H, W].

    weights [N] f32 (training only): image i stands for weights[i] identical images of the minibatch the reference would
    process (frame de-duplication: the depth camera renders every 4th env step).  The batch statistics are those of the full
    minibatch (weighted), and with the upstream gradient summed over the copies (what autograd delivers when the features
    are gathered back per sample) the parameter gradients are the full minibatch's."""
    if bn.training:
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        momentum = float(bn.momentum)      # usable() refuses momentum = None
        if weights is not None:
            weights = weights.to(device=x.device, dtype=torch.float32).contiguous()
            assert weights.shape == (x.shape[0],)
        return _ReluBatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, float(bn.eps),
                                         weights)
    lib = N.load()
    n, c, h, w = x.shape
    with torch.no_grad():
        scale = bn.weight.double() * torch.rsqrt(bn.running_var.double() + bn.eps)
        shift = bn.bias.double() - bn.running_mean.double() * scale
        y = torch.empty_like(x)
        scale32, shift32 = scale.float(), shift.float()
        N.check(lib.ag_relu_bn_apply(x.data_ptr(), scale32.data_ptr(), shift32.data_ptr(), y.data_ptr(), n, c, h * w,
                                     _stream(x)), "ag_relu_bn_apply")
    return y


def relu_batchnorm_torch(x, bn, weights):
    """The same weighted ReLU -> BatchNorm2d out of torch ops (any device; autograd derives the backward): the path for the
    CPU test-suite and for layers the fused node does not take.  weights = None is nn.BatchNorm2d itself."""
    if weights is None or not bn.training:
        return bn(torch.relu(x))
    r = torch.relu(x)
    wv = weights.to(device=x.device, dtype=x.dtype).view(-1, 1, 1, 1)
    hw = x.shape[2] * x.shape[3]
    m = wv.sum() * hw
    mean = (wv * r).sum((0, 2, 3)) / m
    var = (wv * (r - mean.view(1, -1, 1, 1)) ** 2).sum((0, 2, 3)) / m
    y = (r - mean.view(1, -1, 1, 1)) * torch.rsqrt(var + bn.eps).view(1, -1, 1, 1)
    if bn.affine:
        y = y * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)
    with torch.no_grad():
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        # momentum = None is nn.BatchNorm2d's cumulative moving average: factor 1 / num_batches_tracked
        mom = float(bn.momentum) if bn.momentum is not None else 1.0 / max(float(bn.num_batches_tracked), 1.0)
        bn.running_mean.mul_(1.0 - mom).add_(mean.detach(), alpha=mom)
        bn.running_var.mul_(1.0 - mom).add_(var.detach() * (m / torch.clamp(m - 1, min=1.0)), alpha=mom)
    return y
