"""Linear layer whose WEIGHT GRADIENT is a split-K batched GEMM.

Why: in the PPO update the batch dimension is M = minibatch = O(10^5) while the layers are tiny
(18->256, 256->256, 256->5).  dW = dY^T X reduces over M into an output of at most 256x256 = ONE
macro-tile, so hipBLASLt's heuristic launches a handful of workgroups on a 256-CU part: measured on
MI355X at M = 196 608 (profiles/r01_gemm_probe.md): 595 us (256x256), 421 us (18->256), 244 us
(256->5).  Splitting M into S slices (`bmm` of S independent [N, M/S] x [M/S, K] products, then a
sum over S) fills the chip: 213 / 45 / 43 us.  Forward and dX keep the plain hipBLASLt MFMA GEMMs,
which already run near the fp32 matrix peak for these shapes.
"""
import torch
import torch.nn.functional as F

SPLIT_K = 64
MIN_ROWS = 8192


class _SplitKLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = go @ weight
        m = x.shape[0]
        if ctx.needs_input_grad[1]:
            go_c = go.contiguous()
            s = SPLIT_K
            gw = torch.bmm(go_c.view(s, m // s, go.shape[1]).transpose(1, 2), x.view(s, m // s, x.shape[1])).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = go.sum(0)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """F.linear with a split-K wgrad when the batch is large enough to need it."""
    if (x.is_cuda and x.dim() == 2 and x.shape[0] >= MIN_ROWS and x.shape[0] % SPLIT_K == 0
            and x.is_contiguous() and torch.is_grad_enabled() and weight.requires_grad):
        return _SplitKLinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)
