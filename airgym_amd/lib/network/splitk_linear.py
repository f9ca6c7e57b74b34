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
            gw = _splitk_wgrad(go_c, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = go.sum(0)
        return gx, gw, gb


def sum_slices(t):
    """t [S, ...] -> sum over dim 0.  torch's dim-0 reduction of the split-K partials runs in ~8 us on MI355X
    (a plain streaming HIP kernel with 64 strided loads per thread measured 42-97 us, so it was dropped)."""
    return t.sum(0)


def _splitk_wgrad(go, x):
    m, s = x.shape[0], SPLIT_K
    return sum_slices(torch.bmm(go.view(s, m // s, go.shape[1]).transpose(1, 2), x.view(s, m // s, x.shape[1])))


class _LinearEluFn(torch.autograd.Function):
    """h = ELU(x W^T + b) with (a) the activation applied in place on the GEMM output (only h is kept for
    backward: ELU'(z) = h + 1 for z <= 0), (b) ELU-backward and the bias gradient fused in one HIP pass
    (`ag_elu_bwd_bias`), (c) split-K weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        h = F.elu_(F.linear(x, weight, bias))
        ctx.save_for_backward(x, weight, h)
        return h

    @staticmethod
    def backward(ctx, dh):
        import ctypes

        from airgym_amd import _native as N
        x, weight, h = ctx.saved_tensors
        lib = N.load()
        dh = dh.contiguous()
        m, c = h.shape
        dz = torch.empty_like(h)
        rows = lib.ag_elu_bwd_bias_rows_per_block()
        partials = torch.empty((m + rows - 1) // rows, c, dtype=torch.float32, device=h.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        N.check(lib.ag_elu_bwd_bias(dh.data_ptr(), h.data_ptr(), dz.data_ptr(), partials.data_ptr(), m, c, stream),
                "ag_elu_bwd_bias")
        gb = sum_slices(partials) if ctx.needs_input_grad[2] else None
        gx = dz @ weight if ctx.needs_input_grad[0] else None
        gw = _splitk_wgrad(dz, x) if ctx.needs_input_grad[1] else None
        return gx, gw, gb


def _fusable(x, weight):
    return (x.is_cuda and x.dim() == 2 and x.shape[0] >= MIN_ROWS and x.shape[0] % SPLIT_K == 0
            and x.is_contiguous() and torch.is_grad_enabled() and weight.requires_grad)


def linear_elu(x, weight, bias):
    """ELU(F.linear(x, weight, bias)); fused backward on large CUDA batches."""
    c = weight.shape[0]
    if _fusable(x, weight) and bias is not None and c % 4 == 0 and c <= 1024 and 256 % (c // 4) == 0:
        return _LinearEluFn.apply(x, weight, bias)
    return F.elu(linear(x, weight, bias))


def linear(x, weight, bias=None):
    """F.linear with a split-K wgrad when the batch is large enough to need it."""
    if (x.is_cuda and x.dim() == 2 and x.shape[0] >= MIN_ROWS and x.shape[0] % SPLIT_K == 0
            and x.is_contiguous() and torch.is_grad_enabled() and weight.requires_grad):
        return _SplitKLinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)
