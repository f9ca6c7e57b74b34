"""ag_split_gemm (airgym_amd/csrc/split_gemm.hip): the 256 x 256 hidden-layer GEMM of the actor-critic MLP
(lib/network/mlp.py:36-39) on the bf16 matrix cores with an EXACT three-way split of every float32 operand.
The claim to hold it to is "float32 accuracy": measured against a float64 product, its error must not exceed that of the
library's float32 GEMM (which is itself only accumulation-order noise)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def lib():
    from airgym_amd import _native as N
    assert torch.cuda.is_available()
    return N.load()


def _gemm(lib, A, W, transpose, bias=None):
    from airgym_amd import _native as N
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, int(transpose), _stream()), "prepare")
    C = torch.empty(A.shape[0], 256, device="cuda")
    N.check(lib.ag_split_gemm(A.data_ptr(), planes.data_ptr(), bias.data_ptr() if bias is not None else None, C.data_ptr(),
                              A.shape[0], 256, 256, _stream()), "ag_split_gemm")
    return C


def test_operand_layout_with_identity_and_asymmetric_matrix(lib):
    """A = I (padded) against an ASYMMETRIC B catches a swapped row / column or k mapping of the MFMA fragments."""
    g = torch.Generator(device="cuda").manual_seed(0)
    W = torch.randn(256, 256, device="cuda", generator=g)
    A = torch.zeros(300, 256, device="cuda"); A[:256] = torch.eye(256, device="cuda"); A[256:] = torch.eye(256, device="cuda")[:44] * 2.0
    C = _gemm(lib, A, W, transpose=False)                 # C = A W^T  -> rows of W^T
    assert torch.equal(C[:256], W.t().contiguous()) and torch.equal(C[256:], 2.0 * W.t()[:44])
    C2 = _gemm(lib, A, W, transpose=True)                 # C = A W
    assert torch.equal(C2[:256], W)
    # exactness of the split itself: one non-zero per row -> the product of two floats, exact only if all pieces are kept
    x = torch.randn(256, device="cuda", generator=g)
    D = torch.diag(x).contiguous()
    C3 = _gemm(lib, D, W, transpose=True)
    ref = (x.double()[:, None] * W.double())
    # a single product: the six kept terms miss a2 b3 + a3 b2 + a3 b3 < 2^-23 |ab| (round-to-nearest split) and the last f32
    # accumulate rounds once more (2^-24): <= 1.5 ulp in total, against 0.5 ulp for one correctly rounded f32 product
    assert ((C3.double() - ref).abs() <= 1.55 * 2.0 ** -23 * ref.abs() + 1e-30).all()


@pytest.mark.parametrize("M", [1, 127, 128, 4097, 196608])
@pytest.mark.parametrize("transpose", [False, True])
def test_float32_accuracy_against_float64(lib, M, transpose):
    g = torch.Generator(device="cuda").manual_seed(M + int(transpose))
    A = torch.randn(M, 256, device="cuda", generator=g) * torch.exp(2 * torch.randn(M, 1, device="cuda", generator=g))
    W = torch.randn(256, 256, device="cuda", generator=g) / 16.0
    bias = torch.randn(256, device="cuda", generator=g)
    C = _gemm(lib, A, W, transpose, bias)
    Wd = W.double() if transpose else W.double().t()
    n = min(M, 8192)
    ref = A[:n].double() @ Wd + bias.double()
    lib32 = (A[:n] @ (W if transpose else W.t())) + bias
    scale = (A[:n].double().abs() @ Wd.abs()) + bias.double().abs()          # sum |a||b|: the natural error scale
    err_split = ((C[:n].double() - ref).abs() / scale).max().item()
    err_lib = ((lib32.double() - ref).abs() / scale).max().item()
    # f32 accumulation noise is ~1e-7 relative to sum |a||b|; the split GEMM must be in the same class as the library GEMM
    assert err_split < 4e-7, (err_split, err_lib)
    assert err_split <= 3.0 * err_lib + 1e-8, (err_split, err_lib)
    assert torch.isfinite(C).all() and C.shape == (M, 256)
    if M > n:      # tail rows too
        ref_t = A[-64:].double() @ Wd + bias.double()
        assert ((C[-64:].double() - ref_t).abs() / ((A[-64:].double().abs() @ Wd.abs()) + bias.double().abs())).max().item() < 4e-7


def test_rejects_other_shapes(lib):
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    W = torch.zeros(128, 256, device="cuda")
    assert lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 128, 256, 0, _stream()) != 0
    A = torch.zeros(8, 256, device="cuda"); C = torch.zeros(8, 256, device="cuda")
    assert lib.ag_split_gemm(A.data_ptr(), planes.data_ptr(), None, C.data_ptr(), 8, 256, 128, _stream()) != 0


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7])
def test_scheduling_variants_are_the_same_arithmetic(lib, variant):
    """The A/B variants (prefetch placement, wave arrangement) order the same six MFMAs per product identically."""
    g = torch.Generator(device="cuda").manual_seed(7)
    A = torch.randn(4097, 256, device="cuda", generator=g)
    W = torch.randn(256, 256, device="cuda", generator=g) / 16.0
    try:
        assert lib.ag_debug_split_gemm_variant(0) == 0
        ref = _gemm(lib, A, W, False)
        assert lib.ag_debug_split_gemm_variant(variant) == 0
        got = _gemm(lib, A, W, False)
    finally:
        lib.ag_debug_split_gemm_variant(-1)
    assert torch.equal(ref, got)
    exact = A.double() @ W.double().t()
    assert ((got.double() - exact).abs() / (A.double().abs() @ W.double().abs().t())).max().item() < 4e-7
