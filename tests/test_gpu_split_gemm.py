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


@pytest.mark.parametrize("M", [1, 127, 129, 4097, 65536, 196608])
@pytest.mark.parametrize("A1", [5, 6])
def test_fused_elu_heads_epilogue(lib, M, A1):
    """ag_split_gemm_elu_heads = ag_split_gemm (no bias) + ag_elu_heads(zbias) in one launch: Z bit-identical to the plain
    kernel's (same accumulators), heads within float32 summation noise of a float64 evaluation and of the two-launch path."""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(7 * M + A1)
    X = torch.randn(M, 256, device="cuda", generator=g)
    W = torch.randn(256, 256, device="cuda", generator=g) / 16.0
    b = torch.randn(256, device="cuda", generator=g) * 0.3
    Wh = torch.randn(A1, 256, device="cuda", generator=g) / 16.0
    bh = torch.randn(A1, device="cuda", generator=g)
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, 0, _stream()), "prepare")
    Z = torch.full((M, 256), float("nan"), device="cuda")
    H = torch.full((M, A1), float("nan"), device="cuda")
    N.check(lib.ag_split_gemm_elu_heads(X.data_ptr(), planes.data_ptr(), b.data_ptr(), Wh.data_ptr(), bh.data_ptr(),
                                        Z.data_ptr(), H.data_ptr(), M, 256, 256, A1, _stream()), "ag_split_gemm_elu_heads")
    Z2 = torch.empty(M, 256, device="cuda")
    N.check(lib.ag_split_gemm(X.data_ptr(), planes.data_ptr(), None, Z2.data_ptr(), M, 256, 256, _stream()), "ag_split_gemm")
    assert torch.equal(Z, Z2)                                   # bias-free pre-activation, untouched by the head epilogue
    H2 = torch.empty(M, A1, device="cuda")
    N.check(lib.ag_elu_heads(Z2.data_ptr(), Wh.data_ptr(), bh.data_ptr(), H2.data_ptr(), M, 256, A1, 0, b.data_ptr(), _stream()),
            "ag_elu_heads")
    assert torch.isfinite(H).all()
    n = min(M, 8192)
    for sl in (slice(0, n), slice(M - min(M, 300), M)):
        e = torch.nn.functional.elu(Z[sl].double() + b.double())
        ref = e @ Wh.double().t() + bh.double()
        scale = e.abs() @ Wh.double().abs().t() + bh.double().abs()
        err = ((H[sl].double() - ref).abs() / scale).max().item()
        err2 = ((H2[sl].double() - ref).abs() / scale).max().item()
        assert err < 4e-7, (err, err2)                           # hardware exp2 in the ELU: ~1e-7 relative on e
        assert (H[sl] - H2[sl]).abs().max().item() <= 2e-6 * scale.max().item()


def test_fused_elu_heads_rejects_bad_arguments(lib):
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    X = torch.zeros(4, 256, device="cuda"); Z = torch.zeros(4, 256, device="cuda"); H = torch.zeros(4, 7, device="cuda")
    b = torch.zeros(256, device="cuda"); Wh = torch.zeros(7, 256, device="cuda"); bh = torch.zeros(7, device="cuda")
    args = (X.data_ptr(), planes.data_ptr(), b.data_ptr(), Wh.data_ptr(), bh.data_ptr(), Z.data_ptr(), H.data_ptr())
    assert lib.ag_split_gemm_elu_heads(*args, 4, 256, 256, 7, _stream()) != 0       # A1 outside {5, 6}
    assert lib.ag_split_gemm_elu_heads(*args, 4, 128, 256, 5, _stream()) != 0
    assert lib.ag_split_gemm_elu_heads(*args[:2], None, *args[3:], 4, 256, 256, 5, _stream()) != 0
    assert lib.ag_split_gemm_elu_heads(*args, 0, 256, 256, 5, _stream()) != 0


@pytest.mark.parametrize("M", [1, 127, 129, 257, 4097, 196608])
@pytest.mark.parametrize("D", [16, 18, 20, 48])
def test_fused_first_layer_backward_epilogue(lib, M, D):
    """ag_split_gemm_input_wgrad = ag_split_gemm (dX of layer 2) + ag_elu_bwd_input_wgrad (ELU' + dW1 / db1 partials) in one
    launch, with dh1 never written: summed partials against a float64 evaluation and against the two-launch path."""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(11 * M + D)
    dZ = torch.randn(M, 256, device="cuda", generator=g) * 0.1
    W = torch.randn(256, 256, device="cuda", generator=g) / 16.0
    h1 = torch.nn.functional.elu(torch.randn(M, 256, device="cuda", generator=g))
    x = torch.randn(M, D, device="cuda", generator=g)
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, 1, _stream()), "prepare")
    rows = lib.ag_split_gemm_input_wgrad_rows()
    tiles = (M + rows - 1) // rows
    dwp = torch.full((tiles, 256, D), float("nan"), device="cuda")
    dbp = torch.full((tiles, 256), float("nan"), device="cuda")
    N.check(lib.ag_split_gemm_input_wgrad(dZ.data_ptr(), planes.data_ptr(), h1.data_ptr(), x.data_ptr(), dwp.data_ptr(),
                                          dbp.data_ptr(), M, 256, 256, D, _stream()), "ag_split_gemm_input_wgrad")
    assert torch.isfinite(dwp).all() and torch.isfinite(dbp).all()
    dW, db = dwp.double().sum(0), dbp.double().sum(0)
    # float64 reference of the same chain
    dh = dZ.double() @ W.double()
    dz1 = dh * torch.where(h1 > 0, torch.ones_like(h1), h1 + 1.0).double()
    dW_ref, db_ref = dz1.t() @ x.double(), dz1.sum(0)
    sW = (dZ.double().abs() @ W.double().abs()).t() @ x.double().abs() + 1e-30       # sum |terms|: the natural error scale
    sb = (dZ.double().abs() @ W.double().abs()).sum(0) + 1e-30
    assert ((dW - dW_ref).abs() / sW).max().item() < 4e-7
    assert ((db - db_ref).abs() / sb).max().item() < 4e-7
    # the two-launch path it replaces (Hovering's widths; Tracking's 48 has no stand-alone first-layer kernel)
    r2 = lib.ag_input_wgrad_rows(D)
    if r2 > 0:
        dh32 = torch.empty(M, 256, device="cuda")
        N.check(lib.ag_split_gemm(dZ.data_ptr(), planes.data_ptr(), None, dh32.data_ptr(), M, 256, 256, _stream()), "ag_split_gemm")
        b2 = (M + r2 - 1) // r2
        dwp2, dbp2 = torch.empty(b2, 256, D, device="cuda"), torch.empty(b2, 256, device="cuda")
        N.check(lib.ag_elu_bwd_input_wgrad(dh32.data_ptr(), h1.data_ptr(), x.data_ptr(), dwp2.data_ptr(), dbp2.data_ptr(), M, 256, D,
                                           _stream()), "ag_elu_bwd_input_wgrad")
        assert ((dwp2.double().sum(0) - dW).abs() / sW).max().item() < 4e-7
        assert ((dbp2.double().sum(0) - db).abs() / sb).max().item() < 4e-7
    else:
        assert D == 48 and lib.ag_split_gemm_input_wgrad_supported(D) == 1 and lib.ag_split_gemm_input_wgrad_supported(24) == 0
    # per-tile partials: each tile only sees its own rows (tile t of the tail is partial)
    t = tiles - 1
    sl = slice(t * rows, M)
    assert torch.allclose(dbp[t].double(), dz1[sl].sum(0), rtol=0, atol=4e-7 * sb.max().item() + 1e-12)


def test_fused_first_layer_backward_rejects_bad_arguments(lib):
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    dZ = torch.zeros(4, 256, device="cuda"); h1 = torch.zeros(4, 256, device="cuda"); x = torch.zeros(4, 18, device="cuda")
    dw = torch.zeros(1, 256, 18, device="cuda"); db = torch.zeros(1, 256, device="cuda")
    a = (dZ.data_ptr(), planes.data_ptr(), h1.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr())
    assert lib.ag_split_gemm_input_wgrad(*a, 4, 256, 256, 17, _stream()) != 0        # D outside {16, 18, 20}
    assert lib.ag_split_gemm_input_wgrad(*a, 4, 256, 128, 18, _stream()) != 0
    assert lib.ag_split_gemm_input_wgrad(*a[:2], None, *a[3:], 4, 256, 256, 18, _stream()) != 0
    assert lib.ag_split_gemm_input_wgrad(*a, 0, 256, 256, 18, _stream()) != 0


def test_prepare_pair_equals_two_prepares(lib):
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(5)
    W = torch.randn(256, 256, device="cuda", generator=g)
    nb = lib.ag_split_gemm_plane_bytes()
    a, b, c, d = (torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(4))
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), a.data_ptr(), 256, 256, 0, _stream()), "prepare")
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), b.data_ptr(), 256, 256, 1, _stream()), "prepare")
    N.check(lib.ag_split_gemm_prepare_pair(W.data_ptr(), c.data_ptr(), d.data_ptr(), 256, 256, _stream()), "prepare_pair")
    assert torch.equal(a, c) and torch.equal(b, d) and not torch.equal(a, b)
