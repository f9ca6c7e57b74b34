"""ag_mlp_chain_forward (csrc/mlp_chain.hip): the actor-critic MLP [D -> 256 -> 256 -> (A + 1)] in one launch with the
activations in registers, against a float64 evaluation of the same network (ModelA2CContinuousLogStd.forward / MLP,
lib/model/a2c_continuous_logstd_model.py:80-193, lib/network/mlp.py:36-39) and against the two launches it replaces
(ag_mlp_input_layer + ag_split_gemm_elu_heads).  Float32-accurate: the error bound is that of an f32 FMA chain, as for
ag_split_gemm (tests/test_gpu_split_gemm.py)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _net(D, A1, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    return dict(W1=r(256, D) * (0.5 / D ** 0.5) * scale, b1=r(256) * 0.2, W2=r(256, 256) * 0.08 * scale, b2=r(256) * 0.2,
                Wh=r(A1, 256) * 0.05, bh=r(A1) * 0.1,
                mean=(r(D) * 0.3).double(), var=(torch.rand(D, device="cuda", generator=g) * 2 + 0.05).double())


def _reference64(obs, n, eps=1e-5, clip=5.0, normalize=True):
    x = obs.double()
    if normalize:
        # the normaliser itself is float32 arithmetic in both paths: (x - mean_f32) / sqrt(var_f32 + eps), clamp
        x = ((obs - n["mean"].float()) / torch.sqrt(n["var"].float() + eps)).clamp(-clip, clip).double()
    elu = torch.nn.functional.elu
    h1 = elu(x @ n["W1"].double().t() + n["b1"].double())
    h2 = elu(h1 @ n["W2"].double().t() + n["b2"].double())
    return x, h1, h2, h2 @ n["Wh"].double().t() + n["bh"].double()


@pytest.mark.parametrize("M,D,A1,normalize,store", [
    (65536, 18, 5, True, False),      # the rollout's shape (Hovering, CTBR): heads only
    (4096 + 77, 18, 5, True, True),   # ragged last tile, every optional output
    (1000, 18, 6, False, True),       # atti: 5 actions + value; no input normaliser
    (300, 48, 5, True, True),         # Tracking's 48 observations (four K steps of the first layer)
    (31, 16, 5, True, False),         # less than one wave
])
def test_chain_forward_matches_float64_and_the_two_launch_path(M, D, A1, normalize, store):
    from airgym_amd import _native as N
    lib = N.load()
    assert lib.ag_mlp_chain_supported(D, 256, A1) == 1
    n = _net(D, A1, seed=M + D)
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.randn(M, D, device="cuda", generator=g) * 1.5 + 0.2
    image = torch.empty(lib.ag_mlp_chain_image_bytes(D), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_mlp_chain_prepare(n["W1"].data_ptr(), n["b1"].data_ptr(), D, n["W2"].data_ptr(), n["Wh"].data_ptr(), A1,
                                     image.data_ptr(), _stream()), "ag_mlp_chain_prepare")
    heads = torch.full((M, A1), float("nan"), device="cuda")
    xn = torch.full((M, D), float("nan"), device="cuda") if (store and normalize) else None
    h1 = torch.full((M, 256), float("nan"), device="cuda") if store else None
    h2 = torch.full((M, 256), float("nan"), device="cuda") if store else None
    P = lambda t: t.data_ptr() if t is not None else None
    N.check(lib.ag_mlp_chain_forward(obs.data_ptr(), P(n["mean"]) if normalize else None, P(n["var"]) if normalize else None,
                                     1e-5, 5.0, image.data_ptr(), n["b2"].data_ptr(), n["bh"].data_ptr(), heads.data_ptr(),
                                     P(xn), P(h1), P(h2), M, D, A1, _stream()), "ag_mlp_chain_forward")
    torch.cuda.synchronize()
    x64, h1_64, h2_64, heads64 = _reference64(obs, n, normalize=normalize)
    assert torch.isfinite(heads).all()
    # error scale of an f32 evaluation: eps_f32 * sum |a||b| per product (here bounded through the activations' magnitudes)
    tol_h = 4e-6 * (1.0 + h2_64.abs().max().item())
    assert (heads.double() - heads64).abs().max().item() <= 4e-6 * (1.0 + heads64.abs().max().item()) + 2e-6
    if store:
        assert (h1.double() - h1_64).abs().max().item() <= 4e-6 * (1.0 + h1_64.abs().max().item())
        assert (h2.double() - h2_64).abs().max().item() <= tol_h
        if xn is not None:
            assert torch.equal(xn.double(), x64)
    # the two launches it replaces
    if D in (16, 18, 20, 48):
        planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
        N.check(lib.ag_split_gemm_prepare(n["W2"].data_ptr(), planes.data_ptr(), 256, 256, 0, _stream()), "prepare")
        xn2 = torch.empty(M, D, device="cuda") if normalize else None
        hh1, z2, heads2 = torch.empty(M, 256, device="cuda"), torch.empty(M, 256, device="cuda"), torch.empty(M, A1, device="cuda")
        N.check(lib.ag_mlp_input_layer(obs.data_ptr(), P(n["mean"]) if normalize else None, P(n["var"]) if normalize else None,
                                       n["W1"].data_ptr(), n["b1"].data_ptr(), P(xn2), hh1.data_ptr(), M, D, 256, 1e-5, 5.0,
                                       _stream()), "ag_mlp_input_layer")
        N.check(lib.ag_split_gemm_elu_heads(hh1.data_ptr(), planes.data_ptr(), n["b2"].data_ptr(), n["Wh"].data_ptr(),
                                            n["bh"].data_ptr(), z2.data_ptr(), heads2.data_ptr(), M, 256, 256, A1, _stream()),
                "ag_split_gemm_elu_heads")
        torch.cuda.synchronize()
        assert (heads - heads2).abs().max().item() <= 8e-6 * (1.0 + heads64.abs().max().item())
        # both are float32-class: neither is further from float64 than a few times the other
        e_chain = (heads.double() - heads64).abs().max().item()
        e_two = (heads2.double() - heads64).abs().max().item()
        assert e_chain <= 4.0 * e_two + 1e-6, (e_chain, e_two)


def test_chain_forward_is_exact_on_exactly_representable_products():
    """Identity-like weights: every product is exact in bf16 pieces, so the chain must reproduce the inputs bit for bit - a
    layout error (a permuted K order on one side only, a transposed tile) cannot hide behind rounding."""
    from airgym_amd import _native as N
    lib = N.load()
    M, D, A1 = 256, 18, 5
    W1 = torch.zeros(256, D, device="cuda")
    for k in range(D):
        W1[3 * k + 1, k] = 2.0 ** (k % 5)             # feature 3k+1 carries input k, scaled by a power of two
    b1 = torch.zeros(256, device="cuda")
    W2 = torch.zeros(256, 256, device="cuda")
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(3)).cuda()
    W2[torch.arange(256, device="cuda"), perm] = 1.0      # h2[j] = h1[perm[j]]
    b2 = torch.zeros(256, device="cuda")
    Wh = torch.zeros(A1, 256, device="cuda")
    inv = torch.argsort(perm)
    for a in range(A1):
        Wh[a, inv[3 * a + 1]] = 1.0                   # head a = h2[inv[3a+1]] = h1[3a+1] = 2^(a % 5) x[a]
    bh = torch.arange(A1, device="cuda", dtype=torch.float32)
    obs = torch.rand(M, D, device="cuda") + 0.25          # positive: ELU is the identity
    image = torch.empty(lib.ag_mlp_chain_image_bytes(D), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_mlp_chain_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), Wh.data_ptr(), A1, image.data_ptr(), _stream()), "prep")
    heads = torch.empty(M, A1, device="cuda")
    h1 = torch.empty(M, 256, device="cuda")
    h2 = torch.empty(M, 256, device="cuda")
    N.check(lib.ag_mlp_chain_forward(obs.data_ptr(), None, None, 0.0, 5.0, image.data_ptr(), b2.data_ptr(), bh.data_ptr(),
                                     heads.data_ptr(), None, h1.data_ptr(), h2.data_ptr(), M, D, A1, _stream()), "fwd")
    torch.cuda.synchronize()
    want_h1 = obs @ W1.t()
    assert torch.equal(h1, want_h1)
    assert torch.equal(h2, want_h1[:, perm])
    want = torch.stack([obs[:, a] * 2.0 ** (a % 5) + a for a in range(A1)], 1)
    assert torch.equal(heads, want)
