"""ag_split_wgrad (csrc/split_wgrad.hip): the 256 x 256 weight gradient dW = dZ^T X of the hidden layer (autograd of
lib/network/mlp.py:36-39) on the bf16 matrix cores at float32 accuracy - against float64, against the library's f32 GEMM, and
through the hand-scheduled PPO minibatch."""
import ctypes

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


def _wgrad(lib, dz, x, slices=None):
    from airgym_amd import _native as N
    M = dz.shape[0]
    S = slices or lib.ag_split_wgrad_slices(M)
    parts = torch.full((S, 256, 256), float("nan"), device="cuda")
    N.check(lib.ag_split_wgrad(dz.data_ptr(), x.data_ptr(), parts.data_ptr(), M, 256, 256, S, _stream()), "ag_split_wgrad")
    return parts


@pytest.mark.parametrize("M", [196608, 65536, 4096, 16, 1, 777, 4099, 33])
def test_matches_float64(lib, M):
    """max |dW - dW64| / sum_m |dz||x| at the f32 level, and never more than 3x the library f32 GEMM's own error; ragged row
    counts (M % 16 != 0), fewer chunks than CUs and slices without any chunk included."""
    g = torch.Generator(device="cuda").manual_seed(M)
    dz = torch.randn(M, 256, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g)
    x = torch.randn(M, 256, device="cuda", generator=g)
    x[:, ::7] *= 30.0                                      # columns of very different magnitude
    parts = _wgrad(lib, dz, x)
    assert torch.isfinite(parts).all()                     # every slice wrote its whole tile (zeros where it had no rows)
    dw = parts.sum(0, dtype=torch.float64)
    ref = dz.double().t() @ x.double()
    scale = dz.double().abs().t() @ x.double().abs() + 1e-30
    err = ((dw - ref).abs() / scale).max().item()
    lib_err = (((dz.t() @ x).double() - ref).abs() / scale).max().item()
    assert err < 4e-7, (M, err)
    assert err <= 3.0 * lib_err + 1e-9, (M, err, lib_err)


def test_exact_on_representable_products(lib):
    """The split is exact: with a one-hot dZ the gradient is a gather of rows of X (each product is x * 1, every partial sum
    has one non-zero term), reproduced bit for bit - rows and columns land where they belong (the LDS image permutes both)."""
    M = 2048
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, 256, device="cuda", generator=g)
    rows = torch.randperm(M, device="cuda", generator=g)[:256]          # output row co takes input row rows[co]
    dz = torch.zeros(M, 256, device="cuda")
    dz[rows, torch.arange(256, device="cuda")] = 1.0
    dw = _wgrad(lib, dz, x).sum(0)
    assert torch.equal(dw, x[rows])
    # and transposed roles: one-hot X picks columns of dZ
    dz2 = torch.randn(M, 256, device="cuda", generator=g)
    x2 = torch.zeros(M, 256, device="cuda")
    x2[rows, torch.arange(256, device="cuda")] = 1.0
    dw2 = _wgrad(lib, dz2, x2).sum(0)
    assert torch.equal(dw2, dz2[rows].t())


@pytest.mark.parametrize("slices", [1, 7, 256, 300])
def test_any_slice_count(lib, slices):
    M = 5000
    g = torch.Generator(device="cuda").manual_seed(slices)
    dz, x = torch.randn(M, 256, device="cuda", generator=g), torch.randn(M, 256, device="cuda", generator=g)
    dw = _wgrad(lib, dz, x, slices).sum(0, dtype=torch.float64)
    ref = dz.double().t() @ x.double()
    assert ((dw - ref).abs() / (dz.double().abs().t() @ x.double().abs())).max().item() < 4e-7


def test_deterministic_and_slice_partition(lib):
    """Two runs are bit-identical; slice s holds exactly the contribution of its own rows."""
    M = 64 * 48
    g = torch.Generator(device="cuda").manual_seed(11)
    dz, x = torch.randn(M, 256, device="cuda", generator=g), torch.randn(M, 256, device="cuda", generator=g)
    p1, p2 = _wgrad(lib, dz, x, 4), _wgrad(lib, dz, x, 4)
    assert torch.equal(p1, p2)
    rows = M // 4
    for s in range(4):
        ref = dz[s * rows:(s + 1) * rows].double().t() @ x[s * rows:(s + 1) * rows].double()
        sc = dz[s * rows:(s + 1) * rows].double().abs().t() @ x[s * rows:(s + 1) * rows].double().abs()
        assert ((p1[s].double() - ref).abs() / sc).max().item() < 4e-7, s


def test_update_step_uses_it_and_matches_library_path(lib):
    """The hand-scheduled minibatch with the weight gradient on the matrix cores produces the same flat gradient as with the
    library's split-K f32 GEMM (use_split_wgrad: false) to f32 accuracy, and no library GEMM is left on the update path."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent

    grads = {}
    for flag in (1, 0):
        class Args:
            envs = 4096; minibatches = 2; graph = 0; split_wgrad = flag
        torch.manual_seed(0)
        agent = A2CAgent("t", bench.build_params(Args, 1))
        agent.init_tensors()
        agent.obs = agent.env_reset()
        fs = agent._fused_step
        assert fs is not None and (fs.split_wgrad == {1}) == bool(flag)
        assert ("no library GEMM" in fs.gemm_description) == bool(flag)
        batch = agent.play_steps()
        agent.model.train()
        batch.pop("played_frames")
        agent.prepare_dataset(batch)
        agent.model.update_stats = False
        fs.step(agent.dataset[0])
        grads[flag] = agent.flat_grad.clone()
        agent.vec_env.env.hip.close()
    a, b = grads[1], grads[0]
    assert torch.isfinite(a).all()
    denom = b.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-5 * denom, ((a - b).abs().max().item(), denom)
