"""`mixed_precision: true` (the reference's torch.cuda.amp switch, lib/agent/a2c_base.py:236-237,566,582): the one-bf16-MFMA-per-product
twins of the matrix-core kernels (suffix _bf16; csrc/split_common.hpp AG_SPLIT_PLANES = 1).  Operands are rounded to bf16 (round to
nearest even), products are exact, accumulation is float32 - so against a float64 product of the bf16-ROUNDED operands the twins must
be float32-accurate, and against the unrounded float64 product within 2^-7 sum |a||b| (both operands carry 2^-8)."""
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


def _r(x):
    """what the kernels' leading plane holds: bf16 round-to-nearest-even, as float64"""
    return x.bfloat16().double()


@pytest.mark.parametrize("M", [4096, 1000, 196608])
def test_plain_product_is_the_product_of_the_rounded_operands(lib, M):
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(M)
    A = torch.randn(M, 256, device="cuda", generator=g)
    W = torch.randn(256, 256, device="cuda", generator=g) / 16.0
    b = 0.1 * torch.randn(256, device="cuda", generator=g)
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W.data_ptr(), planes.data_ptr(), 256, 256, 0, _stream()), "prepare")
    C = torch.full((M, 256), float("nan"), device="cuda")
    N.check(lib.ag_split_gemm_bf16(A.data_ptr(), planes.data_ptr(), b.data_ptr(), C.data_ptr(), M, 256, 256, _stream()), "bf16")
    scale = A.double().abs() @ W.double().abs().t() + 1e-30
    exact_rounded = _r(A) @ _r(W).t() + b.double()
    assert ((C.double() - exact_rounded).abs() / scale).max().item() < 4e-7          # float32 accumulation of exact products
    full = A.double() @ W.double().t() + b.double()
    err = (C.double() - full).abs() / scale
    assert err.max().item() < 2.0 ** -7 and err.mean().item() < 2.0 ** -10, (err.max().item(), err.mean().item())
    # ... and it really is the cheaper kernel's result, not the float32-accurate one
    C3 = torch.empty_like(C)
    N.check(lib.ag_split_gemm(A.data_ptr(), planes.data_ptr(), b.data_ptr(), C3.data_ptr(), M, 256, 256, _stream()), "f32")
    assert ((C3.double() - full).abs() / scale).max().item() < 4e-7 < err.max().item()


@pytest.mark.parametrize("M,D", [(4096, 18), (196608, 18), (2048, 16)])
def test_update_kernels_bf16_twins_against_float64(lib, M, D):
    """The three launches of the optimizer step in their bf16 form against a float64 evaluation of the same two layers
    (forward -> dz2 is given; weight gradient; dX + first-layer backward), error relative to the products' magnitudes."""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(77 + M + D)
    f = dict(device="cuda", dtype=torch.float32)
    x = (2.0 * torch.randn(M, D, generator=g, **f)).clamp_(-5.0, 5.0)
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.3 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    dz2 = torch.randn(M, 256, generator=g, **f) * torch.rand(M, 1, generator=g, **f)
    image = torch.empty(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    bwd = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_input_prepare_pair(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), image.data_ptr(), bwd.data_ptr(),
                                                 _stream()), "prepare_pair")
    z1 = x.double() @ W1.double().t() + b1.double()
    h1 = torch.where(z1 > 0, z1, torch.expm1(z1))
    zabs = x.double().abs() @ W1.double().abs().t() + b1.double().abs()
    # weight gradient
    S = lib.ag_split_wgrad_input_slices(M)
    parts = torch.full((S, 256, 256), float("nan"), device="cuda")
    N.check(lib.ag_split_wgrad_input_bf16(dz2.data_ptr(), x.data_ptr(), image.data_ptr(), parts.data_ptr(), M, 256, 256, D, S, _stream()),
            "wgrad bf16")
    dw2 = parts.sum(0, dtype=torch.float64)
    ref = dz2.double().t() @ h1
    scale = dz2.double().abs().t() @ (h1.abs() + zabs) + 1e-30
    err = (dw2 - ref).abs() / scale
    assert torch.isfinite(parts).all() and err.max().item() < 2.0 ** -7 and err.mean().item() < 2.0 ** -11, (err.max().item(), err.mean().item())
    # dX + first-layer backward
    if lib.ag_split_gemm_input_wgrad_recompute_supported(D):
        tiles = (M + 255) // 256
        dw1 = torch.full((tiles, 256, D), float("nan"), device="cuda")
        db1 = torch.full((tiles, 256), float("nan"), device="cuda")
        N.check(lib.ag_split_gemm_input_wgrad_recompute_bf16(dz2.data_ptr(), bwd.data_ptr(), image.data_ptr(), x.data_ptr(),
                                                             dw1.data_ptr(), db1.data_ptr(), M, 256, 256, D, 0, _stream()), "dx bf16")
        dh1 = dz2.double() @ W2.double()
        dz1 = dh1 * torch.where(z1 > 0, torch.ones_like(z1), torch.exp(z1))
        habs = dz2.double().abs() @ W2.double().abs()
        sw = habs.t() @ x.double().abs() + 1e-30
        ew = (dw1.sum(0, dtype=torch.float64) - dz1.t() @ x.double()).abs() / sw
        eb = (db1.sum(0, dtype=torch.float64) - dz1.sum(0)).abs() / (habs.sum(0) + 1e-30)
        # three roundings in a row (dz2 and W2; z1's operands inside ELU'; dz1 and x): 3 x 2^-7 is the worst case
        assert ew.max().item() < 3 * 2.0 ** -7 and eb.max().item() < 3 * 2.0 ** -7, (ew.max().item(), eb.max().item())
        assert ew.mean().item() < 2.0 ** -9


@pytest.mark.parametrize("M", [65536, 1024])
def test_policy_forward_bf16_twin(lib, M):
    from airgym_amd import _native as N
    D, A1 = 18, 5
    g = torch.Generator(device="cuda").manual_seed(5 + M)
    f = dict(device="cuda", dtype=torch.float32)
    obs = torch.randn(M, D, generator=g, **f)
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.1 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    b2 = 0.1 * torch.randn(256, generator=g, **f)
    Wh = torch.randn(A1, 256, generator=g, **f) / 16.0
    bh = 0.1 * torch.randn(A1, generator=g, **f)
    image = torch.empty(lib.ag_mlp_chain_image_bytes(D), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_mlp_chain_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), Wh.data_ptr(), A1, image.data_ptr(), _stream()),
            "chain prepare")
    heads = {}
    for name in ("ag_mlp_chain_forward", "ag_mlp_chain_forward_bf16"):
        out = torch.full((M, A1), float("nan"), device="cuda")
        N.check(getattr(lib, name)(obs.data_ptr(), None, None, 0.0, 5.0, image.data_ptr(), b2.data_ptr(), bh.data_ptr(), out.data_ptr(),
                                   None, None, None, M, D, A1, _stream()), name)
        heads[name] = out.double()
    elu = lambda z: torch.where(z > 0, z, torch.expm1(z))
    h1 = elu(obs.double() @ W1.double().t() + b1.double())
    h2 = elu(h1 @ W2.double().t() + b2.double())
    ref = h2 @ Wh.double().t() + bh.double()
    scale = h2.abs() @ Wh.double().abs().t() + 1.0
    e3 = ((heads["ag_mlp_chain_forward"] - ref).abs() / scale).max().item()
    e1 = ((heads["ag_mlp_chain_forward_bf16"] - ref).abs() / scale)
    assert e3 < 2e-5 and e3 < e1.max().item() < 3 * 2.0 ** -7 and e1.mean().item() < 2.0 ** -8, (e3, e1.max().item(), e1.mean().item())


def _agent(mixed, seed=0, **extra):
    import os
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent

    class Args:
        envs = 4096; minibatches = 4; graph = 0; task = "hovering"; ctl = "rate"
    params = bench.build_params(Args, 1)
    params["config"]["bounds_loss_coef"] = 1e-4
    params["config"]["mixed_precision"] = mixed
    params["config"].update(extra)
    params["seed"] = seed
    return A2CAgent("mp", params)


def test_agent_with_mixed_precision_runs_the_bf16_twins_and_its_gradient_is_the_float32_gradient_to_bf16_accuracy():
    agent = _agent(True)
    fs = agent._fused_step
    assert agent.mixed_precision and fs is not None and fs.bf16 and fs.covers_all_products
    agent.init_tensors()
    assert agent._fused_rollout.bf16 and agent._fused_rollout.chain is not None
    agent.obs = agent.env_reset()
    agent.epoch_num = 1
    agent.train_epoch()
    batch = agent.play_steps()
    agent.model.train()
    agent.curr_frames = batch.pop("played_frames")
    agent.prepare_dataset(batch)
    agent.model.running_mean_std.eval()
    agent.model.update_stats = False
    mb = agent.dataset[1]
    mu0, sig0 = mb["mu"].clone(), mb["sigma"].clone()
    fs.begin_epoch()
    fs.step(mb)
    assert fs.last_launches["forward"][0].startswith("ag_split_gemm_input_loss_heads_bwd")
    g_bf16 = agent.flat_grad.clone()
    mb["mu"].copy_(mu0); mb["sigma"].copy_(sig0)
    agent._loss_and_backward(mb)                       # autograd, float32 library GEMMs
    g_f32 = agent.flat_grad.clone()
    rel = ((g_bf16 - g_f32)[:-1].norm() / g_f32[:-1].norm()).item()
    assert 1e-5 < rel < 3e-2, rel                      # bf16-level agreement - and NOT float32-level: the twins really ran
    agent.vec_env.env.hip.close()


def test_configurations_the_twins_do_not_cover_are_refused():
    import pytest as _pt
    with _pt.raises(NotImplementedError, match="mixed_precision"):
        _agent(True, recompute_h1=False)               # part of the products would run elsewhere
    with _pt.raises(NotImplementedError, match="mixed_precision"):
        _agent(True, use_split_gemm=False)


def test_mixed_precision_learns():
    """Same bar as the float32 path's learning-speed test, two seeds: the meter reaches 2 000 within 120 epochs."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.learning_curves import run
    best = []
    for seed in (0, 1):
        out = run(f"bf16 seed {seed}", 65536, 8, 120, 10, seed=seed, extra={"mixed_precision": True})
        assert all(c["kl"] == c["kl"] and c["c_loss"] == c["c_loss"] for c in out["curve"]), "NaN in the losses"
        best.append(max((c["reward"] or 0.0) for c in out["curve"]))
    print("mixed precision: best meter by epoch 120:", best)
    assert all(b >= 2000.0 for b in best), best
