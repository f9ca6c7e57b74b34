"""Round 5: the first layer's activations h1 = ELU(x W1^T + b1) of the [D -> 256 -> 256] trunk are no longer stored by the update's
forward launch; the two backward kernels that consumed them recompute them on the matrix cores (autograd keeps h1 as a saved
tensor: lib/network/mlp.py:36-39 under lib/agent/a2c_continuous.py:299-369).

  ag_split_wgrad_input                 dW2 = dZ^T h1 with the X operand produced from x (csrc/split_wgrad.hip, split_wgrad_fin_kernel)
  ag_split_gemm_input_wgrad_recompute  the dX GEMM + first-layer backward with ELU'(h1) recomputed (csrc/split_gemm.hip, RC epilogue)
  ag_split_gemm_input_loss_heads_bwd   with h1_dev = NULL: the same launch without the h1 stores

Each against float64 and against the stored-h1 kernels they replace, through the C ABI."""
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


def _first_layer(lib, M, D, seed):
    """x [M, D] (already 'normalised': clamped to +-5), W1, b1, W2 and the image of ag_split_gemm_input_prepare"""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(seed)
    f = dict(device="cuda", dtype=torch.float32)
    x = (2.0 * torch.randn(M, D, generator=g, **f)).clamp_(-5.0, 5.0)
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.3 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    image = torch.empty(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_input_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), image.data_ptr(), _stream()), "in_prepare")
    z = x.double() @ W1.double().t() + b1.double()
    h64 = torch.where(z > 0, z, torch.expm1(z))
    # magnitude of the first layer's product (what the float32-level error of h1 is relative to)
    zabs = x.double().abs() @ W1.double().abs().t() + b1.double().abs()
    return g, x, W1, b1, W2, image, z, h64, zabs


@pytest.mark.parametrize("M,D", [(196608, 18), (4096, 18), (2048, 16), (1536, 20), (32, 18), (64, 18), (8224, 18)])
def test_weight_gradient_with_produced_x_operand_matches_float64(lib, M, D):
    from airgym_amd import _native as N
    assert lib.ag_split_wgrad_input_supported(D)
    g, x, W1, b1, W2, image, z, h64, zabs = _first_layer(lib, M, D, 7 + M + D)
    dz = torch.randn(M, 256, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g)
    S = lib.ag_split_wgrad_input_slices(M)
    assert S == min(M // 32, torch.cuda.get_device_properties(0).multi_processor_count)
    parts = torch.full((S, 256, 256), float("nan"), device="cuda")
    N.check(lib.ag_split_wgrad_input(dz.data_ptr(), x.data_ptr(), image.data_ptr(), parts.data_ptr(), M, 256, 256, D, S, _stream()),
            "ag_split_wgrad_input")
    assert torch.isfinite(parts).all()
    dw = parts.sum(0, dtype=torch.float64)
    ref = dz.double().t() @ h64
    # error budget: the split products are float32-accurate (4e-7 of sum |dz||h1|, as ag_split_wgrad) and h1 itself carries a
    # float32-level error of its K = D + 1 product and of v_exp_f32 (relative to sum |x||w| + |b|)
    scale = dz.double().abs().t() @ (h64.abs() + zabs) + 1e-30
    err = ((dw - ref).abs() / scale).max().item()
    assert err < 6e-7, (M, D, err)
    # ... and against the kernel it replaces on stored activations (float32 h1 from the float64 reference)
    h32 = h64.float()
    S0 = lib.ag_split_wgrad_slices(M)
    parts0 = torch.empty(S0, 256, 256, device="cuda")
    N.check(lib.ag_split_wgrad(dz.data_ptr(), h32.data_ptr(), parts0.data_ptr(), M, 256, 256, S0, _stream()), "ag_split_wgrad")
    dw0 = parts0.sum(0, dtype=torch.float64)
    assert ((dw - dw0).abs() / scale).max().item() < 1e-6


def test_weight_gradient_with_produced_x_puts_rows_and_columns_where_they_belong(lib):
    """one-hot dZ: dW2[co, :] = h1[rows[co], :] - a gather of recomputed activations; every (row slot, column) of the permuted
    LDS image and of the register-resident B fragments lands where it belongs, and the values are float32-accurate h1."""
    from airgym_amd import _native as N
    M, D = 4096, 18
    g, x, W1, b1, W2, image, z, h64, zabs = _first_layer(lib, M, D, 11)
    rows = torch.randperm(M, device="cuda", generator=g)[:256]
    dz = torch.zeros(M, 256, device="cuda")
    dz[rows, torch.arange(256, device="cuda")] = 1.0
    S = lib.ag_split_wgrad_input_slices(M)
    parts = torch.empty(S, 256, 256, device="cuda")
    N.check(lib.ag_split_wgrad_input(dz.data_ptr(), x.data_ptr(), image.data_ptr(), parts.data_ptr(), M, 256, 256, D, S, _stream()),
            "ag_split_wgrad_input")
    dw = parts.sum(0)
    assert ((dw.double() - h64[rows]).abs() / zabs[rows]).max().item() < 4e-7
    # deterministic
    parts2 = torch.empty_like(parts)
    N.check(lib.ag_split_wgrad_input(dz.data_ptr(), x.data_ptr(), image.data_ptr(), parts2.data_ptr(), M, 256, 256, D, S, _stream()),
            "ag_split_wgrad_input")
    assert torch.equal(parts, parts2)


def test_weight_gradient_with_produced_x_rejects_bad_arguments(lib):
    from airgym_amd import _native as N
    p = ctypes.c_void_p(256)
    assert not lib.ag_split_wgrad_input_supported(48) and not lib.ag_split_wgrad_input_supported(17)
    assert lib.ag_split_wgrad_input_slices(0) == 0 and lib.ag_split_wgrad_input_slices(64) == 2
    assert lib.ag_split_wgrad_input(None, p, p, p, 64, 256, 256, 18, 2, None) == -1
    assert lib.ag_split_wgrad_input(p, p, p, p, 64, 128, 256, 18, 2, None) == N.AG_ERR_UNSUPPORTED
    assert lib.ag_split_wgrad_input(p, p, p, p, 64, 256, 256, 48, 2, None) == N.AG_ERR_UNSUPPORTED
    assert lib.ag_split_wgrad_input(p, p, p, p, 48, 256, 256, 18, 1, None) == N.AG_ERR_UNSUPPORTED      # M % 32
    assert lib.ag_split_gemm_input_wgrad_recompute(p, p, p, p, p, p, 256, 256, 256, 20, 0, None) == N.AG_ERR_UNSUPPORTED
    assert lib.ag_split_gemm_input_wgrad_recompute(p, p, p, p, p, p, 256, 256, 256, 18, 64, None) == -1      # tile_rows: 0 / 128 / 256
    assert lib.ag_split_gemm_input_wgrad_recompute(p, p, None, p, p, p, 256, 256, 256, 18, 0, None) == -1


@pytest.mark.parametrize("M,D,tile_rows", [(256, 18, 0), (4096, 18, 0), (2048, 16, 0), (1000, 18, 0), (196608, 18, 0),
                                           (4096, 18, 128), (32768, 18, 128), (1000, 16, 128)])
def test_first_layer_backward_with_recomputed_h1(lib, M, D, tile_rows):
    """dW1 / db1 partials of the recomputing epilogue == the stored-h1 epilogue (ag_split_gemm_input_wgrad fed with a float32 h1)
    and == float64 autograd of the two layers."""
    from airgym_amd import _native as N
    if not lib.ag_split_gemm_input_wgrad_recompute_supported(D):
        pytest.skip("needs the 256-row tile build")
    g, x, W1, b1, W2, image, z, h64, zabs = _first_layer(lib, M, D, 23 + M + D)
    dz2 = torch.randn(M, 256, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g)
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W2.data_ptr(), planes.data_ptr(), 256, 256, 1, _stream()), "prepare (transposed)")
    rows = tile_rows or lib.ag_split_gemm_input_wgrad_rows()
    tiles = (M + rows - 1) // rows
    dw = torch.full((tiles, 256, D), float("nan"), device="cuda")
    db = torch.full((tiles, 256), float("nan"), device="cuda")
    N.check(lib.ag_split_gemm_input_wgrad_recompute(dz2.data_ptr(), planes.data_ptr(), image.data_ptr(), x.data_ptr(), dw.data_ptr(),
                                                    db.data_ptr(), M, 256, 256, D, tile_rows, _stream()), "recompute")
    h32 = h64.float()
    t0 = (M + 255) // 256
    dw0, db0 = torch.empty(t0, 256, D, device="cuda"), torch.empty(t0, 256, device="cuda")
    N.check(lib.ag_split_gemm_input_wgrad(dz2.data_ptr(), planes.data_ptr(), h32.data_ptr(), x.data_ptr(), dw0.data_ptr(),
                                          db0.data_ptr(), M, 256, 256, D, _stream()), "stored")
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all() and torch.isfinite(db).all()
    # float64 reference: dh1 = dz2 W2, dz1 = dh1 * ELU'(z1), dW1 = dz1^T x, db1 = sum dz1
    dh1 = dz2.double() @ W2.double()
    dz1 = dh1 * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    ref_w, ref_b = dz1.t() @ x.double(), dz1.sum(0)
    sw = (dz2.double().abs() @ W2.double().abs()).t() @ x.double().abs() + 1e-30
    sb = (dz2.double().abs() @ W2.double().abs()).sum(0) + 1e-30
    got_w, got_b = dw.sum(0, dtype=torch.float64), db.sum(0, dtype=torch.float64)
    assert ((got_w - ref_w).abs() / sw).max().item() < 1e-6, ((got_w - ref_w).abs() / sw).max().item()
    assert ((got_b - ref_b).abs() / sb).max().item() < 1e-6
    # and the stored-h1 kernel: same computation on activations that differ by float32 rounding
    assert ((got_w - dw0.sum(0, dtype=torch.float64)).abs() / sw).max().item() < 1e-6
    assert ((got_b - db0.sum(0, dtype=torch.float64)).abs() / sb).max().item() < 1e-6
    # per tile too (a tile's partial is a sum over its own rows only; 128-row tiles: pairs against the 256-row reference)
    dwt = dw.double() if rows == 256 else torch.nn.functional.pad(dw.double(), (0, 0, 0, 0, 0, dw.shape[0] % 2)).view(-1, 2, 256, D).sum(1)
    assert ((dwt - dw0.double()).abs().amax(0) / sw).max().item() < 1e-6


@pytest.mark.parametrize("M,D,normalize", [(4096, 18, True), (2048, 16, False), (196608, 18, True)])
def test_forward_launch_with_128_row_tiles_equals_256_row_tiles(lib, M, D, normalize):
    """`tile_rows = 128` (4-wave workgroups, for minibatches with fewer 256-row tiles than CUs) changes the tiling, not the
    function: dz bit-identical (a row's arithmetic does not depend on its tile), per-tile partials sum pairwise to the 256-row
    launch's within float32 summation order."""
    from airgym_amd import _native as N
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_fused_input_layer import _loss_args
    A = 4
    g = torch.Generator(device="cuda").manual_seed(900 + M + D)
    f = dict(device="cuda", dtype=torch.float32)
    obs = 3.0 * torch.randn(M, D, generator=g, **f)
    mean = torch.randn(D, generator=g, device="cuda", dtype=torch.float64)
    var = torch.rand(D, generator=g, device="cuda", dtype=torch.float64) + 0.05
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.1 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    b2 = 0.1 * torch.randn(256, generator=g, **f)
    Wh = torch.randn(A + 1, 256, generator=g, **f) / 16.0
    bh = 0.1 * torch.randn(A + 1, generator=g, **f)
    image = torch.empty(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_input_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), image.data_ptr(), _stream()), "in_prepare")
    res = []
    for rows in (256, 128):
        gl = torch.Generator(device="cuda").manual_seed(5)
        L, _, out = _loss_args(N, lib, M, A, gl, M // rows)
        L.tile_rows = rows
        xn = torch.full((M, D), 7.0, **f) if normalize else None
        dz = torch.full((M, 256), 7.0, **f)
        inp = N.AgInputLayerArgs()
        inp.struct_size, inp.D = ctypes.sizeof(N.AgInputLayerArgs), D
        inp.obs_dev = obs.data_ptr()
        inp.mean_dev = mean.data_ptr() if normalize else None
        inp.var_dev = var.data_ptr() if normalize else None
        inp.xn_dev = xn.data_ptr() if normalize else None
        inp.h1_dev = None
        inp.eps, inp.clip = 1e-5, 5.0
        N.check(lib.ag_split_gemm_input_loss_heads_bwd(ctypes.byref(inp), image.data_ptr(), b2.data_ptr(), Wh.data_ptr(), bh.data_ptr(),
                                                       dz.data_ptr(), ctypes.byref(L), M, 256, 256, A + 1, _stream()), "fused")
        torch.cuda.synchronize()
        res.append((dz, xn, out))
    (dz_a, xn_a, out_a), (dz_b, xn_b, out_b) = res
    assert torch.equal(dz_a, dz_b) and torch.equal(out_a["new_mu"], out_b["new_mu"]) and torch.equal(out_a["heads"], out_b["heads"])
    if normalize:
        assert torch.equal(xn_a, xn_b)
    for k in ("loss_partials", "dwh_partials", "db_partials"):
        a, b = out_a[k].double(), out_b[k].double()
        b2s = b.view(a.shape[0], 2, *b.shape[1:]).sum(1)
        tol = 2e-5 * a.abs().amax(0, keepdim=True).clamp_min(1e-6)
        assert ((a - b2s).abs() <= tol).all(), k
    assert lib.ag_split_gemm_pick_tile_rows(196608) == 256 and lib.ag_split_gemm_pick_tile_rows(32768) == 128
    assert lib.ag_split_gemm_pick_tile_rows(32768 + 64) == 256          # not a multiple of 128


@pytest.mark.parametrize("M,D,normalize", [(4096, 18, True), (2048, 16, False), (196608, 18, True)])
def test_forward_launch_without_the_h1_store_is_the_same_launch(lib, M, D, normalize):
    """h1_dev = NULL removes the stores and nothing else: dz, every partial, xn and the written-back mu / sigma are bit-identical."""
    from airgym_amd import _native as N
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_fused_input_layer import _loss_args
    if not lib.ag_split_gemm_input_fwd_supported(D):
        pytest.skip("needs the 256-row tile build")
    A = 4
    g = torch.Generator(device="cuda").manual_seed(300 + M + D)
    f = dict(device="cuda", dtype=torch.float32)
    obs = 3.0 * torch.randn(M, D, generator=g, **f)
    mean = torch.randn(D, generator=g, device="cuda", dtype=torch.float64)
    var = torch.rand(D, generator=g, device="cuda", dtype=torch.float64) + 0.05
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.1 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    b2 = 0.1 * torch.randn(256, generator=g, **f)
    Wh = torch.randn(A + 1, 256, generator=g, **f) / 16.0
    bh = 0.1 * torch.randn(A + 1, generator=g, **f)
    image = torch.empty(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_input_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), image.data_ptr(), _stream()), "in_prepare")
    tiles = M // lib.ag_split_gemm_loss_rows()
    res = []
    for store in (True, False):
        gl = torch.Generator(device="cuda").manual_seed(5)
        L, _, out = _loss_args(N, lib, M, A, gl, tiles)
        xn = torch.full((M, D), 7.0, **f) if normalize else None
        h1 = torch.full((M, 256), 7.0, **f)
        dz = torch.full((M, 256), 7.0, **f)
        inp = N.AgInputLayerArgs()
        inp.struct_size, inp.D = ctypes.sizeof(N.AgInputLayerArgs), D
        inp.obs_dev = obs.data_ptr()
        inp.mean_dev = mean.data_ptr() if normalize else None
        inp.var_dev = var.data_ptr() if normalize else None
        inp.xn_dev = xn.data_ptr() if normalize else None
        inp.h1_dev = h1.data_ptr() if store else None
        inp.eps, inp.clip = 1e-5, 5.0
        N.check(lib.ag_split_gemm_input_loss_heads_bwd(ctypes.byref(inp), image.data_ptr(), b2.data_ptr(), Wh.data_ptr(), bh.data_ptr(),
                                                       dz.data_ptr(), ctypes.byref(L), M, 256, 256, A + 1, _stream()), "fused")
        torch.cuda.synchronize()
        res.append((dz, xn, h1, out))
    (dz_a, xn_a, h1_a, out_a), (dz_b, xn_b, h1_b, out_b) = res
    assert torch.equal(dz_a, dz_b)
    if normalize:
        assert torch.equal(xn_a, xn_b)
    assert (h1_b == 7.0).all() and not (h1_a == 7.0).all()          # the second launch never touched the buffer
    for k in out_a:
        assert torch.equal(out_a[k], out_b[k]), k


@pytest.mark.parametrize("recompute", [True, False])
def test_update_step_with_and_without_stored_h1(recompute):
    """The hand-scheduled minibatch step with `recompute_h1` on / off against autograd (every gradient to 2e-5 of the largest), and
    the switch really selects the path."""
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
    params["config"]["recompute_h1"] = recompute
    agent = A2CAgent("t", params)
    fs = agent._fused_step
    assert fs is not None and fs.fuse_gemm_input and fs.recompute_h1 == recompute
    agent.init_tensors()
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
    fs.h[0].fill_(123.0)
    fs.step(mb)
    g_fused = agent.flat_grad.clone()
    assert bool((fs.h[0] == 123.0).all()) == recompute             # h1 is not written when it is recomputed
    mb["mu"].copy_(mu0); mb["sigma"].copy_(sig0)
    agent._loss_and_backward(mb)
    g_auto = agent.flat_grad.clone()
    scale = g_auto[:-1].abs().max()
    assert (g_fused - g_auto)[:-1].abs().max() <= 2e-5 * scale + 1e-9, ((g_fused - g_auto).abs().max(), scale)
    agent.vec_env.env.hip.close()


def test_weight_images_of_the_step_in_one_launch(lib):
    """ag_split_gemm_input_prepare_pair == ag_split_gemm_input_prepare + ag_split_gemm_prepare(transpose = 1), bit for bit."""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(9)
    f = dict(device="cuda", dtype=torch.float32)
    D = 18
    W1, b1, W2 = torch.randn(256, D, generator=g, **f), torch.randn(256, generator=g, **f), torch.randn(256, 256, generator=g, **f)
    img0 = torch.zeros(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    img1 = torch.ones_like(img0)
    pt0 = torch.zeros(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    pt1 = torch.ones_like(pt0)
    N.check(lib.ag_split_gemm_input_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), img0.data_ptr(), _stream()), "prepare")
    N.check(lib.ag_split_gemm_prepare(W2.data_ptr(), pt0.data_ptr(), 256, 256, 1, _stream()), "prepare_t")
    N.check(lib.ag_split_gemm_input_prepare_pair(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), img1.data_ptr(), pt1.data_ptr(),
                                                 _stream()), "prepare_pair")
    assert torch.equal(img0, img1) and torch.equal(pt0, pt1)
    assert lib.ag_split_gemm_input_prepare_pair(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), img1.data_ptr(), None, _stream()) == -1
