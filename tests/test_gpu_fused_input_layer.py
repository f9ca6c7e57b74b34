"""ag_split_gemm_input_loss_heads_bwd (airgym_amd/csrc/split_gemm.hip, FIN > 0): the first layer of the [D -> 256 -> 256] trunk
(lib/network/mlp.py:36-39 behind the input normaliser, lib/core/running_mean_std.py:78-79) formed inside the forward GEMM's
launch, on the matrix cores.  It must be the same function as ag_mlp_input_layer followed by ag_split_gemm_loss_heads_bwd: the
normalised inputs bit-identical, the first-layer activations float32-accurate against float64 (the product is a split MFMA
product instead of an FMA chain), and everything behind them equal to the two-launch path within float32 noise."""
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


def _loss_args(N, lib, M, A, g, tiles):
    f = dict(device="cuda", dtype=torch.float32)
    t = {
        "logstd": 0.1 * torch.randn(A, generator=g, **f),
        "actions": torch.randn(M, A, generator=g, **f).clamp_(-1.2, 1.2),
        "old_neglogp": 3.0 + torch.randn(M, generator=g, **f),
        "advantages": torch.randn(M, generator=g, **f),
        "returns": torch.randn(M, generator=g, **f),
        "old_values": torch.randn(M, generator=g, **f),
        "old_mu": 0.3 * torch.randn(M, A, generator=g, **f),
        "old_sigma": torch.rand(M, A, generator=g, **f) + 0.5,
    }
    out = {
        "new_mu": torch.zeros(M, A, **f), "new_sigma": torch.zeros(M, A, **f), "heads": torch.zeros(M, A + 1, **f),
        "loss_partials": torch.zeros(tiles, lib.ag_ppo_loss_num_sums(), **f),
        "dwh_partials": torch.zeros(tiles, A + 1, 256, **f), "db_partials": torch.zeros(tiles, 256, **f),
    }
    L = N.AgLossEpilogue()
    L.struct_size = ctypes.sizeof(N.AgLossEpilogue)
    L.logstd_dev, L.actions_dev, L.old_neglogp_dev = t["logstd"].data_ptr(), t["actions"].data_ptr(), t["old_neglogp"].data_ptr()
    L.advantages_dev, L.returns_dev, L.old_values_dev = t["advantages"].data_ptr(), t["returns"].data_ptr(), t["old_values"].data_ptr()
    L.old_mu_dev, L.old_sigma_dev = t["old_mu"].data_ptr(), t["old_sigma"].data_ptr()
    L.new_mu_dev, L.new_sigma_dev, L.heads_dev = out["new_mu"].data_ptr(), out["new_sigma"].data_ptr(), out["heads"].data_ptr()
    L.loss_partials_dev, L.dwh_partials_dev = out["loss_partials"].data_ptr(), out["dwh_partials"].data_ptr()
    L.db_partials_dev = out["db_partials"].data_ptr()
    L.e_clip, L.critic_coef, L.bounds_loss_coef, L.clip_value, L.bound_type = 0.2, 2.0, 1e-4, 1, 1
    return L, t, out


@pytest.mark.parametrize("M,D,normalize", [(256, 18, True), (4096, 18, True), (2048, 16, False), (1536, 20, True), (196608, 18, True)])
def test_first_layer_inside_the_forward_gemm_equals_the_two_launches(lib, M, D, normalize):
    from airgym_amd import _native as N
    if not lib.ag_split_gemm_input_fwd_supported(D):
        pytest.skip("needs the 256-row tile build")
    A = 4
    g = torch.Generator(device="cuda").manual_seed(100 + M + D)
    f = dict(device="cuda", dtype=torch.float32)
    obs = 3.0 * torch.randn(M, D, generator=g, **f)
    obs[::7, 0] = 40.0                       # clamped by the normaliser
    mean = torch.randn(D, generator=g, device="cuda", dtype=torch.float64)
    var = torch.rand(D, generator=g, device="cuda", dtype=torch.float64) + 0.05
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.1 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    b2 = 0.1 * torch.randn(256, generator=g, **f)
    Wh = torch.randn(A + 1, 256, generator=g, **f) / 16.0
    bh = 0.1 * torch.randn(A + 1, generator=g, **f)
    planes = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W2.data_ptr(), planes.data_ptr(), 256, 256, 0, _stream()), "prepare")
    tiles = M // lib.ag_split_gemm_loss_rows()

    # reference: the two launches
    gl = torch.Generator(device="cuda").manual_seed(5)
    L0, _, out0 = _loss_args(N, lib, M, A, gl, tiles)
    xn0 = torch.zeros(M, D, **f) if normalize else None
    h0 = torch.zeros(M, 256, **f)
    dz0 = torch.zeros(M, 256, **f)
    N.check(lib.ag_mlp_input_layer(obs.data_ptr(), mean.data_ptr() if normalize else None, var.data_ptr() if normalize else None,
                                   W1.data_ptr(), b1.data_ptr(), xn0.data_ptr() if normalize else None, h0.data_ptr(), M, D, 256,
                                   1e-5, 5.0, _stream()), "ag_mlp_input_layer")
    N.check(lib.ag_split_gemm_loss_heads_bwd(h0.data_ptr(), planes.data_ptr(), b2.data_ptr(), Wh.data_ptr(), bh.data_ptr(),
                                             dz0.data_ptr(), ctypes.byref(L0), M, 256, 256, A + 1, _stream()), "loss_heads_bwd")
    # the fused launch (same loss inputs: same generator seed)
    gl = torch.Generator(device="cuda").manual_seed(5)
    L1, _, out1 = _loss_args(N, lib, M, A, gl, tiles)
    xn1 = torch.full((M, D), 7.0, **f) if normalize else None
    h1 = torch.full((M, 256), 7.0, **f)
    dz1 = torch.full((M, 256), 7.0, **f)
    image = torch.empty(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_input_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), image.data_ptr(), _stream()), "in_prepare")
    inp = N.AgInputLayerArgs()
    inp.struct_size, inp.D = ctypes.sizeof(N.AgInputLayerArgs), D
    inp.obs_dev = obs.data_ptr()
    inp.mean_dev = mean.data_ptr() if normalize else None
    inp.var_dev = var.data_ptr() if normalize else None
    inp.xn_dev = xn1.data_ptr() if normalize else None
    inp.h1_dev = h1.data_ptr()
    inp.eps, inp.clip = 1e-5, 5.0
    N.check(lib.ag_split_gemm_input_loss_heads_bwd(ctypes.byref(inp), image.data_ptr(), b2.data_ptr(), Wh.data_ptr(), bh.data_ptr(),
                                                   dz1.data_ptr(), ctypes.byref(L1), M, 256, 256, A + 1, _stream()), "fused")
    torch.cuda.synchronize()
    if normalize:
        assert torch.equal(xn1, xn0) and xn0.abs().max() == 5.0
    # h1: both against float64 below; against each other within float32 rounding of an 18-term dot product
    assert (h1 - h0).abs().max() <= 4e-6 * max(1.0, h0.abs().max().item()), (h1 - h0).abs().max()
    # everything behind h1 (GEMM, heads, loss, head backward): the same computation on inputs that differ by float32 rounding
    scale = dz0.abs().max()
    assert (dz1 - dz0).abs().max() <= 2e-5 * scale, ((dz1 - dz0).abs().max(), scale)
    for k in out0:
        ref = out0[k]
        tol = 2e-5 * ref.abs().max().clamp_min(1e-6)
        if k == "loss_partials":      # per-tile sums of 256 rows: compare tile by tile at the scale of each column
            tol = 2e-4 * ref.abs().amax(0, keepdim=True).clamp_min(1e-6)
        assert ((out1[k] - ref).abs() <= tol).all(), (k, (out1[k] - ref).abs().max())
    # and the first layer against float64
    xr = torch.clamp((obs.double() - mean) / torch.sqrt(var.float().double() + 1e-5), -5, 5) if normalize else obs.double()
    z = xr @ W1.double().t() + b1.double()
    ref = torch.where(z > 0, z, torch.expm1(z))
    assert (h1.double() - ref).abs().max() < 2e-5 and (h0.double() - ref).abs().max() < 2e-5


def test_fused_first_layer_rejects_bad_arguments(lib):
    from airgym_amd import _native as N
    if not lib.ag_split_gemm_input_fwd_supported(18):
        pytest.skip("needs the 256-row tile build")
    assert not lib.ag_split_gemm_input_fwd_supported(48) and not lib.ag_split_gemm_input_fwd_supported(17)
    M, D, A = 512, 18, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    L, _, _ = _loss_args(N, lib, M, A, g, M // 256)
    f = dict(device="cuda", dtype=torch.float32)
    bufs = dict(obs=torch.zeros(M, D, **f), W1=torch.zeros(256, D, **f), b1=torch.zeros(256, **f), h1=torch.zeros(M, 256, **f),
                xn=torch.zeros(M, D, **f), mean=torch.zeros(D, device="cuda", dtype=torch.float64),
                var=torch.ones(D, device="cuda", dtype=torch.float64), image=torch.zeros(lib.ag_split_gemm_input_image_bytes(),
                dtype=torch.uint8, device="cuda"), b2=torch.zeros(256, **f), Wh=torch.zeros(A + 1, 256, **f), bh=torch.zeros(A + 1, **f),
                dz=torch.zeros(M, 256, **f))

    def call(M_=M, D_=D, size=None, xn=True, mean=True):
        inp = N.AgInputLayerArgs()
        inp.struct_size = ctypes.sizeof(N.AgInputLayerArgs) if size is None else size
        inp.D = D_
        inp.obs_dev, inp.h1_dev = bufs["obs"].data_ptr(), bufs["h1"].data_ptr()
        inp.mean_dev = bufs["mean"].data_ptr() if mean else None
        inp.var_dev = bufs["var"].data_ptr() if mean else None
        inp.xn_dev = bufs["xn"].data_ptr() if xn else None
        inp.eps, inp.clip = 1e-5, 5.0
        return lib.ag_split_gemm_input_loss_heads_bwd(ctypes.byref(inp), bufs["image"].data_ptr(), bufs["b2"].data_ptr(),
                                                      bufs["Wh"].data_ptr(), bufs["bh"].data_ptr(), bufs["dz"].data_ptr(),
                                                      ctypes.byref(L), M_, 256, 256, A + 1, _stream())
    assert call() == 0
    assert call(M_=500) == N.AG_ERR_UNSUPPORTED            # whole 256-row tiles only
    assert call(D_=48) == N.AG_ERR_UNSUPPORTED
    assert call(size=8) == -1
    assert call(xn=False) == -1          # normaliser statistics without an output for the normalised inputs
    assert call(xn=False, mean=False) == 0
    # the partial tables' capacity (ag_loss_epilogue.partial_tiles): a launch that would write more tiles than they hold is refused
    L.partial_tiles = M // 256
    assert call() == 0
    L.partial_tiles = M // 256 - 1
    assert call() == -1
    L.tile_rows, L.partial_tiles = 128, M // 256          # 128-row tiles: twice as many partial rows
    assert call() == -1
    L.partial_tiles = -3
    assert call() == -1
    L.tile_rows, L.partial_tiles = 0, 0
    h = torch.zeros(M, 256, **f)
    planes = torch.zeros(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")

    def plain():
        return lib.ag_split_gemm_loss_heads_bwd(h.data_ptr(), planes.data_ptr(), bufs["b2"].data_ptr(), bufs["Wh"].data_ptr(),
                                                bufs["bh"].data_ptr(), bufs["dz"].data_ptr(), ctypes.byref(L), M, 256, 256, A + 1, _stream())
    assert plain() == 0
    L.partial_tiles = 1
    assert plain() == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,D,normalize", [(196608, 48, True), (1000, 48, True), (4096, 18, True), (2048, 30, False), (777, 21, True),
                                           (4096, 62, True), (256, 1, False)])
def test_first_layer_as_its_own_launch_on_the_matrix_cores(lib, M, D, normalize):
    """ag_mlp_first_layer (csrc/first_layer.hip; Tracking's 48 inputs and every other width the forward GEMM cannot produce itself):
    the same function as ag_mlp_input_layer - normalised inputs bit-identical, activations float32-accurate against float64 -
    incl. odd widths (scalar loads), the widest supported input (D + 1 = 63) and ragged row counts."""
    from airgym_amd import _native as N
    assert lib.ag_mlp_first_layer_supported(D, 256) and not lib.ag_mlp_first_layer_supported(64, 256)
    assert not lib.ag_mlp_first_layer_supported(D, 128)
    g = torch.Generator(device="cuda").manual_seed(500 + M + D)
    f = dict(device="cuda", dtype=torch.float32)
    obs = 3.0 * torch.randn(M, D, generator=g, **f)
    obs[::7, 0] = 40.0                       # clamped by the normaliser
    mean = torch.randn(D, generator=g, device="cuda", dtype=torch.float64)
    var = torch.rand(D, generator=g, device="cuda", dtype=torch.float64) + 0.05
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.1 * torch.randn(256, generator=g, **f)
    image = torch.empty(lib.ag_mlp_first_layer_image_bytes(D), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_mlp_first_layer_prepare(W1.data_ptr(), b1.data_ptr(), D, image.data_ptr(), _stream()), "prepare")
    xn1 = torch.full((M, D), 7.0, **f) if normalize else None
    h1 = torch.full((M, 256), float("nan"), **f)
    N.check(lib.ag_mlp_first_layer(obs.data_ptr(), mean.data_ptr() if normalize else None, var.data_ptr() if normalize else None, 1e-5, 5.0,
                                   image.data_ptr(), xn1.data_ptr() if normalize else None, h1.data_ptr(), M, D, _stream()), "first_layer")
    xn0 = torch.zeros(M, D, **f) if normalize else None
    h0 = torch.zeros(M, 256, **f)
    rc0 = lib.ag_mlp_input_layer(obs.data_ptr(), mean.data_ptr() if normalize else None, var.data_ptr() if normalize else None,
                                 W1.data_ptr(), b1.data_ptr(), xn0.data_ptr() if normalize else None, h0.data_ptr(), M, D, 256,
                                 1e-5, 5.0, _stream())
    have_ref = rc0 == 0                      # (the vector-ALU kernel holds the weights in registers / LDS: not every width)
    torch.cuda.synchronize()
    assert torch.isfinite(h1).all()
    if normalize and have_ref:
        assert torch.equal(xn1, xn0)
    xr = torch.clamp((obs.double() - mean) / torch.sqrt(var.float().double() + 1e-5), -5, 5) if normalize else obs.double()
    z = xr @ W1.double().t() + b1.double()
    ref = torch.where(z > 0, z, torch.expm1(z))
    scale = xr.abs() @ W1.double().abs().t() + b1.double().abs() + 1e-30
    # split product + v_exp_f32; exp(z) - 1 rounds at 1.0, an ABSOLUTE 2^-23 that only shows against a tiny scale (D = 1)
    assert ((h1.double() - ref).abs() / (scale + 0.25)).max().item() < 6e-7
    if have_ref:
        assert (h1 - h0).abs().max() <= 4e-6 * max(1.0, h0.abs().max().item())
    if normalize:
        assert torch.equal(xn1, xr.float()) or (xn1.double() - xr).abs().max().item() < 1e-6
    # bf16 twin: one MFMA per product
    hb = torch.empty_like(h1)
    N.check(lib.ag_mlp_first_layer_bf16(obs.data_ptr(), mean.data_ptr() if normalize else None, var.data_ptr() if normalize else None, 1e-5,
                                        5.0, image.data_ptr(), xn1.data_ptr() if normalize else None, hb.data_ptr(), M, D, _stream()),
            "first_layer_bf16")
    # header contract: xn_dev is given iff the statistics are - a mismatch is refused, not silently skipped
    bad_xn = None if normalize else h0.data_ptr()
    assert lib.ag_mlp_first_layer(obs.data_ptr(), mean.data_ptr() if normalize else None, var.data_ptr() if normalize else None, 1e-5, 5.0,
                                  image.data_ptr(), bad_xn, h1.data_ptr(), M, D, _stream()) == -1      # AG_ERR_INVALID_ARG
    eb = ((hb.double() - ref).abs() / (scale + 0.25)).max().item()
    assert 1e-6 < eb < 2.0 ** -7, eb
