"""csrc/conv_kernels.hip against torch's conv2d evaluated in float64 on the CPU (reference modules: lib/network/cnn.py:11-13).

Tolerance: the kernels are exact float32 (f32-input MFMA / fmaf) in their own summation order, so the error against float64 is
float32 rounding over K terms: 2e-5 of the tensor's scale for the forward and the input gradient (K <= 288), 1e-4 for the weight
gradients (K = every output pixel of every image)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LAYERS = {
    "conv1": dict(cin=1, cout=16, k=5, hin=212, win=120),
    "conv2": dict(cin=16, cout=32, k=3, hin=106, win=60),
    "conv3": dict(cin=32, cout=64, k=3, hin=53, win=30),
}


def _close(got, ref, tol, what):
    scale = ref.abs().max().item()
    err = (got.double().cpu() - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max error {err:.3e} against scale {scale:.3e} (tolerance {tol:g})"


def _layer(name, seed):
    c = LAYERS[name]
    g = torch.Generator().manual_seed(seed)
    conv = nn.Conv2d(c["cin"], c["cout"], c["k"], stride=2, padding=c["k"] // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.3)
        conv.bias.copy_(torch.randn(conv.bias.shape, generator=g))
    return c, conv, g


@pytest.mark.parametrize("name", ["conv1", "conv2", "conv3"])
@pytest.mark.parametrize("n", [1, 5])
def test_forward_and_gradients_match_float64(name, n):
    from airgym_amd.lib.network import hip_conv
    c, conv, g = _layer(name, 3)
    x = torch.randn(n, c["cin"], c["hin"], c["win"], generator=g)
    ref_conv = nn.Conv2d(c["cin"], c["cout"], c["k"], stride=2, padding=c["k"] // 2).double()
    ref_conv.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    xr = x.double().requires_grad_(c["cin"] > 1)
    yr = ref_conv(xr)
    dy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(dy)

    conv = conv.cuda()
    xg = x.cuda().requires_grad_(c["cin"] > 1)
    assert hip_conv.supported(xg, conv)
    y = hip_conv.conv2d(xg, conv)
    assert y.shape == yr.shape
    _close(y.detach(), yr.detach(), 2e-5, name + " forward")
    y.backward(dy.float().cuda())
    _close(conv.weight.grad, ref_conv.weight.grad, 1e-4, name + " weight gradient")
    _close(conv.bias.grad, ref_conv.bias.grad, 1e-4, name + " bias gradient")
    if c["cin"] > 1:
        _close(xg.grad, xr.grad, 2e-5, name + " input gradient")


@pytest.mark.parametrize("name", ["conv2", "conv3"])
def test_previous_relu_batchnorm_applied_while_staging(name):
    """scale / shift given: the layer convolves relu(x) * scale[c] + shift[c] (padding stays zero) without that tensor existing."""
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    c, conv, g = _layer(name, 5)
    n = 3
    x = torch.randn(n, c["cin"], c["hin"], c["win"], generator=g)
    scale = torch.rand(c["cin"], generator=g) + 0.5
    shift = torch.randn(c["cin"], generator=g)
    act = torch.relu(x.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    yr = F.conv2d(act, conv.weight.double(), conv.bias.double(), stride=2, padding=1)
    dy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    w64 = conv.weight.double().detach().requires_grad_(True)
    b64 = conv.bias.double().detach().requires_grad_(True)
    F.conv2d(act, w64, b64, stride=2, padding=1).backward(dy)

    dev = torch.device("cuda")
    xg, sc, sh = x.to(dev), scale.to(dev), shift.to(dev)
    w, b = conv.weight.detach().to(dev).contiguous(), conv.bias.detach().to(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.empty(yr.shape, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(c["cin"], c["cout"]), dtype=torch.float32, device=dev)
    N.check(lib.ag_cnn_conv_fwd(xg.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, n,
                                c["cin"], c["cout"], c["hin"], c["win"], ws.data_ptr(), stream), "fwd")
    _close(y, yr.detach(), 2e-5, name + " forward with the previous layer's ReLU + BatchNorm")
    gparts = lib.ag_cnn_conv_wgrad_partials(n, c["cin"], c["cout"], c["hin"], c["win"])
    plen = c["cout"] * c["cin"] * 9 + c["cout"]
    partials = torch.empty(gparts, plen, dtype=torch.float32, device=dev)
    dyg = dy.float().to(dev)
    N.check(lib.ag_cnn_conv_wgrad(dyg.data_ptr(), xg.data_ptr(), sc.data_ptr(), sh.data_ptr(), partials.data_ptr(), 1, n, c["cin"],
                                  c["cout"], c["hin"], c["win"], stream), "wgrad")
    s = partials.sum(0)
    _close(s[:plen - c["cout"]].reshape(w.shape), w64.grad, 1e-4, name + " weight gradient with ReLU + BatchNorm")
    _close(s[plen - c["cout"]:], b64.grad, 1e-4, name + " bias gradient")


def test_unsupported_shapes_are_refused():
    from airgym_amd import _native as N
    from airgym_amd.lib.network import hip_conv
    lib = N.load()
    assert not lib.ag_cnn_conv_supported(16, 32, 100, 60)
    conv = nn.Conv2d(16, 32, 3, stride=2, padding=1).cuda()
    assert not hip_conv.supported(torch.zeros(2, 16, 100, 60, device="cuda"), conv)
    assert hip_conv.supported(torch.zeros(2, 16, 106, 60, device="cuda"), conv)
    assert not hip_conv.supported(torch.zeros(2, 16, 106, 60), conv.cpu())
    ws = torch.zeros(16, device="cuda")
    assert lib.ag_cnn_conv_fwd(ws.data_ptr(), None, None, ws.data_ptr(), ws.data_ptr(), ws.data_ptr(), None, 1, 16, 32, 100, 60,
                               ws.data_ptr(), None) == N.AG_ERR_UNSUPPORTED


def test_feature_extractor_is_the_same_function_with_and_without_the_kernels():
    """CNNFeatureExtractor (training mode, batch statistics) through csrc/conv_kernels.hip against the same module on torch's
    conv2d: features, parameter gradients and BatchNorm running statistics."""
    import copy
    from airgym_amd.lib.network.cnn import CNNFeatureExtractor
    torch.manual_seed(1)
    a = CNNFeatureExtractor(12).cuda().train()
    b = copy.deepcopy(a)
    b.hip_convs = False
    x = torch.rand(6, 1, 212, 120, device="cuda")
    g = torch.randn(6, 12, device="cuda")
    fa, fb = a(x), b(x)
    fa.backward(g)
    fb.backward(g)
    scale = fb.abs().max().item()
    assert (fa - fb).abs().max().item() <= 2e-4 * scale
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        s = pb.grad.abs().max().item()
        assert (pa.grad - pb.grad).abs().max().item() <= 5e-4 * s + 1e-7, name
    for (name, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-4, atol=1e-6), name


@pytest.mark.parametrize("weighted", [False, True])
def test_fused_trunk_is_the_layer_sequence(weighted):
    """lib/network/fused_cnn.py (one autograd node; the ReLU + BatchNorm outputs never written) against the same extractor layer by
    layer on torch's conv2d: features, every parameter gradient, running statistics; with and without image multiplicities."""
    import copy
    from airgym_amd.lib.network.cnn import CNNFeatureExtractor
    torch.manual_seed(2)
    a = CNNFeatureExtractor(12).cuda().train()
    with torch.no_grad():
        for mod in a.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_()
    b = copy.deepcopy(a)
    b.fused_trunk = False
    b.hip_convs = False
    before = copy.deepcopy(a.state_dict())
    x = torch.rand(7, 1, 212, 120, device="cuda")
    w = torch.tensor([1., 4., 2., 1., 3., 4., 1.], device="cuda") if weighted else None
    g = torch.randn(7, 12, device="cuda")
    # raw image + the input normaliser's per-pixel statistics (some pixels far enough out to hit the +-5 clamp)
    norm = (torch.rand(212 * 120, device="cuda"), torch.rand(212 * 120, device="cuda") * 0.3 + 0.02) if weighted else None
    fa, fb = a(x, w, norm), b(x, w, norm)
    fa.backward(g)
    fb.backward(g)
    scale = fb.abs().max().item()
    assert (fa - fb).abs().max().item() <= 2e-4 * scale
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        s = pb.grad.abs().max().item()
        assert (pa.grad - pb.grad).abs().max().item() <= 1e-3 * s + 1e-6, (name, (pa.grad - pb.grad).abs().max().item(), s)
    for (name, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-4, atol=1e-6), name
    # direct_grads: the backward writes into pre-allocated .grad tensors instead of handing gradients to autograd
    c = copy.deepcopy(b)
    c.fused_trunk = c.hip_convs = True
    c.direct_grads = True
    c.load_state_dict(before)
    trunk_params = [p for n_, p in c.named_parameters() if n_.startswith("features.")]
    for p in c.parameters():
        p.grad = torch.full_like(p, 7.0) if any(p is q for q in trunk_params) else torch.zeros_like(p)   # overwritten, not accumulated
    c(x, w, norm).backward(g)
    for (name, pa), (_, pc) in zip(a.named_parameters(), c.named_parameters()):
        # (same kernels; the partial sums are added up by differently shaped torch reductions: float32 rounding only)
        assert (pa.grad - pc.grad).abs().max().item() <= 1e-4 * pa.grad.abs().max().item() + 1e-9, name
    # eval mode (running statistics), no gradient: the rollout's path
    a.eval(), b.eval()
    with torch.no_grad():
        ea, eb = a(x, None, norm), b(x, None, norm)
    assert (ea - eb).abs().max().item() <= 2e-4 * eb.abs().max().item()


def test_weighted_indexed_moments_match_the_expanded_batch():
    """RunningMeanStd.update(x, weights=, index=) on the GPU (ag_weighted_moments, one pass) against the plain update on the
    batch written out: x[index] with row i repeated weights[i] times (what the reference's normaliser would see)."""
    from airgym_amd.lib.core.running_mean_std import RunningMeanStd
    torch.manual_seed(4)
    store = torch.rand(40, 1, 212, 120, device="cuda") * 3.0 + 0.5
    index = torch.tensor([3, 7, 8, 20, 21, 39, 0], device="cuda")
    weights = torch.tensor([4., 1., 3., 4., 2., 1., 4.], device="cuda")
    a = RunningMeanStd((1, 212, 120)).cuda()
    b = RunningMeanStd((1, 212, 120)).cuda()
    for _ in range(2):
        a.update(store, None, weights, index)
        b.update(torch.repeat_interleave(store[index].double(), weights.long(), dim=0))
    assert float(a.count) == float(b.count)
    assert torch.allclose(a.running_mean, b.running_mean, rtol=0, atol=1e-9)
    assert torch.allclose(a.running_var, b.running_var, rtol=1e-9, atol=1e-12)


def test_reductions_from_weights_equal_the_reduction_kernel():
    """The trunk's backward with the ReLU + BatchNorm reductions taken from the next convolution's (w, dw) (ag_bn_sums_from_conv,
    default) against the same trunk with ag_relu_bn_bwd_reduce passes over the gradients: every parameter gradient."""
    import copy
    from airgym_amd.lib.network.cnn import CNNFeatureExtractor
    torch.manual_seed(3)
    a = CNNFeatureExtractor(12).cuda().train()
    with torch.no_grad():
        for mod in a.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_()
    b = copy.deepcopy(a)
    b.bn_sums_from_weights = False
    x = torch.rand(9, 1, 212, 120, device="cuda") * 3.0
    w = torch.tensor([1., 4., 2., 1., 3., 4., 1., 2., 2.], device="cuda")
    g = torch.randn(9, 12, device="cuda")
    c = copy.deepcopy(a)
    c.dgrad_epilogue = False        # the second layer's ReLU + BatchNorm backward as a pass of its own
    c.conv1_wgrad_fused = False     # and conv2's input gradient / conv1's weight gradient as two kernels
    a(x, w).backward(g)
    b(x, w).backward(g)
    c(x, w).backward(g)
    for (name, pa), (_, pb), (_, pc) in zip(a.named_parameters(), b.named_parameters(), c.named_parameters()):
        s = pb.grad.abs().max().item()
        assert (pa.grad - pb.grad).abs().max().item() <= 2e-4 * s + 1e-8, (name, (pa.grad - pb.grad).abs().max().item(), s)
        assert (pa.grad - pc.grad).abs().max().item() <= 2e-4 * s + 1e-8, (name, (pa.grad - pc.grad).abs().max().item(), s)


def test_small_batchnorm_weights_fall_back_to_the_reduction_kernels():
    """The (w, dw) identity divides by gamma: a BatchNorm scale near zero would turn the float32 rounding of dw into a huge
    dgamma for that channel (and, through gradient clipping, starve every other parameter).  With one gamma at 1e-6 and one at
    exactly 0 the default trunk must notice (bn_gamma_guard) and produce the gradients of the reduction-kernel path - every
    parameter, to 1e-3 of its scale; the un-guarded identity on the same weights is shown to be off by orders of magnitude."""
    import copy
    from airgym_amd.lib.network.cnn import CNNFeatureExtractor
    torch.manual_seed(5)
    a = CNNFeatureExtractor(12).cuda().train()
    with torch.no_grad():
        for mod in a.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_()
        a.features[2].weight[3] = 1e-6          # first BatchNorm layer: one tiny scale
        a.features[5].weight[7] = 0.0           # second: one exactly zero
    a.reset_gamma_guard()
    ref = copy.deepcopy(a)
    ref.bn_sums_from_weights = False            # ag_relu_bn_bwd_reduce passes over the gradients: exact for any gamma
    raw = copy.deepcopy(a)
    raw.bn_gamma_guard = 0.0                    # the identity, un-guarded
    x = torch.rand(9, 1, 212, 120, device="cuda") * 3.0
    w = torch.tensor([1., 4., 2., 1., 3., 4., 1., 2., 2.], device="cuda")
    g = torch.randn(9, 12, device="cuda")
    a(x, w).backward(g)
    ref(x, w).backward(g)
    raw(x, w).backward(g)
    assert a.bn_fallback_steps == 1 and raw.bn_fallback_steps == 0
    worst_raw = 0.0
    for (name, pa), (_, pr), (_, pw) in zip(a.named_parameters(), ref.named_parameters(), raw.named_parameters()):
        s = pr.grad.abs().max().item()
        assert torch.isfinite(pa.grad).all(), name
        assert (pa.grad - pr.grad).abs().max().item() <= 1e-3 * s + 1e-8, (name, (pa.grad - pr.grad).abs().max().item(), s)
        worst_raw = max(worst_raw, ((pw.grad - pr.grad).abs().max().item() / (s + 1e-30)) if torch.isfinite(pw.grad).all() else 1e30)
    assert worst_raw > 1e-1, worst_raw          # what the guard is there for
    # healthy weights again: the guard lets the identity back in (one step later: the decision uses the previous step's copy)
    with torch.no_grad():
        a.features[2].weight[3] = 1.0
        a.features[5].weight[7] = 1.0
    for p in a.parameters():
        p.grad = None
    a(x, w).backward(g)          # decided on the stale ratio: still the reduction kernels
    a(x, w).backward(g)          # now the identity
    assert a.bn_fallback_steps == 2


@pytest.mark.parametrize("shape", [(5, 32, 53, 30), (4, 64, 27, 15)])
def test_border_sums_in_passing_equal_the_standalone_kernel_and_torch(shape):
    """ag_relu_bn_bwd_dx_weighted's plane / border sums (formed while dx is written) against ag_plane_border_sums on the result and
    against torch slicing."""
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    n, c, h, w = shape
    torch.manual_seed(6)
    dev = torch.device("cuda")
    x, dy = torch.randn(shape, device=dev), torch.randn(shape, device=dev)
    coef = torch.stack((torch.rand(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) + 0.5,
                        torch.full((c,), 1.0 / (n * h * w), device=dev)), 1).contiguous()
    sums = torch.randn(c, 2, device=dev)
    dx = torch.empty_like(x)
    ps, bs = torch.empty(n, c, device=dev), torch.empty(n, c, 5, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N.check(lib.ag_relu_bn_bwd_dx_weighted(dy.data_ptr(), x.data_ptr(), coef.data_ptr(), sums.data_ptr(), None, dx.data_ptr(),
                                           ps.data_ptr(), bs.data_ptr(), w, n, c, h * w, stream), "dx")
    ref = torch.stack((dx[:, :, 0, :].sum(2), dx[:, :, -1, :].sum(2), dx[:, :, :, 0].sum(2), dx[:, :, 0, 0], dx[:, :, -1, 0]), 2)
    alone = torch.empty(n, c, 5, device=dev)
    N.check(lib.ag_plane_border_sums(dx.data_ptr(), alone.data_ptr(), n, c, h, w, stream), "border")
    scale = ref.abs().max().item()
    assert (bs - ref).abs().max().item() <= 1e-5 * scale and (alone - ref).abs().max().item() <= 1e-5 * scale
    assert (ps - dx.sum((2, 3))).abs().max().item() <= 1e-5 * dx.sum((2, 3)).abs().max().item()


@pytest.mark.parametrize("weighted,with_sums", [(False, True), (True, True), (True, False)])
def test_input_gradient_with_the_batchnorm_backward_in_its_epilogue(weighted, with_sums):
    """ag_cnn_conv_dgrad_bn (third layer): [x > 0] (A g + m_i (B x + C)) with g the convolution's input gradient, against float64
    (aten's convolution_backward for g), and the per-workgroup sums of the result against torch slicing."""
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    torch.manual_seed(8)
    dev = torch.device("cuda")
    n = 5
    dz = torch.randn(n, 64, 27, 15, device=dev)
    w = torch.randn(64, 32, 3, 3, device=dev) * 0.1
    x = torch.randn(n, 32, 53, 30, device=dev)
    tab = torch.cat((torch.randn(32, 3, device=dev), torch.zeros(32, 1, device=dev)), 1).contiguous()
    wts = torch.tensor([1., 3., 2., 4., 1.], device=dev) if weighted else None
    g = torch.ops.aten.convolution_backward(dz.double(), x.double(), w.double(), [64], [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                            [True, False, False])[0]
    t = tab.double()
    mi = (wts.double() if weighted else torch.ones(n, device=dev, dtype=torch.float64)).view(n, 1, 1, 1)
    ref = (x.double() > 0) * (t[:, 0].view(1, 32, 1, 1) * g + mi * (t[:, 1].view(1, 32, 1, 1) * x.double() + t[:, 2].view(1, 32, 1, 1)))
    dx = torch.empty_like(x)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(32, 64), device=dev)
    rows = lib.ag_cnn_conv_dgrad_bn_rows(n, 32, 64, 53, 30)
    assert rows > 0
    sums = torch.full((rows, 32, 6), float("nan"), device=dev) if with_sums else None
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N.check(lib.ag_cnn_conv_dgrad_bn(dz.data_ptr(), w.data_ptr(), x.data_ptr(), tab.data_ptr(), wts.data_ptr() if weighted else None,
                                     dx.data_ptr(), sums.data_ptr() if with_sums else None, n, 32, 64, 53, 30, ws.data_ptr(), stream),
            "ag_cnn_conv_dgrad_bn")
    scale = ref.abs().max().item()
    assert (dx.double() - ref).abs().max().item() <= 2e-6 * scale
    if with_sums:
        d = dx.double()
        want = torch.stack((d.sum((0, 2, 3)), d[:, :, 0, :].sum((0, 2)), d[:, :, -1, :].sum((0, 2)), d[:, :, :, 0].sum((0, 2)),
                            d[:, :, 0, 0].sum(0), d[:, :, -1, 0].sum(0)), 1)
        got = sums.double().sum(0)
        assert torch.isfinite(sums).all()
        assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    # other layers are refused
    assert lib.ag_cnn_conv_dgrad_bn_rows(n, 16, 32, 106, 60) == N.AG_ERR_UNSUPPORTED


@pytest.mark.parametrize("weighted,normalised,indexed", [(False, False, False), (True, True, True)])
def test_input_gradient_feeding_the_first_weight_gradient_in_one_kernel(weighted, normalised, indexed):
    """ag_cnn_conv_dgrad_conv1_wgrad against the two kernels it replaces (ag_cnn_conv_dgrad, then ag_cnn_conv1_wgrad with the
    first layer's ReLU + BatchNorm backward in its staging): the same float32 gradient values, summed in a different order."""
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    torch.manual_seed(10)
    dev = torch.device("cuda")
    n = 5
    dz = torch.randn(n, 32, 53, 30, device=dev)
    w = torch.randn(32, 16, 3, 3, device=dev) * 0.1
    x1 = torch.randn(n, 16, 106, 60, device=dev)
    tab = torch.cat((torch.randn(16, 3, device=dev), torch.zeros(16, 1, device=dev)), 1).contiguous()
    wts = torch.tensor([1., 3., 2., 4., 1.], device=dev) if weighted else None
    store = torch.rand(12, 1, 212, 120, device=dev) * 3.0
    index = torch.tensor([7, 0, 3, 11, 4], device=dev) if indexed else None
    img = store if indexed else store[:n].contiguous()
    mean = torch.rand(212 * 120, device=dev) if normalised else None
    std = (torch.rand(212 * 120, device=dev) * 0.3 + 0.02) if normalised else None
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: t.data_ptr() if t is not None else None
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(16, 32), device=dev)
    # the two kernels
    dy1 = torch.empty_like(x1)
    N.check(lib.ag_cnn_conv_dgrad(dz.data_ptr(), w.data_ptr(), dy1.data_ptr(), n, 16, 32, 106, 60, ws.data_ptr(), stream), "dgrad")
    pa = torch.empty(lib.ag_cnn_conv1_wgrad_partials(n), 16, 32, device=dev)
    N.check(lib.ag_cnn_conv1_wgrad(dy1.data_ptr(), x1.data_ptr(), tab.data_ptr(), P(wts), img.data_ptr(), P(index), P(mean), P(std),
                                   pa.data_ptr(), n, stream), "conv1_wgrad")
    # the one kernel
    rows = lib.ag_cnn_conv_dgrad_conv1_wgrad_partials(n)
    assert rows > 0
    pb = torch.full((rows, 16, 32), float("nan"), device=dev)
    N.check(lib.ag_cnn_conv_dgrad_conv1_wgrad(dz.data_ptr(), w.data_ptr(), x1.data_ptr(), tab.data_ptr(), P(wts), img.data_ptr(), P(index),
                                              P(mean), P(std), pb.data_ptr(), n, ws.data_ptr(), stream), "dgrad_conv1_wgrad")
    a, b = pa.double().sum(0)[:, :26], pb.double().sum(0)[:, :26]
    assert torch.isfinite(pb[:, :, :26]).all()
    assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item(), ((a - b).abs().max().item(), a.abs().max().item())


@pytest.mark.parametrize("name", ["conv2", "conv3"])
@pytest.mark.parametrize("apply", [False, True])
@pytest.mark.parametrize("n", [1, 7])
def test_split_bf16_forward_is_float32_accurate(name, apply, n):
    """ag_cnn_conv_fwd_split (round 6: the 3 x 3 layers' forward on the bf16 matrix cores, exact three-way split of both operands, six
    MFMAs per product block): output against float64 within the f32 kernel's own tolerance (2e-5 of scale), per-(image, band) sums of
    relu(y) and relu(y)^2 against float64 sums, with and without the previous layer's ReLU + BatchNorm applied while staging (padding
    stays zero), every band incl. the partial last one; and not further from float64 than 3x the f32-input-MFMA kernel is."""
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    c, conv, g = _layer(name, 11)
    x = torch.randn(n, c["cin"], c["hin"], c["win"], generator=g) * 2.0
    scale = torch.rand(c["cin"], generator=g) + 0.5
    shift = torch.randn(c["cin"], generator=g)
    act = torch.relu(x.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) if apply else x.double()
    yr = F.conv2d(act, conv.weight.double(), conv.bias.double(), stride=2, padding=1)
    dev = torch.device("cuda")
    xg, sc, sh = x.to(dev), scale.to(dev), shift.to(dev)
    w, b = conv.weight.detach().to(dev).contiguous(), conv.bias.detach().to(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (c["cin"], c["cout"], c["hin"], c["win"])
    ho, wo = yr.shape[2], yr.shape[3]
    bands = lib.ag_cnn_conv_fwd_split_bands(*args)
    assert bands == (ho + 3) // 4
    y = torch.full(yr.shape, float("nan"), dtype=torch.float32, device=dev)
    stats = torch.full((n, bands, c["cout"], 2), float("nan"), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(c["cin"], c["cout"]), dtype=torch.float32, device=dev)
    N.check(lib.ag_cnn_conv_fwd_split(xg.data_ptr(), sc.data_ptr() if apply else None, sh.data_ptr() if apply else None, w.data_ptr(),
                                      b.data_ptr(), y.data_ptr(), stats.data_ptr(), n, *args, ws.data_ptr(), stream), "fwd_split")
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(stats).all()
    _close(y, yr, 2e-5, name + " split forward")
    y0 = torch.empty_like(y)
    N.check(lib.ag_cnn_conv_fwd(xg.data_ptr(), sc.data_ptr() if apply else None, sh.data_ptr() if apply else None, w.data_ptr(),
                                b.data_ptr(), y0.data_ptr(), None, n, *args, ws.data_ptr(), stream), "fwd")
    e_split = (y.double().cpu() - yr).abs().max().item()
    e_f32 = (y0.double().cpu() - yr).abs().max().item()
    assert e_split <= 3.0 * e_f32 + 1e-7 * yr.abs().max().item(), (e_split, e_f32)
    # statistics: sums over the band's valid pixels
    rl = torch.relu(yr)
    for band in range(bands):
        rows = rl[:, :, 4 * band:min(4 * band + 4, ho), :]
        s1, s2 = rows.sum((2, 3)), (rows * rows).sum((2, 3))
        got = stats[:, band].double().cpu()
        assert (got[..., 0] - s1).abs().max().item() <= 2e-5 * max(1.0, s1.abs().max().item()), (band, "sum")
        assert (got[..., 1] - s2).abs().max().item() <= 2e-5 * max(1.0, s2.abs().max().item()), (band, "sum of squares")
    # without statistics the same output, bit for bit
    y2 = torch.empty_like(y)
    N.check(lib.ag_cnn_conv_fwd_split(xg.data_ptr(), sc.data_ptr() if apply else None, sh.data_ptr() if apply else None, w.data_ptr(),
                                      b.data_ptr(), y2.data_ptr(), None, n, *args, ws.data_ptr(), stream), "fwd_split")
    assert torch.equal(y, y2)


@pytest.mark.parametrize("name", ["conv2", "conv3"])
def test_split_forward_large_batch_item_order(name):
    """From 8 images per CU on, the split forward's persistent workgroups walk whole images (sequential planes) instead of taking
    (image, band) items round robin - another loop structure around the same arithmetic.  2 100 images (> 8 x 256): output and
    per-(image, band) statistics against the f32-input-MFMA kernel (two float32 evaluations: 3e-6 of scale), every image and band
    written exactly once (NaN-prefilled outputs)."""
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    c, conv, g = _layer(name, 13)
    n = 2100
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, c["cin"], c["hin"], c["win"], generator=gen, device=dev)
    sc = torch.rand(c["cin"], generator=gen, device=dev) + 0.5
    sh = torch.randn(c["cin"], generator=gen, device=dev)
    w, b = conv.weight.detach().to(dev).contiguous(), conv.bias.detach().to(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (c["cin"], c["cout"], c["hin"], c["win"])
    ho, wo = (c["hin"] - 1) // 2 + 1, c["win"] // 2
    ws = torch.empty(lib.ag_cnn_conv_workspace_floats(c["cin"], c["cout"]), dtype=torch.float32, device=dev)
    y = torch.full((n, c["cout"], ho, wo), float("nan"), dtype=torch.float32, device=dev)
    bands = lib.ag_cnn_conv_fwd_split_bands(*args)
    st = torch.full((n, bands, c["cout"], 2), float("nan"), dtype=torch.float32, device=dev)
    N.check(lib.ag_cnn_conv_fwd_split(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(),
                                      n, *args, ws.data_ptr(), stream), "fwd_split")
    y0 = torch.empty_like(y)
    bands0 = lib.ag_cnn_conv_fwd_bands(*args)
    st0 = torch.empty(n, bands0, c["cout"], 2, dtype=torch.float32, device=dev)
    N.check(lib.ag_cnn_conv_fwd(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(), y0.data_ptr(), st0.data_ptr(),
                                n, *args, ws.data_ptr(), stream), "fwd")
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(st).all()
    assert (y - y0).abs().max().item() <= 3e-6 * y0.abs().max().item()
    s, s0 = st.sum(1), st0.sum(1)                                   # per image: the band partitions differ between the two kernels
    assert (s - s0).abs().max().item() <= 2e-5 * s0.abs().max().item()
