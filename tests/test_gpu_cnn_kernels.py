"""ReLU + BatchNorm2d on csrc/cnn_kernels.hip (airgym_amd/lib/network/fused_relu_bn.py) against the two torch modules of the
reference's feature extractor (lib/network/cnn.py:3-33) on the same inputs: forward, every gradient, the running statistics,
eval mode; the three plane sizes of the Planning network (float4 / float2 / scalar paths) and a ragged tail."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(37, 16, 106, 60), (33, 32, 53, 30), (130, 64, 27, 15), (1, 3, 5, 7), (300, 16, 4, 4)])
def test_relu_batchnorm_matches_torch_modules(shape):
    from airgym_amd.lib.network.fused_relu_bn import relu_batchnorm, usable
    torch.manual_seed(shape[0])
    n, c, h, w = shape
    bn_ref = nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.5, 0.5)
        bn_ref.running_mean.uniform_(-0.2, 0.2); bn_ref.running_var.uniform_(0.5, 2.0)
    bn = copy.deepcopy(bn_ref)
    x = (torch.randn(n, c, h, w, device="cuda") * 1.5 + 0.3).requires_grad_(True)
    x_ref = x.detach().clone().requires_grad_(True)
    assert usable(x, bn)
    g = torch.randn(n, c, h, w, device="cuda")
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    # float64 evaluation of the same two modules: the yardstick (MIOpen's float32 batch-norm backward is itself ~1e-3 off)
    bn64 = copy.deepcopy(bn_ref).double()
    x64 = x.detach().double().requires_grad_(True)
    y64 = bn64(torch.relu(x64))
    y64.backward(g.double())
    # training step
    y = relu_batchnorm(x, bn)
    y_ref = bn_ref(torch.relu(x_ref))
    assert torch.allclose(y.double(), y64, rtol=1e-5, atol=1e-5), (y.double() - y64).abs().max()
    assert torch.allclose(y, y_ref, rtol=1e-4, atol=1e-4), (y - y_ref).abs().max()
    y.backward(g)
    y_ref.backward(g)
    sc = x64.grad.abs().max().item() + 1e-12
    err_mine = (x.grad.double() - x64.grad).abs().max().item()
    err_lib = (x_ref.grad.double() - x64.grad).abs().max().item()
    assert err_mine <= 1e-5 * sc + 1e-7, (err_mine, err_lib, sc)
    assert err_mine <= err_lib + 1e-6 * sc                     # at least as close to float64 as the library path
    m = n * h * w
    for got, ref in ((bn.weight.grad, bn64.weight.grad), (bn.bias.grad, bn64.bias.grad)):
        assert torch.allclose(got.double(), ref, rtol=1e-5, atol=1e-5 * m ** 0.5), (got.double() - ref).abs().max()
    assert torch.allclose(bn.running_mean.double(), bn64.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), bn64.running_var, rtol=1e-5, atol=1e-6)
    assert not torch.equal(bn.running_mean, rm0) and not torch.equal(bn.running_var, rv0)
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked) == 1
    # exactly zero inputs take the ReLU' = 0 branch like torch
    # eval mode: running statistics
    bn.eval(); bn_ref.eval()
    with torch.no_grad():
        ye = relu_batchnorm(x.detach(), bn)
        ye_ref = bn_ref(torch.relu(x_ref.detach()))
    assert torch.allclose(ye, ye_ref, rtol=1e-5, atol=2e-5)


def test_feature_extractor_fused_equals_plain_modules():
    """CNNFeatureExtractor with the fused pairs == the same module with plain torch layers: output, all parameter gradients,
    buffers after two training steps; state-dict keys unchanged."""
    from airgym_amd.lib.network.cnn import CNNFeatureExtractor
    torch.manual_seed(0)
    a = CNNFeatureExtractor(30).cuda()
    b = copy.deepcopy(a)
    b.fused_relu_bn = False
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    for step in range(2):
        img = torch.rand(48, 1, 212, 120, device="cuda") * 4.0 - 1.0
        fa, fb = a(img), b(img)
        assert torch.allclose(fa, fb, rtol=1e-4, atol=1e-4), (fa - fb).abs().max()
        (fa.square().mean()).backward()
        (fb.square().mean()).backward()
        # one scale for all parameters: a convolution bias in front of a batch norm has an exactly-zero gradient, what both paths
        # return there is rounding noise.  (The plain path's batch-norm backward is MIOpen's: ~1e-3 of a float64 evaluation.)
        sc = max(pb.grad.abs().max().item() for pb in b.parameters())
        for (na, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert (pa.grad - pb.grad).abs().max().item() <= 1e-2 * sc + 1e-8, (na, step)
        a.zero_grad(); b.zero_grad()
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(va.float(), vb.float(), rtol=1e-4, atol=1e-5), k
    a.eval(); b.eval()
    with torch.no_grad():
        assert torch.allclose(a(img), b(img), rtol=1e-4, atol=1e-4)


def test_relu_bn_rejects_bad_arguments():
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros(2, 65, 4, 4, device="cuda")
    p = torch.zeros(8, 65, 2, device="cuda")
    assert lib.ag_relu_bn_stats(x.data_ptr(), p.data_ptr(), 2, 65, 16, st) != 0          # C > 64
    assert lib.ag_relu_bn_stats(None, p.data_ptr(), 2, 16, 16, st) != 0
    assert lib.ag_relu_bn_stats(x.data_ptr(), p.data_ptr(), 0, 16, 16, st) != 0
