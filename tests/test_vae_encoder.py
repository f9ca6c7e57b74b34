"""Frozen VAE depth encoder (airgym_amd/lib/network/vae.py) against outputs of the reference's own ImgEncoder / VAE.encode
(tests/golden/vae_encoder.npz, made by tests/golden/make_golden.py::gen_vae with a deterministic parameter fill), and its
integration into the actor-critic (reference: lib/model/a2c_continuous_logstd_model.py:32-48,114-126)."""
import math
import os

import numpy as np
import pytest
import torch

from airgym_amd.lib.network.vae import DepthEncoder, FrozenVAEEncoder

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_encoder.npz")


def fill_(module):
    """The fill formula of make_golden.py::vae_fill_ (a data specification: parameter i of the name-sorted list)."""
    with torch.no_grad():
        for i, (name, p) in enumerate(sorted(module.named_parameters())):
            k = torch.arange(p.numel(), dtype=torch.float64)
            v = torch.cos(0.61803 * k + i) * (0.7 / math.sqrt(p[0].numel())) if p.dim() > 1 else 0.02 * torch.sin(k + i)
            p.copy_(v.reshape(p.shape).float())


def _probe_image():
    i = torch.arange(212, dtype=torch.float32).view(1, 1, 212, 1)
    j = torch.arange(120, dtype=torch.float32).view(1, 1, 1, 120)
    b = torch.arange(3, dtype=torch.float32).view(3, 1, 1, 1)
    return 0.5 + 0.5 * torch.sin(0.05 * i + 0.11 * j + b)


def test_encoder_matches_reference_outputs():
    g = np.load(GOLD)
    enc = FrozenVAEEncoder({"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                            "interpolation_mode": "bilinear"}, device="cpu")
    # same parameter names as the reference's encoder => trained/vae_model.pth would load key for key
    assert sorted(n for n, _ in enc.encoder.named_parameters()) == list(g["param_names"])
    fill_(enc.encoder)
    img = _probe_image()
    resized = torch.nn.functional.interpolate(img, (120, 212), mode="bilinear")
    np.testing.assert_allclose(resized[:, 0, ::17, ::23].numpy(), g["resized_probe"], rtol=0, atol=1e-6)
    z = enc.encoder(resized)
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=0, atol=2e-6)
    means = enc.encode(img)                       # 212x120 camera layout -> resized inside, means returned
    np.testing.assert_allclose(means.numpy(), g["means"], rtol=0, atol=2e-6)
    enc.return_sampled_latent = True
    torch.manual_seed(0)
    sampled = enc.encode(img)
    torch.manual_seed(0)
    eps = torch.randn(3, 64)
    np.testing.assert_allclose(sampled.numpy(), g["means"] + eps.numpy() * g["std"], rtol=0, atol=1e-5)


def test_missing_weights_fail_loudly(tmp_path):
    with pytest.raises(FileNotFoundError, match="vae_model.pth"):
        FrozenVAEEncoder({"latent_dims": 64, "model_folder": str(tmp_path), "model_file": "vae_model.pth"}, device="cpu")
    # a checkpoint in the reference's layout (DataParallel / 'dronet.' prefixes, decoder keys present) loads
    src = DepthEncoder(1, 64)
    sd = {"module.dronet." + k: v for k, v in src.state_dict().items()}
    sd["module.img_decoder.fc.weight"] = torch.zeros(3, 3)
    torch.save(sd, tmp_path / "vae_model.pth")
    enc = FrozenVAEEncoder({"latent_dims": 64, "model_folder": str(tmp_path), "model_file": "vae_model.pth"}, device="cpu")
    assert enc.pretrained
    for (k, a), (_, b) in zip(src.state_dict().items(), enc.encoder.state_dict().items()):
        assert torch.equal(a, b), k


def test_actor_critic_with_frozen_vae():
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    params = {"network": {"separate": False, "mlp": {"units": [64, 128, 64], "activation": "elu"},
                          "space": {"continuous": {"fixed_sigma": True}},
                          "vae": {"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                                  "interpolation_mode": "bilinear", "return_sampled_latent": False}},
              "config": {"normalize_input": True, "normalize_value": True}}
    keys = {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}}
    m = ModelA2CContinuousLogStd(params, keys)
    sd = m.state_dict()
    assert not any("conv" in k or "dense" in k for k in sd), "the frozen encoder must stay out of the policy checkpoint"
    assert all(p.requires_grad for p in m.parameters())
    assert m.actor_mlp.layers[0].weight.shape == (64, 16 + 64)
    assert sd["running_mean_std.running_mean_std.observation.running_mean"].shape == (80,)
    obs = {"image": torch.rand(5, 1, 212, 120), "observation": torch.randn(5, 16)}
    m.eval()
    out = m({"is_train": False, "obs": obs})
    assert out["actions"].shape == (5, 4) and out["values"].shape == (5, 1)
    m.train()
    res = m({"is_train": True, "prev_actions": out["actions"], "obs": obs})
    res["prev_neglogp"].sum().backward()
    assert m.actor_mlp.layers[0].weight.grad is not None
    assert all(p.grad is None for p in m._frozen[0].encoder.parameters())
    m.double()                                     # _apply reaches the unregistered encoder too
    assert next(m._frozen[0].encoder.parameters()).dtype == torch.float64


def test_chunked_encoding_equals_one_call():
    """network.vae.encode_chunk only bounds the batch per convolution call: same features, same order."""
    import torch
    from airgym_amd.lib.network.vae import FrozenVAEEncoder
    torch.manual_seed(0)
    cfg = {"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212], "interpolation_mode": "bilinear"}
    a = FrozenVAEEncoder(cfg, device="cpu")
    b = FrozenVAEEncoder(dict(cfg, encode_chunk=3), device="cpu")
    b.encoder.load_state_dict(a.encoder.state_dict())
    img = torch.rand(8, 1, 212, 120)
    za, zb = a.encode(img), b.encode(img)
    assert za.shape == (8, 64) and torch.allclose(za, zb, atol=1e-6)
    prev = torch.backends.cudnn.benchmark
    a.encode(img[:2])
    assert torch.backends.cudnn.benchmark == prev
