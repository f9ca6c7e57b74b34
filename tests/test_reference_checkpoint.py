"""Checkpoint compatibility with the reference's shipped Planning policy, trained/planning_cnn_rate.pth.

tests/golden/planning_checkpoint.npz (made by make_golden.py::gen_planning_checkpoint) holds a manifest of that checkpoint's
tensors and the mu / value outputs of the reference's OWN sub-modules loaded with it, on deterministic inputs.
  * the manifest test needs nothing but the fixture: this build's Planning model has exactly the checkpoint's keys and shapes;
  * the behavioural test loads the real file when the reference checkout is present (build container), strictly, through the
    player, and must reproduce the golden outputs.  The 778 KB checkpoint itself is not copied into the repo."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "planning_checkpoint.npz")
CKPT = "/root/reference/trained/planning_cnn_rate.pth"


def _model():
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    params = {"network": {"separate": False, "mlp": {"units": [64, 128, 64], "activation": "elu"},
                          "space": {"continuous": {"fixed_sigma": True}}, "cnn": {"output_dim": 30}},
              "config": {"normalize_input": True, "normalize_value": True}}
    keys = {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}}
    return ModelA2CContinuousLogStd(params, keys)


def _probe_inputs(n=6):
    i = torch.arange(212, dtype=torch.float32).view(1, 1, 212, 1)
    j = torch.arange(120, dtype=torch.float32).view(1, 1, 1, 120)
    b = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1)
    image = 0.5 + 0.5 * torch.sin(0.031 * i + 0.057 * j + 0.7 * b)
    observation = torch.sin(torch.arange(n * 16, dtype=torch.float32).view(n, 16) * 0.37) * 1.5
    return image, observation


def test_model_layout_equals_the_reference_checkpoint():
    g = np.load(GOLD)
    manifest = json.loads(str(g["manifest"]))
    sd = _model().state_dict()
    assert sorted(sd.keys()) == sorted(manifest.keys())
    for k, meta in manifest.items():
        assert list(sd[k].shape) == meta["shape"], k
        assert str(sd[k].dtype).replace("torch.", "") == meta["dtype"], k
    assert int(g["epoch"]) == 200 and int(g["frame"]) == 4915200


@pytest.mark.skipif(not os.path.isfile(CKPT), reason="reference checkout (trained/planning_cnn_rate.pth) not present on this box")
def test_reference_checkpoint_loads_and_reproduces_reference_outputs():
    g = np.load(GOLD)
    manifest = json.loads(str(g["manifest"]))
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    m = _model()
    m.load_state_dict(ck["model"], strict=True)
    for k, v in m.state_dict().items():          # the file is the one the fixture was made from
        assert abs(float(v.double().sum()) - manifest[k]["sum"]) <= 1e-9 * max(1.0, manifest[k]["abs_sum"]), k
    m.eval()
    image, observation = _probe_inputs()
    with torch.no_grad():
        mu, logstd, value = m.trunk({"image": image, "observation": observation})
        value_denorm = m.denorm_value(value)
    np.testing.assert_allclose(mu.numpy(), g["mu"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(value.numpy(), g["value"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(value_denorm.numpy(), g["value_denorm"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logstd[0].numpy(), g["logstd"], rtol=0, atol=0)
    # and through the player's restore path (players.py:372-388): deterministic actions = clamp(mu)
    from airgym_amd.lib.agent.players import A2CPlayer
    assert hasattr(A2CPlayer, "restore")
