"""The data-parallel update path on ONE GPU (world_size 1 over RCCL): with multi_gpu the minibatch hipGraph is split at the
gradient all-reduce (graph A: forward / backward / reductions; eager all-reduce; graph B: rank average + clip + Adam + LR rule,
reference: trancate_gradients_and_step, lib/agent/a2c_base.py:293-316).  A one-rank group makes the all-reduce the identity and
the division a division by 1.0, so nothing changes numerically; the run with the split graphs must equal, bit for bit, the same data-parallel run with the
update launched eagerly (use_hip_graph_update: false) - which checks the split capture, the eager collective between two
replays and the KL returned from the reduced buffer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import yaml

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(multi_gpu, graph_update=True):
    with open(os.path.join(REPO, "scripts", "config", "ppo_hovering.yaml")) as f:
        params = yaml.safe_load(f)["params"]
    c = params["config"]
    params["network"]["mlp"]["units"] = [256, 256]
    envs = 4096
    c.update(num_actors=envs, minibatch_size=envs * c["horizon_length"] // 24, device="cuda:0", multi_gpu=multi_gpu,
             max_epochs=-1, write_summaries=False, print_stats=False, save_frequency=0, save_best_after=10 ** 9,
             use_hip_graph=True, use_hip_graph_update=graph_update, dist_backend="nccl")
    c["env_config"] = {"use_image": False, "num_envs": envs, "ctl_mode": "rate", "seed": 0, "sim_device": "cuda:0", "headless": True}
    params["seed"] = 0
    return params


def _run(multi_gpu, epochs=4, graph_update=True):
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    from airgym_amd.lib.core import collectives
    torch.manual_seed(0)
    agent = A2CAgent("mg", _params(multi_gpu, graph_update))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.broadcast_parameters()
    collectives.reset()
    kls = []
    for _ in range(epochs):
        agent.epoch_num += 1
        kls.append(agent.train_epoch()["kl"])
    out = {"param": agent.flat_param.clone(), "lr": agent.last_lr, "kls": kls, "graphs": dict(agent._upd_graphs),
           "graph_update": agent._graph_update, "counts": collectives.snapshot(),
           "rms": agent.model.running_mean_std.running_mean.clone()}
    agent.vec_env.env.hip.close()
    return out


def test_split_minibatch_graph_under_multi_gpu_equals_the_eager_update():
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eager = _run(True, graph_update=False)
        multi = _run(True, graph_update=True)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert multi["graph_update"] and not eager["graph_update"] and not eager["graphs"]
    assert "tail" in multi["graphs"]
    # graphs exist for the statistics-off minibatches; the first mini-epoch's (whose forward all-reduces the normaliser
    # moments when there is more than one rank) stayed eager
    assert any(k != "tail" and k[1] is False for k in multi["graphs"])
    assert not any(k != "tail" and k[1] is True for k in multi["graphs"])
    # 4 epochs x 5 mini-epochs x 24 minibatches gradient all-reduces, each issued eagerly, in both runs
    assert multi["counts"]["gradient"]["calls"] == 4 * 5 * 24 == eager["counts"]["gradient"]["calls"]
    assert "normaliser_moments" not in multi["counts"]          # a one-rank group has nothing to merge (running_mean_std.py)
    # same kernels in the same order on the same data: the split-graph run IS the eager run
    assert torch.equal(eager["rms"], multi["rms"])
    assert torch.equal(eager["param"], multi["param"]), (eager["param"] - multi["param"]).abs().max().item()
    assert eager["lr"] == multi["lr"] and eager["kls"] == multi["kls"]
