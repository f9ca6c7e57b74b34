"""Planning task on the MI355X: HIP kernels through the C ABI vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TERMS = ["continous_action_reward", "heading_reward", "speed_reward", "forward_reward", "alive_reward", "ups_reward",
         "z_reward", "esdf_reward", "thrust_reward", "reach_goal_reward", "reward"]


@pytest.fixture(scope="module")
def Handle():
    from airgym_amd.hip_env import HipEnvHandle
    assert torch.cuda.is_available()
    return HipEnvHandle


def _actions(rng, n, t):
    a = rng.uniform(-0.4, 0.4, size=(n, 4)).astype(np.float32)
    a[:, 3] = rng.uniform(-0.72, -0.66, size=n)      # thrust around hover (0.1537 -> -0.6926)
    a[:, 1] = rng.uniform(0.0, 0.5, size=n)          # pitch forward, into the obstacle field
    return a


def test_initial_state_and_scene_match_oracle(Handle):
    from oracle.planning_ref import PlanningRef
    n = 8
    ora = PlanningRef(n, "rate", seed=7)
    env = Handle("planning", "rate", n, seed=7)
    st, ps = env.get_state(), env.planning_get_state()
    np.testing.assert_allclose(st["root_states"].cpu().numpy(), ora.root_states.numpy(), atol=1e-6)
    np.testing.assert_allclose(ps["obstacles"][..., :3].cpu().numpy(), ora.obstacles.numpy(), atol=1e-5)
    assert np.array_equal(ps["obstacles"][..., 3].cpu().numpy().astype(np.int64), ora.variants.numpy())
    np.testing.assert_allclose(ps["goal"].cpu().numpy(), ora.goal_positions.numpy(), atol=1e-6)
    assert (env.reset_buf == 1).all() and env.image.shape == (n, 1, 212, 120)
    env.close()


def test_closed_loop_with_rendering_vs_oracle(Handle):
    from oracle.planning_ref import PlanningRef
    n, seed = 4, 3
    ora = PlanningRef(n, "rate", seed=seed)
    env = Handle("planning", "rate", n, seed=seed)
    rng = np.random.default_rng(0)
    for t in range(12):                      # renders at steps 4, 8, 12 (counter % 4 == 0)
        a = _actions(rng, n, t)
        o, _, rew, done, ex = ora.step(torch.from_numpy(a))
        env.step(torch.from_numpy(a).cuda())
        assert np.array_equal(env.reset_buf.cpu().numpy(), done.numpy()), f"done step {t}"
        np.testing.assert_allclose(env.get_state()["root_states"].cpu().numpy(), ora.root_states.numpy(), atol=1e-5)
        np.testing.assert_allclose(env.obs_buf.cpu().numpy(), o["observation"].numpy(), atol=2e-5, err_msg=f"obs {t}")
        img, ref = env.image.cpu().numpy(), o["image"].numpy()
        bad = np.abs(img - ref) > 2e-3
        assert bad.mean() < 0.01, f"step {t}: {bad.mean():.3%} image pixels differ"
        esdf = env.planning_get_state()["extra"][:, 3].cpu().numpy()
        np.testing.assert_allclose(esdf, ora.esdf_dist.numpy(), atol=5e-3, err_msg=f"esdf {t}")
        np.testing.assert_allclose(esdf, img.reshape(n, -1).min(1), atol=1e-6)      # esdf IS the min pixel (Q16)
        # reward terms that do not depend on the image are tight; esdf/alive/total follow the image tolerance
        for k in TERMS:
            tol = 5e-3 if k in ("esdf_reward", "reward") else 2e-5
            np.testing.assert_allclose(env.reward_terms[k].cpu().numpy(), ex["item_reward_info"][k].numpy(), atol=tol,
                                       err_msg=f"{k} step {t}")
    assert env.image.max() > 1.5       # rendered + blurred (kernel sum ~12): not the initial zeros
    env.close()


def test_parity_mode_reset_uniforms(Handle):
    from oracle.planning_ref import PlanningRef
    n, seed = 64, 9
    ora = PlanningRef(n, "vel", seed=seed)
    env = Handle("planning", "vel", n, seed=seed)
    ora.cam_rate = 10 ** 9
    rng = np.random.default_rng(2)
    n_done = 0
    for t in range(3):                       # keep the step count below the first scheduled render (step 4)
        a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
        a[:, 2] = 3.0 if t == 1 else a[:, 2]             # climb hard: leave the 0.6 m height corridor
        u = rng.random((n, 121)).astype(np.float32)
        o, _, rew, done, _ = ora.step(torch.from_numpy(a), reset_uniforms=torch.from_numpy(u))
        env.planning_step_with_uniforms(torch.from_numpy(a).cuda(), torch.from_numpy(u))
        assert np.array_equal(env.reset_buf.cpu().numpy(), done.numpy())
        n_done += int(done.sum())
        ps = env.planning_get_state()
        np.testing.assert_allclose(ps["obstacles"][..., :3].cpu().numpy(), ora.obstacles.numpy(), atol=1e-5)
        np.testing.assert_allclose(ps["goal"].cpu().numpy(), ora.goal_positions.numpy(), atol=1e-6)
        np.testing.assert_allclose(env.get_state()["root_states"].cpu().numpy(), ora.root_states.numpy(), atol=1e-5)
        np.testing.assert_allclose(env.rew_buf.cpu().numpy(), rew.numpy(), atol=2e-5)
    env.close()


def test_collision_flag_and_done(Handle):
    """Put the robot inside an obstacle: collisions = 1 and the env terminates (planning.py:286)."""
    from oracle import planning_ref as P
    n = 64
    env = Handle("planning", "rate", n, seed=1)
    ps = env.planning_get_state()
    table = torch.from_numpy(P.load_variant_table()).cuda()
    ob = ps["obstacles"]
    centre, axis, r, h = P.world_cylinders(ob[..., :3].cpu(), ob[..., 3].long().cpu(), P.load_variant_table())
    st = env.get_state()["root_states"].clone()
    # robot i sits on the axis of its obstacle 0 at the flight height (if the cylinder reaches it)
    c0, a0 = centre[:, 0], axis[:, 0]
    tpar = (1.5 - c0[:, 2]) / a0[:, 2]
    inside = tpar.abs() <= h[:, 0]
    pos = c0 + tpar[:, None] * a0
    st[:, 0:3] = pos.cuda()
    env.set_state(root_states=st, was_reset=torch.zeros(n, dtype=torch.int32))
    a = torch.zeros(n, 4, device="cuda"); a[:, 3] = -0.69
    env.step(a)
    hit = env.collisions.bool().cpu()
    assert inside.sum() > 10 and (hit[inside]).all()
    assert (env.reset_buf.cpu().bool()[inside]).all()
    env.close()


def test_full_size_properties(Handle):
    """BASELINE config 4 per-GPU size is 16 384 envs; here 4 096 envs x 24 steps: size-independent properties."""
    n = 4096
    env = Handle("planning", "rate", n, seed=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    tot = 0
    for t in range(24):
        a = torch.randn(n, 4, generator=g, device="cuda").clamp(-1, 1) * 0.3
        a[:, 3] = -0.69
        env.step(a)
        assert torch.equal(env.compact_reset_ids().long(), env.reset_buf.nonzero().squeeze(-1))
        tot += int(env.reset_buf.sum())
    assert torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all() and torch.isfinite(env.image).all()
    assert env.image.min() >= 0.0 and env.image.max() < 40.0
    ps, st = env.planning_get_state(), env.get_state()
    assert torch.allclose(ps["extra"][:, 3], env.image.reshape(n, -1).min(1).values)
    assert (ps["obstacles"][..., 0].abs() <= 8.0).all() and (ps["obstacles"][..., 1].abs() <= 4.0).all()
    assert (ps["goal"][:, 0] == 8.5).all() and (ps["goal"][:, 1].abs() <= 1.5).all()
    q = st["root_states"][:, 3:7]
    assert torch.allclose(q.norm(dim=-1), torch.ones(n, device="cuda"), atol=1e-5)
    assert tot > 0
    env.close()


def test_drop_in_api_and_ppo_epoch(Handle):
    """task_registry.make_env('planning') -> dict observations through AirGymRLGPUEnv -> one PPO epoch with the CNN policy."""
    import yaml, os
    from argparse import Namespace
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    from airgym_amd.lib.utils import vecenv
    import airgym_amd.envs  # noqa: F401  (registers the tasks, as `from airgym.envs import *` does in the reference's scripts)
    from airgym_amd.utils.task_registry import task_registry
    env, cfg = task_registry.make_env("planning", Namespace(num_envs=64, ctl_mode="rate", seed=3, sim_device="cuda:0", headless=True))
    obs, priv = env.reset()
    assert set(obs) == {"image", "observation"} and obs["image"].shape == (64, 1, 212, 120) and obs["observation"].shape == (64, 16)
    o2, _, rew, done, extras = env.step(torch.zeros(64, 4, device="cuda"))
    assert o2["image"] is env.full_camera_array and done.dtype == torch.int64
    assert set(extras["item_reward_info"]) >= {"esdf_reward", "reach_goal_reward", "heading_reward", "reward"}
    assert env.cam_resolution == (212, 120) and env.cam_channel == 1
    env.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = yaml.safe_load(open(os.path.join(repo, "scripts", "config", "ppo_planning.yaml")))["params"]
    c = params["config"]
    c.update(num_actors=256, horizon_length=8, minibatch_size=512, mini_epochs=2, max_epochs=1, write_summaries=False,
             print_stats=False, save_frequency=0, save_best_after=10 ** 9, device="cuda:0",
             train_dir="/tmp/airgym_runs_planning",
             env_config={"use_image": True, "num_envs": 256, "ctl_mode": "rate", "seed": 1, "sim_device": "cuda:0", "headless": True})
    agent = A2CAgent("run", params)
    assert agent.obs_shape == {"image": (1, 212, 120), "observation": (16,)}
    before = agent.flat_param.clone()
    agent.train()
    assert agent.epoch_num == 1 and torch.isfinite(agent.flat_param).all() and not torch.equal(before, agent.flat_param)
    assert agent._dedup and agent._frame_stores[0].abs().sum() > 0      # rendered frames reached the rollout's frame store


def test_golden_observations_reward_done(Handle, golden):
    """Planning.compute_observations + compute_quadcopter_reward of the REFERENCE (planning.py:186-307) replayed on the HIP
    post-physics kernel: recorded (state, goal, actions, pre_actions, pre_root_positions, progress, collisions, esdf) ->
    recorded 16-dim obs, 11 reward terms and the reset flags bit-exact (goal-reach rows, esdf 0.2999 / 0.3001 rows,
    progress 1597..1600 rows included).  `vel` handle: its action map is the identity, so the recorded processed
    actions pass through unchanged (the reward code itself has no ctl_mode branch)."""
    g = golden("planning_obs_reward")
    n = g["root_states"].shape[0]
    env = Handle("planning", "vel", n, seed=0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    env.set_state(root_states=t(g["root_states"]), progress=t((g["progress"] - 1).astype(np.int32)),   # the kernel does progress++
                  pre_actions=t(g["pre_actions"]), was_reset=torch.zeros(n, dtype=torch.int32))
    extra = torch.zeros(n, 5)
    extra[:, 0:3] = t(g["pre_root_positions"]); extra[:, 3] = t(g["esdf_dist"])
    env.planning_set_state(goal=t(g["goal"]), extra=extra)
    env.planning_eval_post(t(g["actions"]).cuda(), t(g["collisions"]))
    reset = env.reset_buf.cpu().numpy()
    assert np.array_equal(reset, g["reset"]), np.nonzero(reset != g["reset"])
    assert reset.sum() > 8 and (reset == 0).sum() > 8
    np.testing.assert_allclose(env.obs_buf.cpu().numpy(), g["obs"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(env.rew_buf.cpu().numpy(), g["reward"], rtol=0, atol=2e-5)
    for k in TERMS:
        np.testing.assert_allclose(env.reward_terms[k].cpu().numpy(), g["info_" + k], rtol=0, atol=2e-5, err_msg=k)
    assert env.reward_terms["reach_goal_reward"][0].item() == 200.0 and env.reward_terms["reach_goal_reward"][1].item() == 0.0
    assert env.reward_terms["alive_reward"][8:10].tolist() == [-1.0, 0.0]
    assert np.array_equal(env.collisions.cpu().numpy(), g["collisions"])
    env.close()


def test_config4_size_16384_envs_vs_oracle_slice(Handle):
    """BASELINE config 4 per-GPU size: Planning, 16 384 envs, CTBR.  Envs are independent and the RNG is keyed by the global
    env id, so a 4-env oracle with env_id_offset = 16 380 reproduces the LAST four envs of the full-size job; the rest of
    the batch is held to size-independent properties.  12 steps = 3 camera renders of all 16 384 envs."""
    from oracle.planning_ref import PlanningRef
    n, k, seed = 16384, 4, 5
    env = Handle("planning", "rate", n, seed=seed)
    ora = PlanningRef(k, "rate", seed=seed, env_id_offset=n - k)
    np.testing.assert_allclose(env.get_state()["root_states"][n - k:].cpu().numpy(), ora.root_states.numpy(), atol=1e-6)
    rng = np.random.default_rng(3)
    g = torch.Generator(device="cuda").manual_seed(1)
    tot = 0
    for t in range(12):
        a_full = torch.randn(n, 4, generator=g, device="cuda").clamp(-1, 1) * 0.3
        a_full[:, 3] = -0.69
        a_tail = _actions(rng, k, t)
        a_full[n - k:] = torch.from_numpy(a_tail).cuda()
        o, _, rew, done, ex = ora.step(torch.from_numpy(a_tail))
        env.step(a_full)
        assert np.array_equal(env.reset_buf[n - k:].cpu().numpy(), done.numpy()), f"done step {t}"
        np.testing.assert_allclose(env.get_state()["root_states"][n - k:].cpu().numpy(), ora.root_states.numpy(), atol=1e-5)
        np.testing.assert_allclose(env.obs_buf[n - k:].cpu().numpy(), o["observation"].numpy(), atol=2e-5)
        img, ref = env.image[n - k:].cpu().numpy(), o["image"].numpy()
        assert (np.abs(img - ref) > 2e-3).mean() < 0.01, f"step {t}"
        np.testing.assert_allclose(env.rew_buf[n - k:].cpu().numpy(), rew.numpy(), atol=5e-3)
        assert torch.equal(env.compact_reset_ids().long(), env.reset_buf.nonzero().squeeze(-1))
        tot += int(env.reset_buf.sum())
    assert torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all() and torch.isfinite(env.image).all()
    assert env.image.min() >= 0.0 and env.image.max() < 40.0 and env.image.max() > 1.5
    ps, st = env.planning_get_state(), env.get_state()
    assert torch.allclose(ps["extra"][:, 3], env.image.reshape(n, -1).min(1).values)     # esdf IS the min pixel (Q16)
    assert (ps["obstacles"][..., 0].abs() <= 8.0).all() and (ps["obstacles"][..., 1].abs() <= 4.0).all()
    assert (ps["goal"][:, 0] == 8.5).all() and (ps["goal"][:, 1].abs() <= 1.5).all()
    assert torch.allclose(st["root_states"][:, 3:7].norm(dim=-1), torch.ones(n, device="cuda"), atol=1e-5)
    assert tot > 0
    env.close()


def test_frozen_vae_model_forward_on_gpu(Handle, golden):
    """The frozen-VAE policy path (lib/network/vae_image_encoder.py:34-53, a2c_continuous_logstd_model.py:32-48,114-126)
    on the GPU: the encoder reproduces the reference ImgEncoder's recorded latents (deterministic weight fill), and the
    actor-critic consumes the HIP Planning env's own dict observation."""
    import math
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    from airgym_amd.lib.network.vae import FrozenVAEEncoder
    g = golden("vae_encoder")
    enc = FrozenVAEEncoder({"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                            "interpolation_mode": "bilinear"}, device="cuda")
    with torch.no_grad():
        for i, (name, p) in enumerate(sorted(enc.encoder.named_parameters())):
            kk = torch.arange(p.numel(), dtype=torch.float64)
            v = torch.cos(0.61803 * kk + i) * (0.7 / math.sqrt(p[0].numel())) if p.dim() > 1 else 0.02 * torch.sin(kk + i)
            p.copy_(v.reshape(p.shape).float())
    i_ = torch.arange(212, dtype=torch.float32).view(1, 1, 212, 1)
    j_ = torch.arange(120, dtype=torch.float32).view(1, 1, 1, 120)
    b_ = torch.arange(3, dtype=torch.float32).view(3, 1, 1, 1)
    img = (0.5 + 0.5 * torch.sin(0.05 * i_ + 0.11 * j_ + b_)).cuda()
    means = enc.encode(img)
    np.testing.assert_allclose(means.cpu().numpy(), g["means"], rtol=0, atol=2e-5)       # MIOpen conv vs the CPU recording
    params = {"network": {"separate": False, "mlp": {"units": [64, 128, 64], "activation": "elu"},
                          "space": {"continuous": {"fixed_sigma": True}},
                          "vae": {"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                                  "interpolation_mode": "bilinear", "return_sampled_latent": False}},
              "config": {"normalize_input": True, "normalize_value": True}}
    m = ModelA2CContinuousLogStd(params, {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}}).cuda()
    n = 64
    env = Handle("planning", "rate", n, seed=2)
    env.planning_render_next_step()
    env.step(torch.zeros(n, 4, device="cuda"))
    obs = {"image": env.image, "observation": env.obs_buf}
    m.eval()
    with torch.no_grad():
        out = m({"is_train": False, "obs": obs})
    assert out["actions"].shape == (n, 4) and out["values"].shape == (n, 1)
    assert torch.isfinite(out["mus"]).all() and torch.isfinite(out["values"]).all()
    env.close()


def test_player_plays_a_reference_layout_checkpoint_in_the_hip_planning_env(Handle, golden, tmp_path):
    """SURVEY 8(f)-2 on the GPU: a checkpoint in the key layout of the reference's trained/planning_cnn_rate.pth (names and
    shapes from the fixture's manifest; deterministic fill, the 778 KB file itself does not travel) -> A2CPlayer.restore
    (players.py:372-388) -> deterministic play loop (players.py:204-290) on the HIP Planning env with dict observations."""
    import json
    import os
    import yaml
    from airgym_amd.lib.agent.players import A2CPlayer
    manifest = json.loads(str(golden("planning_checkpoint")["manifest"]))
    sd = {}
    for i, (k, meta) in enumerate(sorted(manifest.items())):
        n = int(np.prod(meta["shape"])) if meta["shape"] else 1
        v = 0.05 * torch.cos(0.61803 * torch.arange(n, dtype=torch.float64) + i)
        if k.endswith("running_var") or k.endswith(".count"):
            v = v.abs() + 1.0
        if "num_batches_tracked" in k:
            v = torch.full((n,), 7.0, dtype=torch.float64)
        sd[k] = v.reshape(meta["shape"]).to(getattr(torch, meta["dtype"]))
    fn = str(tmp_path / "planning_like.pth")
    torch.save({"model": sd, "epoch": 200, "frame": 4915200}, fn)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = yaml.safe_load(open(os.path.join(repo, "scripts", "config", "ppo_planning.yaml")))["params"]
    params["config"].update(num_actors=64, device="cuda:0",
                            env_config={"use_image": True, "num_envs": 64, "ctl_mode": "rate", "seed": 2, "sim_device": "cuda:0",
                                        "headless": True},
                            player={"deterministic": True, "games_num": 10 ** 6, "max_steps": 24, "print_stats": False})
    p = A2CPlayer(params)
    assert isinstance(p.obs_shape, dict) and p.obs_shape["image"] == (1, 212, 120) and p.obs_shape["observation"] == (16,)
    p.restore(fn)
    for k, v in p.model.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k                        # strict load, key for key
    obs = p.env.reset()
    act = p.get_action(obs)
    assert act.shape == (64, 4) and act.abs().max() <= 1.0 and torch.isfinite(act).all()
    assert torch.equal(act, p.get_action(obs))                       # deterministic: clamp(mu)
    res = p.run(print_every=8)
    assert np.isfinite(res["av_reward"]) and res["games"] > 0        # an untrained fill crashes or leaves the corridor quickly


def _vae_agent(envs, cache, horizon=8, seed=0):
    import os
    import yaml
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = yaml.safe_load(open(os.path.join(repo, "scripts", "config", "ppo_planning.yaml")))["params"]
    c = params["config"]
    c.update(num_actors=envs, horizon_length=horizon, mini_epochs=2, minibatch_size=envs * horizon // 2, device="cuda:0",
             max_epochs=-1, write_summaries=False, print_stats=False, save_frequency=0, save_best_after=10 ** 9,
             cache_frozen_features=cache, use_hip_graph=False)
    c["env_config"] = {"use_image": True, "num_envs": envs, "ctl_mode": "rate", "seed": seed, "sim_device": "cuda:0",
                       "headless": True}
    params["network"].pop("cnn", None)
    params["network"]["vae"] = {"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                                "interpolation_mode": "bilinear", "return_sampled_latent": False}
    params["seed"] = seed
    torch.manual_seed(seed)
    agent = A2CAgent("vae_cache_test", params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    return agent


def test_frozen_encoder_features_are_cached_per_rendered_image():
    """Planning with the frozen depth VAE: the rollout keeps [N, 64] features instead of images, re-encodes only on the steps
    the camera ran (every 4th, planning.py:153-156), and the cached features equal a fresh encoding of the image the env
    holds; the update runs on the features (no encoder), trains, and the image normaliser still sees the rendered images."""
    agent = _vae_agent(64, True)
    assert agent._cache_latents and set(agent.obs_buf) == {"observation", "latent"}
    assert agent.obs_buf["latent"].shape == (9, 64, 64)
    calls = []
    enc = agent.model._frozen[0]
    orig = enc.encode
    enc.encode = lambda im: (calls.append(im.shape[0]), orig(im))[1]
    rms = agent.model.running_mean_std.running_mean_std["image"]
    count0 = float(rms.count)
    agent.epoch_num = 1
    st = agent.train_epoch()
    # 8 steps -> 2 camera steps (+ slot 0 re-encoded at the start of the rollout); nothing in the update
    assert calls == [64, 64, 64], calls
    assert float(rms.count) == count0 + 2 * 64
    # the rollout's last slot (carried into slot 0) == a fresh encoding of the env's current image under the normaliser state
    # the ROLLOUT used; the update has moved the normaliser since, so compare through a second rollout start instead
    lat_last = agent.obs_buf["latent"][0].clone()
    agent.model.eval()
    fresh = agent.model.encode_image(agent._hip_env.image)
    assert fresh.shape == lat_last.shape and torch.isfinite(fresh).all()
    calls.clear()
    batch = agent.play_steps()
    assert torch.allclose(batch["obses"]["latent"].view(64, 8, 64)[:, 0], fresh, atol=1e-6)
    lat = batch["obses"]["latent"].view(64, 8, 64)
    ren = [t for t in range(1, 8) if not torch.equal(lat[:, t], lat[:, t - 1])]
    assert len(ren) <= 2 and len(calls) == 1 + len(ren) + (2 - len(ren))          # features change only on camera steps
    for k in ("a_loss", "c_loss", "kl"):
        assert st[k] == st[k]
    assert "image" not in batch["obses"]


def test_cached_features_match_uncached_training_step():
    """Same seeds, cache on / off: with the input normaliser frozen (normalize_input false) the two paths see identical
    features, so one PPO epoch produces the same losses and the same parameters."""
    outs = []
    for cache in (True, False):
        agent = _vae_agent(32, cache, horizon=4, seed=3)
        agent.normalize_input = False
        agent.model.normalize_input = False
        agent.epoch_num = 1
        st = agent.train_epoch()
        outs.append((st, agent.flat_param.clone()))
    (a, pa), (b, pb) = outs
    for k in ("a_loss", "c_loss", "entropy"):
        assert abs(float(a[k]) - float(b[k])) <= 1e-5 * max(1.0, abs(float(b[k]))), (k, a[k], b[k])
    assert (pa - pb).abs().max().item() <= 1e-5


def test_last_step_rendered_follows_the_camera_schedule(Handle):
    """ag_planning_last_step_rendered: true exactly on the steps that rewrote the depth image (every 4th step, planning.py:153-156,
    or when a render was forced), and the image buffer is bit-identical across the steps in between; Balloon has no camera."""
    n = 32
    env = Handle("planning", "rate", n, seed=5)
    a = torch.zeros(n, 4, device="cuda"); a[:, 3] = -0.69
    flags, changed = [], []
    prev = env.image.clone()
    for t in range(9):
        env.step(a)
        flags.append(env.last_step_rendered())
        changed.append(not torch.equal(env.image, prev))
        prev = env.image.clone()
    assert sum(flags) == 2 and flags == changed, (flags, changed)
    idx = [i for i, f in enumerate(flags) if f]
    assert idx[1] - idx[0] == 4
    env.planning_render_next_step()
    env.step(a)
    assert env.last_step_rendered()
    env.close()
    b = Handle("balloon", "rate", 8, seed=1)
    with pytest.raises(RuntimeError):
        b.last_step_rendered()
    b.close()


def _cnn_agent(envs, dedup, horizon=8, seed=3, minibatches=3):
    import os
    import yaml
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = yaml.safe_load(open(os.path.join(repo, "scripts", "config", "ppo_planning.yaml")))["params"]
    c = params["config"]
    c.update(num_actors=envs, horizon_length=horizon, mini_epochs=2, minibatch_size=envs * horizon // minibatches,
             device="cuda:0", max_epochs=-1, write_summaries=False, print_stats=False, save_frequency=0,
             save_best_after=10 ** 9, dedup_frames=dedup, use_hip_graph=False)
    c["env_config"] = {"use_image": True, "num_envs": envs, "ctl_mode": "rate", "seed": seed, "sim_device": "cuda:0",
                       "headless": True}
    params["seed"] = seed
    torch.manual_seed(seed)
    agent = A2CAgent("dedup_test", params)
    agent.init_tensors()
    torch.manual_seed(seed + 1)
    agent.obs = agent.env_reset()
    return agent


def test_weighted_relu_batchnorm_kernels():
    """ReLU + BatchNorm2d with per-image multiplicities on csrc/cnn_kernels.hip == the same layer applied to the expanded
    batch (every image repeated weights[i] times, upstream gradients of the copies summed): outputs, running statistics,
    input / gamma / beta gradients, against a float64 evaluation."""
    from airgym_amd.lib.network.fused_relu_bn import relu_batchnorm
    g = torch.Generator(device="cuda").manual_seed(0)
    for (n, c, h, w) in [(13, 16, 53, 30), (9, 64, 27, 15), (5, 32, 7, 3)]:
        x = torch.randn(n, c, h, w, device="cuda", generator=g)
        cnt = torch.randint(1, 5, (n,), device="cuda", generator=g)
        inv = torch.repeat_interleave(torch.arange(n, device="cuda"), cnt)
        up = torch.randn(int(cnt.sum()), c, h, w, device="cuda", generator=g)        # one upstream gradient per COPY
        # float64 reference on the expanded batch
        bn64 = torch.nn.BatchNorm2d(c).cuda().double()
        with torch.no_grad():
            bn64.weight.uniform_(0.5, 1.5); bn64.bias.uniform_(-0.5, 0.5)
        x64 = x.double().requires_grad_(True)
        y64 = bn64(torch.relu(x64[inv]))
        (y64 * up.double()).sum().backward()
        # the HIP node on the distinct images, upstream gradient summed over the copies
        bn = torch.nn.BatchNorm2d(c).cuda()
        with torch.no_grad():
            bn.weight.copy_(bn64.weight); bn.bias.copy_(bn64.bias)
        xg = x.clone().requires_grad_(True)
        y = relu_batchnorm(xg, bn, cnt.float())
        gsum = torch.zeros_like(x).index_add_(0, inv, up)
        (y * gsum).sum().backward()
        assert torch.allclose(y.double()[inv], y64, atol=2e-5)
        assert torch.allclose(bn.running_mean.double(), bn64.running_mean, atol=1e-6)
        assert torch.allclose(bn.running_var.double(), bn64.running_var, atol=1e-6)
        sc = x64.grad.abs().max().item()
        assert (xg.grad.double() - x64.grad).abs().max().item() <= 2e-5 * sc + 1e-6
        assert torch.allclose(bn.weight.grad.double(), bn64.weight.grad, rtol=1e-4, atol=1e-3 * bn64.weight.grad.abs().max().item())
        assert torch.allclose(bn.bias.grad.double(), bn64.bias.grad, rtol=1e-4, atol=1e-3 * bn64.bias.grad.abs().max().item())


def test_frame_dedup_is_the_same_update_on_a_quarter_of_the_images():
    """dedup_frames (default for camera tasks with the trainable CNN): the rollout keeps only the rendered frames and runs the
    CNN once per frame; a minibatch runs the CNN on its DISTINCT images with statistics weighted by their multiplicities.
    Against dedup_frames: false (every sample's image stored and convolved, as the reference does): identical rollouts, and
    the same loss, the same gradient of every parameter and the same normaliser / BatchNorm statistics for a minibatch."""
    out = []
    for dedup in (True, False):
        agent = _cnn_agent(48, dedup, horizon=8, seed=5, minibatches=3)      # 128-sample minibatches cut through env trajectories
        assert bool(getattr(agent, "_dedup", False)) == dedup
        calls = []
        cnn = agent.model.actor_cnn
        orig = cnn.forward
        cnn.forward = lambda x, weights=None, norm=None, index=None: (
            calls.append(x.shape[0] if index is None else index.shape[0]), orig(x, weights, norm, index))[1]
        torch.manual_seed(11)
        batch = agent.play_steps()
        rollout_calls = list(calls)
        agent.model.train()
        agent.curr_frames = batch.pop("played_frames")
        agent.prepare_dataset(batch)
        agent.model.running_mean_std.eval()
        agent.model.update_stats = True
        calls.clear()
        mb = agent.dataset[1]
        a, c, e, b_, _, _ = agent._loss_and_backward(mb)
        out.append({"dedup": dedup, "actions": batch["actions"].clone(), "values": batch["values"].clone(),
                    "grad": agent.flat_grad.clone(), "loss": (float(a), float(c)), "rollout_calls": rollout_calls,
                    "update_images": list(calls),
                    "img_mean": agent.model.running_mean_std.running_mean_std["image"].running_mean.clone(),
                    "img_count": float(agent.model.running_mean_std.running_mean_std["image"].count),
                    "bn": [m.running_var.clone() for m in agent.model.actor_cnn.features if isinstance(m, torch.nn.BatchNorm2d)],
                    "mem": sum(v.numel() for v in (agent._frame_stores if dedup else [agent.obs_buf["image"]]))})
        agent.vec_env.env.hip.close()
    d, f = out
    # the rollouts are the same rollout (same policy outputs from cached vs recomputed features)
    assert torch.allclose(d["actions"], f["actions"], atol=1e-5) and torch.allclose(d["values"], f["values"], atol=1e-4)
    # CNN work: per rendered frame in the rollout (2-3 of 8 steps + slot 0) instead of per step; ~1/3 of the images in the update
    assert len(d["rollout_calls"]) <= 4 and len(f["rollout_calls"]) >= 8
    assert d["update_images"][0] < 0.5 * f["update_images"][0], (d["update_images"], f["update_images"])
    # the minibatch: same losses, same gradients, same statistics
    assert abs(d["loss"][0] - f["loss"][0]) <= 1e-5 and abs(d["loss"][1] - f["loss"][1]) <= 1e-4 * max(1.0, abs(f["loss"][1]))
    scale = f["grad"][:-1].abs().max().item()
    assert (d["grad"] - f["grad"])[:-1].abs().max().item() <= 2e-4 * scale, ((d["grad"] - f["grad"]).abs().max().item(), scale)
    assert d["img_count"] == f["img_count"] and torch.allclose(d["img_mean"], f["img_mean"], atol=1e-6)
    for x, y in zip(d["bn"], f["bn"]):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-6)
    # image memory: two stores of ceil((H + 1) / 4) + 2 frames instead of H + 1 images per env (H = 24: 18 vs 25; this H = 8 test: 10 vs 9)
    assert d["mem"] == 2 * ((8 + 1 + 3) // 4 + 2) * 48 * 212 * 120 and f["mem"] == 9 * 48 * 212 * 120


def test_frame_dedup_rollout_after_an_update_runs_the_updated_cnn():
    """Across epochs: slot 0 of rollout k + 1 shows the image the env held when rollout k ended, and the PPO update in between
    moved the CNN weights, the BatchNorm running statistics and the image normaliser.  The reference runs the CURRENT CNN on
    every step (lib/agent/a2c_base.py:357-369), so the cached features of that frame must be recomputed before the first step:
    with dedup on and off the SECOND rollout must be the same rollout too (stale features differ at the 1e-1 level - six
    optimizer steps move the eval-mode BatchNorm statistics almost half way from their initial values)."""
    out = []
    for dedup in (True, False):
        agent = _cnn_agent(32, dedup, horizon=8, seed=7, minibatches=2)
        for ep in range(2):
            torch.manual_seed(100 + ep)
            agent.epoch_num += 1
            agent.train_epoch()
        torch.manual_seed(200)
        batch = agent.play_steps()                     # third rollout, behind two updates
        out.append({k: batch[k].clone() for k in ("values", "mus", "neglogpacs")})
        agent.vec_env.env.hip.close()
    d, f = out
    H = 8
    v_d, v_f = d["values"].view(32, H), f["values"].view(32, H)            # env-major: [env, step]
    m_d, m_f = d["mus"].view(32, H, -1), f["mus"].view(32, H, -1)
    # the steps that show slot 0's frame (before the first render of the rollout) are the ones stale features would hit
    assert (v_d[:, :2] - v_f[:, :2]).abs().max().item() < 5e-3, (v_d[:, :2] - v_f[:, :2]).abs().max().item()
    assert (m_d[:, :2] - m_f[:, :2]).abs().max().item() < 5e-3
    assert (v_d - v_f).abs().max().item() < 2e-2 and (m_d - m_f).abs().max().item() < 2e-2
