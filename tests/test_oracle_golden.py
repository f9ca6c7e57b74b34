"""Pin the oracle against golden vectors produced by the reference's own code
(tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import ppo_ref
from oracle import rigid_body
from oracle.hovering_ref import HoveringRef, compute_yaw_diff, quat_axis, tensor_clamp
from oracle.tracking_ref import TrackingRef

CLS = {"hovering": HoveringRef, "tracking": TrackingRef}


def t(a):
    return torch.from_numpy(np.asarray(a))


def make(cls, n, ctl_mode, g, target=None):
    env = cls(n, ctl_mode=ctl_mode, seed=0, target_state=None if target is None else list(target))
    env.root_states = t(g["root_states"]).clone()
    env.progress_buf = t(g["progress"]).clone()
    return env


def test_helpers(golden):
    g = golden("helpers")
    q = t(g["q"])
    for ax in range(3):
        assert torch.equal(quat_axis(q, ax), t(g[f"quat_axis{ax}"]))
    assert torch.equal(compute_yaw_diff(t(g["a"]), t(g["b"])), t(g["yaw_diff"]))
    assert torch.equal(tensor_clamp(t(g["t"]), t(g["lo"]), t(g["hi"])), t(g["clamp"]))


@pytest.mark.parametrize("task", ["hovering", "tracking"])
def test_observations(golden, task):
    g = golden(f"{task}_obs")
    n = g["root_states"].shape[0]
    env = make(CLS[task], n, "rate", g, g["target_state"])
    obs = env.compute_observations(t(g["noise"]))
    assert obs.shape == g["obs"].shape
    assert torch.equal(obs, t(g["obs"]))
    if task == "tracking":
        assert torch.equal(env.ref_positions, t(g["ref_positions"]))


def test_lemniscate(golden):
    g = golden("lemniscate")
    env = TrackingRef(64, "vel")
    env.progress_buf = t(g["progress"])
    assert torch.equal(env.compute_traj_lemniscate(), t(g["ref"]))


@pytest.mark.parametrize("task", ["hovering", "tracking"])
@pytest.mark.parametrize("ctl_mode", ["rate", "vel", "atti", "pos", "prop"])
def test_reward_and_done(golden, task, ctl_mode):
    g = golden(f"{task}_reward_{ctl_mode}")
    n = g["root_states"].shape[0]
    env = make(CLS[task], n, ctl_mode, g)
    env.actions = t(g["actions"])
    env.pre_actions = t(g["pre_actions"])
    env.cmd_thrusts = t(g["cmd_thrusts"]).to(torch.float32)   # oracle is f32; reference controller is f64 (Q5)
    if task == "tracking":
        env.ref_positions = env.compute_traj_lemniscate()
    reward, reset, info = env.compute_quadcopter_reward()
    # termination flags are integers: bit-exact
    assert torch.equal(reset, t(g["reset"]))
    assert reset.sum() > 4 and (reset == 0).sum() > 4
    # reward: the reference promotes to f64 through the effort term; f32 oracle within 1 ulp-ish
    np.testing.assert_allclose(reward.numpy(), g["reward"], rtol=0, atol=1e-6)
    for k, v in info.items():
        if torch.is_tensor(v):
            np.testing.assert_allclose(v.numpy(), g["info_" + k], rtol=0, atol=1e-6, err_msg=k)


def test_hovering_boundary_rows(golden):
    """rows 4..11 of the hovering reward fixtures sit either side of each termination threshold."""
    g = golden("hovering_reward_rate")
    assert list(g["reset"][4:12]) == [0, 1, 0, 1, 0, 1, 0, 1]
    # progress rows 0..3 = max_len-3, -2, -1, max_len -> done from max_len-1 on (hovering.py:435)
    pos_ok = np.linalg.norm(g["root_states"][:4, :3], axis=1) <= 4
    exp = np.array([0, 0, 1, 1])
    got = g["reset"][:4]
    assert ((got == exp) | ~pos_ok | (got == 1)).all()


@pytest.mark.parametrize("task", ["hovering", "tracking"])
def test_reset_distribution(golden, task):
    g = golden(f"{task}_reset")
    n = g["uniforms"].shape[0]
    env = CLS[task](n, "rate")
    env.progress_buf[:] = 7
    env.pre_actions[:] = 1
    env.reset_buf[:] = 0
    env.reset_idx(torch.arange(n), t(g["uniforms"]))
    np.testing.assert_allclose(env.root_states.numpy(), g["root_states"], rtol=0, atol=1e-7)
    assert torch.equal(env.reset_buf, t(g["reset_buf"]))
    assert torch.equal(env.progress_buf, t(g["progress"]))
    assert torch.equal(env.pre_actions, t(g["pre_actions"]))


def test_wrench_assembly(golden):
    """forces/torques the reference hands to PhysX (hovering.py:256-281), reduced to the
    composite-body wrench about the COM, must equal rigid_body.body_wrench_from_cmd."""
    g = golden("wrench")
    cmd = t(g["cmd"]).to(torch.float32)
    was_reset = t(g["reset_buf"])
    fz, tau = rigid_body.body_wrench_from_cmd(cmd, (was_reset == 0).float())
    forces = g["forces"].astype(np.float64)      # [N,5,3] body 0 = base, 1..4 props, LOCAL frame
    torques = g["torques"].astype(np.float64)
    assert np.abs(forces[:, 0]).max() == 0 and np.abs(forces[:, :, :2]).max() == 0
    assert np.abs(torques[:, :, :2]).max() == 0
    arm = rigid_body.ROTOR_ARM
    r = np.array([[arm, -arm, 0.024], [-arm, arm, 0.024], [arm, arm, 0.024], [-arm, -arm, 0.024]])
    fz_ref = forces[:, 1:5, 2].sum(1)
    tau_ref = np.cross(r[None], forces[:, 1:5]).sum(1) + torques[:, 1:5].sum(1)
    np.testing.assert_allclose(fz.numpy(), fz_ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(tau.numpy(), tau_ref, rtol=0, atol=1e-6)
    # thrust is zeroed for flagged envs, reaction torque is not (Q2)
    flagged = g["reset_buf"] == 1
    assert flagged.any() and np.abs(fz.numpy()[flagged]).max() == 0
    assert np.abs(tau.numpy()[flagged, 2]).max() > 0


@pytest.mark.parametrize("mode", ["rate", "atti", "vel", "pos"])
def test_action_map_and_quat_canonicalisation(golden, mode):
    g = golden("action_map")
    a_in = t(g[f"{mode}_in"])
    env = HoveringRef(a_in.shape[0], mode)
    env.root_states[:, 3:7] = t(g[f"{mode}_quat_in"])
    env.pre_physics_step(a_in)
    assert torch.equal(env.actions, t(g[f"{mode}_out"]))
    assert torch.equal(env.root_states[:, 3:7], t(g[f"{mode}_quat_out"]))


def test_ppo_numerics(golden):
    g = golden("ppo")
    a = ppo_ref.actor_loss(t(g["old_nlp"]), t(g["new_nlp"]), t(g["adv"]), 0.2)
    assert torch.equal(a, t(g["a_loss"]))
    c = ppo_ref.critic_loss(t(g["vp"]), t(g["v"]), 0.2, t(g["ret"]), False)
    assert torch.equal(c, t(g["c_loss"]))
    c2 = ppo_ref.critic_loss(t(g["vp"]), t(g["v"]), 0.2, t(g["ret"]), True)
    assert torch.equal(c2, t(g["c_loss_clip"]))
    kl = ppo_ref.policy_kl(t(g["mu0"]), t(g["s0"]), t(g["mu1"]), t(g["s1"]), True)
    assert torch.equal(kl, t(g["kl"]))
    klnr = ppo_ref.policy_kl(t(g["mu0"]), t(g["s0"]), t(g["mu1"]), t(g["s1"]), False)
    assert torch.equal(klnr, t(g["kl_nr"]))
    assert "b_loss" in g.files
    assert torch.equal(ppo_ref.bound_loss(t(g["mu_big"])), t(g["b_loss"]))
    rms = ppo_ref.RunningMeanStdRef((6,))
    for i in range(3):
        rms.update(t(g[f"rms_x{i}"]))
        assert torch.equal(rms.normalize(t(g[f"rms_x{i}"])), t(g[f"rms_y{i}"]))
    assert torch.equal(rms.running_mean, t(g["rms_mean"]))
    assert torch.equal(rms.running_var, t(g["rms_var"]))
    assert torch.equal(rms.count, t(g["rms_count"]))
    assert torch.equal(rms.normalize(t(g["rms_x0"])), t(g["rms_y_eval"]))
    kls = g["sched_kls"]
    lrs = []
    for start in (3e-4, 1e-6, 1e-2):
        for k in kls:
            lrs.append(ppo_ref.adaptive_lr(start, float(k)))
    assert lrs == list(g["sched_lrs"])
    ws = [t(g[f"mlp_w{i}"]) for i in range(3)]
    bs = [t(g[f"mlp_b{i}"]) for i in range(3)]
    assert torch.equal(ppo_ref.mlp_forward(t(g["mlp_x"]), ws, bs), t(g["mlp_y"]))


def test_gae(golden):
    g = golden("gae")
    advs = ppo_ref.gae(t(g["fdones"]), t(g["last_values"]), t(g["mb_fdones"]), t(g["mb_values"]),
                       t(g["mb_rewards"]), 0.99, 0.95)
    assert torch.equal(advs, t(g["advs"]))


# ----------------------------------------------------------------------------- Planning
def test_planning_observations_reward_done(golden):
    from oracle.planning_ref import PlanningRef
    g = golden("planning_obs_reward")
    n = g["root_states"].shape[0]
    env = PlanningRef(n, "rate")
    env.root_states = t(g["root_states"]).clone()
    env.goal_positions = t(g["goal"]).clone()
    env.actions = t(g["actions"]).clone()
    env.pre_actions = t(g["pre_actions"]).clone()
    env.pre_root_positions = t(g["pre_root_positions"]).clone()
    env.progress_buf = t(g["progress"]).clone()
    env.collisions = t(g["collisions"]).clone()
    env.esdf_dist = t(g["esdf_dist"]).clone()
    env.compute_observations()
    assert torch.equal(env.obs_buf, t(g["obs"]))
    assert torch.equal(env.related_dist, t(g["related_dist"]))
    reward, reset, info = env.compute_quadcopter_reward()
    assert torch.equal(reset, t(g["reset"])) and reset.sum() > 8 and (reset == 0).sum() > 8
    assert torch.equal(reward, t(g["reward"]))
    for k, v in info.items():
        assert torch.equal(v, t(g["info_" + k])), k
    # goal within 0.3 m -> +200 and done (rows 0, 2); just outside (row 1) -> no bonus
    assert g["info_reach_goal_reward"][0] == 200.0 and g["info_reach_goal_reward"][1] == 0.0
    assert list(g["info_alive_reward"][8:10]) == [-1.0, 0.0]          # esdf 0.2999 / 0.3001


def test_planning_reset(golden):
    from oracle.planning_ref import NUM_OBSTACLES, PlanningRef
    g = golden("planning_reset")
    k = g["ux"].shape[0]
    env = PlanningRef(k, "rate")
    env.progress_buf[:] = 9; env.pre_actions[:] = 1; env.prev_related_dist[:] = 1; env.pre_root_positions[:] = 1
    env.reset_buf[:] = 0
    # reference draws for 41 assets (asset 0 = goal ball, overwritten afterwards); obstacles are assets 1..40
    u = np.concatenate([np.stack((g["ux"][:, 1:, 0], g["uy"][:, 1:, 0], g["uyaw"][:, 1:, 0]), -1).reshape(k, -1),
                        g["ugoal"]], axis=1).astype(np.float32)
    env.reset_idx(torch.arange(k), t(u))
    a = g["asset_states"]
    np.testing.assert_allclose(env.obstacles[:, :, 0].numpy(), a[:, 1:, 0], atol=1e-6)
    np.testing.assert_allclose(env.obstacles[:, :, 1].numpy(), a[:, 1:, 1], atol=1e-6)
    assert np.abs(a[:, 1:, 2]).max() == 0
    # obstacle yaw: the reference stores a quaternion (0, 0, sin(yaw/2), cos(yaw/2)) up to sign
    half = 0.5 * env.obstacles[:, :, 2].numpy()
    qz, qw = a[:, 1:, 5], a[:, 1:, 6]
    sgn = np.sign(qw * np.cos(half) + qz * np.sin(half))
    np.testing.assert_allclose(np.sin(half), sgn * qz, atol=2e-6)
    np.testing.assert_allclose(np.cos(half), sgn * qw, atol=2e-6)
    np.testing.assert_allclose(env.goal_positions.numpy(), a[:, 0, 0:3], atol=1e-6)
    np.testing.assert_allclose(env.root_states.numpy(), g["root_states"], atol=1e-6)
    for name, ref in (("reset_buf", "reset_buf"), ("progress_buf", "progress"), ("pre_actions", "pre_actions"),
                      ("pre_root_positions", "pre_root_positions"), ("prev_related_dist", "prev_related_dist")):
        assert torch.equal(getattr(env, name), t(g[ref])), name


def test_planning_depth_post_processing(golden):
    from oracle.planning_ref import post_process_depth
    g = golden("planning_images")
    cam = -t(g["cam"])                                   # the reference negates IsaacGym's depth tensor
    for e in range(cam.shape[0]):
        img = post_process_depth(cam[e], t(g["add"][e]), t(g["mul"][e]), t(g["kernel"][e]))
        assert torch.equal(img, t(g["image"][e]))


def test_planning_scene_and_raycast_sanity():
    """The ray-caster is the build's spec (IsaacGym's rasteriser is closed): geometric self-checks."""
    from oracle import planning_ref as P
    tab = P.load_variant_table()
    assert tab.shape == (100, 8) and np.allclose(np.linalg.norm(tab[:, 3:6], axis=1), 1.0, atol=1e-6)
    # one vertical cylinder 2 m ahead of a level camera at the origin height 1.5
    centre = torch.tensor([[2.15, 0.0, 1.5]]); axis = torch.tensor([[0.0, 0.0, 1.0]])
    r = torch.tensor([0.1]); h = torch.tensor([2.0])
    img = P.render_depth_one(torch.tensor([0.0, 0.0, 1.4]), torch.tensor([0.0, 0, 0, 1]), centre, axis, r, h,
                             torch.tensor([100.0, 0, 0]))
    c = img[P.CAM_H // 2, P.CAM_W // 2].item()
    assert abs(c - 1.9) < 5e-3                           # camera sits 0.15 m ahead of the body: 2.15 - 0.15 - 0.1
    assert torch.isinf(img[P.CAM_H // 2, 5])             # far left column looks past the cylinder (and past 5 m)
    assert img[-1, P.CAM_W // 2] < 4.0                   # bottom row sees the ground plane
    d = P.point_capped_cylinder_distance(torch.tensor([[2.15, 0.3, 1.5], [2.15, 0.0, 3.8]]),
                                         centre[None].expand(2, 1, 3), axis[None].expand(2, 1, 3), r[None].expand(2, 1),
                                         h[None].expand(2, 1))
    assert abs(d[0, 0].item() - 0.2) < 1e-6 and abs(d[1, 0].item() - 0.3) < 1e-6
