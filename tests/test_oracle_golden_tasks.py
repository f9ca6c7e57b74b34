"""Pin oracle/custom_ref.py (Balloon, Avoid) against golden vectors recorded from the
REFERENCE's own methods (tests/golden/make_golden_tasks.py)."""
import numpy as np
import torch

from oracle.custom_ref import AvoidRef, BalloonRef


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_balloon_observations_reward_done(golden):
    g = golden("balloon_obs_reward")
    n = g["root_states"].shape[0]
    env = BalloonRef(n, "rate")
    env.root_states = t(g["root_states"]).clone()
    env.balloon_positions = t(g["balloon"]).clone()
    env.actions = t(g["actions"]).clone()
    env.pre_actions = t(g["pre_actions"]).clone()
    env.pre_root_positions = t(g["pre_root_positions"]).clone()
    env.progress_buf = t(g["progress"]).clone()
    obs = env.compute_observations(t(g["noise"]))
    assert torch.equal(obs, t(g["obs"]))
    reward, reset, info = env.compute_quadcopter_reward()
    assert torch.equal(reset, t(g["reset"])) and reset.sum() > 20 and (reset == 0).sum() > 20
    assert torch.equal(reward, t(g["reward"]))
    for k, v in info.items():
        assert torch.equal(v, t(g["info_" + k])), k
    # threshold rows: hit radius, x-overshoot, range, altitude band, backwards flight, thrust channel outside [-1, 1]
    assert list(g["reset"][8:20]) == [1, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1]
    assert list(g["info_hit_reward"][8:10]) == [800, 0]
    assert list(g["reset"][4:8]) == [0, 0, 1, 1] or (np.asarray(g["reset"][4:8]) >= np.array([0, 0, 1, 1])).all()


def test_balloon_reset(golden):
    g = golden("balloon_reset")
    k = g["uniforms"].shape[0]
    env = BalloonRef(k, "rate")
    env.progress_buf[:] = 9; env.pre_actions[:] = 1; env.pre_root_positions[:] = 1; env.reset_buf[:] = 0
    env.reset_idx(torch.arange(k), t(g["uniforms"]))
    np.testing.assert_allclose(env.root_states.numpy(), g["root_states"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(env.balloon_positions.numpy(), g["balloon"], rtol=0, atol=1e-7)
    for name, ref in (("reset_buf", "reset_buf"), ("progress_buf", "progress"), ("pre_actions", "pre_actions"),
                      ("pre_root_positions", "pre_root_positions")):
        assert torch.equal(getattr(env, name), t(g[ref])), name


def test_avoid_observations_reward_done(golden):
    g = golden("avoid_obs_reward")
    n = g["root_states"].shape[0]
    env = AvoidRef(n, "rate")
    env.root_states = t(g["root_states"]).clone()
    env.actions = t(g["actions"]).clone()
    env.pre_actions = t(g["pre_actions"]).clone()
    env.progress_buf = t(g["progress"]).clone()
    env.collisions = t(g["collisions"]).clone()
    obs = env.compute_observations()
    assert torch.equal(obs, t(g["obs"]))
    reward, reset, info = env.compute_quadcopter_reward()
    assert torch.equal(reset, t(g["reset"])) and reset.sum() > 20 and (reset == 0).sum() > 20
    assert torch.equal(reward, t(g["reward"]))
    for k, v in info.items():
        assert torch.equal(v, t(g["info_" + k])), k
    assert list(g["reset"][8:16]) == [0, 1, 0, 1, 0, 1, 0, 1]          # z 0.3 / 1.7, range 2 m, roll 90 deg: either side
    assert set(np.unique(g["info_alive_reward"])) == {-500.0, 0.5}


def test_avoid_reset_and_throw(golden):
    g = golden("avoid_reset")
    k = g["uniforms"].shape[0]
    env = AvoidRef(k, "rate")
    env.progress_buf[:] = 9; env.pre_actions[:] = 1; env.pre_root_positions[:] = 1; env.reset_buf[:] = 0
    env.reset_idx(torch.arange(k), t(g["uniforms"]))
    np.testing.assert_allclose(env.root_states.numpy(), g["root_states"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(env.object_positions.numpy(), g["object_pos"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(env.object_linvels.numpy(), g["object_vel"], rtol=0, atol=2e-6)
    parked = g["object_pos"][:, 0] == -999
    assert 10 < parked.sum() < 50 and list(parked[:4]) == [False, True, False, True]       # mask 0.7999 / 0.8001 / 0 / 0.95
    for name, ref in (("reset_buf", "reset_buf"), ("progress_buf", "progress"), ("pre_actions", "pre_actions"),
                      ("pre_root_positions", "pre_root_positions")):
        assert torch.equal(getattr(env, name), t(g[ref])), name
    # the throw is aimed: a ballistic flight from the release point passes within the 0.3 m aiming box around (0, 0, 1)
    p, v = g["object_pos"][~parked].astype(np.float64), g["object_vel"][~parked].astype(np.float64)
    tt = np.linalg.norm(p[:, :2], axis=1) / 4.5
    at = p + v * tt[:, None] + np.array([0, 0, -0.5 * 9.81])[None] * tt[:, None] ** 2
    assert np.abs(at - np.array([0, 0, 1.0])).max() < 0.7     # (time of flight taken to the origin, not to the jittered aim point)


def test_avoid_closed_loop_sanity():
    """The build-defined pieces: the cube flies a parabola, lands and stays; a cube on the robot is a collision."""
    env = AvoidRef(8, "rate", seed=3)
    env.cam_rate = 10 ** 9
    z0 = env.object_positions[:, 2].clone()
    thrown = env.object_positions[:, 0] != -999
    a = torch.zeros(8, 4); a[:, 3] = -0.69
    for _ in range(5):
        env.step(a)
    assert (env.object_positions[thrown, 2] != z0[thrown]).all() or not thrown.any()
    env.object_positions[:] = env.root_positions
    env.object_linvels[:] = 0
    env.reset_buf[:] = 0
    _, _, rew, done, ex = env.step(a)
    assert (done == 1).all() and (ex["item_reward_info"]["alive_reward"] == -500).all()
