"""CPU check of the kernel arithmetic: airgym_amd/csrc/env_math.hpp (the source the gfx950 kernel
inlines) compiled with g++ by tests/host_harness, stepped side by side with the oracle.

This is what lets the kernel math be debugged on the GPU-less build box; the GPU parity tests proper
(tests/test_gpu_parity.py, -m gpu) go through the C-ABI of libairgym_hip.so.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import philox
from oracle.hovering_ref import HoveringRef
from oracle.tracking_ref import TrackingRef

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "harness.cpp")
OUT = os.path.join(HERE, "host_harness", "_build", "libagh.so")
HDR = os.path.join(os.path.dirname(HERE), "airgym_amd", "csrc", "env_math.hpp")
HDR2 = os.path.join(os.path.dirname(HERE), "airgym_amd", "csrc", "planning_math.hpp")

TASKS = {"hovering": (0, HoveringRef, 18), "tracking": (1, TrackingRef, 48)}
CTLS = {"pos": 0, "vel": 1, "atti": 2, "rate": 3, "prop": 4}


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if (not os.path.exists(OUT)) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-o", OUT])
    return ctypes.CDLL(OUT)


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class HarnessEnv:
    def __init__(self, lib, task, ctl, n, seed, env_id_offset=0):
        self.lib = lib
        self.task_id, _, self.nobs = TASKS[task]
        self.ctl_id = CTLS[ctl]
        self.A = 5 if ctl == "atti" else 4
        self.n = n
        self.seed = seed
        self.off = env_id_offset
        self.max_len = 3600 if task == "tracking" else 2400
        self.target = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1] + [0] * 9, dtype=np.float32)
        self.rs = np.zeros((n, 13), np.float32)
        self.cs = np.zeros((n, 12), np.float32)
        self.pa = np.zeros((n, self.A), np.float32)
        self.progress = np.zeros(n, np.int32)
        self.was_reset = np.zeros(n, np.int32)
        self.tick = 0
        self.reset_all()

    def reset_all(self):
        rc = self.lib.agh_reset_all(self.task_id, self.A, self.n, ctypes.c_double(0.01), self.max_len, fp(self.target),
                                    ctypes.c_uint64(self.seed), ctypes.c_uint32(self.tick), ctypes.c_uint32(self.off),
                                    fp(self.rs), fp(self.cs), fp(self.pa), fp(self.progress), fp(self.was_reset))
        assert rc == 0
        self.tick += 1

    def step(self, actions, noise=None, uniforms=None):
        n = self.n
        actions = np.ascontiguousarray(actions, np.float32)
        self.obs = np.zeros((n, self.nobs), np.float32)
        self.rew = np.zeros(n, np.float32)
        self.done = np.zeros(n, np.int32)
        self.timeout = np.zeros(n, np.int32)
        self.terms = np.zeros((n, 9), np.float32)
        self.cmd = np.zeros((n, 4), np.float32)
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float32)
            uniforms = np.ascontiguousarray(uniforms, np.float32)
        rc = self.lib.agh_step(self.task_id, self.ctl_id, n, ctypes.c_double(0.01), self.max_len, fp(self.target),
                               ctypes.c_uint64(self.seed), ctypes.c_uint32(self.tick), ctypes.c_uint32(self.off), 0,
                               fp(self.rs), fp(self.cs), fp(self.pa), fp(self.progress), fp(self.was_reset),
                               fp(actions), fp(noise) if noise is not None else None,
                               fp(uniforms) if uniforms is not None else None,
                               fp(self.obs), fp(self.rew), fp(self.done), fp(self.timeout), fp(self.terms), fp(self.cmd))
        assert rc == 0
        self.tick += 1


def test_philox_matches_oracle(lib):
    out = (ctypes.c_uint32 * 4)()
    for ctr, key in [((0, 0, 0, 0), (0, 0)), ((5, 77, 1, 3), (123, 456)),
                     ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0))]:
        lib.agh_philox(*[ctypes.c_uint32(c) for c in ctr], *[ctypes.c_uint32(k) for k in key], out)
        ref = philox.philox4x32_10(*ctr, *key)
        assert [int(x) for x in out] == [int(r) for r in ref]


def scripted_actions(rng, n, A, t, ctl):
    """random + scripted mix, kept away from clamp edges (SURVEY section 7 'hard parts')."""
    a = rng.uniform(-0.8, 0.8, size=(n, A)).astype(np.float32)
    if ctl in ("rate", "atti"):
        a[:, -1] = rng.uniform(-0.9, -0.3, size=n)   # thrust 0.05 .. 0.35 after the 0.5+0.5a map
    if ctl == "atti":
        a[:, 0] = rng.uniform(0.6, 0.95, size=n)     # qw > 0
        a[:, 1:4] *= 0.3
    if ctl == "prop":
        a = rng.uniform(0.05, 0.3, size=(n, A)).astype(np.float32)
    if t % 7 == 0:
        a[: n // 4] = a[0]                            # a block of identical actions
    return a


@pytest.mark.parametrize("task", ["hovering", "tracking"])
@pytest.mark.parametrize("ctl", ["rate", "vel", "atti", "pos", "prop"])
def test_100_step_trajectory_matches_oracle(lib, task, ctl):
    n, steps, seed = 64, 100, 1234
    _, cls, nobs = TASKS[task]
    ora = cls(n, ctl_mode=ctl, seed=seed)
    har = HarnessEnv(lib, task, ctl, n, seed)
    # identical initial state from the counter RNG (reset keyed by (seed, env, tick 0))
    np.testing.assert_allclose(har.rs, ora.root_states.numpy(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(7)
    n_resets = 0
    for t in range(steps):
        a = scripted_actions(rng, n, har.A, t, ctl)
        obs, _, rew, reset, extras = ora.step(torch.from_numpy(a))
        har.step(a)
        # integer reset indices bit-exact
        assert np.array_equal(np.nonzero(har.done)[0], ora.last_reset_env_ids.numpy()), f"step {t}"
        n_resets += int(har.done.sum())
        np.testing.assert_allclose(har.rs, ora.root_states.numpy(), rtol=0, atol=1e-5, err_msg=f"state step {t}")
        np.testing.assert_allclose(har.obs, obs.numpy(), rtol=0, atol=2e-5, err_msg=f"obs step {t}")
        np.testing.assert_allclose(har.rew, rew.numpy(), rtol=0, atol=1e-5, err_msg=f"rew step {t}")
        np.testing.assert_allclose(har.cmd, ora.cmd_thrusts.numpy(), rtol=0, atol=1e-5, err_msg=f"cmd step {t}")
        assert np.array_equal(har.progress, ora.progress_buf.numpy().astype(np.int32))
        assert np.array_equal(har.was_reset, ora.reset_buf.numpy().astype(np.int32))
    assert har.tick == ora.tick
    if task == "tracking":
        assert n_resets > 0   # random actions leave the 1 m tube quickly: the reset path is exercised


@pytest.mark.parametrize("task", ["hovering", "tracking"])
def test_parity_mode_with_supplied_randoms(lib, task):
    n, seed = 64, 5
    _, cls, nobs = TASKS[task]
    ora = cls(n, ctl_mode="rate", seed=seed)
    har = HarnessEnv(lib, task, "rate", n, seed)
    rng = np.random.default_rng(3)
    for t in range(30):
        a = scripted_actions(rng, n, 4, t, "rate")
        noise = rng.standard_normal((n, 18)).astype(np.float32)
        uni = rng.random((n, 12)).astype(np.float32)
        obs, _, rew, reset, _ = ora.step(torch.from_numpy(a), noise=torch.from_numpy(noise),
                                         reset_uniforms=torch.from_numpy(uni))
        har.step(a, noise, uni)
        assert np.array_equal(har.done, reset.numpy().astype(np.int32))
        np.testing.assert_allclose(har.obs, obs.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(har.rs, ora.root_states.numpy(), rtol=0, atol=1e-5)


def test_episode_end_reset_and_thrust_zero_quirk(lib):
    """progress >= max_len-1 terminates (hovering.py:435); the following step runs with zero thrust (Q2)."""
    n, seed = 64, 9
    ora = HoveringRef(n, "rate", seed=seed)
    har = HarnessEnv(lib, "hovering", "rate", n, seed)
    ora.progress_buf[:] = 2397
    har.progress[:] = 2397
    a = np.zeros((n, 4), np.float32)
    a[:, 3] = -0.7
    ora.step(torch.from_numpy(a)); har.step(a)
    assert har.done.sum() == 0 or (har.done == ora.reset_buf.numpy()).all()
    ora.step(torch.from_numpy(a)); har.step(a)
    assert (har.done == 1).all() and (ora.reset_buf == 1).all()      # progress hit 2399
    assert (har.progress == 0).all() and (har.pa == 0).all()
    np.testing.assert_allclose(har.rs, ora.root_states.numpy(), rtol=0, atol=1e-6)
    vz0 = har.rs[:, 9].copy()
    ora.step(torch.from_numpy(a)); har.step(a)
    # free fall for one step: dv_z = -9.81 * 0.01 regardless of the commanded thrust
    np.testing.assert_allclose(har.rs[:, 9] - vz0, -0.0981, atol=2e-5)
    np.testing.assert_allclose(har.rs, ora.root_states.numpy(), rtol=0, atol=1e-5)


# ----------------------------------------------------------------------------- Planning (planning_math.hpp)
def _plan_pair(lib, ctl, n, seed):
    from oracle.planning_ref import PlanningRef
    ora = PlanningRef(n, ctl, seed=seed)
    st = dict(rs=ora.root_states.numpy().copy(), cs=np.zeros((n, 12), np.float32),
              pa=np.zeros((n, 4), np.float32), progress=np.zeros(n, np.int32), was_reset=np.ones(n, np.int32),
              obst=np.concatenate([ora.obstacles.numpy(), ora.variants.numpy()[..., None].astype(np.float32)], -1).astype(np.float32).copy(),
              goal=ora.goal_positions.numpy().copy(), extra=np.zeros((n, 5), np.float32))
    st["extra"][:, 3] = 10.0      # esdf: these tests pin the image min at 10 on both sides
    cs = st["cs"]
    cs[:, 3:6] = 0.0
    cs[:, 9:12] = ora.root_states[:, 7:10].numpy()
    return ora, st


def test_planning_raycast_matches_oracle(lib):
    from oracle import planning_ref as P
    ora, st = _plan_pair(lib, "rate", 3, 11)
    # move the robots into the obstacle field so that cylinders are in view
    ora.root_states[:, 0] = torch.tensor([-4.0, 0.0, 3.0]); ora.root_states[:, 1] = torch.tensor([0.5, -1.0, 1.0])
    q = torch.tensor([[0.02, -0.03, 0.1, 1.0], [0.0, 0.05, -0.3, 1.0], [0.03, 0.0, 0.6, 1.0]])
    ora.root_states[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
    table = P.load_variant_table()
    centre, axis, r, h = ora.scene()
    for e in range(3):
        ref = P.render_depth_one(ora.root_positions[e], ora.root_quats[e], centre[e], axis[e], r[e], h[e], ora.goal_positions[e]).numpy()
        out = np.zeros((P.CAM_H, P.CAM_W), np.float32)
        pos = ora.root_states[e, 0:3].numpy().copy(); quat = ora.root_states[e, 3:7].numpy().copy()
        assert lib.agh_plan_render(fp(pos), fp(quat), fp(st["obst"][e]), fp(table), fp(st["goal"][e]), fp(out)) == 0
        both_inf = np.isinf(ref) & np.isinf(out)
        close = np.abs(np.where(both_inf, 0, ref) - np.where(both_inf, 0, out)) < 1e-4
        frac_bad = 1.0 - (both_inf | close).mean()
        assert frac_bad < 2e-3, f"env {e}: {frac_bad:.4%} pixels differ (silhouette flips only are tolerated)"
        assert np.isfinite(out).mean() > 0.1          # something is actually in view


@pytest.mark.parametrize("ctl", ["rate", "vel", "pos", "prop"])
def test_planning_step_matches_oracle(lib, ctl):
    """physics + collision + obs + reward/done + reset (planning.py:138-307), esdf held fixed (no render)."""
    from oracle import planning_ref as P
    n, seed = 32, 5
    ora, st = _plan_pair(lib, ctl, n, seed)
    table = P.load_variant_table()
    ora.cam_rate = 10 ** 9                       # never render: esdf stays 10 in both
    ora.full_camera_array[:] = 10.0
    rng = np.random.default_rng(1)
    tick = ora.tick
    n_done = 0
    for t in range(120):
        a = rng.uniform(-0.5, 0.5, size=(n, 4)).astype(np.float32)
        if ctl == "rate":
            a[:, 3] = rng.uniform(-0.85, -0.5, size=n); a[:, 1] = rng.uniform(0.0, 0.4, size=n)
        if ctl == "prop":
            a = rng.uniform(0.14, 0.17, size=(n, 4)).astype(np.float32)
        if ctl in ("vel", "pos"):
            a[:, 0] = 0.8
        obs = np.zeros((n, 16), np.float32); rew = np.zeros(n, np.float32); done = np.zeros(n, np.int32)
        terms = np.zeros((n, 11), np.float32); coll = np.zeros(n, np.float32)
        o, _, r_ref, d_ref, ex = ora.step(torch.from_numpy(a))
        rc = lib.agh_plan_step(CTLS[ctl], n, ctypes.c_double(0.01), 1600, ctypes.c_uint64(seed), ctypes.c_uint32(tick),
                               ctypes.c_uint32(0), fp(st["rs"]), fp(st["cs"]), fp(st["pa"]), fp(st["progress"]),
                               fp(st["was_reset"]), fp(a), fp(st["obst"]), fp(st["goal"]), fp(st["extra"]), fp(table),
                               None, fp(obs), fp(rew), fp(done), fp(terms), fp(coll))
        assert rc == 0
        tick += 1
        assert np.array_equal(done, d_ref.numpy().astype(np.int32)), f"step {t}"
        n_done += int(done.sum())
        np.testing.assert_allclose(st["rs"], ora.root_states.numpy(), atol=1e-5, err_msg=f"state {t}")
        np.testing.assert_allclose(obs, o["observation"].numpy(), atol=2e-5, err_msg=f"obs {t}")
        np.testing.assert_allclose(rew, r_ref.numpy(), atol=2e-5, err_msg=f"rew {t}")
        np.testing.assert_allclose(coll, ora.collisions.numpy(), atol=0)
        np.testing.assert_allclose(st["obst"][..., :3], ora.obstacles.numpy(), atol=1e-5)
        np.testing.assert_allclose(st["goal"], ora.goal_positions.numpy(), atol=1e-6)
        np.testing.assert_allclose(st["extra"][:, 0:3], ora.pre_root_positions.numpy(), atol=1e-5)
        np.testing.assert_allclose(st["extra"][:, 4], ora.prev_related_dist.numpy(), atol=1e-5)
        for j, k in enumerate(["continous_action_reward", "heading_reward", "speed_reward", "forward_reward",
                               "alive_reward", "ups_reward", "z_reward", "esdf_reward", "thrust_reward",
                               "reach_goal_reward", "reward"]):
            np.testing.assert_allclose(terms[:, j], ex["item_reward_info"][k].numpy(), atol=2e-5, err_msg=k)
    # the z corridor is 0.6 m wide: random rate / thrust actions leave it, so resets are exercised; the velocity and position
    # cascades hold altitude (more so since the mixer stopped turning saturated torque demands into lift)
    assert n_done > 0 or ctl in ("vel", "pos")


# ---------------------------------------------------------------------------------------- Balloon / Avoid (row f3)
def _custom_post(lib, task_id, n, max_len, target, g, goal, objvel, noise, coll, uniforms=None, nobs=18):
    A = g["actions"].shape[1]
    rs = np.ascontiguousarray(g["root_states"], np.float32).copy()
    pa = np.ascontiguousarray(g["pre_actions"], np.float32).copy()
    progress = (g["progress"] - 1).astype(np.int32)
    actions = np.ascontiguousarray(g["actions"], np.float32)
    prepos = np.ascontiguousarray(g.get("pre_root_positions", np.zeros((n, 3))), np.float32).copy()
    obs = np.zeros((n, nobs), np.float32); rew = np.zeros(n, np.float32); done = np.zeros(n, np.int32)
    terms = np.zeros((n, 8), np.float32)
    tgt = np.asarray(target, np.float32)
    rc = lib.agh_custom_post(task_id, CTLS["vel"], n, max_len, fp(tgt), fp(rs), fp(pa), fp(progress), fp(actions), fp(goal),
                             fp(objvel), fp(prepos), fp(coll), fp(noise), fp(uniforms) if uniforms is not None else None,
                             fp(obs), fp(rew), fp(done), fp(terms))
    assert rc == 0
    return dict(obs=obs, rew=rew, done=done, terms=terms, rs=rs, pa=pa, progress=progress, goal=goal, objvel=objvel, prepos=prepos)


def test_balloon_kernel_math_matches_reference_recordings(lib):
    """balloon_post / balloon_reset (planning_math.hpp) under g++ against the vectors recorded from the reference's own
    Balloon methods (tests/golden/balloon_*.npz)."""
    g = dict(np.load(os.path.join(HERE, "golden", "balloon_obs_reward.npz")))
    n = g["root_states"].shape[0]
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1] + [0] * 9
    r = _custom_post(lib, 3, n, 800, ident, g, np.ascontiguousarray(g["balloon"], np.float32).copy(), np.zeros((n, 3), np.float32),
                     np.ascontiguousarray(g["noise"], np.float32), np.zeros(n, np.float32))
    assert np.array_equal(r["done"], g["reset"].astype(np.int32))
    np.testing.assert_allclose(r["obs"], g["obs"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["rew"], g["reward"], rtol=0, atol=3e-5)
    for j, k in enumerate(("guidance_reward", "hit_reward", "action_smoothness_reward", "effort_reward", "ups_reward", "reward")):
        np.testing.assert_allclose(r["terms"][:, j], g["info_" + k].astype(np.float64), rtol=0, atol=3e-5, err_msg=k)
    # reset with the recorded uniforms: force every row to terminate (z above the ceiling)
    gr = dict(np.load(os.path.join(HERE, "golden", "balloon_reset.npz")))
    k = gr["uniforms"].shape[0]
    rs = np.zeros((k, 13), np.float32); rs[:, 6] = 1; rs[:, 2] = 2.0
    fake = dict(root_states=rs, pre_actions=np.ones((k, 4), np.float32), progress=np.full(k, 10), actions=np.zeros((k, 4), np.float32))
    r = _custom_post(lib, 3, k, 800, ident, fake, np.zeros((k, 3), np.float32), np.zeros((k, 3), np.float32),
                     np.zeros((k, 18), np.float32), np.zeros(k, np.float32), uniforms=np.ascontiguousarray(gr["uniforms"], np.float32))
    assert r["done"].all()
    np.testing.assert_allclose(r["rs"], gr["root_states"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["goal"], gr["balloon"], rtol=0, atol=1e-6)
    assert (r["pa"] == 0).all() and (r["progress"] == 0).all() and (r["prepos"] == 0).all()


def test_avoid_kernel_math_matches_reference_recordings(lib):
    """avoid_post / avoid_reset (planning_math.hpp) under g++ against tests/golden/avoid_*.npz."""
    g = dict(np.load(os.path.join(HERE, "golden", "avoid_obs_reward.npz")))
    n = g["root_states"].shape[0]
    target = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    r = _custom_post(lib, 4, n, 600, target, g, np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32),
                     np.zeros((n, 18), np.float32), np.ascontiguousarray(g["collisions"], np.float32), nobs=16)
    assert np.array_equal(r["done"], np.maximum(g["reset"], (g["collisions"] > 0)).astype(np.int32))
    np.testing.assert_allclose(r["obs"], g["obs"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["rew"], g["reward"], rtol=0, atol=6e-5)
    for j, k in enumerate(("pose_reward", "ups_reward", "spin_reward", "effort_reward", "action_smoothness_reward",
                           "thrust_reward", "alive_reward")):
        np.testing.assert_allclose(r["terms"][:, j], g["info_" + k], rtol=0, atol=2e-6, err_msg=k)
    gr = dict(np.load(os.path.join(HERE, "golden", "avoid_reset.npz")))
    k = gr["uniforms"].shape[0]
    rs = np.zeros((k, 13), np.float32); rs[:, 6] = 1; rs[:, 2] = 2.0
    fake = dict(root_states=rs, pre_actions=np.ones((k, 4), np.float32), progress=np.full(k, 10), actions=np.zeros((k, 4), np.float32))
    r = _custom_post(lib, 4, k, 600, target, fake, np.zeros((k, 3), np.float32), np.zeros((k, 3), np.float32),
                     np.zeros((k, 18), np.float32), np.zeros(k, np.float32), uniforms=np.ascontiguousarray(gr["uniforms"], np.float32),
                     nobs=16)
    assert r["done"].all()
    np.testing.assert_allclose(r["rs"], gr["root_states"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["goal"], gr["object_pos"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["objvel"], gr["object_vel"], rtol=0, atol=1e-5)


def test_ray_aabb_matches_oracle(lib):
    from oracle.planning_ref import _ray_aabb
    lib.agh_ray_aabb.restype = ctypes.c_float
    rng = np.random.default_rng(0)
    c = np.array([2.0, 0.3, 1.1], np.float32)
    for _ in range(200):
        o = rng.uniform(-1, 1, 3).astype(np.float32); o[2] += 1
        d = np.array([1.0, rng.uniform(-1, 1), rng.uniform(-0.6, 0.6)], np.float32)
        ref = _ray_aabb(torch.from_numpy(o), torch.from_numpy(d)[None], torch.from_numpy(c), 0.15)[0].item()
        got = lib.agh_ray_aabb(fp(o), fp(d), fp(c), ctypes.c_float(0.15))
        assert (np.isinf(ref) and np.isinf(got)) or abs(ref - got) < 1e-5


def test_stagger_progress_matches_oracle(lib):
    """AG_FLAG_STAGGER_PHASE: the kernel's stagger_progress() (g++ build of env_math.hpp) == HoveringRef._reset_all's phases."""
    n, max_len, seed, off = 513, 2400, 77, 1 << 20
    out = np.zeros(n, np.int32)
    assert lib.agh_stagger_progress(n, max_len, ctypes.c_uint64(seed), ctypes.c_uint32(0), ctypes.c_uint32(off), fp(out)) == 0
    ora = HoveringRef(n, "rate", seed=seed, env_id_offset=off, stagger_episode_phase=True)
    # the constructor's full reset ran at tick 0
    raw = philox.raw_blocks(seed, np.arange(off, off + n, dtype=np.uint32), 0, philox.STREAM_PHASE, 1)[:, 0]
    assert np.array_equal(out, (raw % np.uint32(max_len - 1)).astype(np.int32))
    assert np.array_equal(out.astype(np.int64), ora.progress_buf.numpy())
    assert out.min() >= 0 and out.max() <= max_len - 2 and len(np.unique(out)) > n // 2
    # default (flag off): all zero, the reference's hovering.py:333
    assert HoveringRef(8, "rate", seed=seed).progress_buf.sum() == 0


def test_tracking_lookahead_over_the_whole_episode(lib):
    """Round 6: the ten lemniscate look-ahead points share ONE sinf / cosf pair (env_math.hpp::lemniscate_refs: angle addition with
    the small angle t_k - t_0, t_k rounded to float32 exactly as tracking.py:196-197 rounds it).  Sweep progress over the whole
    36 s episode - t up to 9.1 rad, where one float32 ulp of t is 1e-6 - against the oracle's ten sin / cos pairs: observation columns
    18:48 within 2e-6 (the tolerance the reference recordings are held to on the GPU)."""
    n, seed = 3600, 11
    ora = TrackingRef(n, ctl_mode="vel", seed=seed)
    har = HarnessEnv(lib, "tracking", "vel", n, seed)
    prog = np.arange(n, dtype=np.int32)
    prog[-8:] = 3580                                       # (keep away from the time limit: no reset inside this step)
    ora.progress_buf[:] = torch.from_numpy(prog.astype(np.int64))
    har.progress[:] = prog
    # put every env ON its reference point (one step later), so that nobody leaves the 1 m tube and resets
    t = (prog + 1).astype(np.float64) * 0.01 * 0.25
    pos = np.stack([3 * np.sin(t) / (1 + np.cos(t) ** 2), 3 * np.sin(t) * np.cos(t) / (1 + np.cos(t) ** 2), np.ones_like(t)], 1)
    rs = har.rs.copy()
    rs[:, 0:3] = pos.astype(np.float32)
    rs[:, 3:7] = (0, 0, 0, 1)
    rs[:, 7:13] = 0
    har.rs[:] = rs
    ora.root_states[:] = torch.from_numpy(rs)
    a = np.zeros((n, 4), np.float32)
    obs, _, rew, reset, _ = ora.step(torch.from_numpy(a))
    har.step(a)
    keep = (reset.numpy() == 0) & (har.done == 0)          # envs that reset show the post-reset observation
    assert keep.sum() > 3000
    d = np.abs(har.obs[keep, 18:48] - obs.numpy()[keep, 18:48])
    assert d.max() < 2e-6, d.max()
    # the first point is the reward's ref_positions[:, 0]: the same sinf / cosf as before this change
    np.testing.assert_allclose(har.rew[keep], rew.numpy()[keep], rtol=0, atol=1e-5)
