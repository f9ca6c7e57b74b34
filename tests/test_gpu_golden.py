"""The HIP code object itself against (a) the golden vectors recorded from the REFERENCE's own methods
(tests/golden/*.npz, made by tests/golden/make_golden.py) and (b) the oracle on the edge cases the trajectory tests never
reach: saturating actions (every clamp of hovering.py:93-121), w < 0 quaternions (hovering.py:224-226), `atti` a0 < 0
termination (hovering.py:445-446), the zero-velocity NaN quirk (Q8), the termination thresholds (d = 4 +- eps, dz = +-2 +- eps,
roll ~ 90 deg, progress 2397..2400).  Everything goes through the C ABI (HipEnvHandle -> libairgym_hip.so).

The CPU suite pins oracle == reference on these fixtures (tests/test_oracle_golden.py); this module pins HIP == reference
directly, without the oracle in between, wherever the reference recorded an output.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ppo_ref
from oracle.hovering_ref import HoveringRef
from oracle.tracking_ref import TrackingRef

pytestmark = pytest.mark.gpu

CLS = {"hovering": HoveringRef, "tracking": TrackingRef}
MODES = ["rate", "vel", "atti", "pos", "prop"]


@pytest.fixture(scope="module")
def Handle():
    from airgym_amd.hip_env import HipEnvHandle
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return HipEnvHandle


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------------------------- env: golden vectors
@pytest.mark.parametrize("task", ["hovering", "tracking"])
def test_golden_observations(Handle, golden, task):
    """compute_observations + add_noise of the reference (hovering.py:337-358, tracking.py:202-214) replayed on the HIP
    kernel's device functions: recorded states + recorded noise -> recorded obs."""
    g = golden(f"{task}_obs")
    n = g["root_states"].shape[0]
    env = Handle(task, "rate", n, seed=0, target_state=g["target_state"])
    env.set_state(root_states=t(g["root_states"]), progress=t(g["progress"].astype(np.int32)))
    env.eval_obs_reward(torch.zeros(n, 4), torch.zeros(n, 4), noise=t(g["noise"]))
    obs = env.obs_buf.cpu().numpy()
    assert obs.shape == g["obs"].shape
    # v_rcp (1 ulp) in quaternion_to_matrix / the lemniscate denominator: a few 1e-7 on values of order 1..3
    np.testing.assert_allclose(obs, g["obs"], rtol=0, atol=2e-6)
    env.close()


@pytest.mark.parametrize("task", ["hovering", "tracking"])
@pytest.mark.parametrize("ctl", MODES)
def test_golden_reward_and_done(Handle, golden, task, ctl):
    """compute_quadcopter_reward of the reference (hovering.py:371-459, tracking.py:223-296), all five modes: recorded
    (state, progress, actions, pre_actions, cmd_thrusts) -> recorded reward, every reward term, and the reset flags
    BIT-EXACT, including the rows that straddle each termination threshold."""
    g = golden(f"{task}_reward_{ctl}")
    n = g["root_states"].shape[0]
    env = Handle(task, ctl, n, seed=0)
    env.set_state(root_states=t(g["root_states"]), progress=t(g["progress"].astype(np.int32)),
                  pre_actions=t(g["pre_actions"]))
    env.eval_obs_reward(t(g["actions"]), t(g["cmd_thrusts"].astype(np.float32)))
    reset = env.reset_buf.cpu().numpy()
    assert np.array_equal(reset, g["reset"]), np.nonzero(reset != g["reset"])
    assert reset.sum() > 4 and (reset == 0).sum() > 4
    np.testing.assert_allclose(env.rew_buf.cpu().numpy(), g["reward"], rtol=0, atol=1e-5, equal_nan=True)
    for name, buf in env.reward_terms.items():
        key = "info_" + name
        if key in g.files:      # thrust_reward is only emitted by the rate / atti branches of the reference
            np.testing.assert_allclose(buf.cpu().numpy(), g[key], rtol=0, atol=1e-5, equal_nan=True, err_msg=name)
    if task == "hovering":
        # rows 4..11: |rel| = 3.999 / 4.001, rel_z = +-1.999 / +-2.001, roll just below / above 90 deg; in atti mode a row
        # also ends when its (random) a0 is negative (hovering.py:445-446)
        a0_neg = (g["actions"][4:12, 0] < 0) if ctl == "atti" else np.zeros(8, bool)
        assert list(reset[4:12]) == list(np.array([0, 1, 0, 1, 0, 1, 0, 1]) | a0_neg)
    env.close()


@pytest.mark.parametrize("task", ["hovering", "tracking"])
def test_golden_reset_distribution(Handle, golden, task):
    """reset_idx of the reference (hovering.py:310-335, tracking.py:159-192) with the recorded uniforms, executed by the
    step kernel's own reset path: every env is driven to the episode-length termination, the uniforms are supplied."""
    g = golden(f"{task}_reset")
    n = g["uniforms"].shape[0]
    env = Handle(task, "rate", n, seed=4)
    max_len = env.max_episode_length
    rs = torch.zeros(n, 13); rs[:, 6] = 1.0; rs[:, 2] = 1.0 if task == "tracking" else 0.0
    env.set_state(root_states=rs, progress=torch.full((n,), max_len - 2, dtype=torch.int32),
                  pre_actions=torch.ones(n, 4))
    a = torch.zeros(n, 4); a[:, 3] = -0.7
    env.step_with_inputs(a.cuda(), torch.zeros(n, 18), t(g["uniforms"]))
    st = env.get_state()
    assert (env.reset_buf.cpu().numpy() == g["reset_buf"]).all() and (env.reset_buf == 1).all()
    np.testing.assert_allclose(st["root_states"].cpu().numpy(), g["root_states"], rtol=0, atol=1e-6)
    assert np.array_equal(st["progress"].cpu().numpy(), g["progress"].astype(np.int32))
    assert np.array_equal(st["pre_actions"].cpu().numpy(), g["pre_actions"])
    assert (st["was_reset"] == 1).all()
    # u = 0 and u = 1 - 2^-24 rows: the reset box is half-open like torch_rand_float's
    assert np.abs(st["root_states"][:2, 0:3].cpu().numpy()).max() <= (0.1 + 1.0 if task == "tracking" else 1.0)
    env.close()


@pytest.mark.parametrize("ctl", ["rate", "atti", "vel", "pos"])
def test_golden_action_map_and_quat_canonicalisation(Handle, golden, ctl):
    """pre_physics_step of the reference (hovering.py:212-226): recorded raw actions (|a| up to 14: every clamp binds,
    the thrust remap saturates at both ends) -> recorded processed actions BIT-EXACT (read back as pre_actions after the
    step); quaternions with w < 0 are flipped before anything uses them, so an env started from q and its twin started
    from the reference's canonicalised q stay bit-identical."""
    g = golden("action_map")
    a_in, a_out = g[f"{ctl}_in"], g[f"{ctl}_out"]
    q_in, q_out = g[f"{ctl}_quat_in"], g[f"{ctl}_quat_out"]
    n = a_in.shape[0]
    assert (q_in[:, 3] < 0).sum() > 20 and (q_out[:, 3] >= 0).all()
    gen = torch.Generator().manual_seed(5)
    base = torch.zeros(n, 13)
    base[:, 0:3] = 0.2 * torch.randn(n, 3, generator=gen)
    base[:, 7:13] = 0.2 * torch.randn(n, 6, generator=gen)
    envs = []
    for q in (q_in, q_out):
        e = Handle("hovering", ctl, n, seed=1, obs_noise=False)
        rs = base.clone(); rs[:, 3:7] = t(q)
        e.set_state(root_states=rs, progress=torch.full((n,), 10, dtype=torch.int32),
                    ctl_state=torch.zeros(n, 12), pre_actions=torch.zeros(n, e.num_actions),
                    was_reset=torch.zeros(n, dtype=torch.int32))
        e.step(t(a_in).cuda())
        envs.append(e)
    sa, sb = envs[0].get_state(), envs[1].get_state()
    alive = (envs[0].reset_buf == 0).cpu().numpy()
    assert alive.sum() > n // 4
    assert np.array_equal(sa["pre_actions"].cpu().numpy()[alive], a_out[alive])     # the reference's clamp, bit-exact
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k                                          # q and -q are the same attitude
    assert torch.equal(envs[0].obs_buf, envs[1].obs_buf) and torch.equal(envs[0].rew_buf, envs[1].rew_buf)
    # and the same step against the oracle (which test_oracle_golden pins to these fixtures)
    ora = HoveringRef(n, ctl, seed=1)
    rs = base.clone(); rs[:, 3:7] = t(q_in)
    ora.root_states = rs.clone(); ora.progress_buf[:] = 10; ora.reset_buf[:] = 0
    ora.ctl_state.reset(torch.arange(n), ora.root_states)
    for name in ("rate_int", "prev_rate", "vel_int", "prev_vel"):
        if hasattr(ora.ctl_state, name):
            getattr(ora.ctl_state, name).zero_()
    obs, _, rew, reset, _ = ora.step(t(a_in), noise=torch.zeros(n, 18))
    assert np.array_equal(envs[0].reset_buf.cpu().numpy(), reset.numpy())
    # rotor commands: 1e-5, except rows whose mixer output is saturated (a rotor at 0 or 1): there the sequential desaturation
    # subtracts torque demands of O(10) from each other and float32 rounding of either side (the oracle rounds every operation,
    # the kernel fuses a * b + c inside one expression: -ffp-contract=on) is amplified to a few 1e-5 - 2 of 512 commands in `vel`
    got_cmd, ref_cmd = envs[0].cmd_thrusts.cpu().numpy(), ora.cmd_thrusts.numpy()
    saturated = ((ref_cmd <= 0.0) | (ref_cmd >= 1.0)).any(axis=1)
    np.testing.assert_allclose(got_cmd[~saturated], ref_cmd[~saturated], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got_cmd[saturated], ref_cmd[saturated], rtol=0, atol=5e-5)
    assert (np.abs(got_cmd - ref_cmd) > 1e-5).mean() < 0.01
    keep = reset.numpy() == 0
    np.testing.assert_allclose(sa["root_states"].cpu().numpy()[keep], ora.root_states.numpy()[keep], rtol=0, atol=1e-5)
    np.testing.assert_allclose(envs[0].rew_buf.cpu().numpy(), rew.numpy(), rtol=0, atol=1e-5)
    for e in envs:
        e.close()


# ---------------------------------------------------------------------------------------------- env: edge cases vs oracle
def _saturating_actions(rng, n, A, ctl, t_):
    a = rng.uniform(-8.0, 8.0, size=(n, A)).astype(np.float32)
    a[n // 2:] = rng.uniform(-1.5, 1.5, size=(n - n // 2, A)).astype(np.float32)   # half just around the limits
    if ctl in ("rate", "atti"):
        a[0::7, -1] = -1.0      # thrust remap lands exactly on 0
        a[1::7, -1] = 1.0       # ... and exactly on 1
        a[2::7, -1] = -3.0      # clamp below
        a[3::7, -1] = 5.0       # clamp above
    if ctl == "atti" and t_ % 2 == 0:
        a[:, 0] = np.abs(a[:, 0])   # every other step: no a0 < 0 termination, so episodes continue
    return a


@pytest.mark.parametrize("task", ["hovering", "tracking"])
@pytest.mark.parametrize("ctl", MODES)
def test_saturating_actions_match_oracle(Handle, task, ctl):
    """|a| up to 8 on every channel: every action clamp of hovering.py:93-121 binds, the mixer saturates, episodes end on
    the bounds; states start with w < 0 quaternions.  Per-step comparison with the oracle incl. reset ids bit-exact."""
    n, steps, seed = 256, 12, 77
    ora = CLS[task](n, ctl_mode=ctl, seed=seed)
    env = Handle(task, ctl, n, seed=seed)
    # flip the sign of every other quaternion: same attitude, exercises the canonicalisation inside the step
    rs = ora.root_states.clone(); rs[0::2, 3:7] *= -1.0
    if task == "hovering":      # a quarter of the envs close to the altitude bound and climbing: they terminate within the run
        rs[0::4, 2] = 1.93; rs[0::4, 9] = 2.0
    else:                       # ... or drifting out of the 1 m tube around the reference curve (tracking.py:275)
        rs[0::4, 0] = 0.96; rs[0::4, 7] = 1.0
    ora.root_states = rs.clone()
    env.set_state(root_states=rs)
    rng = np.random.default_rng(11)
    n_resets, n_sat = 0, 0
    for t_ in range(steps):
        a = _saturating_actions(rng, n, env.num_actions, ctl, t_)
        obs, _, rew, reset, extras = ora.step(torch.from_numpy(a))
        env.step(torch.from_numpy(a).cuda())
        st = env.get_state()
        ids = env.compact_reset_ids().cpu().numpy()
        assert np.array_equal(ids, ora.last_reset_env_ids.numpy()), f"reset ids differ at step {t_}"
        assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy())
        np.testing.assert_allclose(st["root_states"].cpu().numpy(), ora.root_states.numpy(), rtol=0, atol=2e-5,
                                   err_msg=f"state step {t_}")
        # saturated set-points put 1-ulp differences of v_rcp / v_rsq through gains of 10^2 and the desaturation branches
        np.testing.assert_allclose(env.cmd_thrusts.cpu().numpy(), ora.cmd_thrusts.numpy(), rtol=0, atol=1e-4)
        np.testing.assert_allclose(env.rew_buf.cpu().numpy(), rew.numpy(), rtol=0, atol=2e-5, equal_nan=True)
        np.testing.assert_allclose(st["pre_actions"].cpu().numpy(), ora.pre_actions.numpy(), rtol=0, atol=0)
        np.testing.assert_allclose(env.obs_buf.cpu().numpy(), obs.numpy(), rtol=0, atol=5e-5)
        cmd = ora.cmd_thrusts.numpy()
        n_sat += int(((cmd <= 0.0) | (cmd >= 1.0)).any(1).sum())
        n_resets += len(ids)
    assert n_resets > 0 and (ctl == "prop" or n_sat > n)      # the mixer really saturated
    env.close()


def test_atti_negative_a0_terminates(Handle):
    """hovering.py:445-446: in atti mode a processed action with a0 (= q_w of the attitude set-point) < 0 ends the episode."""
    n = 192
    ora = HoveringRef(n, "atti", seed=3)
    env = Handle("hovering", "atti", n, seed=3)
    rng = np.random.default_rng(0)
    a = rng.uniform(-0.3, 0.3, size=(n, 5)).astype(np.float32)
    a[:, 0] = rng.uniform(0.5, 0.9, size=n)
    a[::3, 0] = rng.uniform(-0.9, -1e-3, size=len(a[::3]))
    a[1::9, 0] = 0.0                                   # exactly zero does NOT terminate (strict <)
    a[:, 4] = -0.7
    _, _, rew, reset, _ = ora.step(torch.from_numpy(a))
    env.step(torch.from_numpy(a).cuda())
    got = env.reset_buf.cpu().numpy()
    assert np.array_equal(got, reset.numpy())
    assert (got[::3] == 1).all() and got[1::9].sum() == 0 and 0 < got.sum() < n
    assert np.array_equal(env.compact_reset_ids().cpu().numpy(), np.nonzero(got)[0])
    np.testing.assert_allclose(env.rew_buf.cpu().numpy(), rew.numpy(), rtol=0, atol=1e-5)
    env.close()


def test_zero_velocity_nan_quirk(Handle):
    """Q8 (hovering.py:391-396): the velocity-direction term divides by |v|; at exactly zero velocity the reference
    yields NaN for that term and for the reward.  Reproduced, not patched."""
    n = 64
    rs = torch.zeros(n, 13); rs[:, 6] = 1.0; rs[:, 0] = 0.5
    rs[1::2, 7] = 0.3
    env = Handle("hovering", "rate", n, seed=0)
    env.set_state(root_states=rs, progress=torch.full((n,), 5, dtype=torch.int32), pre_actions=torch.zeros(n, 4))
    ora = HoveringRef(n, "rate", seed=0)
    ora.root_states = rs.clone(); ora.progress_buf[:] = 5
    ora.actions = torch.full((n, 4), 0.2); ora.pre_actions = torch.zeros(n, 4); ora.cmd_thrusts = torch.full((n, 4), 0.15)
    reward, reset, info = ora.compute_quadcopter_reward()
    env.eval_obs_reward(ora.actions, ora.cmd_thrusts)
    r = env.rew_buf.cpu().numpy()
    assert np.isnan(reward.numpy()[0::2]).all() and np.isnan(r[0::2]).all()
    assert np.isfinite(r[1::2]).all()
    np.testing.assert_allclose(r[1::2], reward.numpy()[1::2], rtol=0, atol=1e-5)
    assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy())
    env.close()


def test_episode_length_threshold_rows(Handle, golden):
    """progress max_len-3 .. max_len: done from max_len-1 on (hovering.py:435 `progress_buf >= max_episode_length - 1`)."""
    for task, ctl in (("hovering", "rate"), ("tracking", "vel")):
        n = 64
        env = Handle(task, ctl, n, seed=0)
        L = env.max_episode_length
        rs = torch.zeros(n, 13); rs[:, 6] = 1.0
        prog = torch.full((n,), 7, dtype=torch.int32)
        prog[:4] = torch.tensor([L - 3, L - 2, L - 1, L], dtype=torch.int32)
        if task == "tracking":
            # sit on the reference curve so the distance rule (d > 1 m) does not fire
            tt = prog.float() * 0.01 * 0.25
            rs[:, 0] = 3 * torch.sin(tt) / (1 + torch.cos(tt) ** 2)
            rs[:, 1] = 3 * torch.sin(tt) * torch.cos(tt) / (1 + torch.cos(tt) ** 2)
            rs[:, 2] = 1.0
        rs[:, 7] = 0.1
        env.set_state(root_states=rs, progress=prog, pre_actions=torch.zeros(n, 4))
        env.eval_obs_reward(torch.zeros(n, 4), torch.full((n, 4), 0.15))
        got = env.reset_buf.cpu().numpy()
        assert list(got[:4]) == [0, 0, 1, 1] and got[4:].sum() == 0, (task, got[:8])
        env.close()


# ---------------------------------------------------------------------------------------------- rollout form
@pytest.mark.parametrize("task,ctl,n", [("hovering", "rate", 1000), ("tracking", "vel", 777), ("hovering", "atti", 64)])
def test_step_rollout_matches_step(Handle, task, ctl, n):
    """ag_step_rollout (u8 done flags, per-tile reward-term sums) == ag_step on a twin handle: obs / reward bit-identical,
    done flags equal, tile sums == sums of the per-env item_reward_info arrays over each 64-env tile."""
    a = Handle(task, ctl, n, seed=23)
    b = Handle(task, ctl, n, seed=23)
    tiles = (n + 63) // 64
    obs = torch.zeros(n, a.num_obs, device="cuda"); rew = torch.zeros(n, device="cuda")
    done = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    sums = torch.full((tiles, 12), float("nan"), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    names = list(b.reward_terms.keys())
    for t_ in range(30):
        act = torch.randn(n, a.num_actions, generator=g, device="cuda").clamp(-1, 1)
        if t_ == 20:        # force some terminations through the altitude / distance rule
            st = a.get_state()["root_states"].clone(); st[::5, 2] += 5.0
            a.set_state(root_states=st); b.set_state(root_states=st)
        a.step_rollout(act, obs, rew, done, sums)
        b.step(act)
        assert torch.equal(obs, b.obs_buf) and torch.equal(rew, b.rew_buf), t_
        assert torch.equal(done.long(), b.reset_buf)
        assert torch.equal(a.reset_mask, b.reset_mask)
        per_env = torch.stack([b.reward_terms[k] for k in names])                   # [9, n]
        pad = torch.zeros(len(names), tiles * 64, device="cuda"); pad[:, :n] = per_env
        ref = pad.view(len(names), tiles, 64).double().sum(-1).t()                   # [tiles, 9]
        assert torch.allclose(sums[:, :len(names)].double(), ref, rtol=2e-6, atol=1e-5), t_
    assert done.sum() >= 0
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    a.close(); b.close()



# ---------------------------------------------------------------------------------------------- PPO kernels: golden vectors
def test_golden_gae_kernel(golden):
    """ag_gae on the inputs recorded around the reference's A2CBase.discount_values (a2c_base.py:463-478)."""
    from airgym_amd import _native as N
    lib = N.load()
    g = golden("gae")
    H, n = g["mb_fdones"].shape
    dones = torch.cat((t(g["mb_fdones"]), t(g["fdones"]).view(1, n)), 0).to(torch.uint8).cuda().contiguous()
    values, rewards = t(g["mb_values"]).cuda().contiguous(), t(g["mb_rewards"]).cuda().contiguous()
    last = t(g["last_values"]).cuda().contiguous()
    advs, rets = torch.empty_like(values), torch.empty_like(values)
    N.check(lib.ag_gae(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), last.data_ptr(), 0.99, 0.95,
                       advs.data_ptr(), rets.data_ptr(), H, n, _stream()), "ag_gae")
    np.testing.assert_allclose(advs.cpu().numpy(), g["advs"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rets.cpu().numpy(), g["advs"] + g["mb_values"], rtol=1e-6, atol=1e-6)


def test_golden_running_mean_std_kernels(golden):
    """ag_rms_update / ag_normalize_rows on the batches recorded around the reference's RunningMeanStd
    (lib/core/running_mean_std.py:31-79): float64 running statistics and the normalised outputs."""
    from airgym_amd import _native as N
    lib = N.load()
    g = golden("ppo")
    D = g["rms_x0"].shape[1]
    mean = torch.zeros(D, dtype=torch.float64, device="cuda")
    var = torch.ones(D, dtype=torch.float64, device="cuda")
    count = torch.ones(1, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(lib.ag_rms_scratch_doubles(D), dtype=torch.float64, device="cuda")
    for i in range(3):
        x = t(g[f"rms_x{i}"]).cuda().contiguous()
        N.check(lib.ag_rms_update(x.data_ptr(), x.shape[0], D, mean.data_ptr(), var.data_ptr(), count.data_ptr(),
                                  scratch.data_ptr(), _stream()), "ag_rms_update")
        y = torch.empty_like(x)
        N.check(lib.ag_normalize_rows(x.data_ptr(), mean.data_ptr(), var.data_ptr(), y.data_ptr(), x.shape[0], D, 1e-5, 5.0,
                                      _stream()), "ag_normalize_rows")
        np.testing.assert_allclose(y.cpu().numpy(), g[f"rms_y{i}"], rtol=0, atol=2e-6)
    # the reference forms the BATCH moments in float32 (input.mean / input.var) and merges them in float64; the kernel forms
    # them in float64 from the same float32 data: agreement to float32 rounding of the batch moments
    np.testing.assert_allclose(mean.cpu().numpy(), g["rms_mean"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(var.cpu().numpy(), g["rms_var"], rtol=1e-6, atol=0)
    assert count.item() == float(g["rms_count"])
    x0 = t(g["rms_x0"]).cuda().contiguous(); y = torch.empty_like(x0)
    N.check(lib.ag_normalize_rows(x0.data_ptr(), mean.data_ptr(), var.data_ptr(), y.data_ptr(), x0.shape[0], D, 1e-5, 5.0,
                                  _stream()), "ag_normalize_rows")
    np.testing.assert_allclose(y.cpu().numpy(), g["rms_y_eval"], rtol=0, atol=2e-6)


def test_golden_ppo_loss_kernel(golden):
    """ag_ppo_loss + ag_ppo_loss_finalize fed straight from ppo.npz: the critic loss (plain and clipped) and the bound
    loss against the reference's RECORDED per-sample outputs; the actor loss / KL / neglogp against oracle.ppo_ref (which
    the CPU suite pins to the same fixture), including d loss / d heads against autograd of the oracle composition."""
    from airgym_amd import _native as N
    lib = N.load()
    g = golden("ppo")
    M, A = g["mu_big"].shape
    logstd = torch.tensor([-0.2, 0.1, 0.0, 0.3])
    sigma = torch.exp(logstd)
    gen = torch.Generator().manual_seed(9)
    mu = t(g["mu_big"]).clone()                                   # |mu| > 1.1 in places: the bound loss is active
    actions = mu + sigma * torch.randn(M, A, generator=gen)
    vp, v, ret = t(g["vp"]), t(g["v"]), t(g["ret"])      # vp = value_preds_batch (OLD values), v = the model's new values
    adv, old_nlp = t(g["adv"]), t(g["old_nlp"])
    old_mu, old_sigma = t(g["mu0"]), t(g["s0"])
    e_clip, critic_coef, ent_coef, b_coef = 0.2, 2.0, 0.01, 1e-4
    for clip_value, c_key in ((False, "c_loss"), (True, "c_loss_clip")):
        heads = torch.cat((mu, v), 1).contiguous()
        # ---- oracle composition (a2c_continuous.py:299-350), autograd for the gradients
        hq = heads.clone().requires_grad_(True)
        lq = logstd.clone().requires_grad_(True)
        mu_q, val_q = hq[:, :A], hq[:, A:]
        ls = mu_q * 0.0 + lq
        sg = torch.exp(ls)
        nlp = ppo_ref.neglogp(actions, mu_q, sg, ls)
        a_l = ppo_ref.actor_loss(old_nlp, nlp, adv, e_clip)
        c_l = ppo_ref.critic_loss(vp, val_q, e_clip, ret, clip_value)
        b_l = ppo_ref.bound_loss(mu_q)
        ent = (0.5 + 0.5 * np.log(2 * np.pi) + ls).sum(-1).mean()
        loss = a_l.mean() + 0.5 * c_l.mean() * critic_coef - ent * ent_coef + b_l.mean() * b_coef
        loss.backward()
        kl = ppo_ref.policy_kl(mu_q.detach(), sg.detach(), old_mu, old_sigma, True)
        # the fixture's recorded outputs for the pieces that can be fed verbatim
        assert torch.equal(c_l.detach(), t(g[c_key])) and torch.equal(b_l.detach(), t(g["b_loss"]))
        # ---- HIP
        f = dict(dtype=torch.float32, device="cuda")
        d_heads = torch.empty(M, A + 1, **f)
        new_mu, new_sigma = torch.empty(M, A, **f), torch.empty(M, A, **f)
        parts = torch.zeros(lib.ag_ppo_loss_max_blocks(), lib.ag_ppo_loss_num_sums(), **f)
        nb = ctypes.c_int(0)
        dev = [x.cuda().contiguous() for x in (heads, logstd, actions, old_nlp, adv, ret, vp, old_mu, old_sigma)]
        N.check(lib.ag_ppo_loss(*[x.data_ptr() for x in dev], M, A, e_clip, critic_coef, b_coef, int(clip_value), 1,
                                d_heads.data_ptr(), new_mu.data_ptr(), new_sigma.data_ptr(), parts.data_ptr(),
                                ctypes.byref(nb), _stream()), "ag_ppo_loss")
        g_ls, g_hb = torch.empty(A, **f), torch.empty(A + 1, **f)
        kl_out, stats = torch.empty(1, **f), torch.empty(8, **f)
        N.check(lib.ag_ppo_loss_finalize(parts.data_ptr(), nb.value, M, A, dev[1].data_ptr(), ent_coef, critic_coef, b_coef,
                                         g_ls.data_ptr(), g_hb.data_ptr(), kl_out.data_ptr(), stats.data_ptr(), _stream()),
                "ag_ppo_loss_finalize")
        s = stats.cpu()
        assert abs(s[1].item() - float(g[c_key].mean())) < 2e-6 * max(1.0, abs(float(g[c_key].mean())))   # recorded c_loss
        assert abs(s[3].item() - float(g["b_loss"].mean())) < 2e-6 * max(1.0, float(g["b_loss"].mean()))   # recorded b_loss
        assert abs(s[0].item() - a_l.mean().item()) < 1e-5 and abs(s[2].item() - ent.item()) < 1e-6
        assert abs(s[4].item() - kl.item()) < 1e-5 * max(1.0, abs(kl.item())) and abs(kl_out.item() - kl.item()) < 1e-5 * max(1.0, abs(kl.item()))
        assert abs(s[5].item() - loss.item()) < 2e-5 * max(1.0, abs(loss.item()))
        lr_ = old_nlp - nlp.detach()       # policy_clip_fraction, lib/core/torch_ext.py:168-178
        clip_ref = ((lr_ < np.log(1.0 - e_clip)) | (lr_ > np.log(1.0 + e_clip))).float().mean().item()
        assert abs(s[6].item() - clip_ref) <= 1.0 / M + 1e-7 and clip_ref > 0.05
        np.testing.assert_allclose(d_heads.cpu().numpy(), hq.grad.numpy(), rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(g_ls.cpu().numpy(), lq.grad.numpy(), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(g_hb.cpu().numpy(), hq.grad.sum(0).numpy(), rtol=1e-4, atol=1e-7)
        assert torch.equal(new_mu.cpu(), mu) and torch.allclose(new_sigma.cpu(), sigma.expand(M, A))


def test_golden_mlp_forward_kernels(golden):
    """The reference's MLP (lib/network/mlp.py:4-39, ELU after every layer) with the recorded weights: first layer through
    ag_mlp_input_layer, last layer's ELU through ag_elu_heads -> recorded output."""
    from airgym_amd import _native as N
    lib = N.load()
    g = golden("ppo")
    x = t(g["mlp_x"]).cuda().contiguous()
    ws = [t(g[f"mlp_w{i}"]).cuda().contiguous() for i in range(3)]
    bs = [t(g[f"mlp_b{i}"]).cuda().contiguous() for i in range(3)]
    M, D = x.shape
    h0 = torch.empty(M, ws[0].shape[0], device="cuda")
    N.check(lib.ag_mlp_input_layer(x.data_ptr(), None, None, ws[0].data_ptr(), bs[0].data_ptr(), None, h0.data_ptr(), M, D,
                                   ws[0].shape[0], 0.0, 5.0, _stream()), "ag_mlp_input_layer")
    h1 = torch.nn.functional.elu(torch.addmm(bs[1], h0, ws[1].t()))
    z2 = torch.mm(h1, ws[2].t()).contiguous()                      # pre-activation WITHOUT bias; the kernel adds it
    C = ws[2].shape[0]
    Wh = torch.zeros(5, C, device="cuda"); bh = torch.zeros(5, device="cuda")
    Wh[:4, :4] = torch.eye(4, device="cuda")                       # heads = first four ELU outputs
    heads = torch.empty(M, 5, device="cuda")
    N.check(lib.ag_elu_heads(z2.data_ptr(), Wh.data_ptr(), bh.data_ptr(), heads.data_ptr(), M, C, 5, 1, bs[2].data_ptr(),
                             _stream()), "ag_elu_heads")
    np.testing.assert_allclose(z2.cpu().numpy(), g["mlp_y"], rtol=1e-5, atol=2e-6)       # written back: ELU(z + b)
    np.testing.assert_allclose(heads[:, :4].cpu().numpy(), g["mlp_y"][:, :4], rtol=1e-5, atol=2e-6)


def test_golden_adaptive_lr_in_adam_kernel(golden):
    """AdaptiveScheduler of the reference (lib/core/schedulers.py:19-32): recorded (start lr, kl) -> lr sequences, applied by
    ag_adam_clip_step's device-side schedule; the parameter update itself against torch.optim.Adam + clip_grad_norm_."""
    from airgym_amd import _native as N
    lib = N.load()
    g = golden("ppo")
    kls, lrs = g["sched_kls"], g["sched_lrs"]
    n = 1024
    gen = torch.Generator().manual_seed(0)
    k = 0
    for start in (3e-4, 1e-6, 1e-2):
        p = torch.randn(n, generator=gen)
        ref_p = torch.nn.Parameter(p.clone())
        opt = torch.optim.Adam([ref_p], lr=start, betas=(0.9, 0.999), eps=1e-8)
        pd = p.clone().cuda()
        m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
        state = torch.zeros(lib.ag_adam_state_bytes() // 8, dtype=torch.float64, device="cuda")
        for kl in kls:
            state[0] = start          # the fixture applies the rule to (start, kl) pairs independently
            lr = start
            grad = torch.randn(n, generator=gen) * 0.3
            gd = torch.cat((grad, torch.tensor([np.float32(kl)]))).cuda()
            N.check(lib.ag_adam_clip_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(), n,
                                          0.9, 0.999, 1e-8, 0.0, 1.5, 0.008, 1e-6, 1e-2, _stream()), "ag_adam_clip_step")
            # reference order (a2c_continuous.py:350-356, a2c_base.py:293-316): clip, Adam step with the CURRENT lr, then
            # the scheduler turns this minibatch's KL into the NEXT lr
            ref_p.grad = grad.clone()
            torch.nn.utils.clip_grad_norm_([ref_p], 1.5)
            for grp in opt.param_groups:
                grp["lr"] = lr
            opt.step()
            assert ppo_ref.adaptive_lr(lr, float(kl)) == lrs[k]                    # oracle == recorded (float64 KL)
            # the kernel sees the KL as float32 (it lives in the gradient buffer) and compares it with the float32 threshold:
            # kl = 2 * thr and kl = thr / 2 exactly stay "inside" on both sides, like the reference's float64 comparison
            # (min_lr / max_lr cross the C ABI as float: the 1e-2 bound is 0.00999999978 there)
            assert abs(state[0].item() - lrs[k]) <= 1e-7 * lrs[k], (start, kl, state[0].item(), lrs[k])
            k += 1
            np.testing.assert_allclose(pd.cpu().numpy(), ref_p.detach().numpy(), rtol=0, atol=3e-6)
    assert k == len(lrs)


# ---------------------------------------------------------------------------------------------- host reset_idx(subset)
@pytest.mark.parametrize("task", ["hovering", "tracking"])
def test_host_reset_of_a_subset_matches_oracle(Handle, task):
    """reset_idx(env_ids) called from the host on a subset (hovering.py:310-335 / tracking.py:159-192 -> ag_reset_envs): the
    listed envs are re-randomised exactly like the oracle's reset_idx with the same counter tick, flagged reset and cleared;
    every other env is untouched; the next step treats them like any freshly reset env (zero thrust for one step, Q2)."""
    n, seed = 300, 31
    ora = CLS[task](n, "rate", seed=seed)
    env = Handle(task, "rate", n, seed=seed)
    rng = np.random.default_rng(2)
    for t_ in range(5):
        a = rng.uniform(-0.5, 0.5, size=(n, 4)).astype(np.float32); a[:, 3] = -0.7
        ora.step(torch.from_numpy(a)); env.step(torch.from_numpy(a).cuda())
    ids = torch.tensor([0, 7, 63, 64, 65, 128, 299, 7], dtype=torch.int64)          # a duplicate is harmless
    before = env.get_state()
    ora.reset_idx(torch.unique(ids)); ora.tick += 1
    env.reset_envs(ids)
    st = env.get_state()
    np.testing.assert_allclose(st["root_states"].cpu().numpy(), ora.root_states.numpy(), rtol=0, atol=1e-6)
    assert np.array_equal(env.reset_buf.cpu().numpy(), ora.reset_buf.numpy())
    assert np.array_equal(st["progress"].cpu().numpy(), ora.progress_buf.numpy().astype(np.int32))
    assert np.array_equal(st["pre_actions"].cpu().numpy(), ora.pre_actions.numpy())
    mask = torch.ones(n, dtype=torch.bool); mask[ids] = False
    for k in before:
        assert torch.equal(before[k][mask.cuda()], st[k][mask.cuda()]), k              # the others are untouched
    assert np.array_equal(env.compact_reset_ids().cpu().numpy(), np.nonzero(ora.reset_buf.numpy())[0])
    for t_ in range(3):
        a = rng.uniform(-0.5, 0.5, size=(n, 4)).astype(np.float32); a[:, 3] = -0.7
        obs, _, rew, reset, _ = ora.step(torch.from_numpy(a)); env.step(torch.from_numpy(a).cuda())
        assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy())
        np.testing.assert_allclose(env.get_state()["root_states"].cpu().numpy(), ora.root_states.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(env.rew_buf.cpu().numpy(), rew.numpy(), rtol=0, atol=1e-5)
    env.close()


def test_host_reset_subset_through_the_task_class(Handle):
    """Hovering.reset_idx(env_ids) of the drop-in class no longer raises for a subset."""
    from argparse import Namespace
    import airgym_amd.envs  # noqa: F401  (registers the tasks, as `from airgym.envs import *` does in the reference's scripts)
    from airgym_amd.utils.task_registry import task_registry
    env, _ = task_registry.make_env("hovering", Namespace(num_envs=128, ctl_mode="rate", seed=3, sim_device="cuda:0", headless=True))
    env.reset()
    env.step(torch.zeros(128, 4, device="cuda"))
    p0 = env.root_states.clone()
    env.reset_idx(torch.tensor([3, 77], device="cuda"))
    p1 = env.root_states
    changed = (p0 != p1).any(dim=1).cpu()
    assert changed[3] and changed[77] and int(changed.sum()) == 2
    assert env.reset_buf[3] == 1 and env.reset_buf[77] == 1 and int(env.progress_buf[3]) == 0
    env.close()
