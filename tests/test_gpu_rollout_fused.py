"""ag_step_rollout_fused (policy sampling + env step + reward / episode accounting in ONE launch,
csrc/step_kernel.hip step_kernel_ws2<.., true>) against the three launches it replaces
(ag_policy_sample -> ag_step_rollout -> ag_rollout_account) on twin handles: one whole step of A2CBase.play_steps
(lib/agent/a2c_base.py:651-695) must come out the same, step after step, resets included (the sampler's outputs bit-identical;
the env step's to 2e-6: two compilations of the same expressions) - and the opt-in time-out flag (AG_FLAG_FIX_TIME_OUTS)
against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle.hovering_ref import HoveringRef
from oracle.tracking_ref import TrackingRef

pytestmark = pytest.mark.gpu


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def Handle():
    from airgym_amd.hip_env import HipEnvHandle
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return HipEnvHandle


class _Rollout:
    """The rollout-side buffers of one env handle (what the agent owns)."""

    def __init__(self, env, H, lib):
        n, A = env.num_envs, env.num_actions
        f = dict(device="cuda", dtype=torch.float32)
        self.obs = torch.zeros(H + 1, n, env.num_obs, **f)
        self.raw = torch.zeros(H, n, **f)
        self.done = torch.zeros(H + 1, n, dtype=torch.uint8, device="cuda")
        self.tiles = torch.zeros(H, lib.ag_term_sum_tiles(n), 12, **f)
        self.actions, self.mus, self.sigmas = (torch.zeros(H, n, A, **f) for _ in range(3))
        self.nlp, self.values, self.shaped = (torch.zeros(H, n, **f) for _ in range(3))
        self.cur_r, self.cur_s, self.cur_l = (torch.zeros(n, **f) for _ in range(3))
        self.env_actions = torch.zeros(n, A, **f)


@pytest.mark.parametrize("task,ctl,n,norm_value,bootstrap,fix", [
    ("hovering", "rate", 1000, True, True, False),     # ragged tail tile, value de-normalisation, (never-firing) bootstrap
    ("hovering", "atti", 130, False, False, False),    # 5 actions (scalar stores, odd LDS stride)
    ("tracking", "vel", 777, True, True, True),        # 48 observations; time-outs that fire
    ("hovering", "rate", 4096, True, True, True),
])
def test_fused_step_equals_three_launches(Handle, task, ctl, n, norm_value, bootstrap, fix):
    from airgym_amd import _native as N
    lib = N.load()
    H, seed_env, seed_pol, offset = 6, 23, 0x0123456789ABCDEF, 512
    max_len = 12 if fix else 0                   # short episodes so that resets and time-outs happen inside the test
    kw = dict(seed=seed_env, env_id_offset=offset, max_episode_length=max_len, fix_time_outs=fix)
    a, b = Handle(task, ctl, n, **kw), Handle(task, ctl, n, **kw)
    A = a.num_actions
    ra, rb = _Rollout(a, H, lib), _Rollout(b, H, lib)
    g = torch.Generator(device="cuda").manual_seed(1)
    logstd = 0.2 * torch.randn(A, device="cuda", generator=g) - 0.5
    vmean = torch.tensor([0.4], dtype=torch.float64, device="cuda")
    vvar = torch.tensor([3.0], dtype=torch.float64, device="cuda")
    counter = torch.tensor([3], dtype=torch.int64, device="cuda")
    pa = torch.zeros(H, lib.ag_rollout_account_blocks(n), 4, dtype=torch.float64, device="cuda")
    pb = torch.zeros(H, lib.ag_term_sum_tiles(n), 4, dtype=torch.float64, device="cuda")
    scale, shift, lo, hi, gamma = 0.1, 0.05, -2.0, 2.0, 0.99
    a.reset_all(); b.reset_all()
    total_done = 0
    for rollout in range(4):                     # 24 steps: the 12-step time limit fires twice
        for slot in range(H):
            # the two instantiations of the kernel are separate compilations of the same expressions (LLVM may contract a
            # different product of a sum of products into the FMA: <= 1 ulp per op), so every step starts from ONE state - the
            # comparison is per step, ulp-level differences cannot pile up or flip a termination threshold later on
            sa = a.get_state()
            b.set_state(**sa)
            rb.cur_r.copy_(ra.cur_r); rb.cur_s.copy_(ra.cur_s); rb.cur_l.copy_(ra.cur_l)
            heads = torch.randn(n, A + 1, device="cuda", generator=g)
            heads[:, A] *= 4
            if ctl in ("rate", "atti"):
                heads[:, A - 1] -= 0.6           # thrust command around hover so that episodes last
            # ---- three launches
            N.check(lib.ag_policy_sample(heads.data_ptr(), logstd.data_ptr(), vmean.data_ptr() if norm_value else None,
                                         vvar.data_ptr() if norm_value else None, 1e-5, seed_pol, counter.data_ptr(), H, slot,
                                         offset, ra.actions[slot].data_ptr(), ra.nlp[slot].data_ptr(), ra.values[slot].data_ptr(),
                                         ra.mus[slot].data_ptr(), ra.sigmas[slot].data_ptr(), ra.env_actions.data_ptr(), n, A,
                                         _stream()), "ag_policy_sample")
            a.step_rollout(ra.env_actions, ra.obs[slot + 1], ra.raw[slot], ra.done[slot + 1], ra.tiles[slot])
            tmo = a.time_out_buf.view(torch.uint8)
            N.check(lib.ag_rollout_account(ra.raw[slot].data_ptr(), ra.done[slot + 1].data_ptr(),
                                           tmo.data_ptr() if bootstrap else None, ra.values[slot].data_ptr() if bootstrap else None,
                                           scale, shift, lo, hi, 0, gamma, ra.shaped[slot].data_ptr(), ra.cur_r.data_ptr(),
                                           ra.cur_s.data_ptr(), ra.cur_l.data_ptr(), pa[slot].data_ptr(), n, _stream()),
                    "ag_rollout_account")
            # ---- one launch
            t = N.AgRolloutTail()
            t.struct_size = ctypes.sizeof(N.AgRolloutTail)
            t.heads_dev, t.logstd_dev = heads.data_ptr(), logstd.data_ptr()
            t.vmean_dev = vmean.data_ptr() if norm_value else None
            t.vvar_dev = vvar.data_ptr() if norm_value else None
            t.veps, t.seed, t.counter_dev = 1e-5, seed_pol, counter.data_ptr()
            t.horizon, t.slot, t.id_offset = H, slot, offset
            t.actions_dev, t.neglogp_dev, t.values_dev = rb.actions[slot].data_ptr(), rb.nlp[slot].data_ptr(), rb.values[slot].data_ptr()
            t.mus_dev, t.sigmas_dev = rb.mus[slot].data_ptr(), rb.sigmas[slot].data_ptr()
            t.scale, t.shift, t.min_val, t.max_val, t.log_val, t.gamma = scale, shift, lo, hi, 0, gamma
            t.bootstrap_timeouts = int(bootstrap)
            t.shaped_dev, t.cur_rew_dev = rb.shaped[slot].data_ptr(), rb.cur_r.data_ptr()
            t.cur_shaped_dev, t.cur_len_dev, t.partials_dev = rb.cur_s.data_ptr(), rb.cur_l.data_ptr(), pb[slot].data_ptr()
            b.step_rollout_fused(t, rb.obs[slot + 1], rb.raw[slot], rb.done[slot + 1], rb.tiles[slot])
            torch.cuda.synchronize()
            # the sampler is a pure function of (heads, counters): bit-identical
            for name in ("actions", "mus", "sigmas", "nlp", "values"):
                assert torch.equal(getattr(ra, name)[slot], getattr(rb, name)[slot]), (name, rollout, slot)
            # the env step: same expressions, separately compiled
            for name, tol in (("raw", 2e-6), ("shaped", 2e-6), ("tiles", 2e-4)):
                d = (getattr(ra, name)[slot] - getattr(rb, name)[slot]).abs().max().item()
                assert d <= tol, (name, d, rollout, slot)
            dobs = (ra.obs[slot + 1] - rb.obs[slot + 1]).abs().max().item()
            assert dobs <= 2e-6, (dobs, rollout, slot)
            assert torch.equal(ra.done[slot + 1], rb.done[slot + 1])
            assert torch.allclose(ra.cur_r, rb.cur_r, rtol=0, atol=1e-5) and torch.allclose(ra.cur_s, rb.cur_s, rtol=0, atol=1e-5)
            assert torch.equal(ra.cur_l, rb.cur_l)
            assert torch.equal(a.time_out_buf, b.time_out_buf) and torch.equal(a.reset_mask, b.reset_mask)
            # episode sums: per 256-env block vs per 64-env tile - same totals (sums of f32 values in f64)
            assert torch.allclose(pa[slot].sum(0), pb[slot].sum(0), rtol=1e-6, atol=1e-4), (rollout, slot)
            total_done += int(ra.done[slot + 1].sum())
            if fix:
                # the time-out flag marks exactly the envs whose episode ran to the limit; with the bootstrap their shaped
                # reward carries gamma * value
                tm = b.time_out_buf
                assert torch.equal(tm & (rb.done[slot + 1] == 0), torch.zeros_like(tm))        # a time-out is a done
                if bootstrap and tm.any():
                    plain = torch.clamp((rb.raw[slot] + shift) * scale, lo, hi)
                    assert torch.allclose(rb.shaped[slot][tm], (plain + gamma * rb.values[slot])[tm], rtol=1e-6, atol=1e-6)
            else:
                assert not b.time_out_buf.any()                                               # quirk Q3: never true
        counter.add_(1)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.allclose(sa[k].float(), sb[k].float(), rtol=0, atol=2e-6), k
    assert total_done > 0 or not fix          # the 12-step time limit of the `fix` cases fires inside the test
    a.close(); b.close()


@pytest.mark.parametrize("task,ctl", [("hovering", "rate"), ("tracking", "vel")])
def test_fix_time_outs_matches_oracle(Handle, task, ctl):
    """AG_FLAG_FIX_TIME_OUTS: time_out_buf = 'the episode reached the time limit this step' (progress >= max - 1 before the
    reset); everything else is unchanged.  Oracle: HoveringRef(fix_time_outs=True)."""
    n, max_len = 300, 9
    cls = {"hovering": HoveringRef, "tracking": TrackingRef}[task]
    ora = cls(n, ctl, seed=9, fix_time_outs=True)
    ora.max_episode_length = max_len
    env = Handle(task, ctl, n, seed=9, max_episode_length=max_len, fix_time_outs=True)
    plain = Handle(task, ctl, n, seed=9, max_episode_length=max_len)
    rng = np.random.default_rng(4)
    fired = 0
    for t in range(30):
        act = rng.uniform(-0.3, 0.3, size=(n, 4)).astype(np.float32)
        if ctl == "rate":
            act[:, 3] = rng.uniform(-0.75, -0.65, size=n)
        obs, _, rew, reset, extras = ora.step(torch.from_numpy(act))
        env.step(torch.from_numpy(act).cuda()); plain.step(torch.from_numpy(act).cuda())
        assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy()), t
        assert np.array_equal(env.time_out_buf.cpu().numpy(), extras["time_outs"].numpy()), t
        assert torch.equal(env.obs_buf, plain.obs_buf) and torch.equal(env.rew_buf, plain.rew_buf)      # nothing else moves
        assert not plain.time_out_buf.any()
        fired += int(env.time_out_buf.sum())
    assert fired > 0
    env.close(); plain.close()


def test_fused_rollout_entry_point_validates(Handle):
    from airgym_amd import _native as N
    env = Handle("hovering", "rate", 64)
    t = N.AgRolloutTail()
    buf = torch.zeros(64 * 18, device="cuda")
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    t.struct_size = 4                                     # ABI guard
    with pytest.raises(ValueError):
        env.step_rollout_fused(t, buf, buf[:64].contiguous(), d)
    t.struct_size = ctypes.sizeof(N.AgRolloutTail)        # required pointers missing
    with pytest.raises(ValueError):
        env.step_rollout_fused(t, buf, buf[:64].contiguous(), d)
    env.close()
    pl = Handle("balloon", "rate", 64)
    with pytest.raises(RuntimeError):
        pl.step_rollout_fused(t, torch.zeros(64 * 18, device="cuda"), buf[:64].contiguous(), d)
    pl.close()


def test_stagger_episode_phase_matches_oracle(Handle):
    """AG_FLAG_STAGGER_PHASE (opt-in): a full reset gives every env its own progress, U{0 .. max_len - 2} from the counter RNG
    keyed by the GLOBAL env id; the episodes then end spread over max_len steps instead of all in the same one; in-step resets
    still start at 0 (hovering.py:333).  Oracle: HoveringRef(stagger_episode_phase=True)."""
    n, max_len, off = 700, 40, 4096
    ora = HoveringRef(n, "rate", seed=11, env_id_offset=off, stagger_episode_phase=True)
    # the oracle reads max_episode_length at reset time: rebuild its initial state with the short limit
    ora.max_episode_length = max_len
    ora.tick = 0
    ora._reset_all()
    ora.tick = 1
    env = Handle("hovering", "rate", n, seed=11, env_id_offset=off, max_episode_length=max_len, stagger_episode_phase=True)
    p0 = env.get_state()["progress"].cpu().numpy().astype(np.int64)
    assert np.array_equal(p0, ora.progress_buf.numpy())
    assert p0.min() >= 0 and p0.max() <= max_len - 2 and len(np.unique(p0)) > max_len // 2
    # sharding-invariant: the same global ids on a handle with another offset give the same phases
    half = Handle("hovering", "rate", n // 2, seed=11, env_id_offset=off + n // 2, max_episode_length=max_len,
                  stagger_episode_phase=True)
    assert np.array_equal(half.get_state()["progress"].cpu().numpy().astype(np.int64), p0[n // 2:])
    half.close()
    rng = np.random.default_rng(5)
    per_step = []
    for t in range(max_len + 5):
        act = rng.uniform(-0.2, 0.2, size=(n, 4)).astype(np.float32)
        act[:, 3] = rng.uniform(-0.75, -0.65, size=n)
        obs, _, rew, reset, _ = ora.step(torch.from_numpy(act))
        env.step(torch.from_numpy(act).cuda())
        assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy()), t
        assert np.array_equal(env.get_state()["progress"].cpu().numpy().astype(np.int64), ora.progress_buf.numpy()), t
        per_step.append(int(reset.sum()))
    # the time limit fires on many different steps, never for (nearly) all envs at once
    assert sum(1 for c in per_step if c > 0) > max_len // 2 and max(per_step) < n // 4
    env.close()
    with pytest.raises(ValueError):
        Handle("tracking", "vel", 64, stagger_episode_phase=True)
