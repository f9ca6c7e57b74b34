"""ag_step_multi (K env steps per launch, state in registers between them; csrc/step_kernel.hip step_kernel_multi) against K
calls of ag_step_rollout (since round 5 the dedicated one-step kernel step_kernel_ws2<.., false>: TWO instantiations are compared
here, for every K including K = 1) on a twin handle: `for t in range(K): env.step(actions[t])`
(Hovering.step, hovering.py:286-308) must come out IDENTICAL, bit for bit - observations (noise included: Philox tick
tick0 + t), rewards, done flags, per-tile reward-term sums, the time-out flags, the state left behind and the ballot mask /
reset ids of the last step - with in-step resets happening inside the launch."""
import numpy as np
import pytest
import torch

from oracle.hovering_ref import HoveringRef

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Handle():
    from airgym_amd.hip_env import HipEnvHandle
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return HipEnvHandle


def _bufs(env, K, lib):
    n = env.num_envs
    f = dict(device="cuda", dtype=torch.float32)
    return dict(obs=torch.full((K, n, env.num_obs), -7.0, **f), rew=torch.full((K, n), -7.0, **f),
                done=torch.full((K, n), 9, dtype=torch.uint8, device="cuda"),
                tmo=torch.full((K, n), 9, dtype=torch.uint8, device="cuda"),
                tiles=torch.full((K, lib.ag_term_sum_tiles(n), 12), -7.0, **f))


@pytest.mark.parametrize("task,ctl,n,K,max_len,fix", [
    ("hovering", "rate", 1000, 1, 0, False),      # K = 1: the K-step kernel against the one-step kernel; ragged tail tile
    ("hovering", "rate", 4096, 1, 3, True),       # K = 1 with the time limit firing every third launch
    ("tracking", "vel", 520, 1, 2, False),        # K = 1, 48 observations
    ("hovering", "atti", 200, 1, 5, False),       # K = 1, five actions
    ("hovering", "rate", 1000, 4, 0, False),
    ("hovering", "rate", 4096, 24, 16, True),     # the time limit fires inside the launch (twice per env), flags on
    ("hovering", "atti", 130, 24, 9, False),      # 5 actions (scalar action loads), resets
    ("tracking", "vel", 776, 24, 0, False),       # 48 observations, LV cascade (4 controller arrays); d > 1 resets
    ("tracking", "pos", 200, 4, 6, True),
    ("hovering", "prop", 322, 7, 5, False),       # odd K, no controller memory
])
def test_multi_equals_k_single_steps(Handle, task, ctl, n, K, max_len, fix):
    from airgym_amd import _native as N
    lib = N.load()
    kw = dict(seed=31, env_id_offset=2048, max_episode_length=max_len, fix_time_outs=fix)
    a, b = Handle(task, ctl, n, **kw), Handle(task, ctl, n, **kw)
    A = a.num_actions
    g = torch.Generator(device="cuda").manual_seed(3)
    total_done = total_tmo = 0
    for launch in range(3):                        # the state written back by one launch feeds the next
        acts = (torch.randn(K, n, A, generator=g, device="cuda") * 0.6).clamp_(-1, 1)
        if ctl in ("rate", "atti"):
            acts[..., -1] = acts[..., -1] * 0.1 - 0.7      # near hover thrust: episodes survive a few steps
        ba, bb = _bufs(a, K, lib), _bufs(b, K, lib)
        a.step_multi(acts, ba["obs"], ba["rew"], ba["done"], ba["tmo"], ba["tiles"])
        for t in range(K):
            b.step_rollout(acts[t], bb["obs"][t], bb["rew"][t], bb["done"][t], bb["tiles"][t])
            bb["tmo"][t].copy_(b.time_out_buf)
        torch.cuda.synchronize()
        for key in ("obs", "rew", "done", "tmo", "tiles"):
            x, y = ba[key], bb[key]
            same = torch.equal(x, y) if x.dtype == torch.uint8 else bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())
            assert same, (key, launch, (x.float() - y.float()).abs().max().item())
        sa, sb = a.get_state(), b.get_state()
        for key in sa:
            assert torch.equal(sa[key], sb[key]), (key, launch)
        assert torch.equal(a.time_out_buf, b.time_out_buf)
        assert a.tick == b.tick
        ia, ib = a.compact_reset_ids(), b.compact_reset_ids()
        assert torch.equal(ia, ib)
        total_done += int(ba["done"].sum())
        total_tmo += int(ba["tmo"].sum())
    if max_len or task == "tracking":
        assert total_done > 0
    if fix and max_len:
        assert total_tmo > 0          # the opt-in time-out flag fired inside a launch
    a.close(); b.close()


def test_multi_matches_oracle_trajectory(Handle):
    """... and against the oracle directly: 24 steps in one launch, 1e-5 on the state, flags bit-exact (north_star's bar)."""
    from airgym_amd import _native as N
    lib = N.load()
    n, K = 512, 24
    env = Handle("hovering", "rate", n, seed=5)
    ora = HoveringRef(n, "rate", seed=5)
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(K, n, 4, generator=g) * 1.6 - 0.8
    b = _bufs(env, K, lib)
    env.step_multi(acts.cuda(), b["obs"], b["rew"], b["done"], None, None)
    for t in range(K):
        obs, _, rew, reset, _ = ora.step(acts[t])
        assert torch.equal(b["done"][t].cpu().long(), reset), t
        assert (b["obs"][t].cpu() - obs).abs().max().item() < 2e-5, t
        assert (b["rew"][t].cpu() - rew).abs().max().item() < 1e-5, t
    assert (env.get_state()["root_states"].cpu() - ora.root_states).abs().max().item() < 1e-5
    env.close()


def test_multi_validates(Handle):
    env = Handle("hovering", "rate", 64)
    f = dict(device="cuda", dtype=torch.float32)
    obs, rew = torch.zeros(2, 64, 18, **f), torch.zeros(2, 64, **f)
    done = torch.zeros(2, 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(AssertionError):
        env.step_multi(torch.zeros(2, 64, 5, **f), obs, rew, done)          # wrong action width
    with pytest.raises(AssertionError):
        env.step_multi(torch.zeros(3, 64, 4, **f), obs, rew, done)          # K = 3 but buffers hold 2 slices
    odd = Handle("hovering", "rate", 65)                                       # 65 * 18 % 4 != 0: slices would be misaligned
    with pytest.raises((ValueError, RuntimeError)):
        odd.step_multi(torch.zeros(2, 65, 4, **f), torch.zeros(2, 65, 18, **f), torch.zeros(2, 65, **f),
                       torch.zeros(2, 65, dtype=torch.uint8, device="cuda"))
    odd.close(); env.close()
