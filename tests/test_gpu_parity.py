"""GPU parity tests proper: libairgym_hip.so (through its C ABI, via HipEnvHandle) against the
oracle on identical seeded inputs.  Bar (BASELINE.json north_star): state trajectories within 1e-5
abs over 100 steps, integer reset indices bit-exact.

Run on the MI355X box:  python -m pytest tests -m gpu -q
"""
import numpy as np
import pytest
import torch

from oracle.hovering_ref import HoveringRef
from oracle.tracking_ref import TrackingRef

pytestmark = pytest.mark.gpu

CLS = {"hovering": HoveringRef, "tracking": TrackingRef}
STATE_TOL = 1e-5     # north_star: 1e-5 abs over 100 steps
OBS_TOL = 2e-5       # obs = state + sigma*N(0,1): Box-Muller via OCML vs numpy log/sin/cos, sigma up to 0.4
REW_TOL = 1e-5


def scripted_actions(rng, n, A, t, ctl):
    a = rng.uniform(-0.8, 0.8, size=(n, A)).astype(np.float32)
    if ctl in ("rate", "atti"):
        a[:, -1] = rng.uniform(-0.9, -0.3, size=n)
    if ctl == "atti":
        a[:, 0] = rng.uniform(0.6, 0.95, size=n)
        a[:, 1:4] *= 0.3
    if ctl == "prop":
        a = rng.uniform(0.05, 0.3, size=(n, A)).astype(np.float32)
    if t % 7 == 0:
        a[: n // 4] = a[0]
    return a


@pytest.fixture(scope="module")
def Handle():
    from airgym_amd.hip_env import HipEnvHandle
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return HipEnvHandle


def _compare_step(env, ora, a, t, check_terms=True):
    obs, _, rew, reset, extras = ora.step(torch.from_numpy(a))
    env.step(torch.from_numpy(a).cuda())
    st = env.get_state()
    ids = env.compact_reset_ids().cpu().numpy()
    assert np.array_equal(ids, ora.last_reset_env_ids.numpy()), f"reset ids differ at step {t}"
    assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy()), f"reset_buf step {t}"
    np.testing.assert_allclose(st["root_states"].cpu().numpy(), ora.root_states.numpy(), rtol=0, atol=STATE_TOL,
                               err_msg=f"state step {t}")
    np.testing.assert_allclose(env.obs_buf.cpu().numpy(), obs.numpy(), rtol=0, atol=OBS_TOL, err_msg=f"obs step {t}")
    np.testing.assert_allclose(env.rew_buf.cpu().numpy(), rew.numpy(), rtol=0, atol=REW_TOL, err_msg=f"rew step {t}")
    assert np.array_equal(st["progress"].cpu().numpy(), ora.progress_buf.numpy().astype(np.int32))
    assert not env.time_out_buf.any()
    if check_terms:
        for k, v in extras["item_reward_info"].items():
            if torch.is_tensor(v):
                np.testing.assert_allclose(env.reward_terms[k].cpu().numpy(), v.numpy(), rtol=0, atol=REW_TOL,
                                           err_msg=f"{k} step {t}")
        np.testing.assert_allclose(env.cmd_thrusts.cpu().numpy(), ora.cmd_thrusts.numpy(), rtol=0, atol=1e-5)
    return len(ids)


@pytest.mark.parametrize("task", ["hovering", "tracking"])
@pytest.mark.parametrize("ctl", ["rate", "vel", "atti", "pos", "prop"])
def test_100_step_trajectory(Handle, task, ctl):
    n, steps, seed = 64, 100, 1234
    ora = CLS[task](n, ctl_mode=ctl, seed=seed)
    env = Handle(task, ctl, n, seed=seed)
    np.testing.assert_allclose(env.get_state()["root_states"].cpu().numpy(), ora.root_states.numpy(), rtol=0, atol=1e-6)
    assert (env.reset_buf == 1).all()            # base_task.py:75
    rng = np.random.default_rng(7)
    n_resets = 0
    for t in range(steps):
        n_resets += _compare_step(env, ora, scripted_actions(rng, n, env.num_actions, t, ctl), t)
    assert env.tick == ora.tick
    if task == "tracking":
        assert n_resets > 0
    env.close()


@pytest.mark.parametrize("n", [1, 63, 65, 300, 1000])
def test_ragged_sizes(Handle, n):
    """num_envs not a multiple of the wavefront / block: tail lanes must not leak."""
    ora = HoveringRef(n, "rate", seed=3)
    env = Handle("hovering", "rate", n, seed=3)
    rng = np.random.default_rng(1)
    for t in range(10):
        _compare_step(env, ora, scripted_actions(rng, n, 4, t, "rate"), t)
    env.close()


def test_tracking_ragged_env_count_matches_oracle(Handle):
    n = 700
    ora = TrackingRef(n, "vel", seed=11)
    env = Handle("tracking", "vel", n, seed=11)
    rng = np.random.default_rng(2)
    for t in range(12):
        _compare_step(env, ora, scripted_actions(rng, n, 4, t, "vel"), t)
    env.close()


def test_parity_mode_supplied_randoms(Handle):
    n = 256
    ora = HoveringRef(n, "rate", seed=5)
    env = Handle("hovering", "rate", n, seed=5)
    rng = np.random.default_rng(3)
    for t in range(30):
        a = scripted_actions(rng, n, 4, t, "rate")
        noise = rng.standard_normal((n, 18)).astype(np.float32)
        uni = rng.random((n, 12)).astype(np.float32)
        obs, _, rew, reset, _ = ora.step(torch.from_numpy(a), noise=torch.from_numpy(noise),
                                         reset_uniforms=torch.from_numpy(uni))
        env.step_with_inputs(torch.from_numpy(a).cuda(), torch.from_numpy(noise), torch.from_numpy(uni))
        assert np.array_equal(env.reset_buf.cpu().numpy(), reset.numpy())
        # with the noise supplied the observation tolerance is the state tolerance
        np.testing.assert_allclose(env.obs_buf.cpu().numpy(), obs.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(env.get_state()["root_states"].cpu().numpy(), ora.root_states.numpy(), rtol=0, atol=1e-5)
    env.close()


def test_sharding_is_invisible(Handle):
    """env ids are global: a shard [256, 512) of a 512-env job reproduces rows 256.. of the full job."""
    full = Handle("hovering", "rate", 512, seed=21)
    shard = Handle("hovering", "rate", 256, seed=21, env_id_offset=256)
    g = torch.Generator().manual_seed(0)
    for t in range(20):
        a = (torch.rand(512, 4, generator=g) * 1.6 - 0.8).cuda()
        full.step(a)
        shard.step(a[256:].contiguous())
        assert torch.equal(full.obs_buf[256:], shard.obs_buf)
        assert torch.equal(full.rew_buf[256:], shard.rew_buf)
        assert torch.equal(full.reset_buf[256:], shard.reset_buf)
    full.close(); shard.close()


def test_step_into_rollout_slot(Handle):
    n, H = 300, 4
    env = Handle("hovering", "rate", n, seed=2)
    twin = Handle("hovering", "rate", n, seed=2)
    obs = torch.zeros(H, n, 18, device="cuda")
    rew = torch.zeros(H, n, device="cuda")
    done = torch.zeros(H, n, dtype=torch.int64, device="cuda")
    g = torch.Generator().manual_seed(1)
    for t in range(H):
        a = (torch.rand(n, 4, generator=g) * 1.6 - 0.8).cuda()
        env.step_into(a, obs[t], rew[t], done[t])
        twin.step(a)
        assert torch.equal(obs[t], twin.obs_buf) and torch.equal(rew[t], twin.rew_buf)
        assert torch.equal(done[t], twin.reset_buf)
    env.close(); twin.close()


def test_episode_end_and_thrust_zero_quirk(Handle):
    n = 128
    ora = HoveringRef(n, "rate", seed=9)
    env = Handle("hovering", "rate", n, seed=9)
    ora.progress_buf[:] = 2397
    env.set_state(progress=torch.full((n,), 2397, dtype=torch.int32))
    a = np.zeros((n, 4), np.float32)
    a[:, 3] = -0.7
    _compare_step(env, ora, a, 0)
    k = _compare_step(env, ora, a, 1)
    assert k == n                                   # every env hit progress >= 2399 (hovering.py:435)
    st = env.get_state()
    assert (st["progress"] == 0).all() and (st["pre_actions"] == 0).all() and (st["was_reset"] == 1).all()
    vz0 = st["root_states"][:, 9].clone()
    _compare_step(env, ora, a, 2)
    dv = env.get_state()["root_states"][:, 9] - vz0
    assert torch.allclose(dv, torch.full_like(dv, -0.0981), atol=2e-5)   # zero thrust for one step (Q2)
    env.close()


def test_full_size_properties(Handle):
    """BASELINE config 1 size (65 536 envs): size-independent properties instead of the oracle."""
    n = 65536
    env = Handle("hovering", "rate", n, seed=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    tot_done = 0
    for t in range(200):
        a = torch.randn(n, 4, generator=g, device="cuda").clamp(-1, 1)
        env.step(a)
        ids = env.compact_reset_ids()
        nz = env.reset_buf.nonzero().squeeze(-1)
        assert torch.equal(ids.long(), nz)          # ascending and complete: bit-exact vs nonzero
        tot_done += len(ids)
    st = env.get_state()
    q = st["root_states"][:, 3:7]
    assert torch.allclose(q.norm(dim=-1), torch.ones(n, device="cuda"), atol=1e-5)   # unit quaternions
    assert torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all()
    assert (st["progress"] >= 0).all() and (st["progress"] < 2400).all()
    # idempotent outputs: envs flagged done were re-randomised inside the reset box (hovering.py:316-317)
    done = env.reset_buf.bool()
    if done.any():
        assert (st["root_states"][done, 0:3].abs() <= 1.0).all() and (st["progress"][done] == 0).all()
    assert tot_done > 0
    # observation = state (+noise) - target: statistical check of the noise sigmas on the position block
    d = env.obs_buf[~done, 9:12] - st["root_states"][~done, 0:3]
    assert abs(d.std().item() - 5e-3) < 2e-4 and abs(d.mean().item()) < 1e-4
    env.close()


def test_splitk_linear_matches_plain_autograd(Handle):
    """split-K weight gradient (airgym_amd/lib/network/splitk_linear.py) == torch autograd of F.linear."""
    import torch.nn.functional as F
    from airgym_amd.lib.network.splitk_linear import linear
    g = torch.Generator(device="cuda").manual_seed(0)
    for K, N in [(18, 256), (256, 256), (256, 5)]:
        x = torch.randn(16384, K, device="cuda", generator=g, requires_grad=True)
        w = torch.randn(N, K, device="cuda", generator=g, requires_grad=True)
        b = torch.randn(N, device="cuda", generator=g, requires_grad=True)
        go = torch.randn(16384, N, device="cuda", generator=g)
        y = linear(x, w, b); y.backward(go)
        got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        for t_ in (x, w, b):
            t_.grad = None
        y2 = F.linear(x, w, b); y2.backward(go)
        ref = (y2.detach(), x.grad, w.grad, b.grad)
        for a_, r_ in zip(got, ref):
            assert torch.allclose(a_, r_, rtol=1e-4, atol=1e-3 * r_.abs().max().item())


def test_fused_ppo_loss_matches_composed_torch_ops(Handle):
    """ag_ppo_loss (one kernel) vs the composed torch ops that restate calc_gradients: loss terms, KL,
    gradients w.r.t. heads and logstd, and the mu/sigma write-back."""
    from airgym_amd.lib.core import common_losses, torch_ext
    from airgym_amd.lib.core.fused_loss import fused_ppo_loss
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    g = torch.Generator(device="cuda").manual_seed(0)
    for A, clip_value, btype in [(4, False, "bound"), (5, True, "regularisation"), (4, True, "bound")]:
        M = 70001
        heads = torch.randn(M, A + 1, device="cuda", generator=g)
        heads[:, :A] *= 1.5                                   # some |mu| > 1.1 so the bound loss is active
        heads.requires_grad_(True)
        logstd = (0.3 * torch.randn(A, device="cuda", generator=g)).requires_grad_(True)
        actions = torch.randn(M, A, device="cuda", generator=g)
        old_mu = torch.randn(M, A, device="cuda", generator=g) * 0.1
        old_sigma = torch.rand(M, A, device="cuda", generator=g) + 0.5
        adv = torch.randn(M, device="cuda", generator=g)
        returns = torch.randn(M, 1, device="cuda", generator=g)
        old_values = returns + 0.3 * torch.randn(M, 1, device="cuda", generator=g)
        cfg = dict(e_clip=0.2, critic_coef=2.0, entropy_coef=0.01, bounds_loss_coef=1e-4)
        # composed reference path
        mu, value = heads[:, :A], heads[:, A:]
        ls = mu * 0.0 + logstd
        sigma = torch.exp(ls)
        nlp = ModelA2CContinuousLogStd.neglogp(actions, mu, sigma, ls)
        old_nlp = (nlp.detach() + 0.3 * torch.randn(M, device="cuda", generator=g)).contiguous()
        a = common_losses.actor_loss(old_nlp, nlp, adv, True, 0.2).mean()
        c = common_losses.critic_loss(old_values, value, 0.2, returns, clip_value).mean()
        b = (common_losses.bound_loss(mu) if btype == "bound" else common_losses.reg_loss(mu)).mean()
        ent = (0.5 + 0.5 * np.log(2 * np.pi) + ls).sum(-1).mean()
        loss_ref = a + 0.5 * c * 2.0 - ent * 0.01 + b * 1e-4
        loss_ref.backward()
        g_heads_ref, g_ls_ref = heads.grad.clone(), logstd.grad.clone()
        kl_ref = torch_ext.policy_kl(mu.detach(), sigma.detach(), old_mu, old_sigma)
        heads.grad = None; logstd.grad = None
        om, osig = old_mu.clone(), old_sigma.clone()
        loss, stats = fused_ppo_loss(heads, logstd, actions, old_nlp, adv, returns, old_values, om, osig,
                                     clip_value=clip_value, bound_loss_type=btype, write_back=True, **cfg)
        loss.backward()
        assert torch.allclose(loss, loss_ref, rtol=1e-5, atol=1e-6)
        for got, ref in zip(stats, (a, c, ent, b, kl_ref)):
            assert torch.allclose(got, ref.detach(), rtol=2e-5, atol=1e-6), (got.item(), ref.item())
        assert torch.allclose(heads.grad, g_heads_ref, rtol=1e-4, atol=1e-10)
        assert torch.allclose(logstd.grad, g_ls_ref, rtol=1e-4, atol=1e-7)
        assert torch.equal(om, mu.detach()) and torch.allclose(osig, sigma.detach())


def test_fused_linear_elu_backward(Handle):
    """ag_elu_bwd_bias + split-K wgrad inside _LinearEluFn == autograd of F.elu(F.linear(...))."""
    import torch.nn.functional as F
    from airgym_amd.lib.network.splitk_linear import linear_elu
    g = torch.Generator(device="cuda").manual_seed(0)
    for K, C, M in [(18, 256, 16384), (256, 256, 24576), (32, 64, 8192)]:
        x = torch.randn(M, K, device="cuda", generator=g, requires_grad=True)
        w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).requires_grad_(True)
        b = torch.randn(C, device="cuda", generator=g, requires_grad=True)
        go = torch.randn(M, C, device="cuda", generator=g)
        y = linear_elu(x, w, b); y.backward(go)
        got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        for t_ in (x, w, b):
            t_.grad = None
        y2 = F.elu(F.linear(x, w, b)); y2.backward(go)
        for a_, r_ in zip(got, (y2.detach(), x.grad, w.grad, b.grad)):
            assert torch.allclose(a_, r_, rtol=1e-4, atol=1e-3 * r_.abs().max().item())


def test_fused_adam_clip_lr_step(Handle):
    """ag_adam_clip_step == clip_grad_norm + FlatAdam.step + AdaptiveScheduler (the python/torch path)."""
    from airgym_amd.lib.agent.a2c_continuous import FlatAdam
    from airgym_amd.lib.core.schedulers import AdaptiveScheduler
    sch = AdaptiveScheduler(0.008)
    g = torch.Generator(device="cuda").manual_seed(0)
    n = 71945
    p0 = torch.randn(n, device="cuda", generator=g)
    gbuf_a = torch.zeros(n + 1, device="cuda"); gbuf_b = torch.zeros(n + 1, device="cuda")
    a = FlatAdam(p0.clone(), gbuf_a[:-1], 3e-4)
    b = FlatAdam(p0.clone(), gbuf_b[:-1], 3e-4)
    lr_host = 3e-4
    for it, kl in enumerate([0.0, 0.02, 0.003, 0.009, 0.1, 0.001, 0.001]):
        grad = torch.randn(n, device="cuda", generator=g) * (0.001 if it % 2 else 0.05)
        gbuf_a[:-1] = grad; gbuf_a[-1] = kl
        gbuf_b[:-1] = grad; gbuf_b[-1] = kl
        a.fused_clip_step(gbuf_a, 1.5, 0.008, sch.min_lr, sch.max_lr)
        gb = gbuf_b[:-1]
        gb.mul_(torch.clamp(1.5 / (torch.linalg.vector_norm(gb) + 1e-6), max=1.0))
        b.step()
        lr_host = sch.update(lr_host, 0, 0, 0, float(np.float32(kl)))[0]
        b.lr.fill_(lr_host)
        assert torch.allclose(a.p, b.p, rtol=0, atol=2e-6), it
        assert abs(a.lr.item() - lr_host) < 1e-12 and a.step_t.item() == it + 1
        assert torch.allclose(gbuf_a[:-1], gbuf_b[:-1], rtol=1e-5, atol=1e-9)     # clipped gradient left in place


def test_wave_specialised_kernel_matches_single_wave(Handle):
    """The shipped kernel (physics wave + noise wave per 64 envs, Philox in the kernel) and the parity-mode kernel (one wave
    does everything, random numbers from the caller) are two schedules of the same device functions: fed the Philox numbers
    the shipped kernel draws (oracle/philox.py, same key / counters), they must agree up to FMA contraction differences
    between the two compilations (<= 1 ulp per op)."""
    from oracle import philox
    for task, ctl, n in [("hovering", "rate", 1000), ("tracking", "vel", 777), ("hovering", "atti", 130)]:
        a = Handle(task, ctl, n, seed=17)
        b = Handle(task, ctl, n, seed=17)
        ids = np.arange(n, dtype=np.uint32)
        g = torch.Generator(device="cuda").manual_seed(3)
        for t in range(40):
            act = torch.randn(n, a.num_actions, generator=g, device="cuda").clamp(-1, 1)
            tick = a.tick
            a.step(act)
            z = torch.from_numpy(philox.normals(17, ids, tick, philox.STREAM_OBS_NOISE, 18))
            u = torch.from_numpy(philox.reset_uniforms(17, ids, tick))
            b.step_with_inputs(act, z, u)
            assert torch.allclose(a.obs_buf, b.obs_buf, rtol=0, atol=2e-5), (task, t)      # sigma * (libm vs v_log / v_sin normals)
            assert torch.allclose(a.rew_buf, b.rew_buf, rtol=0, atol=2e-6) and torch.equal(a.reset_buf, b.reset_buf)
        sa, sb = a.get_state(), b.get_state()
        for k_ in sa:
            assert torch.allclose(sa[k_].float(), sb[k_].float(), rtol=0, atol=2e-6), k_
        a.close(); b.close()


def test_tracking_full_size_properties(Handle):
    """BASELINE config 2 size: Tracking, 65 536 envs, LV control - size-independent properties."""
    n = 65536
    env = Handle("tracking", "vel", n, seed=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    tot = 0
    for t in range(120):
        a = torch.randn(n, 4, generator=g, device="cuda").clamp(-1, 1)
        env.step(a)
        if t % 10 == 0:
            assert torch.equal(env.compact_reset_ids().long(), env.reset_buf.nonzero().squeeze(-1))
        tot += int(env.reset_buf.sum())
    st = env.get_state()
    assert env.obs_buf.shape == (n, 48) and torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all()
    assert torch.allclose(st["root_states"][:, 3:7].norm(dim=-1), torch.ones(n, device="cuda"), atol=1e-5)
    # obs[18:21] = lemniscate(progress) - position (tracking.py:194-214), no noise on these columns
    t0 = st["progress"].float() * 0.01 * 0.25
    ref = torch.stack((3 * torch.sin(t0) / (1 + torch.cos(t0) ** 2), 3 * torch.sin(t0) * torch.cos(t0) / (1 + torch.cos(t0) ** 2),
                       torch.ones_like(t0)), -1)
    done = env.reset_buf.bool()
    assert torch.allclose(env.obs_buf[~done, 18:21], (ref - st["root_states"][:, 0:3])[~done], atol=1e-4)
    assert tot > 0 and (st["progress"] < 3600).all()
    # dist_norm term == |ref0 - pos| for envs that did not reset this step; envs farther than 1 m are flagged done
    dn = env.reward_terms["dist_norm"]
    assert (dn[done] > 1.0).float().mean() > 0.9 or done.sum() == 0
    env.close()


@pytest.mark.parametrize("task,ctl,units,gemm_loss,gemm_input", [
    ("hovering", "rate", [256, 256], True, True), ("hovering", "rate", [256, 256], True, False),
    ("hovering", "rate", [256, 256], False, False), ("hovering", "rate", [128, 64, 32], True, True),
    ("hovering", "rate", [512], True, True), ("tracking", "vel", [256, 256], True, True)])
def test_hand_scheduled_update_matches_autograd(Handle, task, ctl, units, gemm_loss, gemm_input):
    """FusedMLPStep (hand-scheduled forward/backward writing into the flat gradient buffer) == the autograd path on the
    same minibatch: every gradient, the KL slot, the logged scalars and the mu/sigma write-back."""
    import os
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent

    class Args:
        envs = 4096; minibatches = 4; graph = 0
    Args.task, Args.ctl = task, ctl
    params = bench.build_params(Args, 1)
    params["config"]["bounds_loss_coef"] = 1e-4
    params["network"]["mlp"]["units"] = units
    params["config"]["fuse_gemm_loss"] = gemm_loss
    params["config"]["fuse_gemm_input"] = gemm_input
    agent = A2CAgent("t", params)
    assert agent._fused_step is not None, "the bench configuration must take the hand-scheduled path"
    if units == [256, 256]:     # Hovering's 18-wide and Tracking's 48-wide first layer: backward in the dX GEMM's epilogue
        assert agent._fused_step.fuse_gemm_input_wgrad and agent._fused_step.split_wgrad == {1}
        # ... and the PPO loss + the head layer's backward in the forward GEMM's epilogue (ag_split_gemm_loss_heads_bwd)
        assert agent._fused_step.fuse_gemm_loss == gemm_loss
        # ... with the first layer formed inside that launch too (Hovering's 18 inputs; Tracking's 48 keep ag_mlp_input_layer)
        assert agent._fused_step.fuse_gemm_input == (gemm_input and gemm_loss and task == "hovering")
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.epoch_num = 1
    agent.train_epoch()                      # moves the policy away from init so ratios/KL are non-trivial
    batch = agent.play_steps()
    agent.model.train()
    agent.curr_frames = batch.pop("played_frames")
    agent.prepare_dataset(batch)
    agent.model.running_mean_std.eval()
    agent.model.update_stats = False
    mb = agent.dataset[1]
    mu0, sig0 = mb["mu"].clone(), mb["sigma"].clone()
    agent._fused_step.begin_epoch()
    st = agent._fused_step.step(mb).clone()
    g_fused = agent.flat_grad.clone()
    mu_f, sig_f = mb["mu"].clone(), mb["sigma"].clone()
    mb["mu"].copy_(mu0); mb["sigma"].copy_(sig0)
    a, c, e, b, _, _ = agent._loss_and_backward(mb)
    g_auto = agent.flat_grad.clone()
    scale = g_auto[:-1].abs().max()
    assert (g_fused - g_auto)[:-1].abs().max() <= 2e-5 * scale + 1e-9, ((g_fused - g_auto).abs().max(), scale)
    assert torch.allclose(g_fused[-1], g_auto[-1], rtol=1e-4, atol=1e-8)
    for got, ref in zip(st[:4], (a, c, e, b)):
        assert torch.allclose(got, ref, rtol=2e-5, atol=1e-6), (got.item(), ref.item())
    assert torch.allclose(mu_f, mb["mu"], atol=1e-6) and torch.allclose(sig_f, mb["sigma"])
    # the two paths train identically for a full epoch (same LR schedule decisions)
    lr_before = agent.optimizer.lr.item()
    agent.epoch_num += 1
    out = agent.train_epoch()
    assert out["kl"] == out["kl"] and agent.optimizer.lr.item() > 0 and lr_before > 0
    agent.vec_env.env.hip.close() if hasattr(agent.vec_env.env.hip, "close") else None


def test_normalize_rows_kernel(Handle):
    import ctypes
    from airgym_amd import _native as N
    lib = N.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    for rows, D in [(1, 18), (1000, 18), (70001, 48), (513, 272)]:
        x = 4 * torch.randn(rows, D, device="cuda", generator=g)
        mean = torch.randn(D, device="cuda", generator=g, dtype=torch.float64)
        var = torch.rand(D, device="cuda", generator=g, dtype=torch.float64) + 0.01
        out = torch.empty_like(x)
        N.check(lib.ag_normalize_rows(x.data_ptr(), mean.data_ptr(), var.data_ptr(), out.data_ptr(), rows, D, 1e-5, 5.0,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ag_normalize_rows")
        ref = torch.clamp((x - mean.float()) / torch.sqrt(var.float() + 1e-5), -5.0, 5.0)
        assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
        assert (out.abs().max() <= 5.0) and (out.abs() == 5.0).any()
