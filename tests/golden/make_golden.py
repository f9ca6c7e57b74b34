#!/usr/bin/env python3
"""Generate golden fixtures by CALLING THE REFERENCE'S OWN FUNCTIONS.

Run once in the build container (needs /root/reference; never runs on the GPU box):

    python tests/golden/make_golden.py

The reference (emNavi/AirGym) cannot be imported as-is here: isaacgym,
rlPx4Controller, pytorch3d, rospy, cv2 and gym are absent.  This harness installs
`sys.modules` stubs for them, imports `airgym.envs.base.hovering`,
`airgym.envs.task.tracking` and `lib.core.*` from /root/reference, and invokes
their methods unbound on a hand-built `self` (CPU tensors).  Only *data* (inputs
and the reference's outputs) is written to tests/golden/*.npz - no reference
source is copied.

`pytorch3d.transforms` is supplied by `oracle/rotations.py` (its four functions
are themselves checked against scipy in tests/test_oracle_rotations.py).
Random draws the reference takes from torch's global generator are replaced by
recorded arrays (stored in the fixture) so the oracle can replay them.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)


def install_stubs():
    np.float = float  # airgym/utils/torch_utils.py:135 uses the removed alias

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    ig = stub("isaacgym")
    ig.gymtorch = stub("isaacgym.gymtorch", unwrap_tensor=lambda t: t, wrap_tensor=lambda t: t)
    ig.gymapi = stub("isaacgym.gymapi", LOCAL_SPACE=0)
    ig.gymutil = stub("isaacgym.gymutil")
    import airgym.utils.torch_utils as tu
    sys.modules["isaacgym.torch_utils"] = tu
    stub("rlPx4Controller")
    stub("rlPx4Controller.pyParallelControl", ParallelRateControl=_Dummy, ParallelVelControl=_Dummy,
         ParallelAttiControl=_Dummy, ParallelPosControl=_Dummy)
    from oracle import rotations
    p3d = stub("pytorch3d")
    p3d.transforms = stub(
        "pytorch3d.transforms",
        quaternion_to_matrix=rotations.quaternion_to_matrix,
        euler_angles_to_matrix=rotations.euler_angles_to_matrix,
        matrix_to_quaternion=rotations.matrix_to_quaternion,
        matrix_to_euler_angles=lambda m, convention="XYZ": rotations.matrix_to_euler_angles_xyz(m),
    )
    stub("rospy")
    stub("std_msgs")
    stub("std_msgs.msg", Float64MultiArray=_Dummy)
    stub("cv2")
    stub("gym", Wrapper=object, spaces=stub("gym.spaces"))
    stub("tensorboardX", SummaryWriter=_Dummy)


class CpuTo:
    """hovering.py:373 / tracking.py:225 hard-code `.to('cuda')`; route it to CPU."""

    def __enter__(self):
        self.orig = torch.Tensor.to

        def to(t, *a, **k):
            a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
            return self.orig(t, *a, **k)

        torch.Tensor.to = to

    def __exit__(self, *e):
        torch.Tensor.to = self.orig


def rand_quat_xyzw(g, n):
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    return q


def make_states(g, n, pos_scale, max_len):
    rs = torch.zeros(n, 13)
    rs[:, 0:3] = (torch.rand(n, 3, generator=g) * 2 - 1) * pos_scale
    rs[:, 3:7] = rand_quat_xyzw(g, n)
    # make most of them roughly upright so that not everything terminates
    up = torch.rand(n, generator=g) < 0.7
    small = torch.zeros(n, 4)
    small[:, :3] = 0.2 * torch.randn(n, 3, generator=g)
    small[:, 3] = 1.0
    small = small / small.norm(dim=-1, keepdim=True)
    rs[up, 3:7] = small[up]
    rs[:, 7:10] = torch.randn(n, 3, generator=g)
    rs[:, 10:13] = 0.5 * torch.randn(n, 3, generator=g)
    progress = torch.randint(0, max_len - 3, (n,), generator=g)
    progress[0:4] = torch.tensor([max_len - 3, max_len - 2, max_len - 1, max_len])
    return rs, progress.long()


def fake_task(cls, n, ctl_mode, rs, progress, target_state, max_len):
    s = object.__new__(cls)
    s.num_envs = n
    s.device = "cpu"
    s.ctl_mode = ctl_mode
    s.dt = 0.01
    s.max_episode_length = max_len
    s.root_states = rs.clone()
    s.root_positions = s.root_states[..., 0:3]
    s.root_quats = s.root_states[..., 3:7]
    s.root_linvels = s.root_states[..., 7:10]
    s.root_angvels = s.root_states[..., 10:13]
    s.progress_buf = progress.clone()
    s.reset_buf = torch.zeros(n, dtype=torch.long)
    s.target_states = torch.tensor(target_state, dtype=torch.float32).repeat(n, 1)
    return s


def gen_helpers(H, out):
    g = torch.Generator().manual_seed(11)
    q = rand_quat_xyzw(g, 512)
    a = (torch.rand(512, generator=g) * 2 - 1) * torch.pi
    b = (torch.rand(512, generator=g) * 2 - 1) * torch.pi
    t = torch.randn(512, 4, generator=g) * 4
    lo = torch.tensor([-6.0, -6, -6, 0])
    hi = torch.tensor([6.0, 6, 6, 1])
    q2 = rand_quat_xyzw(g, 512)
    out["helpers"] = dict(
        q=q, a=a, b=b, t=t, lo=lo, hi=hi, q2=q2,
        quat_axis0=H.quat_axis(q, 0), quat_axis1=H.quat_axis(q, 1), quat_axis2=H.quat_axis(q, 2),
        yaw_diff=H.compute_yaw_diff(a, b),
        clamp=H.tensor_clamp(t, lo, hi),
        qmul=H.quaternion_multiply(q, q2),
        rand_float_u=torch.linspace(0, 1, 17)[:16].reshape(16, 1),
    )


def gen_obs(H, cls, name, num_obs, max_len, out):
    g = torch.Generator().manual_seed(21)
    n = 512
    rs, progress = make_states(g, n, 3.0, max_len)
    target = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0.3, -0.2, 0.5, 0, 0, 0, 0, 0, 0] if name == "hovering" else \
        [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    s = fake_task(cls, n, "rate", rs, progress, target, max_len)
    s.obs_buf = torch.zeros(n, num_obs)
    noise = torch.randn(n, 18, generator=g)
    chunks = [noise[:, 0:9], noise[:, 9:12], noise[:, 12:15], noise[:, 15:18]]
    # add_noise is inherited from Hovering by Tracking: patch the globals of the function that runs
    glb = cls.add_noise.__globals__
    orig = glb["torch_normal_float"]
    it = iter(chunks)
    glb["torch_normal_float"] = lambda shape, device: next(it).clone()
    try:
        cls.compute_observations(s)
    finally:
        glb["torch_normal_float"] = orig
    d = dict(root_states=rs, progress=progress, target_state=torch.tensor(target, dtype=torch.float32),
             noise=noise, obs=s.obs_buf)
    if hasattr(s, "ref_positions"):
        d["ref_positions"] = s.ref_positions
    out[f"{name}_obs"] = d


def gen_reward(H, cls, name, max_len, pos_scale, out):
    for ctl_mode in ("rate", "vel", "atti", "pos", "prop"):
        g = torch.Generator().manual_seed(31 + len(ctl_mode) + ord(ctl_mode[0]))
        n = 512
        A = 5 if ctl_mode == "atti" else 4
        rs, progress = make_states(g, n, pos_scale, max_len)
        target = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]
        if name == "hovering":
            # boundary rows: |rel| = 4 -+ eps, rel_z = -+2 -+ eps, ups_z = 0 -+ eps
            ident = torch.tensor([0, 0, 0, 1.0])
            rs[4, 0:3] = torch.tensor([3.999, 0, 0]); rs[4, 3:7] = ident
            rs[5, 0:3] = torch.tensor([4.001, 0, 0]); rs[5, 3:7] = ident
            rs[6, 0:3] = torch.tensor([0, 0, 1.999]); rs[6, 3:7] = ident
            rs[7, 0:3] = torch.tensor([0, 0, 2.001]); rs[7, 3:7] = ident
            rs[8, 0:3] = torch.tensor([0, 0, -1.999]); rs[8, 3:7] = ident
            rs[9, 0:3] = torch.tensor([0, 0, -2.001]); rs[9, 3:7] = ident
            for k, ang in enumerate((1.5608, 1.5808)):  # roll just below / above 90 deg
                rs[10 + k, 0:3] = 0.1
                rs[10 + k, 3:7] = torch.tensor([np.sin(ang / 2), 0, 0, np.cos(ang / 2)], dtype=torch.float32)
        s = fake_task(cls, n, ctl_mode, rs, progress, target, max_len)
        s.actions = torch.rand(n, A, generator=g) * 2 - 1
        s.pre_actions = torch.rand(n, A, generator=g) * 2 - 1
        # the external controller returns float64 (hovering.py:238-250)
        s.cmd_thrusts = (torch.rand(n, 4, generator=g, dtype=torch.float64) * 1.4 - 0.2)
        if ctl_mode == "prop":
            s.cmd_thrusts = s.actions
        if name == "tracking":
            s.progress_buf = progress
            s.ref_positions = cls.compute_traj_lemniscate(s)
        with CpuTo():
            reward, reset, info = cls.compute_quadcopter_reward(s)
        d = dict(root_states=rs, progress=progress, actions=s.actions, pre_actions=s.pre_actions,
                 cmd_thrusts=s.cmd_thrusts.to(torch.float64), reward=reward.to(torch.float64), reset=reset)
        for k, v in info.items():
            if torch.is_tensor(v):
                d["info_" + k] = v.to(torch.float64)
        out[f"{name}_reward_{ctl_mode}"] = d


def gen_reset(H, cls, name, out):
    g = torch.Generator().manual_seed(41)
    n = 256
    u = torch.rand(n, 12, generator=g)
    u[0] = 0.0
    u[1] = 1.0 - 2 ** -24
    s = object.__new__(cls)
    s.device = "cpu"
    s.root_states = torch.zeros(n, 13)
    s.initial_root_states = torch.zeros(n, 13)
    s.initial_root_states[:, 6] = 1
    s.reset_buf = torch.zeros(n, dtype=torch.long)
    s.progress_buf = torch.full((n,), 7, dtype=torch.long)
    s.pre_actions = torch.ones(n, 4)
    s.gym = types.SimpleNamespace(set_actor_root_state_tensor=lambda *a: None)
    s.sim = None
    s.root_tensor = None
    for nm in ("thrust_cmds_damp", "thrust_rot_damp"):
        setattr(s, nm, torch.ones(n, 4))
    for nm in ("int_pos_error", "int_yaw_error"):
        setattr(s, nm, torch.ones(n, 10))
    s.pre_root_positions = torch.ones(n, 3)
    # reference draw order (hovering.py:316-329): pos xy (K,2), pos z (K,1), euler xy (K,2), euler z (K,1),
    # linvel (K,3), angvel (K,3)
    draws = iter([u[:, 0:2], u[:, 2:3], u[:, 3:5], u[:, 5:6], u[:, 6:9], u[:, 9:12]])
    mod = sys.modules[cls.__module__]
    orig = mod.torch_rand_float
    mod.torch_rand_float = lambda lo, hi, shape, device: (hi - lo) * next(draws).clone() + lo
    try:
        cls.reset_idx(s, torch.arange(n))
    finally:
        mod.torch_rand_float = orig
    out[f"{name}_reset"] = dict(uniforms=u, root_states=s.root_states, reset_buf=s.reset_buf,
                                progress=s.progress_buf, pre_actions=s.pre_actions)


def gen_wrench(H, out):
    """Run the reference's own pre_physics_step in 'prop' mode (cmd = actions) and capture
    the force/torque tensors it hands to PhysX (hovering.py:256-281)."""
    g = torch.Generator().manual_seed(51)
    n = 256
    s = object.__new__(H.Hovering)
    s.counter = 1
    s.num_envs = n
    s.device = "cpu"
    s.ctl_mode = "prop"
    s.reset_buf = (torch.rand(n, generator=g) < 0.25).long()
    s.reset_idx = lambda ids: None
    s.action_upper_limits = torch.tensor([1.0, 1, 1, 1])
    s.action_lower_limits = torch.tensor([0.0, 0, 0, 0])
    s.root_states = torch.zeros(n, 13)
    s.root_states[:, 3:7] = rand_quat_xyzw(g, n)
    rs_in = s.root_states.clone()
    s.forces = torch.zeros(n, 5, 3)
    s.torques = torch.zeros(n, 5, 3)
    captured = {}
    s.gym = types.SimpleNamespace(
        apply_rigid_body_force_tensors=lambda sim, f, t, space: captured.update(f=f.clone(), t=t.clone()))
    s.sim = None
    actions = torch.rand(n, 4, generator=g) * 1.4 - 0.2
    H.Hovering.pre_physics_step(s, actions.clone())
    out["wrench"] = dict(actions=actions, reset_buf=s.reset_buf, root_states_in=rs_in,
                         root_states_out=s.root_states, cmd=s.cmd_thrusts,
                         forces=captured["f"], torques=captured["t"], clamped_actions=s.actions)


def gen_action_map(H, out):
    """Action pre-processing of pre_physics_step for rate mode: last -> 0.5+0.5a, clamp (hovering.py:212-216).
    The controller object is stubbed to return zeros; only self.actions is recorded."""
    g = torch.Generator().manual_seed(61)
    n = 128
    res = {}
    for mode, lim in (("rate", ([-6, -6, -6, 0], [6, 6, 6, 1])), ("atti", ([-1, -1, -1, -1, 0.], [1, 1, 1, 1, 1])),
                      ("vel", ([-6, -6, -6, -6], [6, 6, 6, 6])), ("pos", ([-3, -3, -3, -6.0], [3, 3, 3, 6.0]))):
        A = len(lim[0])
        s = object.__new__(H.Hovering)
        s.counter = 1
        s.num_envs = n
        s.device = "cpu"
        s.ctl_mode = mode
        s.reset_buf = torch.zeros(n, dtype=torch.long)
        s.action_lower_limits = torch.tensor(lim[0], dtype=torch.float32)
        s.action_upper_limits = torch.tensor(lim[1], dtype=torch.float32)
        s.root_states = torch.zeros(n, 13)
        s.root_states[:, 3:7] = rand_quat_xyzw(g, n)
        s.forces = torch.zeros(n, 5, 3)
        s.torques = torch.zeros(n, 5, 3)
        s.gym = types.SimpleNamespace(apply_rigid_body_force_tensors=lambda *a: None)
        s.sim = None

        class Ctl:
            def set_status(self, *a): pass
            def set_q_world(self, *a): pass
            def update(self, *a): return np.zeros((n, 4))
        for nm in ("parallel_pos_control", "parallel_vel_control", "parallel_atti_control", "parallel_rate_control"):
            setattr(s, nm, Ctl())
        actions = torch.randn(n, A, generator=g) * 4
        rs_in = s.root_states.clone()
        H.Hovering.pre_physics_step(s, actions.clone())
        res[f"{mode}_in"] = actions
        res[f"{mode}_out"] = s.actions
        res[f"{mode}_quat_in"] = rs_in[:, 3:7]
        res[f"{mode}_quat_out"] = s.root_states[:, 3:7]
    out["action_map"] = res


def gen_lemniscate(T, out):
    s = object.__new__(T.Tracking)
    s.num_envs = 64
    s.device = "cpu"
    s.dt = 0.01
    s.progress_buf = torch.arange(0, 3600, 57)[:64].long()
    ref = T.Tracking.compute_traj_lemniscate(s)
    out["lemniscate"] = dict(progress=s.progress_buf, ref=ref)


def gen_ppo(out):
    from lib.core import common_losses, schedulers, torch_ext
    from lib.core.running_mean_std import RunningMeanStd
    from lib.network.mlp import MLP
    g = torch.Generator().manual_seed(71)
    n = 256
    old_nlp = torch.randn(n, generator=g)
    new_nlp = old_nlp + 0.3 * torch.randn(n, generator=g)
    adv = torch.randn(n, generator=g)
    a_loss = common_losses.actor_loss(old_nlp, new_nlp, adv, True, 0.2)
    vp = torch.randn(n, 1, generator=g)
    v = vp + 0.3 * torch.randn(n, 1, generator=g)
    ret = torch.randn(n, 1, generator=g)
    c_loss = common_losses.default_critic_loss(vp, v, 0.2, ret, False)
    c_loss_clip = common_losses.default_critic_loss(vp, v, 0.2, ret, True)
    mu0 = torch.randn(n, 4, generator=g)
    s0 = torch.rand(n, 4, generator=g) + 0.5
    mu1 = mu0 + 0.1 * torch.randn(n, 4, generator=g)
    s1 = s0 * (1 + 0.1 * torch.randn(n, 4, generator=g))
    kl = torch_ext.policy_kl(mu0, s0, mu1, s1, True)
    kl_nr = torch_ext.policy_kl(mu0, s0, mu1, s1, False)
    # bound loss (a2c_continuous.py:382-390) - method needs self.bounds_loss_coef only
    from types import SimpleNamespace
    sys.modules.setdefault("lib.utils.vecenv", types.ModuleType("lib.utils.vecenv"))
    mu_big = torch.randn(n, 4, generator=g) * 1.5
    soft_bound = 1.1
    # call the reference implementation without importing gym-dependent modules: read the function object
    import importlib.util
    b_loss = None
    try:
        import lib.agent.a2c_continuous as ac
        b_loss = ac.ContinuousA2CBase.bound_loss(SimpleNamespace(bounds_loss_coef=1e-4), mu_big)
    except Exception as e:  # gym etc. missing
        print("bound_loss via reference import failed:", repr(e))
    # RunningMeanStd
    rms = RunningMeanStd((6,))
    rms.train()
    xs = [torch.randn(64, 6, generator=g) * (i + 1) + i for i in range(3)]
    ys = [rms(x) for x in xs]
    rms.eval()
    y_eval = rms(xs[0])
    y_denorm = rms(torch.randn(64, 6, generator=g) * 3, denorm=True)
    # scheduler
    sch = schedulers.AdaptiveScheduler(0.008)
    kls = [0.0, 0.003, 0.004, 0.008, 0.016, 0.017, 0.5]
    lrs = []
    for start in (3e-4, 1e-6, 1e-2):
        for k in kls:
            lrs.append(sch.update(start, 0.0, 0, 0, k)[0])
    # MLP forward with fixed weights
    torch.manual_seed(5)
    mlp = MLP(18, [64, 128, 64], "elu")
    x = torch.randn(32, 18, generator=g)
    y = mlp(x)
    d = dict(old_nlp=old_nlp, new_nlp=new_nlp, adv=adv, a_loss=a_loss, vp=vp, v=v, ret=ret, c_loss=c_loss,
             c_loss_clip=c_loss_clip, mu0=mu0, s0=s0, mu1=mu1, s1=s1, kl=kl, kl_nr=kl_nr, mu_big=mu_big,
             rms_x0=xs[0], rms_x1=xs[1], rms_x2=xs[2], rms_y0=ys[0], rms_y1=ys[1], rms_y2=ys[2],
             rms_mean=rms.running_mean, rms_var=rms.running_var, rms_count=rms.count, rms_y_eval=y_eval,
             rms_denorm_in=torch.zeros(1), sched_kls=torch.tensor(kls, dtype=torch.float64),
             sched_lrs=torch.tensor(lrs, dtype=torch.float64), mlp_x=x, mlp_y=y.detach())
    if b_loss is not None:
        d["b_loss"] = b_loss
    for i, layer in enumerate(mlp.layers):
        d[f"mlp_w{i}"] = layer.weight.detach()
        d[f"mlp_b{i}"] = layer.bias.detach()
    out["ppo"] = d


def gen_gae(out):
    """discount_values (a2c_base.py:463-478) called unbound on a stub self."""
    try:
        import lib.agent.a2c_base as ab
    except Exception as e:
        print("a2c_base import failed:", repr(e))
        return
    g = torch.Generator().manual_seed(81)
    H, N = 24, 64
    s = types.SimpleNamespace(horizon_length=H, gamma=0.99, tau=0.95)
    fdones = (torch.rand(N, generator=g) < 0.1).float()
    last_values = torch.randn(N, 1, generator=g)
    mb_fdones = (torch.rand(H, N, generator=g) < 0.05).float()
    mb_values = torch.randn(H, N, 1, generator=g)
    mb_rewards = torch.randn(H, N, 1, generator=g) * 0.1
    advs = ab.A2CBase.discount_values(s, fdones, last_values, mb_fdones, mb_values, mb_rewards)
    out["gae"] = dict(fdones=fdones, last_values=last_values, mb_fdones=mb_fdones, mb_values=mb_values,
                      mb_rewards=mb_rewards, advs=advs)


def gen_planning(out):
    """Planning's own tensor code (airgym/envs/task/planning.py) + the depth post-processing of
    airgym/envs/base/customized.py:399-435, called unbound on a hand-built self."""
    import airgym.envs.task.planning as P
    import airgym.envs.base.customized as C
    g = torch.Generator().manual_seed(91)
    n = 256
    s = object.__new__(P.Planning)
    s.num_envs, s.device, s.ctl_mode, s.max_episode_length = n, "cpu", "rate", 1600
    rs = torch.zeros(n, 13)
    rs[:, 0] = (torch.rand(n, generator=g) * 2 - 1) * 9.0
    rs[:, 1] = (torch.rand(n, generator=g) * 2 - 1) * 4.4
    rs[:, 2] = 1.5 + (torch.rand(n, generator=g) * 2 - 1) * 0.4
    q = torch.zeros(n, 4); q[:, :3] = 0.25 * torch.randn(n, 3, generator=g); q[:, 3] = 1.0
    rs[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
    rs[:, 7:10] = torch.randn(n, 3, generator=g)
    rs[:, 10:13] = 0.5 * torch.randn(n, 3, generator=g)
    s.root_states = rs.clone()
    s.root_positions, s.root_quats = s.root_states[..., 0:3], s.root_states[..., 3:7]
    s.root_linvels, s.root_angvels = s.root_states[..., 7:10], s.root_states[..., 10:13]
    goal = torch.zeros(n, 3); goal[:, 0] = 8.5; goal[:, 1] = (torch.rand(n, generator=g) * 2 - 1) * 1.5; goal[:, 2] = 1.5
    goal[0:4] = s.root_positions[0:4] + torch.tensor([[0.29, 0, 0], [0.31, 0, 0], [0.1, 0.1, 0.1], [0.2, 0.2, 0.15]])
    s.goal_positions = goal
    s.obs_buf = torch.zeros(n, 16)
    s.actions_local = torch.rand(n, 4, generator=g) * 2 - 1
    s.actions = s.actions_local
    s.pre_actions = torch.rand(n, 4, generator=g) * 2 - 1
    s.pre_root_positions = s.root_positions + 0.02 * torch.randn(n, 3, generator=g)
    s.progress_buf = torch.randint(0, 1597, (n,), generator=g).long()
    s.progress_buf[4:8] = torch.tensor([1597, 1598, 1599, 1600])
    s.reset_buf = torch.zeros(n, dtype=torch.long)
    s.collisions = (torch.rand(n, generator=g) < 0.1).float()
    s.esdf_dist = torch.rand(n, generator=g) * 1.2
    s.esdf_dist[8:10] = torch.tensor([0.2999, 0.3001])
    P.Planning.compute_observations(s)
    reward, reset, info = P.Planning.compute_quadcopter_reward(s)
    d = dict(root_states=rs, goal=goal, actions=s.actions, pre_actions=s.pre_actions,
             pre_root_positions=s.pre_root_positions, progress=s.progress_buf, collisions=s.collisions,
             esdf_dist=s.esdf_dist, obs=s.obs_buf, reward=reward, reset=reset, related_dist=s.related_dist)
    for k, v in info.items():
        d["info_" + k] = v
    out["planning_obs_reward"] = d

    # ---- reset_idx with recorded draws (planning.py:63-136); 41 assets = goal ball + 40 thin obstacles
    k, na = 64, 41
    s2 = object.__new__(P.Planning)
    s2.device, s2.num_envs, s2.num_assets = "cpu", k, na
    s2.env_asset_root_states = torch.zeros(k, na, 13)
    s2.goal_states = s2.env_asset_root_states[:, 0, :]
    s2.root_states = torch.zeros(k, 13)
    s2.root_quats = s2.root_states[..., 3:7]
    s2.reset_buf = torch.zeros(k, dtype=torch.long)
    s2.progress_buf = torch.full((k,), 9, dtype=torch.long)
    s2.pre_actions = torch.ones(k, 4)
    s2.prev_related_dist = torch.ones(k)
    s2.pre_root_positions = torch.ones(k, 3)
    s2.pre_root_angvels = torch.ones(k, 3)
    s2.gym = types.SimpleNamespace(set_actor_root_state_tensor=lambda *a: None)
    s2.sim = s2.root_tensor = None
    ux, uy, uyaw = (torch.rand(k, na, 1, generator=g) for _ in range(3))
    ugoal = torch.rand(k, 1, generator=g)
    junk = lambda *shape: torch.rand(*shape, generator=g)
    draws = iter([ux, uy, junk(k, na, 2), uyaw, ugoal, junk(k, 1), junk(k, 1), junk(k, 2), junk(k, 1), junk(k, 3), junk(k, 3)])
    orig = P.torch_rand_float
    P.torch_rand_float = lambda lo, hi, shape, device: (hi - lo) * next(draws).clone() + lo
    try:
        P.Planning.reset_idx(s2, torch.arange(k))
    finally:
        P.torch_rand_float = orig
    out["planning_reset"] = dict(ux=ux, uy=uy, uyaw=uyaw, ugoal=ugoal, asset_states=s2.env_asset_root_states,
                                 root_states=s2.root_states, reset_buf=s2.reset_buf, progress=s2.progress_buf,
                                 pre_actions=s2.pre_actions, pre_root_positions=s2.pre_root_positions,
                                 prev_related_dist=s2.prev_related_dist)

    # ---- dump_images (customized.py:399-435): depth post-processing with recorded randoms
    ne, W, H = 1, 212, 120
    s3 = object.__new__(C.Customized)
    s3.num_envs = ne
    cam = -(torch.rand(ne, H, W, generator=g) * 7.0)          # IsaacGym depth tensor: negative z, [H, W]
    cam[0, :10] = -float("inf")                                 # no-hit pixels
    s3.camera_tensors = [cam[e] for e in range(ne)]
    s3.full_camera_array = torch.zeros(ne, 1, W, H)
    add = torch.randn(ne, 1, W, H, generator=g)
    mul = torch.randn(ne, 1, W, H, generator=g)
    ker = torch.randint(0, 256, (ne, 5, 5), generator=g).float()
    seq_n = iter([x for e in range(ne) for x in (add[e], mul[e])])
    seq_k = iter([ker[e] for e in range(ne)])
    o_normal, o_randint = torch.normal, torch.randint
    torch.normal = lambda mean, std, size=None, device=None, **kw: mean + std * next(seq_n).clone()
    torch.randint = lambda lo, hi, size, dtype=None, **kw: next(seq_k).clone()
    C.cv2.normalize = lambda *a, **k_: np.zeros((H, W), np.uint8)
    C.cv2.applyColorMap = lambda *a, **k_: None
    C.cv2.NORM_MINMAX = C.cv2.CV_8UC1 = C.cv2.COLORMAP_PLASMA = 0
    try:
        C.Customized.dump_images(s3)
    finally:
        torch.normal, torch.randint = o_normal, o_randint
    out["planning_images"] = dict(cam=cam, add=add[:, 0], mul=mul[:, 0], kernel=ker / 256.0, image=s3.full_camera_array)



def vae_fill_(module):
    """Deterministic parameter fill shared with tests/test_oracle_golden.py (the 7 MB of encoder weights cannot be a
    fixture): parameter i (sorted by name) <- cos(0.61803 * k + i) * 0.7 / sqrt(fan_in), biases 0.02 * sin(k + i)."""
    import math
    with torch.no_grad():
        for i, (name, p) in enumerate(sorted(module.named_parameters())):
            k = torch.arange(p.numel(), dtype=torch.float64)
            if p.dim() > 1:
                fan_in = p[0].numel()
                v = torch.cos(0.61803 * k + i) * (0.7 / math.sqrt(fan_in))
            else:
                v = 0.02 * torch.sin(k + i)
            p.copy_(v.reshape(p.shape).float())


def vae_test_image():
    i = torch.arange(212, dtype=torch.float32).view(1, 1, 212, 1)
    j = torch.arange(120, dtype=torch.float32).view(1, 1, 1, 120)
    b = torch.arange(3, dtype=torch.float32).view(3, 1, 1, 1)
    return 0.5 + 0.5 * torch.sin(0.05 * i + 0.11 * j + b)


def gen_vae(out):
    """lib/network/VAE.py ImgEncoder + VAE.encode through the VAEImageEncoder.encode recipe (bilinear resize to 120x212,
    means returned), with deterministic weights."""
    import contextlib
    import io
    from lib.network.VAE import VAE
    with contextlib.redirect_stdout(io.StringIO()):
        vae = VAE(input_dim=1, latent_dim=64)
    vae.eval()
    vae_fill_(vae.encoder)
    img = vae_test_image()
    with torch.no_grad():
        resized = torch.nn.functional.interpolate(img, [120, 212], mode="bilinear")
        z = vae.encoder(resized)
        _, means, std = vae.encode(resized)
    np.savez_compressed(os.path.join(out, "vae_encoder.npz"), z=z.numpy(), means=means.numpy(), std=std.numpy(),
                        resized_probe=resized[:, 0, ::17, ::23].numpy(),
                        param_names=np.array(sorted(n for n, _ in vae.encoder.named_parameters())))
    print("vae_encoder.npz", z.shape, float(z.abs().mean()))


def planning_probe_inputs(n=6):
    """Deterministic policy inputs (formula shared with tests/test_reference_checkpoint.py)."""
    i = torch.arange(212, dtype=torch.float32).view(1, 1, 212, 1)
    j = torch.arange(120, dtype=torch.float32).view(1, 1, 1, 120)
    b = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1)
    image = 0.5 + 0.5 * torch.sin(0.031 * i + 0.057 * j + 0.7 * b)
    observation = torch.sin(torch.arange(n * 16, dtype=torch.float32).view(n, 16) * 0.37) * 1.5
    return image, observation


def gen_planning_checkpoint(out):
    """trained/planning_cnn_rate.pth (the reference's shipped Planning policy) through the reference's OWN sub-modules
    (lib.network.cnn / mlp, lib.core.running_mean_std), wired as a2c_continuous_logstd_model.py:140-170 wires them (the model
    class itself needs torchvision): golden mu / value for deterministic inputs + a manifest of the checkpoint's tensors."""
    import contextlib
    import io
    import json
    from lib.core.running_mean_std import RunningMeanStd, RunningMeanStdObs
    from lib.network.cnn import CNNFeatureExtractor
    from lib.network.mlp import MLP
    ck = torch.load(os.path.join(REF, "trained", "planning_cnn_rate.pth"), map_location="cpu", weights_only=False)
    sd = ck["model"]
    manifest = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                    "sum": float(v.double().sum()), "abs_sum": float(v.double().abs().sum())} for k, v in sd.items()}

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = CNNFeatureExtractor(feature_dim=30)
        mlp = MLP(46, [64, 128, 64], "elu")
    cnn.load_state_dict(sub("actor_cnn."))
    mlp.load_state_dict(sub("actor_mlp."))
    rms = RunningMeanStdObs({"image": (1, 212, 120), "observation": (46,)})
    rms.load_state_dict(sub("running_mean_std."))
    vms = RunningMeanStd((1,))
    vms.load_state_dict(sub("value_mean_std."))
    mu_l, v_l = torch.nn.Linear(64, 4), torch.nn.Linear(64, 1)
    mu_l.load_state_dict(sub("mu."))
    v_l.load_state_dict(sub("value_head."))
    for m in (cnn, mlp, rms, vms):
        m.eval()
    image, observation = planning_probe_inputs()
    with torch.no_grad():
        feat = cnn(rms.running_mean_std["image"](image))
        trunk = mlp(rms.running_mean_std["observation"](torch.cat((observation, feat), dim=-1)))
        mu, value = mu_l(trunk), v_l(trunk)
        value_denorm = vms(value, denorm=True)
    np.savez_compressed(os.path.join(out, "planning_checkpoint.npz"), mu=mu.numpy(), value=value.numpy(),
                        value_denorm=value_denorm.numpy(), features=feat.numpy(), logstd=sd["logstd"].numpy(),
                        manifest=np.array(json.dumps(manifest)), epoch=int(ck["epoch"]), frame=int(ck["frame"]))
    print("planning_checkpoint.npz", mu.shape, float(mu.abs().mean()), float(value_denorm.mean()))


def main():
    install_stubs()
    import airgym.envs.base.hovering as H
    import airgym.envs.task.tracking as T
    out = {}
    gen_helpers(H, out)
    gen_obs(H, H.Hovering, "hovering", 18, 2400, out)
    gen_obs(T, T.Tracking, "tracking", 48, 3600, out)
    gen_reward(H, H.Hovering, "hovering", 2400, 3.0, out)
    gen_reward(T, T.Tracking, "tracking", 3600, 0.8, out)
    gen_reset(H, H.Hovering, "hovering", out)
    gen_reset(T, T.Tracking, "tracking", out)
    gen_wrench(H, out)
    gen_action_map(H, out)
    gen_lemniscate(T, out)
    gen_ppo(out)
    gen_gae(out)
    gen_planning(out)
    gen_vae(out)
    gen_planning_checkpoint(out)
    for name, d in out.items():
        arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KB, {len(arrs)} arrays)")


if __name__ == "__main__":
    main()
