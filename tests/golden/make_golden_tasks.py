#!/usr/bin/env python3
"""Golden fixtures for the Balloon and Avoid tasks, made by CALLING THE REFERENCE'S OWN METHODS.

Same harness as make_golden.py (sys.modules stubs for isaacgym / rlPx4Controller / pytorch3d / cv2; methods invoked
unbound on a hand-built `self`; random draws replaced by recorded arrays).  Run once in the build container:

    python tests/golden/make_golden_tasks.py

Writes tests/golden/{balloon,avoid}_{obs_reward,reset}.npz - data only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def views(s):
    s.root_positions, s.root_quats = s.root_states[..., 0:3], s.root_states[..., 3:7]
    s.root_linvels, s.root_angvels = s.root_states[..., 7:10], s.root_states[..., 10:13]


def upright_states(g, n, centre, spread):
    rs = torch.zeros(n, 13)
    rs[:, 0:3] = torch.tensor(centre) + (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor(spread)
    q = torch.zeros(n, 4); q[:, :3] = 0.3 * torch.randn(n, 3, generator=g); q[:, 3] = 1.0
    rs[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
    flip = torch.rand(n, generator=g) < 0.08                       # a few upside-down attitudes (ups < 0)
    rs[flip, 3:7] = torch.tensor([1.0, 0.0, 0.0, 0.05]) / np.sqrt(1 + 0.05 ** 2)
    rs[:, 7:10] = torch.randn(n, 3, generator=g) * 0.8
    rs[:, 10:13] = 0.5 * torch.randn(n, 3, generator=g)
    return rs


def gen_balloon(out):
    import airgym.envs.task.balloon as B
    g = torch.Generator().manual_seed(101)
    n = 256
    s = object.__new__(B.Balloon)
    s.num_envs, s.device, s.ctl_mode, s.max_episode_length = n, "cpu", "rate", 800
    s.root_states = upright_states(g, n, (0.8, 0.0, 1.0), (1.5, 1.5, 0.7))
    s.root_states[:, 7] = s.root_states[:, 7].abs() * (torch.rand(n, generator=g) < 0.85).float() * 2 - 0.2   # mostly forward
    views(s)
    s.balloon_states = torch.zeros(n, 13); s.balloon_states[:, 6] = 1.0
    s.balloon_states[:, 0] = 2.5 + 0.5 * (torch.rand(n, generator=g) * 2 - 1)
    s.balloon_states[:, 1] = 2.0 * (torch.rand(n, generator=g) * 2 - 1)
    s.balloon_states[:, 2] = 1.0 + 0.3 * (torch.rand(n, generator=g) * 2 - 1)
    # threshold rows: hit radius 0.1 -+ eps; x-overshoot -0.2 -+ eps; range 4 -+ eps; z 0.5 / 1.5 -+ eps; v_x = 0 -+ eps
    ident = torch.tensor([0.0, 0, 0, 1])

    def put(row, pos, bal, vx=0.5):
        s.root_states[row, 0:3] = torch.tensor(pos); s.root_states[row, 3:7] = ident
        s.root_states[row, 7:10] = torch.tensor([vx, 0.0, 0.0]); s.balloon_states[row, 0:3] = torch.tensor(bal)
    put(8, (1.0, 0.0, 1.0), (1.099, 0.0, 1.0)); put(9, (1.0, 0.0, 1.0), (1.101, 0.0, 1.0))
    put(10, (1.199, 0.5, 1.0), (1.0, 0.0, 1.0)); put(11, (1.201, 0.5, 1.0), (1.0, 0.0, 1.0))
    put(12, (0.0, 0.0, 1.0), (3.999, 0.0, 1.0)); put(13, (0.0, 0.0, 1.0), (4.001, 0.0, 1.0))
    put(14, (0.0, 0.0, 0.501), (2.0, 0.0, 1.0)); put(15, (0.0, 0.0, 0.499), (2.0, 0.0, 1.0))
    put(16, (0.0, 0.0, 1.499), (2.0, 0.0, 1.0)); put(17, (0.0, 0.0, 1.501), (2.0, 0.0, 1.0))
    put(18, (0.0, 0.0, 1.0), (2.0, 0.0, 1.0), vx=1e-4); put(19, (0.0, 0.0, 1.0), (2.0, 0.0, 1.0), vx=-1e-4)
    s.balloon_positions, s.balloon_quats = s.balloon_states[..., 0:3], s.balloon_states[..., 3:7]
    s.obs_buf = torch.zeros(n, 18)
    s.actions = torch.rand(n, 4, generator=g) * 2 - 1
    s.actions[20, 3], s.actions[21, 3], s.actions[22, 3], s.actions[23, 3] = 1.001, 0.999, -1.001, -0.999
    s.pre_actions = torch.rand(n, 4, generator=g) * 2 - 1
    s.pre_root_positions = s.root_positions + 0.02 * torch.randn(n, 3, generator=g)
    s.progress_buf = torch.randint(0, 797, (n,), generator=g).long()
    s.progress_buf[4:8] = torch.tensor([797, 798, 799, 800])
    s.reset_buf = torch.zeros(n, dtype=torch.long)
    noise = torch.randn(n, 18, generator=g)
    chunks = iter([noise[:, 0:9], noise[:, 9:12], noise[:, 12:15], noise[:, 15:18]])
    import airgym.envs.base.customized as C
    glb = C.Customized.add_noise.__globals__
    orig = glb["torch_normal_float"]
    glb["torch_normal_float"] = lambda shape, device: next(chunks).clone()
    try:
        B.Balloon.compute_observations(s)
    finally:
        glb["torch_normal_float"] = orig
    reward, reset, info = B.Balloon.compute_quadcopter_reward(s)
    d = dict(root_states=s.root_states.clone(), balloon=s.balloon_states[:, 0:3].clone(), actions=s.actions, pre_actions=s.pre_actions,
             pre_root_positions=s.pre_root_positions, progress=s.progress_buf, noise=noise, obs=s.obs_buf.clone(),
             reward=reward, reset=reset)
    for k, v in info.items():
        d["info_" + k] = v
    out["balloon_obs_reward"] = d

    # ---- reset_idx with recorded draws (balloon.py:57-99)
    k = 128
    s2 = object.__new__(B.Balloon)
    s2.device, s2.num_envs = "cpu", k
    s2.balloon_states = torch.zeros(k, 13); s2.balloon_states[:, 6] = 1
    s2.root_states = torch.zeros(k, 13)
    views(s2)
    s2.reset_buf = torch.zeros(k, dtype=torch.long)
    s2.progress_buf = torch.full((k,), 9, dtype=torch.long)
    s2.pre_actions = torch.ones(k, 4)
    s2.pre_root_positions = torch.ones(k, 3); s2.pre_root_angvels = torch.ones(k, 3)
    s2.initial_root_pos = torch.zeros(k, 3)
    s2.gym = types.SimpleNamespace(set_actor_root_state_tensor=lambda *a: None)
    s2.sim = s2.root_tensor = None
    u = torch.rand(k, 15, generator=g)
    u[0] = 0.0; u[1] = 1.0 - 2 ** -24
    # draw order: balloon x, y, z | root xy, z | euler x, y, z | linvel(3) | angvel(3)
    draws = iter([u[:, 0:1], u[:, 1:2], u[:, 2:3], u[:, 3:5], u[:, 5:6], u[:, 6:7], u[:, 7:8], u[:, 8:9], u[:, 9:12], u[:, 12:15]])
    orig = B.torch_rand_float
    B.torch_rand_float = lambda lo, hi, shape, device: (hi - lo) * next(draws).clone() + lo
    try:
        B.Balloon.reset_idx(s2, torch.arange(k))
    finally:
        B.torch_rand_float = orig
    out["balloon_reset"] = dict(uniforms=u, root_states=s2.root_states, balloon=s2.balloon_states[:, 0:3], reset_buf=s2.reset_buf,
                                progress=s2.progress_buf, pre_actions=s2.pre_actions, pre_root_positions=s2.pre_root_positions)


def gen_avoid(out):
    import airgym.envs.task.avoid as A
    g = torch.Generator().manual_seed(202)
    n = 256
    s = object.__new__(A.Avoid)
    s.num_envs, s.device, s.ctl_mode, s.max_episode_length = n, "cpu", "rate", 600
    s.root_states = upright_states(g, n, (0.0, 0.0, 1.0), (1.3, 1.3, 0.8))
    views(s)
    target = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    s.target_states = torch.tensor(target, dtype=torch.float32).repeat(n, 1)
    ident = torch.tensor([0.0, 0, 0, 1])

    def put(row, pos):
        s.root_states[row, 0:3] = torch.tensor(pos); s.root_states[row, 3:7] = ident
    put(8, (0.0, 0.0, 0.301)); put(9, (0.0, 0.0, 0.299)); put(10, (0.0, 0.0, 1.699)); put(11, (0.0, 0.0, 1.701))
    put(12, (1.999, 0.0, 1.0)); put(13, (2.001, 0.0, 1.0))
    for k_, ang in enumerate((1.5608, 1.5808)):                    # roll just below / above 90 deg
        put(14 + k_, (0.1, 0.1, 1.0))
        s.root_states[14 + k_, 3:7] = torch.tensor([np.sin(ang / 2), 0, 0, np.cos(ang / 2)], dtype=torch.float32)
    s.obs_buf = torch.zeros(n, 16)
    s.actions_local = torch.rand(n, 4, generator=g) * 2 - 1
    s.actions = s.actions_local
    s.pre_actions = torch.rand(n, 4, generator=g) * 2 - 1
    s.progress_buf = torch.randint(0, 597, (n,), generator=g).long()
    s.progress_buf[4:8] = torch.tensor([597, 598, 599, 600])
    s.reset_buf = torch.zeros(n, dtype=torch.long)
    s.collisions = (torch.rand(n, generator=g) < 0.1).float()
    A.Avoid.compute_observations(s)
    reward, reset, info = A.Avoid.compute_quadcopter_reward(s)
    d = dict(root_states=s.root_states.clone(), actions=s.actions, pre_actions=s.pre_actions, progress=s.progress_buf,
             collisions=s.collisions, obs=s.obs_buf.clone(), reward=reward, reset=reset)
    for k, v in info.items():
        d["info_" + k] = v
    out["avoid_obs_reward"] = d

    # ---- reset_idx with recorded draws (avoid.py:91-163 incl. calculate_object_velocity :58-90)
    k = 128
    s2 = object.__new__(A.Avoid)
    s2.device, s2.num_envs = "cpu", k
    s2.object_states = torch.zeros(k, 13); s2.object_states[:, 6] = 1
    s2.root_states = torch.zeros(k, 13)
    s2.initial_root_states = torch.zeros(k, 13); s2.initial_root_states[:, 6] = 1
    views(s2)
    s2.reset_buf = torch.zeros(k, dtype=torch.long)
    s2.progress_buf = torch.full((k,), 9, dtype=torch.long)
    s2.pre_actions = torch.ones(k, 4)
    s2.pre_root_positions = torch.ones(k, 3); s2.pre_root_angvels = torch.ones(k, 3)
    s2.gym = types.SimpleNamespace(set_actor_root_state_tensor=lambda *a: None)
    s2.sim = s2.root_tensor = None
    u = torch.rand(k, 11, generator=g)            # mask | theta | aim xyz | root xy | root z | euler xy | euler z
    u[0, 0], u[1, 0], u[2, 0], u[3, 0] = 0.7999, 0.8001, 0.0, 0.95
    thrown = u[:, 0] < 0.8
    kt = int(thrown.sum())
    junk = lambda *shape: torch.rand(*shape, generator=g)
    # reference draw order: mask (K,1); [thrown rows only] theta (Kt,1), z-jitter (Kt,1, scaled by 0), aim (Kt,3);
    # root xy (K,2), root z (K,1), euler xy (K,2), euler z (K,1), linvel (K,3, x0), angvel (K,3, x0)
    draws = iter([u[:, 0:1], u[thrown, 1:2], junk(kt, 1), u[thrown, 2:5], u[:, 5:7], u[:, 7:8], u[:, 8:10], u[:, 10:11],
                  junk(k, 3), junk(k, 3)])
    orig = A.torch_rand_float
    A.torch_rand_float = lambda lo, hi, shape, device: (hi - lo) * next(draws).clone() + lo
    try:
        A.Avoid.reset_idx(s2, torch.arange(k))
    finally:
        A.torch_rand_float = orig
    out["avoid_reset"] = dict(uniforms=u, root_states=s2.root_states, object_pos=s2.object_states[:, 0:3],
                              object_vel=s2.object_states[:, 7:10], reset_buf=s2.reset_buf, progress=s2.progress_buf,
                              pre_actions=s2.pre_actions, pre_root_positions=s2.pre_root_positions)


def main():
    MG.install_stubs()
    out = {}
    gen_balloon(out)
    gen_avoid(out)
    for name, d in out.items():
        arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KB, {len(arrs)} arrays)")


if __name__ == "__main__":
    main()
