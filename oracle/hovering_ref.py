"""Op-for-op PyTorch-CPU restatement of the reference Hovering task (oracle;
test infrastructure; also the `cpu_baseline` "port" timed by bench.py).

Follows `/root/reference/airgym/envs/base/hovering.py`:
  pre_physics_step :203-281   (action map, clamp, quat canonicalisation, controller, wrench)
  step             :286-308   (order: physics, progress++, obs, reward, reset, time_outs)
  reset_idx        :310-335   (distributions)
  compute_observations / add_noise :337-358
  compute_quadcopter_reward :371-459
  quat_rotate / quat_axis :464-481,  compute_yaw_diff :33-38
and `airgym/envs/base/base_task.py:72-76,107-111` (buffers, reset() = reset_idx(all) + step(zeros)).

The two external pieces are the build's spec: `rigid_body.rk4_step` (PhysX in
the reference) and `px4_cascade.controller_update` (rlPx4Controller).

Documented deviations from the reference (see DESIGN.md "Quirks"):
  * RNG: counter-based Philox (oracle/philox.py) instead of torch's global
    generator; uniforms/normals can also be supplied explicitly (parity mode).
  * Q1 double reset: the reference re-randomises a done env at the end of step t
    and again at the start of step t+1.  Only the second draw is observable, so
    the env is randomised once (end of step t); the *observable* side effect -
    rotor thrust zeroed on step t+1 while the reaction torque is kept, Q2 - is
    reproduced through `reset_buf`.
  * Q4: the caller's action tensor is not mutated.
  * Q5: cmd_thrusts are float32 (float64 on the reference's CPU controller).
  * Q6: the multiplied-by-zero RNG draw at :256 is not consumed.
  * Q18: controller state is cleared on reset.
"""
import math

import numpy as np
import torch

from . import philox
from . import rotations as T
from .px4_cascade import CascadeState, controller_update
from .rigid_body import body_wrench_from_cmd, rk4_step, semi_implicit_euler_step

ACTION_LIMITS = {
    # hovering.py:93-121  (lower, upper)
    "pos": ([-3, -3, -3, -6.0], [3, 3, 3, 6.0]),
    "vel": ([-6, -6, -6, -6], [6, 6, 6, 6]),
    "atti": ([-1, -1, -1, -1, 0.0], [1, 1, 1, 1, 1]),
    "rate": ([-6, -6, -6, 0], [6, 6, 6, 1]),
    "prop": ([0, 0, 0, 0], [1, 1, 1, 1]),
}

NOISE_SIGMA = (1e-3, 5e-3, 2e-2, 4e-1)  # matrix, pos, linvel, angvel; hovering.py:350-353


def compute_yaw_diff(a, b):
    """hovering.py:33-38"""
    diff = b - a
    diff = torch.where(diff < -torch.pi, diff + 2 * torch.pi, diff)
    diff = torch.where(diff > torch.pi, diff - 2 * torch.pi, diff)
    return diff


def quat_rotate(q, v):
    """hovering.py:464-474 (xyzw)."""
    shape = q.shape
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(shape[0], 1, 3), v.view(shape[0], 3, 1)).squeeze(-1) * 2.0
    return a + b + c


def quat_axis(q, axis=0):
    """hovering.py:476-481"""
    basis_vec = torch.zeros(q.shape[0], 3)
    basis_vec[:, axis] = 1
    return quat_rotate(q, basis_vec)


def tensor_clamp(t, min_t, max_t):
    """airgym/utils/torch_utils.py:199-201"""
    return torch.max(torch.min(t, max_t), min_t)


class HoveringRef:
    task = "hovering"
    num_obs = 18
    episode_length_s = 24          # hovering_config.py:17
    reset_pos_scale = (1.0, 1.0, 1.0)
    reset_pos_offset = (0.0, 0.0, 0.0)
    reset_euler_scale = (0.01, 0.01, 0.05)   # hovering.py:320-321
    reset_linvel_scale = 0.5                 # :328
    reset_angvel_scale = 0.2                 # :329
    action_limits = ACTION_LIMITS

    def __init__(self, num_envs, ctl_mode="rate", seed=0, env_id_offset=0, dt=0.01,
                 target_state=None, integrator="rk4", fix_time_outs=False, stagger_episode_phase=False):
        assert ctl_mode in ACTION_LIMITS, f"unknown ctl_mode {ctl_mode!r}"
        # opt-in of the BUILD, not of the reference (AG_FLAG_FIX_TIME_OUTS): time_out_buf = "reached the time limit this step"
        # instead of the reference's never-true expression (hovering.py:304 after the reset of :300-302 zeroed the progress)
        self.fix_time_outs = bool(fix_time_outs)
        # opt-in of the BUILD (AG_FLAG_STAGGER_PHASE): a full reset starts env i at progress ~ U{0 .. max_len - 2} (counter
        # RNG stream 2, global env id) instead of 0 (hovering.py:333 progress_buf[env_ids] = 0)
        self.stagger_episode_phase = bool(stagger_episode_phase)
        self.num_envs = num_envs
        self.ctl_mode = ctl_mode
        self.num_actions = 5 if ctl_mode == "atti" else 4          # hovering.py:47
        self.dt = dt
        self.max_episode_length = int(self.episode_length_s / dt)  # hovering.py:48
        self.seed = seed
        self.env_ids_global = np.arange(env_id_offset, env_id_offset + num_envs, dtype=np.uint32)
        self.tick = 0
        self.integrator = rk4_step if integrator == "rk4" else semi_implicit_euler_step

        lo, hi = self.action_limits[ctl_mode]
        self.action_lower_limits = torch.tensor(lo, dtype=torch.float32)
        self.action_upper_limits = torch.tensor(hi, dtype=torch.float32)

        # base_task.py:72-76
        self.obs_buf = torch.zeros(num_envs, self.num_obs, dtype=torch.float32)
        self.rew_buf = torch.zeros(num_envs, dtype=torch.float32)
        self.reset_buf = torch.ones(num_envs, dtype=torch.long)
        self.time_out_buf = torch.zeros(num_envs, dtype=torch.bool)
        self.progress_buf = torch.zeros(num_envs, dtype=torch.long)   # hovering.py:164-165
        self.extras = {}

        self.root_states = torch.zeros(num_envs, 13, dtype=torch.float32)
        self.root_states[:, 6] = 1.0
        if target_state is None:
            target_state = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]  # hovering_config.py:12
        self.target_states = torch.tensor(target_state, dtype=torch.float32).repeat(num_envs, 1)
        self.actions = torch.zeros(num_envs, self.num_actions)
        self.pre_actions = torch.zeros(num_envs, self.num_actions)
        self.cmd_thrusts = torch.zeros(num_envs, 4)
        self.ctl_state = CascadeState(num_envs)
        self.item_reward_info = {}
        # like ag_create: the state is valid (randomised, flagged reset) from the start
        self._reset_all()
        self.tick += 1

    # views, hovering.py:73-77
    @property
    def root_positions(self):
        return self.root_states[:, 0:3]

    @property
    def root_quats(self):
        return self.root_states[:, 3:7]

    @property
    def root_linvels(self):
        return self.root_states[:, 7:10]

    @property
    def root_angvels(self):
        return self.root_states[:, 10:13]

    # ------------------------------------------------------------------ reset
    def reset_state_from_uniforms(self, u):
        """u [K,12] in [0,1): pos(3) euler(3) linvel(3) angvel(3) -> [K,13] states.
        hovering.py:316-329 with torch_rand_float = (hi-lo)*u + lo."""
        def rf(lo, hi, x):
            return (hi - lo) * x + lo
        k = u.shape[0]
        st = torch.zeros(k, 13, dtype=torch.float32)
        ps, po = self.reset_pos_scale, self.reset_pos_offset
        for i in range(3):
            st[:, i] = ps[i] * rf(-1.0, 1.0, u[:, i]) + po[i]
        es = self.reset_euler_scale
        root_angle = torch.stack([es[i] * rf(-torch.pi, torch.pi, u[:, 3 + i]) for i in range(3)], dim=-1)
        matrix = T.euler_angles_to_matrix(root_angle, "XYZ")
        root_quats = T.matrix_to_quaternion(matrix)  # w,x,y,z
        st[:, 3:7] = root_quats[:, [1, 2, 3, 0]]
        st[:, 7:10] = self.reset_linvel_scale * rf(-1.0, 1.0, u[:, 6:9])
        st[:, 10:13] = self.reset_angvel_scale * rf(-1.0, 1.0, u[:, 9:12])
        return st

    def reset_idx(self, env_ids, uniforms=None):
        if uniforms is None:
            uniforms = torch.from_numpy(
                philox.reset_uniforms(self.seed, self.env_ids_global[env_ids.numpy()], self.tick))
        self.root_states[env_ids] = self.reset_state_from_uniforms(uniforms)
        self.reset_buf[env_ids] = 1
        self.progress_buf[env_ids] = 0
        self.pre_actions[env_ids] = 0
        self.ctl_state.reset(env_ids, self.root_states)
        self._reset_extra(env_ids)

    def _reset_extra(self, env_ids):
        pass

    def _reset_all(self):
        """reset_idx(all envs) (hovering.py:310-335); with the build's stagger opt-in the progress counters then start at
        philox(env, tick, stream 2, block 0).x mod (max_len - 1) instead of 0."""
        self.reset_idx(torch.arange(self.num_envs))
        if self.stagger_episode_phase:
            raw = philox.raw_blocks(self.seed, self.env_ids_global, self.tick, philox.STREAM_PHASE, 1)[:, 0]
            span = np.uint32(max(self.max_episode_length - 1, 1))
            self.progress_buf[:] = torch.from_numpy((raw % span).astype(np.int64))

    def reset(self):
        """base_task.py:107-111"""
        self._reset_all()
        self.tick += 1
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions))
        return obs, priv

    # ------------------------------------------------------------------- step
    def pre_physics_step(self, _actions):
        was_reset = self.reset_buf.clone()
        self.actions = _actions.clone().to(torch.float32)
        if self.ctl_mode == "rate" or self.ctl_mode == "atti":
            self.actions[..., -1] = 0.5 + 0.5 * self.actions[..., -1]
        self.actions = tensor_clamp(self.actions, self.action_lower_limits, self.action_upper_limits)
        # quat. if w is negative, then set it to positive. x,y,z,w   (hovering.py:224-226)
        self.root_states[..., 3:7] = torch.where(self.root_states[..., 6:7] < 0,
                                                 -self.root_states[..., 3:7],
                                                 self.root_states[..., 3:7])
        self.cmd_thrusts = controller_update(self.ctl_mode, self.ctl_state, self.actions, self.root_states)
        thrust_mask = (was_reset == 0).to(torch.float32)     # hovering.py:268
        self.fz, self.tau_b = body_wrench_from_cmd(self.cmd_thrusts, thrust_mask)

    def step(self, actions, noise=None, reset_uniforms=None):
        """noise [N,18+] standard normals and reset_uniforms [N,12] may be supplied
        (parity mode); otherwise they come from Philox keyed by (seed, env, tick)."""
        self.pre_physics_step(actions)
        self.root_states = self.integrator(self.root_states, self.fz, self.tau_b, self.dt)
        self.progress_buf += 1
        if noise is None:
            noise = torch.from_numpy(
                philox.normals(self.seed, self.env_ids_global, self.tick, philox.STREAM_OBS_NOISE, 18))
        self.compute_observations(noise)
        self.compute_reward()
        reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        self.last_reset_env_ids = reset_env_ids
        progress_end = self.progress_buf.clone()
        if len(reset_env_ids) > 0:
            u = None if reset_uniforms is None else reset_uniforms[reset_env_ids]
            self.reset_idx(reset_env_ids, u)
        self.time_out_buf = self.progress_buf > self.max_episode_length      # hovering.py:304: never true (quirk Q3)
        if self.fix_time_outs:
            self.time_out_buf = progress_end >= self.max_episode_length - 1
        self.extras["time_outs"] = self.time_out_buf
        self.extras["item_reward_info"] = self.item_reward_info
        self.tick += 1
        return self.obs_buf, None, self.rew_buf, self.reset_buf, self.extras

    # -------------------------------------------------------------------- obs
    def compute_observations(self, noise):
        self.root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]]).reshape(self.num_envs, 9)
        self.obs_buf[..., 0:9] = self.root_matrix
        self.obs_buf[..., 9:12] = self.root_positions
        self.obs_buf[..., 12:15] = self.root_linvels
        self.obs_buf[..., 15:18] = self.root_angvels
        self.add_noise(noise)
        self.obs_buf[..., 0:18] -= self.target_states
        return self.obs_buf

    def add_noise(self, noise):
        self.obs_buf[..., 0:9] += NOISE_SIGMA[0] * noise[:, 0:9]
        self.obs_buf[..., 9:12] += NOISE_SIGMA[1] * noise[:, 9:12]
        self.obs_buf[..., 12:15] += NOISE_SIGMA[2] * noise[:, 12:15]
        self.obs_buf[..., 15:18] += NOISE_SIGMA[3] * noise[:, 15:18]

    # ----------------------------------------------------------------- reward
    def compute_reward(self):
        self.rew_buf[:], self.reset_buf[:], self.item_reward_info = self.compute_quadcopter_reward()
        self.pre_actions = self.actions.clone()

    def compute_quadcopter_reward(self):
        thrust_cmds = torch.clamp(self.cmd_thrusts, min=0.0, max=1.0)
        effort_reward = .1 * (1 - thrust_cmds).sum(-1) / 4

        action_diff = self.actions - self.pre_actions
        thrust_reward = 0
        if self.ctl_mode == "pos" or self.ctl_mode == 'vel' or self.ctl_mode == 'prop':
            continous_action_reward = .2 * torch.exp(-torch.norm(action_diff[..., :], dim=-1))
        else:
            continous_action_reward = .2 * torch.exp(-torch.norm(action_diff[..., :-1], dim=-1)) \
                + .5 / (1.0 + torch.square(3 * action_diff[..., -1]))
            thrust = self.actions[..., -1]
            thrust_reward = .1 * (1 - torch.abs(0.1533 - thrust))

        target_positions = self.target_states[..., 9:12]
        relative_positions = target_positions - self.root_positions
        pos_diff = torch.norm(relative_positions, dim=-1)
        pos_reward = .7 / (1.0 + torch.square(1.6 * pos_diff))

        tar_direction = relative_positions / torch.norm(relative_positions, dim=1, keepdim=True)
        vel_direction = self.root_linvels / torch.norm(self.root_linvels, dim=1, keepdim=True)
        dot_product = (tar_direction * vel_direction).sum(dim=1)
        angle_diff = torch.acos(dot_product.clamp(-1.0, 1.0)).abs()
        vel_direction_reward = .1 * torch.exp(-angle_diff / torch.pi)

        target_matrix = self.target_states[..., 0:9].reshape(self.num_envs, 3, 3)
        target_euler = T.matrix_to_euler_angles_xyz(target_matrix)
        root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]])
        root_euler = T.matrix_to_euler_angles_xyz(root_matrix)
        yaw_diff = compute_yaw_diff(target_euler[..., 2], root_euler[..., 2]) / torch.pi
        yaw_reward = 1.0 / (1.0 + torch.square(3 * yaw_diff))

        spinnage = torch.square(self.root_angvels[:, -1])
        spin_reward = 1.0 / (1.0 + torch.square(3 * spinnage))

        ups = quat_axis(self.root_quats, 2)
        ups_reward = torch.square((ups[..., 2] + 1) / 2)

        if self.ctl_mode == "pos" or self.ctl_mode == 'vel' or self.ctl_mode == 'prop':
            reward = (continous_action_reward + effort_reward + pos_reward
                      + pos_reward * (vel_direction_reward + ups_reward + spin_reward + yaw_reward))
        else:
            reward = (continous_action_reward + effort_reward + thrust_reward + pos_reward
                      + pos_reward * (vel_direction_reward + ups_reward + spin_reward + yaw_reward))

        ones = torch.ones_like(self.reset_buf)
        die = torch.zeros_like(self.reset_buf)
        reset = torch.where(self.progress_buf >= self.max_episode_length - 1, ones, die)
        reset = torch.where(torch.norm(relative_positions, dim=1) > 4, ones, reset)
        reset = torch.where(relative_positions[..., 2] < -2, ones, reset)
        reset = torch.where(relative_positions[..., 2] > 2, ones, reset)
        reset = torch.where(ups[..., 2] < 0.0, ones, reset)
        if self.ctl_mode == "atti":
            reset = torch.where(self.actions[..., 0] < 0, ones, reset)

        item_reward_info = {}
        item_reward_info["continous_action_reward"] = continous_action_reward
        item_reward_info["effort_reward"] = effort_reward
        item_reward_info["thrust_reward"] = thrust_reward if self.ctl_mode == "atti" or self.ctl_mode == 'rate' else 0
        item_reward_info["pos_reward"] = pos_reward
        item_reward_info["vel_direction_reward"] = vel_direction_reward
        item_reward_info["ups_reward"] = ups_reward
        item_reward_info["spin_reward"] = spin_reward
        item_reward_info["yaw_reward"] = yaw_reward
        item_reward_info["reward"] = reward
        return reward, reset, item_reward_info
