"""PyTorch-CPU restatement of the reference Tracking task (oracle; test infrastructure).

Follows `/root/reference/airgym/envs/task/tracking.py`:
  reset_idx :159-192, compute_traj_lemniscate :194-200, compute_observations :202-214,
  compute_reward :216-221, compute_quadcopter_reward :223-296;
  config `airgym/envs/task/tracking_config.py:7-46` (48 obs, 36 s episodes).
Tracking inherits pre_physics_step / step / add_noise from Hovering
(`tracking.py:44` subclassing), so does this class.
"""
import torch

from . import rotations as T
from .hovering_ref import ACTION_LIMITS, HoveringRef, compute_yaw_diff, quat_axis

TRACKING_ACTION_LIMITS = dict(ACTION_LIMITS)
TRACKING_ACTION_LIMITS["pos"] = ([-6, -6, -6, -6.0], [6, 6, 6, 6.0])   # tracking.py:95-99


class TrackingRef(HoveringRef):
    task = "tracking"
    num_obs = 48
    episode_length_s = 36
    reset_pos_scale = (0.1, 0.1, 0.1)        # tracking.py:166-167
    reset_pos_offset = (0.0, 0.0, 1.0)
    reset_euler_scale = (0.1, 0.1, 0.2)      # :170-171
    action_limits = TRACKING_ACTION_LIMITS

    def __init__(self, *a, **k):
        self.pre_root_positions = None
        super().__init__(*a, **k)

    def _reset_extra(self, env_ids):
        if self.pre_root_positions is None:
            self.pre_root_positions = torch.zeros(self.num_envs, 3)
        self.pre_root_positions[env_ids] = 0

    def compute_traj_lemniscate(self, n_steps=10, step_size=5, scale=0.25):
        step = self.progress_buf.unsqueeze(1).expand(-1, n_steps) \
            + torch.arange(n_steps).repeat(self.num_envs, 1) * step_size
        t = step * self.dt * scale
        ref_x = 3 * torch.sin(t) / (1 + torch.cos(t) ** 2)
        ref_y = 3 * torch.sin(t) * torch.cos(t) / (1 + torch.cos(t) ** 2)
        ref_z = torch.ones_like(ref_x)
        return torch.stack((ref_x, ref_y, ref_z), dim=-1)

    def compute_observations(self, noise):
        self.root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]]).reshape(self.num_envs, 9)
        self.obs_buf[..., 0:9] = self.root_matrix
        self.obs_buf[..., 9:12] = self.root_positions
        self.obs_buf[..., 12:15] = self.root_linvels
        self.obs_buf[..., 15:18] = self.root_angvels
        self.ref_positions = self.compute_traj_lemniscate()
        self.related_future_pos = (self.ref_positions - self.root_positions.clone().unsqueeze(1)).reshape(self.num_envs, -1)
        self.obs_buf[..., 18:48] = self.related_future_pos
        self.add_noise(noise)
        return self.obs_buf

    def compute_reward(self):
        self.rew_buf[:], self.reset_buf[:], self.item_reward_info = self.compute_quadcopter_reward()
        self.pre_actions = self.actions.clone()
        self.pre_root_positions = self.root_positions.clone()

    def compute_quadcopter_reward(self):
        thrust_cmds = torch.clamp(self.cmd_thrusts, min=0.0, max=1.0)
        effort_reward = .1 * (1 - thrust_cmds).sum(-1) / 4

        action_diff = self.actions - self.pre_actions
        thrust_reward = 0
        if self.ctl_mode == "pos" or self.ctl_mode == 'vel' or self.ctl_mode == 'prop':
            continous_action_reward = .2 * torch.exp(-torch.norm(action_diff[..., :], dim=-1))
        else:
            continous_action_reward = .1 * torch.exp(-torch.norm(action_diff[..., :-1], dim=-1)) \
                + .5 / (1.0 + torch.square(2 * action_diff[..., -1]))
            thrust = self.actions[..., -1]
            thrust_reward = .1 * (1 - torch.abs(0.1533 - thrust))

        dist_diff = self.ref_positions[:, 0] - self.root_positions
        dist_norm = torch.norm(dist_diff, dim=-1)
        dist_reward = 1. / (1.0 + torch.square(1.8 * dist_norm))

        target_matrix = self.target_states[..., 0:9].reshape(self.num_envs, 3, 3)
        target_euler = T.matrix_to_euler_angles_xyz(target_matrix)
        root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]])
        root_euler = T.matrix_to_euler_angles_xyz(root_matrix)
        yaw_diff = compute_yaw_diff(target_euler[..., 2], root_euler[..., 2]) / torch.pi
        yaw_reward = 1 / (1.0 + torch.square(4 * yaw_diff))

        spinnage = torch.square(self.root_angvels[:, -1])
        spin_reward = 1 / (1.0 + torch.square(2 * spinnage))

        ups = quat_axis(self.root_quats, 2)
        ups_reward = torch.square((ups[..., 2] + 1) / 2)

        if self.ctl_mode == "pos" or self.ctl_mode == 'vel' or self.ctl_mode == 'prop':
            reward = (continous_action_reward + effort_reward + dist_reward
                      + dist_reward * (spin_reward + yaw_reward + ups_reward))
        else:
            reward = (continous_action_reward + effort_reward + thrust_reward + dist_reward
                      + dist_reward * (spin_reward + yaw_reward + ups_reward))

        ones = torch.ones_like(self.reset_buf)
        die = torch.zeros_like(self.reset_buf)
        reset = torch.where(self.progress_buf >= self.max_episode_length - 1, ones, die)
        reset = torch.where(dist_norm > 1.0, ones, reset)
        if self.ctl_mode == "atti":
            reset = torch.where(self.actions[..., 0] < 0, ones, reset)

        item_reward_info = {}
        item_reward_info["dist_norm"] = dist_norm
        item_reward_info["dist_reward"] = dist_reward
        item_reward_info["yaw_reward"] = yaw_reward
        item_reward_info["spin_reward"] = spin_reward
        item_reward_info["continous_action_reward"] = continous_action_reward
        item_reward_info["thrust_reward"] = thrust_reward if self.ctl_mode == "atti" or self.ctl_mode == 'rate' else 0
        item_reward_info["effort_reward"] = effort_reward
        item_reward_info["ups_reward"] = ups_reward
        item_reward_info["reward"] = reward
        return reward, reset, item_reward_info
