"""PyTorch-CPU restatement of the reference Balloon and Avoid tasks (oracle; test infrastructure - only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package).

Reference-owned tensor code restated op for op (pinned by tests/golden/{balloon,avoid}_*.npz, which were recorded from the
reference's own methods by tests/golden/make_golden_tasks.py):
  airgym/envs/task/balloon.py   reset_idx :57-99, step :101-143, compute_observations :145-158, compute_reward :160-165,
                                hit_reward :167-170, compute_quadcopter_reward :172-237
  airgym/envs/task/avoid.py     calculate_object_velocity :58-90, reset_idx :91-163, step :165-206,
                                compute_observations :208-233, compute_reward :235-240, compute_quadcopter_reward :242-300
  airgym/envs/base/customized.py pre_physics_step :216-298 (shared with Planning: rate limits +-1, the clamped copy drives the
                                controller, self.actions keeps the thrust-remapped raw action), add_noise :450-459,
                                dump_images :399-435
  airgym/envs/task/{balloon,avoid}_config.py: 18 / 16 observations, 8 s / 6 s episodes, reset_on_collision, ground plane,
                                Balloon: static ball r = 0.2 with the robot's collision mask (never collides with it);
                                Avoid: one 0.3 m cube, density 0.5, free body, collision mask 0 (collides), depth camera on.

Build-defined spec, PARITY UNPINNED (PhysX / the IsaacGym rasteriser in the reference):
  * integrator + cascade as in hovering_ref (rigid_body.py, px4_cascade.py);
  * Avoid's thrown cube: semi-implicit Euler ballistic flight (v_z -= g dt; p += v dt), no rotation (it is released with zero
    angular velocity), inelastic landing on the ground plane (p_z = 0.15, v = 0); the parked cube of the 20 % "no throw"
    episodes stays at (-999, -999, 0);
  * collisions: robot collision sphere r = 0.2 (X152b/model.urdf:13-18) against the cube (exact sphere-vs-box distance) and
    the ground plane; Balloon: ground plane only;
  * Avoid's depth image: the analytic ray-caster of planning_ref with the cube as an axis-aligned box and the ground plane.
Random numbers: counter-based Philox (reset stream 0, observation noise stream 1, image streams 2-4); every draw can be
supplied explicitly.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox
from . import rotations as T
from .hovering_ref import compute_yaw_diff, quat_axis, tensor_clamp
from .planning_ref import (CAM_H, CAM_W, PLANNING_ACTION_LIMITS, ROBOT_RADIUS, STREAM_IMG_ADD, STREAM_IMG_KERNEL,
                           STREAM_IMG_MUL, post_process_depth, render_depth_one)
from .px4_cascade import CascadeState, controller_update
from .rigid_body import body_wrench_from_cmd, rk4_step

NOISE_SIGMA = (1e-3, 5e-3, 2e-2, 4e-1)     # customized.py:451-454
CUBE_HALF = 0.15                           # env_assets/cubes/1x1/1x1dae.dae: unit cube vertices +-1 under a 0.15 scale node
GRAVITY = 9.81
BALLOON_RESET_UNIFORMS = 15
AVOID_RESET_UNIFORMS = 11


def rf(lo, hi, x):
    """torch_rand_float(lo, hi) with the uniform supplied: (hi - lo) * u + lo (airgym/utils/torch_utils.py:192-193)."""
    return (hi - lo) * x + lo


class CustomizedRef:
    """What Balloon and Avoid inherit from Customized (customized.py): buffers, action limits, pre_physics_step."""
    num_obs = 18
    episode_length_s = 8
    reward_terms = ()

    def __init__(self, num_envs, ctl_mode="rate", seed=0, env_id_offset=0, dt=0.01):
        assert ctl_mode in PLANNING_ACTION_LIMITS
        self.num_envs, self.ctl_mode, self.dt, self.seed = num_envs, ctl_mode, dt, seed
        self.num_actions = 5 if ctl_mode == "atti" else 4
        self.max_episode_length = int(self.episode_length_s / dt)
        self.env_ids_global = np.arange(env_id_offset, env_id_offset + num_envs, dtype=np.uint32)
        self.tick = 0
        self.counter = 0
        lo, hi = PLANNING_ACTION_LIMITS[ctl_mode]
        self.action_lower_limits = torch.tensor(lo, dtype=torch.float32)
        self.action_upper_limits = torch.tensor(hi, dtype=torch.float32)
        n = num_envs
        self.obs_buf = torch.zeros(n, self.num_obs)
        self.rew_buf = torch.zeros(n)
        self.reset_buf = torch.ones(n, dtype=torch.long)
        self.progress_buf = torch.zeros(n, dtype=torch.long)
        self.time_out_buf = torch.zeros(n, dtype=torch.bool)
        self.root_states = torch.zeros(n, 13)
        self.root_states[:, 6] = 1
        self.actions = torch.zeros(n, self.num_actions)
        self.pre_actions = torch.zeros(n, self.num_actions)
        self.cmd_thrusts = torch.zeros(n, 4)
        self.ctl_state = CascadeState(n)
        self.pre_root_positions = torch.zeros(n, 3)
        self.collisions = torch.zeros(n)
        self.extras = {}
        self.item_reward_info = {}

    root_positions = property(lambda s: s.root_states[:, 0:3])
    root_quats = property(lambda s: s.root_states[:, 3:7])
    root_linvels = property(lambda s: s.root_states[:, 7:10])
    root_angvels = property(lambda s: s.root_states[:, 10:13])

    def _reset_uniforms(self, env_ids, count):
        nb = (count + 3) // 4
        raw = philox.raw_blocks(self.seed, self.env_ids_global[env_ids.numpy()], self.tick, philox.STREAM_RESET, nb)
        return torch.from_numpy(philox.u32_to_unit_float(raw)[:, :count])

    def pre_physics_step(self, _actions):
        """customized.py:216-298"""
        self.counter += 1
        was_reset = self.reset_buf.clone()
        self.actions = _actions.clone().to(torch.float32)
        if self.ctl_mode in ("rate", "atti"):
            self.actions[..., -1] = 0.5 + 0.5 * self.actions[..., -1]
        clamped = tensor_clamp(self.actions, self.action_lower_limits, self.action_upper_limits)
        self.root_states[..., 3:7] = torch.where(self.root_states[..., 6:7] < 0, -self.root_states[..., 3:7],
                                                 self.root_states[..., 3:7])
        self.cmd_thrusts = controller_update(self.ctl_mode, self.ctl_state, clamped, self.root_states)
        self.fz, self.tau_b = body_wrench_from_cmd(self.cmd_thrusts, (was_reset == 0).float())

    def _finish_step(self, reset_uniforms):
        if True:   # reset_on_collision (balloon_config.py:19, avoid_config.py:19)
            ones = torch.ones_like(self.reset_buf)
            self.reset_buf = torch.where(self.collisions > 0, ones, self.reset_buf)
        reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        self.last_reset_env_ids = reset_env_ids
        if len(reset_env_ids) > 0:
            self.reset_idx(reset_env_ids, None if reset_uniforms is None else reset_uniforms[reset_env_ids])
        self.time_out_buf = self.progress_buf > self.max_episode_length
        self.extras["time_outs"] = self.time_out_buf
        self.extras["item_reward_info"] = self.item_reward_info
        self.tick += 1


# ----------------------------------------------------------------------------------------------------------- Balloon
class BalloonRef(CustomizedRef):
    task = "balloon"
    num_obs = 18
    episode_length_s = 8          # balloon_config.py:17
    reward_terms = ("guidance_reward", "hit_reward", "action_smoothness_reward", "effort_reward", "ups_reward", "reward")

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.balloon_positions = torch.zeros(self.num_envs, 3)
        self.reset_idx(torch.arange(self.num_envs))
        self.tick += 1

    def reset_idx(self, env_ids, uniforms=None):
        """balloon.py:57-99.  u[15]: balloon x y z | root x y | root z | euler x y z | linvel(3) | angvel(3)."""
        k = len(env_ids)
        u = self._reset_uniforms(env_ids, BALLOON_RESET_UNIFORMS) if uniforms is None else uniforms
        bal = torch.zeros(k, 3)
        bal[:, 0] = .5 * rf(-1.0, 1.0, u[:, 0]) + 2.5
        bal[:, 1] = 2. * rf(-1.0, 1.0, u[:, 1]) + 0.
        bal[:, 2] = .3 * rf(-1., 1., u[:, 2]) + 1.
        self.balloon_positions[env_ids] = bal
        st = torch.zeros(k, 13)
        st[:, 0:2] = 0.1 * rf(-1.0, 1.0, u[:, 3:5]) + 0.
        st[:, 2] = 0.2 * rf(-1., 1., u[:, 5]) + 1.
        root_angle = torch.stack((0.1 * rf(-torch.pi, torch.pi, u[:, 6]), 0.1 * rf(0., torch.pi, u[:, 7]),
                                  0.2 * rf(-torch.pi, torch.pi, u[:, 8])), -1)
        root_quats = T.matrix_to_quaternion(T.euler_angles_to_matrix(root_angle, "XYZ"))
        st[:, 3:7] = root_quats[:, [1, 2, 3, 0]]
        st[:, 7:10] = 0.5 * rf(-1.0, 1.0, u[:, 9:12])
        st[:, 10:13] = 0.2 * rf(-1.0, 1.0, u[:, 12:15])
        self.root_states[env_ids] = st
        self.reset_buf[env_ids] = 1
        self.progress_buf[env_ids] = 0
        self.pre_actions[env_ids] = 0
        self.pre_root_positions[env_ids] = 0
        self.ctl_state.reset(env_ids, self.root_states)

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs))
        self.tick += 1
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions))
        return obs, priv

    def check_collisions(self):
        """The balloon shares the robot's collision mask (no contact); what is left is the ground plane."""
        self.collisions = (self.root_positions[:, 2] <= ROBOT_RADIUS).float()

    def step(self, actions, noise=None, reset_uniforms=None):
        """balloon.py:101-143"""
        self.pre_physics_step(actions)
        self.root_states = rk4_step(self.root_states, self.fz, self.tau_b, self.dt)
        self.progress_buf += 1
        self.check_collisions()
        if noise is None:
            noise = torch.from_numpy(philox.normals(self.seed, self.env_ids_global, self.tick, philox.STREAM_OBS_NOISE, 18))
        self.compute_observations(noise)
        self.compute_reward()
        self._finish_step(reset_uniforms)
        return self.obs_buf, None, self.rew_buf, self.reset_buf, self.extras

    def compute_observations(self, noise):
        """balloon.py:145-158 (+ Customized.add_noise :450-459); the static balloon keeps its identity orientation."""
        self.root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]]).reshape(self.num_envs, 9)
        self.obs_buf[..., 0:9] = self.root_matrix
        self.obs_buf[..., 9:12] = self.root_positions
        self.obs_buf[..., 12:15] = self.root_linvels
        self.obs_buf[..., 15:18] = self.root_angvels
        self.obs_buf[..., 0:9] += NOISE_SIGMA[0] * noise[:, 0:9]
        self.obs_buf[..., 9:12] += NOISE_SIGMA[1] * noise[:, 9:12]
        self.obs_buf[..., 12:15] += NOISE_SIGMA[2] * noise[:, 12:15]
        self.obs_buf[..., 15:18] += NOISE_SIGMA[3] * noise[:, 15:18]
        balloon_matrix = torch.eye(3).reshape(1, 9).repeat(self.num_envs, 1)
        self.obs_buf[..., 0:9] -= balloon_matrix
        self.obs_buf[..., 9:12] -= self.balloon_positions
        return self.obs_buf

    def compute_reward(self):
        self.rew_buf[:], self.reset_buf[:], self.item_reward_info = self.compute_quadcopter_reward()
        self.pre_actions = self.actions.clone()
        self.pre_root_positions = self.root_positions.clone()

    def compute_quadcopter_reward(self):
        """balloon.py:172-237"""
        relative_positions = self.balloon_positions - self.root_positions
        direction_vector = F.normalize(relative_positions, dim=-1)
        direction_yaw = torch.atan2(direction_vector[..., 1], direction_vector[..., 0])
        root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]])
        root_euler = T.matrix_to_euler_angles_xyz(root_matrix)
        relative_heading = compute_yaw_diff(root_euler[..., 2], direction_yaw)
        yaw_distance = torch.norm(relative_heading.unsqueeze(-1), dim=1)
        yaw_reward = 1.0 / (1.0 + torch.square(1.6 * yaw_distance))
        guidance_reward = 30 * (torch.norm(self.balloon_positions - self.pre_root_positions, dim=-1)
                                - torch.norm(self.balloon_positions - self.root_positions, dim=-1))
        ups = quat_axis(self.root_quats, axis=2)
        ups_reward = 0.5 * torch.pow((ups[..., 2] + 1) / 2, 2)
        check = torch.norm(self.balloon_positions - self.root_positions, dim=-1)
        hit_reward = 800 * torch.where(check < 0.1, torch.tensor(1), torch.tensor(0))
        effort_reward = .1 * torch.exp(-self.actions.pow(2).sum(-1))
        action_diff = torch.norm(self.actions - self.pre_actions, dim=-1)
        action_smoothness_reward = .1 * torch.exp(-action_diff)
        reward = guidance_reward + yaw_reward + hit_reward + action_smoothness_reward + ups_reward + effort_reward
        ones = torch.ones_like(self.reset_buf)
        die = torch.zeros_like(self.reset_buf)
        reset = torch.where(self.progress_buf >= self.max_episode_length - 1, ones, die)
        reset = torch.where(self.actions[..., -1] < -1, ones, reset)
        reset = torch.where(self.actions[..., -1] > 1, ones, reset)
        reset = torch.where(relative_positions[..., 0] < -0.2, ones, reset)
        reset = torch.where(self.root_linvels[..., 0] < 0, ones, reset)
        reset = torch.where(torch.norm(relative_positions, dim=1) > 4, ones, reset)
        reset = torch.where(self.root_positions[..., 2] < 0.5, ones, reset)
        reset = torch.where(self.root_positions[..., 2] > 1.5, ones, reset)
        reset = torch.where(check < 0.1, ones, reset)
        info = {"guidance_reward": guidance_reward, "hit_reward": hit_reward,
                "action_smoothness_reward": action_smoothness_reward, "effort_reward": effort_reward,
                "ups_reward": ups_reward, "reward": reward}
        return reward, reset, info


# ------------------------------------------------------------------------------------------------------------- Avoid
def sphere_box_distance(p, centre, half):
    """Distance from points p [N,3] to axis-aligned cubes (centre [N,3], half extent)."""
    d = torch.clamp((p - centre).abs() - half, min=0.0)
    return torch.sqrt((d * d).sum(-1))


class AvoidRef(CustomizedRef):
    task = "avoid"
    num_obs = 16
    episode_length_s = 6          # avoid_config.py:17
    cam_rate = 4
    reward_terms = ("pose_reward", "ups_reward", "spin_reward", "effort_reward", "action_smoothness_reward",
                    "thrust_reward", "alive_reward", "reward")
    target_state = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0]    # avoid_config.py:11

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        n = self.num_envs
        self.target_states = torch.tensor(self.target_state, dtype=torch.float32).repeat(n, 1)
        self.object_positions = torch.zeros(n, 3)
        self.object_linvels = torch.zeros(n, 3)
        self.full_camera_array = torch.zeros(n, 1, CAM_W, CAM_H)
        self.reset_idx(torch.arange(n))
        self.tick += 1

    def calculate_object_velocity(self, positions, v_e, aim_u, g=GRAVITY):
        """avoid.py:58-90 for the thrown rows; aim_u [K,3] are the uniforms of the 0.3-m aiming jitter."""
        drone_position = 0.3 * rf(-1.0, 1.0, aim_u) + torch.tensor([0.0, 0.0, 1.0])
        direction = drone_position - positions
        distance_xy = torch.norm(direction[:, :2], dim=1, keepdim=True)
        unit_direction_xy = direction[:, :2] / distance_xy
        v_e = torch.tensor(v_e).expand_as(distance_xy)
        t = distance_xy / v_e
        z_c = positions[:, 2].unsqueeze(1)
        z_u = drone_position[:, 2].unsqueeze(1)
        v_z = (z_u - z_c + 0.5 * g * t ** 2) / t
        v_x = unit_direction_xy[:, 0].unsqueeze(1) * v_e
        v_y = unit_direction_xy[:, 1].unsqueeze(1) * v_e
        return torch.cat([v_x, v_y, v_z], dim=1)

    def reset_idx(self, env_ids, uniforms=None):
        """avoid.py:91-163.  u[11]: throw mask | theta | aim xyz | root x y | root z | euler x y | euler z."""
        k = len(env_ids)
        u = self._reset_uniforms(env_ids, AVOID_RESET_UNIFORMS) if uniforms is None else uniforms
        thrown = u[:, 0] < 0.8
        R = 4.2
        theta = torch.pi / 6 * rf(-1.0, 1.0, u[:, 1])
        pos = torch.stack((R * torch.cos(theta), R * torch.sin(theta), torch.full((k,), 1.4)), -1)
        vel = self.calculate_object_velocity(pos, 4.5, u[:, 2:5])
        parked = torch.tensor([-999., -999., 0.]).expand(k, 3)
        self.object_positions[env_ids] = torch.where(thrown[:, None], pos, parked)
        self.object_linvels[env_ids] = torch.where(thrown[:, None], vel, torch.zeros(k, 3))
        st = torch.zeros(k, 13)
        st[:, 6] = 1.0                                                 # initial_root_states
        st[:, 0:2] = 0.2 * rf(-1.0, 1.0, u[:, 5:7]) + 0.
        st[:, 2] = 0.2 * rf(-1., 1., u[:, 7]) + 1.
        root_angle = torch.cat((0.01 * rf(-torch.pi, torch.pi, u[:, 8:10]), 0.05 * rf(-torch.pi, torch.pi, u[:, 10:11])), -1)
        root_quats = T.matrix_to_quaternion(T.euler_angles_to_matrix(root_angle, "XYZ"))
        st[:, 3:7] = root_quats[:, [1, 2, 3, 0]]
        self.root_states[env_ids] = st
        self.reset_buf[env_ids] = 1
        self.progress_buf[env_ids] = 0
        self.pre_actions[env_ids] = 0
        self.pre_root_positions[env_ids] = 0
        self.ctl_state.reset(env_ids, self.root_states)

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs))
        self.tick += 1
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions))
        return obs, priv

    def step_object(self):
        """The thrown cube (build-defined, see the module docstring)."""
        fly = (self.object_positions[:, 0] != -999.0) & ~((self.object_positions[:, 2] <= CUBE_HALF)
                                                          & (self.object_linvels.abs().sum(-1) == 0))
        v = self.object_linvels.clone()
        v[:, 2] = v[:, 2] - GRAVITY * self.dt
        p = self.object_positions + v * self.dt
        landed = p[:, 2] <= CUBE_HALF
        p[:, 2] = torch.where(landed, torch.tensor(CUBE_HALF), p[:, 2])
        v = torch.where(landed[:, None], torch.zeros_like(v), v)
        self.object_positions = torch.where(fly[:, None], p, self.object_positions)
        self.object_linvels = torch.where(fly[:, None], v, self.object_linvels)

    def check_collisions(self):
        d = sphere_box_distance(self.root_positions, self.object_positions, CUBE_HALF)
        self.collisions = ((d <= ROBOT_RADIUS) | (self.root_positions[:, 2] <= ROBOT_RADIUS)).float()

    def render_cameras(self, image_randoms=None):
        npix = CAM_W * CAM_H
        for e in range(self.num_envs):
            raw = render_depth_one(self.root_positions[e], self.root_quats[e], None, None, None, None, None,
                                   box=(self.object_positions[e], CUBE_HALF))
            if image_randoms is None:
                ids = self.env_ids_global[e:e + 1]
                add = torch.from_numpy(philox.normals(self.seed, ids, self.tick, STREAM_IMG_ADD, npix)).reshape(CAM_W, CAM_H)
                mul = torch.from_numpy(philox.normals(self.seed, ids, self.tick, STREAM_IMG_MUL, npix)).reshape(CAM_W, CAM_H)
                kraw = philox.raw_blocks(self.seed, ids, self.tick, STREAM_IMG_KERNEL, 7)[0, :25]
                ker = torch.from_numpy((kraw >> np.uint32(24)).astype(np.float32) / np.float32(256.0)).reshape(5, 5)
            else:
                add, mul, ker = image_randoms[0][e], image_randoms[1][e], image_randoms[2][e]
            self.full_camera_array[e] = post_process_depth(raw, add, mul, ker)

    def step(self, actions, reset_uniforms=None, image_randoms=None):
        """avoid.py:165-206"""
        self.pre_physics_step(actions)
        self.root_states = rk4_step(self.root_states, self.fz, self.tau_b, self.dt)
        self.step_object()
        if self.counter % self.cam_rate == 0:
            self.render_cameras(image_randoms)
        self.progress_buf += 1
        self.check_collisions()
        self.compute_observations()
        self.compute_reward()
        self._finish_step(reset_uniforms)
        obs = {"image": self.full_camera_array, "observation": self.obs_buf}
        return obs, None, self.rew_buf, self.reset_buf, self.extras

    def compute_observations(self):
        """avoid.py:208-233 (actions_local aliases the thrust-remapped action tensor, customized.py:222-229)"""
        rot_matrix_global = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]])
        yaw = torch.atan2(rot_matrix_global[:, 1, 0], rot_matrix_global[:, 0, 0])
        cos_yaw, sin_yaw = torch.cos(yaw), torch.sin(yaw)
        z, o = torch.zeros_like(yaw), torch.ones_like(yaw)
        self.world_to_local = torch.stack([torch.stack([cos_yaw, -sin_yaw, z], dim=1),
                                           torch.stack([sin_yaw, cos_yaw, z], dim=1),
                                           torch.stack([z, z, o], dim=1)], dim=2)
        rot_matrix_local = torch.bmm(self.world_to_local, rot_matrix_global)
        self.euler_angles_local = T.matrix_to_euler_angles_xyz(rot_matrix_local)
        self.vel_local = torch.einsum("bij,bj->bi", self.world_to_local, self.root_linvels)
        self.ang_vel_local = torch.einsum("bij,bj->bi", self.world_to_local, self.root_angvels)
        self.obs_buf[..., 0:3] = self.root_positions - self.target_states[..., 9:12]
        self.obs_buf[..., 3:6] = self.euler_angles_local
        self.obs_buf[..., 6:9] = self.vel_local
        self.obs_buf[..., 9:12] = self.ang_vel_local
        self.obs_buf[..., 12:16] = self.actions[..., :4]
        return self.obs_buf

    def compute_reward(self):
        self.rew_buf[:], self.reset_buf[:], self.item_reward_info = self.compute_quadcopter_reward()
        self.pre_actions = self.actions.clone()
        self.pre_root_positions = self.root_positions.clone()

    def compute_quadcopter_reward(self):
        """avoid.py:242-300"""
        target_positions = self.target_states[..., 9:12]
        relative_positions = target_positions - self.root_positions
        target_matrix = self.target_states[..., 0:9].reshape(self.num_envs, 3, 3)
        target_euler = T.matrix_to_euler_angles_xyz(target_matrix)
        root_matrix = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]])
        root_euler = T.matrix_to_euler_angles_xyz(root_matrix)
        relative_heading = compute_yaw_diff(target_euler[..., 2], root_euler[..., 2])
        distance = torch.norm(torch.cat((relative_positions, relative_heading.unsqueeze(-1)), dim=-1), dim=1)
        pose_reward = 1.0 / (1.0 + torch.square(1.6 * distance))
        ups = quat_axis(self.root_quats, axis=2)
        ups_reward = torch.square((ups[..., 2] + 1) / 2)
        spinnage = torch.square(self.root_angvels[:, -1])
        spin_reward = 1.0 / (1.0 + torch.square(spinnage))
        effort_reward = .1 * torch.exp(-self.actions.pow(2).sum(-1))
        action_diff = torch.norm(self.actions[..., :-1] - self.pre_actions[..., :-1], dim=-1)
        thrust_reward = .05 * (1 - torch.abs(0.1533 - self.actions[..., -1]))
        action_smoothness_reward = .1 * torch.exp(-action_diff)
        alive_reward = torch.where(self.collisions > 0, -500., 0.5)
        reward = (pose_reward + pose_reward * (ups_reward + spin_reward) + effort_reward + action_smoothness_reward
                  + thrust_reward + alive_reward)
        ones = torch.ones_like(self.reset_buf)
        die = torch.zeros_like(self.reset_buf)
        reset = torch.where(self.progress_buf >= self.max_episode_length - 1, ones, die)
        reset = torch.where(self.root_positions[..., 2] < .3, ones, reset)
        reset = torch.where(self.root_positions[..., 2] > 1.7, ones, reset)
        reset = torch.where(relative_positions.norm(dim=-1) > 2.0, ones, reset)
        reset = torch.where(ups[..., 2] < 0.0, ones, reset)
        info = {"pose_reward": pose_reward, "ups_reward": ups_reward, "spin_reward": spin_reward,
                "effort_reward": effort_reward, "action_smoothness_reward": action_smoothness_reward,
                "thrust_reward": thrust_reward, "alive_reward": alive_reward, "reward": reward}
        return reward, reset, info
