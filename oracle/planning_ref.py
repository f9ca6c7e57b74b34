"""PyTorch-CPU restatement of the reference Planning task (oracle; test infrastructure).

Reference-owned tensor code restated op for op (pinned by tests/golden/planning_*.npz):
  airgym/envs/task/planning.py   reset_idx :63-136, step :138-184, compute_observations :186-214,
                                 compute_quadcopter_reward :223-307
  airgym/envs/base/customized.py pre_physics_step :216-298 (rate-mode limits +-1, the clamped COPY goes to the
                                 controller while self.actions keeps the thrust-remapped raw action, Q17),
                                 check_collisions :393-397, dump_images :399-435 (depth post-processing)
  airgym/envs/task/planning_config.py:7-80 (16 obs, 16 s episodes, camera 212x120, hfov 87, far 5 m, cam_dt 0.04)

Build-defined spec, PARITY UNPINNED (the reference gets these from IsaacGym's PhysX + rasteriser):
  * the scene: 40 capped cylinders per env (numeric parameters of env_assets/thin/tree_<k>.urdf in
    airgym_amd/assets/thin_trees.json; variant per (env, slot) drawn once from the counter RNG), the goal sphere
    (r = 0.2, rendered, not collidable), the ground plane z = 0;
  * `render_depth`: pin-hole ray-cast, z-depth along the camera axis (camera at body (0.15, 0, 0.1), looking along
    body +x, 212 x 120, square pixels, hfov 87 deg, far plane 5 m; no hit -> +inf);
  * `check_collisions`: robot collision sphere r = 0.2 (X152b/model.urdf:13-18) against cylinders and ground;
  * integrator and cascade as in hovering_ref (rigid_body.py, px4_cascade.py).
Random numbers: counter-based Philox streams (reset 0, image additive noise 2, multiplicative 3, blur kernel 4,
obstacle variants 5); every draw can also be supplied explicitly for parity tests.
"""
import json
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import philox
from . import rotations as T
from .hovering_ref import quat_axis, tensor_clamp
from .px4_cascade import CascadeState, controller_update
from .rigid_body import body_wrench_from_cmd, rk4_step

LENGTH = 8.0
WIDTH = 4.0
FLY_HEIGHT = 1.5
NUM_OBSTACLES = 40
CAM_W, CAM_H = 212, 120
CAM_HFOV_DEG = 87.0
CAM_FAR = 5.0
CAM_OFFSET = (0.15, 0.0, 0.1)
ROBOT_RADIUS = 0.2
GOAL_RADIUS = 0.2

STREAM_IMG_ADD = 2
STREAM_IMG_MUL = 3
STREAM_IMG_KERNEL = 4
STREAM_VARIANT = 5
RESET_UNIFORMS = 3 * NUM_OBSTACLES + 1      # per obstacle x, y, yaw; goal y

PLANNING_ACTION_LIMITS = {   # customized.py:93-123
    "pos": ([-3, -3, -3, -6.0], [3, 3, 3, 6.0]),
    "vel": ([-6, -6, -6, -6], [6, 6, 6, 6]),
    "atti": ([-1, -1, -1, -1, 0.0], [1, 1, 1, 1, 1]),
    "rate": ([-1, -1, -1, 0], [1, 1, 1, 1]),
    "prop": ([0, 0, 0, 0], [1, 1, 1, 1]),
}

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "airgym_amd", "assets",
                       "thin_trees.json")


def load_variant_table():
    """[100, 8] float32: centre (3), unit axis (3), radius, half length - in the obstacle's own frame."""
    raw = np.asarray(json.load(open(_ASSETS))["variants"], dtype=np.float64)
    r, length, ox, oy, oz, roll, pitch, yaw = raw.T
    assert np.abs(roll).max() < 1e-12
    n = np.stack((np.cos(yaw) * np.sin(pitch), np.sin(yaw) * np.sin(pitch), np.cos(pitch)), -1)
    tab = np.concatenate((np.stack((ox, oy, oz), -1), n, r[:, None], 0.5 * length[:, None]), -1)
    return tab.astype(np.float32)


def camera_fx():
    return np.float32((CAM_W / 2.0) / math.tan(math.radians(CAM_HFOV_DEG) / 2.0))


def world_cylinders(obst, variants, table):
    """obst [N,40,3] (x, y, yaw), variants [N,40] int -> centre [N,40,3], axis [N,40,3], r [N,40], h [N,40]."""
    tab = torch.from_numpy(table)[variants]                  # [N,40,8]
    c, s = torch.cos(obst[..., 2]), torch.sin(obst[..., 2])
    o, n = tab[..., 0:3], tab[..., 3:6]
    centre = torch.stack((obst[..., 0] + (c * o[..., 0] - s * o[..., 1]),
                          obst[..., 1] + (s * o[..., 0] + c * o[..., 1]), o[..., 2]), -1)
    axis = torch.stack((c * n[..., 0] - s * n[..., 1], s * n[..., 0] + c * n[..., 1], n[..., 2]), -1)
    return centre, axis, tab[..., 6], tab[..., 7]


def _ray_capped_cylinders(orig, dirs, centre, axis, r, h):
    """orig [3], dirs [P,3] (not normalised), cylinders [K,...] -> smallest ray parameter t > 0 per ray (inf = miss)."""
    oc = orig[None, None, :] - centre[None, :, :]                    # [1,K,3]
    dn = (dirs[:, None, :] * axis[None]).sum(-1)                     # [P,K]
    on = (oc * axis[None]).sum(-1)                                   # [1,K]
    dperp = dirs[:, None, :] - dn[..., None] * axis[None]
    operp = oc - on[..., None] * axis[None]
    a = (dperp * dperp).sum(-1)
    b = (dperp * operp).sum(-1)
    c = (operp * operp).sum(-1) - (r * r)[None]
    disc = b * b - a * c
    inf = torch.full_like(a, float("inf"))
    ok = (disc >= 0) & (a > 1e-12)
    sq = torch.sqrt(torch.clamp(disc, min=0.0))
    a_safe = torch.where(a > 1e-12, a, torch.ones_like(a))
    best = inf
    for t in ((-b - sq) / a_safe, (-b + sq) / a_safe):
        y = on + t * dn
        hit = ok & (t > 0) & (y.abs() <= h[None])
        best = torch.minimum(best, torch.where(hit, t, inf))
    for sgn in (1.0, -1.0):                                          # end caps
        dn_safe = torch.where(dn.abs() > 1e-12, dn, torch.ones_like(dn))
        t = (sgn * h[None] - on) / dn_safe
        p = operp + t[..., None] * dperp
        hit = (dn.abs() > 1e-12) & (t > 0) & ((p * p).sum(-1) <= (r * r)[None])
        best = torch.minimum(best, torch.where(hit, t, inf))
    return best.min(dim=1).values


def _ray_aabb(orig, dirs, centre, half):
    """Slab test: smallest t > 0 where orig + t dirs enters the axis-aligned box (centre [3], half extent), inf = miss.
    A ray starting inside the box returns its exit parameter."""
    inf = torch.full((dirs.shape[0],), float("inf"))
    tmin = torch.full_like(inf, -float("inf"))
    tmax = inf.clone()
    ok = torch.ones_like(inf, dtype=torch.bool)
    for ax in range(3):
        d = dirs[:, ax]
        par = d.abs() <= 1e-12
        inv = 1.0 / torch.where(par, torch.ones_like(d), d)
        t0 = (centre[ax] - half - orig[ax]) * inv
        t1 = (centre[ax] + half - orig[ax]) * inv
        lo, hi = torch.minimum(t0, t1), torch.maximum(t0, t1)
        inside = (orig[ax] >= centre[ax] - half) & (orig[ax] <= centre[ax] + half)
        tmin = torch.where(par, tmin, torch.maximum(tmin, lo))
        tmax = torch.where(par, tmax, torch.minimum(tmax, hi))
        ok = ok & (~par | inside)
    hit = ok & (tmax >= tmin) & (tmax > 0)
    t = torch.where(tmin > 0, tmin, tmax)
    return torch.where(hit, t, inf)


def render_depth_one(pos, quat_xyzw, centre, axis, r, h, goal, box=None):
    """z-depth image [CAM_H, CAM_W] float32 for one env (inf where nothing is hit within the far plane).
    goal = None skips the goal sphere; box = (centre [3], half extent) adds an axis-aligned cube (Avoid's thrown object)."""
    fx = camera_fx()
    R = T.quaternion_to_matrix(quat_xyzw[[3, 0, 1, 2]])
    orig = pos + R @ torch.tensor(CAM_OFFSET)
    u = torch.arange(CAM_W, dtype=torch.float32) + 0.5
    v = torch.arange(CAM_H, dtype=torch.float32) + 0.5
    dy = (CAM_W / 2.0 - u) / fx
    dz = (CAM_H / 2.0 - v) / fx
    d_body = torch.stack((torch.ones(CAM_H, CAM_W), dy[None, :].expand(CAM_H, CAM_W), dz[:, None].expand(CAM_H, CAM_W)), -1)
    dirs = (d_body.reshape(-1, 3) @ R.T)                             # world directions, x-component of body dir = 1
    inf = torch.full((dirs.shape[0],), float("inf"))
    t = _ray_capped_cylinders(orig, dirs, centre, axis, r, h) if centre is not None and centre.shape[0] > 0 else inf.clone()
    if box is not None:
        t = torch.minimum(t, _ray_aabb(orig, dirs, box[0], box[1]))
    # ground plane z = 0
    tg = torch.where(dirs[:, 2] < -1e-12, -orig[2] / torch.where(dirs[:, 2] < -1e-12, dirs[:, 2], -torch.ones_like(inf)), inf)
    t = torch.minimum(t, torch.where(tg > 0, tg, inf))
    if goal is not None:    # goal sphere
        oc = orig - goal
        a = (dirs * dirs).sum(-1)
        b = (dirs * oc[None]).sum(-1)
        c = (oc * oc).sum() - GOAL_RADIUS ** 2
        disc = b * b - a * c
        ts = (-b - torch.sqrt(torch.clamp(disc, min=0.0))) / a
        t = torch.minimum(t, torch.where((disc >= 0) & (ts > 0), ts, inf))
    t = torch.where(t <= CAM_FAR, t, inf)                            # the ray parameter IS the z-depth (dir_x = 1)
    return t.reshape(CAM_H, CAM_W)


def point_capped_cylinder_distance(p, centre, axis, r, h):
    """Euclidean distance from points p [N,3] to capped cylinders [N,K,...] -> [N,K]."""
    d = p[:, None, :] - centre
    y = (d * axis).sum(-1)
    rad = torch.sqrt(torch.clamp((d * d).sum(-1) - y * y, min=0.0))
    dr = torch.clamp(rad - r, min=0.0)
    dy = torch.clamp(y.abs() - h, min=0.0)
    return torch.sqrt(dr * dr + dy * dy)


def post_process_depth(cam, add_noise, mul_noise, kernel):
    """dump_images, customized.py:399-435, for ONE env.  cam [H, W] raw z-depth (IsaacGym hands -depth; the
    reference negates and transposes).  add_noise / mul_noise [W, H] standard normals, kernel [5,5] in [0,1)."""
    img = cam.T.unsqueeze(0)                                              # (1, W, H), positive depth
    img = torch.where(img > 4.5, torch.tensor(4.5), img)
    img = torch.clamp(img, 0, 4.5) / 4.5
    noisy = img + (0.0 + 0.1 * add_noise.unsqueeze(0))                    # torch.normal(mean, std) = mean + std*z
    img = torch.clamp(noisy, 0.0, img.max())
    noisy = img * (1.0 + 0.3 * mul_noise.unsqueeze(0))
    img = torch.clamp(noisy, 0.0, img.max())
    k = kernel.unsqueeze(0).unsqueeze(0)
    return F.conv2d(img.unsqueeze(0), k, padding=2).squeeze(0)           # (1, W, H)


class PlanningRef:
    task = "planning"
    num_obs = 16
    episode_length_s = 16
    cam_rate = 4                      # cam_dt / dt = 0.04 / 0.01, planning.py:153

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
        self.table = load_variant_table()
        raw = philox.raw_blocks(seed, self.env_ids_global, 0xFFFFFFFF, STREAM_VARIANT, NUM_OBSTACLES // 4)
        self.variants = torch.from_numpy((raw % 100).astype(np.int64))       # [N,40], fixed for the env's lifetime
        self.obs_buf = torch.zeros(n, self.num_obs)
        self.rew_buf = torch.zeros(n)
        self.reset_buf = torch.ones(n, dtype=torch.long)
        self.progress_buf = torch.zeros(n, dtype=torch.long)
        self.time_out_buf = torch.zeros(n, dtype=torch.bool)
        self.root_states = torch.zeros(n, 13)
        self.root_states[:, 6] = 1
        self.obstacles = torch.zeros(n, NUM_OBSTACLES, 3)                    # x, y, yaw (z = 0)
        self.goal_positions = torch.zeros(n, 3)
        self.actions = torch.zeros(n, self.num_actions)
        self.pre_actions = torch.zeros(n, self.num_actions)
        self.cmd_thrusts = torch.zeros(n, 4)
        self.ctl_state = CascadeState(n)
        self.pre_root_positions = torch.zeros(n, 3)
        self.prev_related_dist = torch.zeros(n)
        self.collisions = torch.zeros(n)
        self.esdf_dist = torch.ones(n) * 10
        self.full_camera_array = torch.zeros(n, 1, CAM_W, CAM_H)
        self.extras = {}
        self.item_reward_info = {}
        self.reset_idx(torch.arange(n))
        self.tick += 1

    root_positions = property(lambda s: s.root_states[:, 0:3])
    root_quats = property(lambda s: s.root_states[:, 3:7])
    root_linvels = property(lambda s: s.root_states[:, 7:10])
    root_angvels = property(lambda s: s.root_states[:, 10:13])

    # ------------------------------------------------------------------ reset, planning.py:63-136
    def reset_idx(self, env_ids, uniforms=None):
        k = len(env_ids)
        if uniforms is None:
            nb = (RESET_UNIFORMS + 3) // 4
            raw = philox.raw_blocks(self.seed, self.env_ids_global[env_ids.numpy()], self.tick, philox.STREAM_RESET, nb)
            uniforms = torch.from_numpy(philox.u32_to_unit_float(raw)[:, :RESET_UNIFORMS])

        def rf(lo, hi, x):
            return (hi - lo) * x + lo
        u = uniforms[:, :3 * NUM_OBSTACLES].reshape(k, NUM_OBSTACLES, 3)
        self.obstacles[env_ids, :, 0] = LENGTH * rf(-1.0, 1.0, u[..., 0]) + 0.0
        self.obstacles[env_ids, :, 1] = WIDTH * rf(-1.0, 1.0, u[..., 1]) + 0.0
        self.obstacles[env_ids, :, 2] = rf(-torch.pi, torch.pi, u[..., 2])
        goal = torch.zeros(k, 3)
        goal[:, 0] = LENGTH + 0.5
        goal[:, 1] = 1.5 * rf(-1.0, 1.0, uniforms[:, 3 * NUM_OBSTACLES]) + 0.0
        goal[:, 2] = FLY_HEIGHT
        self.goal_positions[env_ids] = goal
        st = torch.zeros(k, 13)
        st[:, 0] = -LENGTH - 0.5
        st[:, 2] = FLY_HEIGHT
        init_yaw = torch.atan2(goal[:, 1] - st[:, 1], goal[:, 0] - st[:, 0])
        root_angle = torch.stack((torch.zeros(k), torch.zeros(k), init_yaw), -1)
        root_quats = T.matrix_to_quaternion(T.euler_angles_to_matrix(root_angle, "XYZ"))
        st[:, 3:7] = root_quats[:, [1, 2, 3, 0]]
        self.root_states[env_ids] = st
        self.reset_buf[env_ids] = 1
        self.progress_buf[env_ids] = 0
        self.pre_actions[env_ids] = 0
        self.prev_related_dist[env_ids] = 0
        self.pre_root_positions[env_ids] = 0
        self.ctl_state.reset(env_ids, self.root_states)

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs))
        self.tick += 1
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions))
        return obs, priv

    # ------------------------------------------------------------------ step, planning.py:138-184
    def pre_physics_step(self, _actions):
        """customized.py:216-298: self.actions = thrust-remapped RAW action; a clamped copy drives the controller."""
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

    def scene(self, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        return world_cylinders(self.obstacles[ids], self.variants[ids], self.table)

    def render_cameras(self, image_randoms=None):
        """render_all_camera_sensors + dump_images.  image_randoms = (add [N,W,H], mul [N,W,H], kernel [N,5,5])."""
        centre, axis, r, h = self.scene()
        npix = CAM_W * CAM_H
        for e in range(self.num_envs):
            raw = render_depth_one(self.root_positions[e], self.root_quats[e], centre[e], axis[e], r[e], h[e],
                                   self.goal_positions[e])
            if image_randoms is None:
                ids = self.env_ids_global[e:e + 1]
                add = torch.from_numpy(philox.normals(self.seed, ids, self.tick, STREAM_IMG_ADD, npix)).reshape(CAM_W, CAM_H)
                mul = torch.from_numpy(philox.normals(self.seed, ids, self.tick, STREAM_IMG_MUL, npix)).reshape(CAM_W, CAM_H)
                kraw = philox.raw_blocks(self.seed, ids, self.tick, STREAM_IMG_KERNEL, 7)[0, :25]
                ker = torch.from_numpy((kraw >> np.uint32(24)).astype(np.float32) / np.float32(256.0)).reshape(5, 5)
            else:
                add, mul, ker = image_randoms[0][e], image_randoms[1][e], image_randoms[2][e]
            self.full_camera_array[e] = post_process_depth(raw, add, mul, ker)

    def check_collisions(self):
        centre, axis, r, h = self.scene()
        d = point_capped_cylinder_distance(self.root_positions, centre, axis, r, h)
        hit = (d.min(dim=1).values <= ROBOT_RADIUS) | (self.root_positions[:, 2] <= ROBOT_RADIUS)
        self.collisions = hit.float()

    def step(self, actions, reset_uniforms=None, image_randoms=None):
        self.pre_physics_step(actions)
        self.root_states = rk4_step(self.root_states, self.fz, self.tau_b, self.dt)
        if self.counter % self.cam_rate == 0:
            self.render_cameras(image_randoms)
        self.progress_buf += 1
        self.check_collisions()
        self.compute_observations()
        self.esdf_dist = self.full_camera_array.reshape(self.num_envs, -1).min(dim=1).values
        self.compute_reward()
        reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        self.last_reset_env_ids = reset_env_ids
        if len(reset_env_ids) > 0:
            self.reset_idx(reset_env_ids, None if reset_uniforms is None else reset_uniforms[reset_env_ids])
        self.time_out_buf = self.progress_buf > self.max_episode_length
        self.extras["time_outs"] = self.time_out_buf
        self.extras["item_reward_info"] = self.item_reward_info
        self.prev_related_dist = self.related_dist
        self.tick += 1
        obs = {"image": self.full_camera_array, "observation": self.obs_buf}
        return obs, None, self.rew_buf, self.reset_buf, self.extras

    # ------------------------------------------------------------------ planning.py:186-214
    def compute_observations(self):
        forward_global = self.goal_positions - self.root_positions
        rot_matrix_global = T.quaternion_to_matrix(self.root_quats[:, [3, 0, 1, 2]])
        yaw = torch.atan2(rot_matrix_global[:, 1, 0], rot_matrix_global[:, 0, 0])
        cos_yaw, sin_yaw = torch.cos(yaw), torch.sin(yaw)
        z, o = torch.zeros_like(yaw), torch.ones_like(yaw)
        self.world_to_local = torch.stack([torch.stack([cos_yaw, -sin_yaw, z], dim=1),
                                           torch.stack([sin_yaw, cos_yaw, z], dim=1),
                                           torch.stack([z, z, o], dim=1)], dim=2)
        rot_matrix_local = torch.bmm(self.world_to_local, rot_matrix_global)
        self.euler_angles_local = T.matrix_to_euler_angles_xyz(rot_matrix_local)
        self.pos_diff_local = torch.einsum("bij,bj->bi", self.world_to_local, forward_global)
        self.vel_local = torch.einsum("bij,bj->bi", self.world_to_local, self.root_linvels)
        self.ang_vel_local = torch.einsum("bij,bj->bi", self.world_to_local, self.root_angvels)
        self.goal_dir = self.pos_diff_local / torch.norm(self.pos_diff_local, dim=-1, keepdim=True)
        self.related_dist = torch.norm(forward_global, dim=-1)
        self.obs_buf[..., 0:3] = self.goal_dir
        self.obs_buf[..., 3:6] = self.euler_angles_local
        self.obs_buf[..., 6:9] = self.vel_local
        self.obs_buf[..., 9:12] = self.ang_vel_local
        self.obs_buf[..., 12:16] = self.actions[..., :4] if self.num_actions == 4 else self.actions[..., :4]

    def compute_reward(self):
        self.rew_buf[:], self.reset_buf[:], self.item_reward_info = self.compute_quadcopter_reward()
        self.pre_actions = self.actions.clone()
        self.pre_root_positions = self.root_positions.clone()

    # ------------------------------------------------------------------ planning.py:223-307
    def compute_quadcopter_reward(self):
        action_diff = self.actions - self.pre_actions
        continous_action_reward = .2 * torch.norm(self.ang_vel_local, dim=-1) + .2 * torch.norm(action_diff, dim=-1)
        thrust_reward = .5 * (1 - torch.abs(0.1533 - self.actions[..., -1]))
        forward_reward = .1 * (torch.norm(self.goal_positions - self.pre_root_positions, dim=-1)
                               - torch.norm(self.goal_positions - self.root_positions, dim=-1))
        forward_vec = self.pos_diff_local / torch.norm(self.pos_diff_local, dim=-1, keepdim=True)
        heading_vec = torch.tensor([1.0, 0.0, 0.0]).repeat(self.num_envs, 1)
        heading_reward = torch.sum(forward_vec * heading_vec, dim=-1)
        speed_reward = -0.5 * (1 - torch.exp(- 2 * torch.square(self.vel_local[..., 0] - 1.0)))
        z_reward = torch.min(torch.min(self.root_positions[..., 2] - 1.8, torch.tensor(0.0)), 1.2 - self.root_positions[..., 2])
        ups = quat_axis(self.root_quats, axis=2)
        ups_reward = torch.square((ups[..., 2] + 1) / 2)
        esdf_reward = 0.5 * (1 - torch.exp(- 0.5 * torch.square(self.esdf_dist)))
        alive_reward = torch.where(self.esdf_dist > 0.3, torch.tensor(0.0), torch.tensor(-1.0))
        reach_goal = self.related_dist < 0.3
        reach_goal_reward = torch.where(reach_goal, torch.tensor(200.0), torch.tensor(0.0))
        reward = (continous_action_reward + forward_reward + alive_reward + esdf_reward + ups_reward + z_reward
                  + speed_reward + heading_reward + thrust_reward + reach_goal_reward)
        ones = torch.ones_like(self.reset_buf)
        die = torch.zeros_like(self.reset_buf)
        reset = torch.where(self.root_positions[..., 2] < FLY_HEIGHT - 0.3, ones, die)
        reset = torch.where(self.root_positions[..., 2] > FLY_HEIGHT + 0.3, ones, reset)
        reset = torch.where(self.root_positions[..., 0] < -LENGTH - 0.5, ones, reset)
        reset = torch.where(self.root_positions[..., 0] > LENGTH + 0.5, ones, reset)
        reset = torch.where(self.root_positions[..., 1] < -WIDTH, ones, reset)
        reset = torch.where(self.root_positions[..., 1] > WIDTH, ones, reset)
        reset = torch.where(self.collisions > 0, ones, reset)
        reset = torch.where(reach_goal, ones, reset)
        reset = torch.where(heading_reward < 0.25, ones, reset)
        reset = torch.where(self.progress_buf >= self.max_episode_length - 1, ones, reset)
        info = {"continous_action_reward": continous_action_reward, "heading_reward": heading_reward,
                "speed_reward": speed_reward, "forward_reward": forward_reward, "alive_reward": alive_reward,
                "ups_reward": ups_reward, "z_reward": z_reward, "esdf_reward": esdf_reward,
                "thrust_reward": thrust_reward, "reach_goal_reward": reach_goal_reward, "reward": reward}
        return reward, reset, info
