"""Planning task - host-side mirror of the reference's airgym/envs/task/planning.py (class Planning on
airgym/envs/base/customized.py): fly from x = -8.5 to the goal ball at x = +8.5 through 40 tilted cylinders, seeing
a 212x120 depth image (refreshed every cam_dt/dt = 4 steps) and a 16-dim yaw-local state.

`step` returns ({'image': full_camera_array [N,1,212,120], 'observation': obs_buf [N,16]}, privileged_obs, rew,
reset, extras) like planning.py:177-184.  Everything per-env runs in the kernels of
airgym_amd/csrc/planning_kernel.hip (physics / render + post-processing / observation-reward-reset).
"""
import torch

from airgym_amd.envs.base.hovering import Hovering
from airgym_amd.envs.task.planning_scene import CAM_CHANNEL, CAM_RESOLUTION

PLANNING_ACTION_LIMITS = {   # customized.py:93-123
    "pos": ([-3, -3, -3, -6.0], [3, 3, 3, 6.0]),
    "vel": ([-6, -6, -6, -6], [6, 6, 6, 6]),
    "rate": ([-1, -1, -1, 0], [1, 1, 1, 1]),
    "prop": ([0, 0, 0, 0], [1, 1, 1, 1]),
}


class Planning(Hovering):
    TASK_NAME = "planning"
    action_limits = PLANNING_ACTION_LIMITS

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        if cfg.env.ctl_mode == "atti":
            raise ValueError("planning observes a 4-dim action (planning.py:214): ctl_mode 'atti' is not supported")
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.enable_onboard_cameras = True
        self.cam_channel = CAM_CHANNEL
        self.cam_resolution = CAM_RESOLUTION
        self.full_camera_array = self.hip.image          # [N, 1, 212, 120], written by the render kernel
        self.collisions = self.hip.collisions

    # scene state, materialised on demand from the kernel's SoA arrays
    @property
    def goal_positions(self):
        return self.hip.planning_get_state()["goal"]

    @property
    def env_asset_root_states(self):
        """Obstacle root poses as (x, y, yaw, variant) [N, 40, 4] (the reference exposes 13-dim actor states)."""
        return self.hip.planning_get_state()["obstacles"]

    @property
    def esdf_dist(self):
        return self.hip.planning_get_state()["extra"][:, 3]

    def step(self, actions):
        self.counter += 1
        self.actions = actions
        self.hip.step(actions)
        self.extras["time_outs"] = self.time_out_buf
        self.extras["item_reward_info"] = self.item_reward_info
        obs = {"image": self.full_camera_array, "observation": self.obs_buf}
        return obs, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras
