"""Avoid task - host-side mirror of the reference's airgym/envs/task/avoid.py (class Avoid on
airgym/envs/base/customized.py): hold (0, 0, 1) while a cube is thrown at the vehicle from 4.2 m (80 % of the episodes,
avoid.py:58-131); 16-dim yaw-local observation + 212x120 depth image every cam_dt/dt = 4 steps, eight reward terms, 6 s
episodes, -500 and reset on collision.  `step` returns ({'image', 'observation'}, privileged_obs, rew, reset, extras)
like avoid.py:199-206.  Everything per-env runs in custom_step_kernel<TASK_AVOID, ...> and
planning_render_kernel<SCENE_AVOID> (airgym_amd/csrc/planning_kernel.hip)."""
from airgym_amd.envs.base.hovering import Hovering
from airgym_amd.envs.task.balloon import CUSTOMIZED_ACTION_LIMITS
from airgym_amd.envs.task.planning_scene import CAM_CHANNEL, CAM_RESOLUTION


class Avoid(Hovering):
    TASK_NAME = "avoid"
    action_limits = {k: v for k, v in CUSTOMIZED_ACTION_LIMITS.items() if k != "atti"}

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        if cfg.env.ctl_mode == "atti":
            # the reference fails here too: `obs_buf[..., 12:16] = actions_local` with [N,5] actions (avoid.py:232)
            raise ValueError("avoid observes a 4-dim action (avoid.py:232): ctl_mode 'atti' is not supported")
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.enable_onboard_cameras = True
        self.cam_channel = CAM_CHANNEL
        self.cam_resolution = CAM_RESOLUTION
        self.full_camera_array = self.hip.image
        self.collisions = self.hip.collisions

    @property
    def object_positions(self):
        return self.hip.planning_get_state()["goal"]

    @property
    def object_linvels(self):
        return self.hip.planning_get_state()["object_vel"]

    def step(self, actions):
        self.counter += 1
        self.actions = actions
        self.hip.step(actions)
        self.extras["time_outs"] = self.time_out_buf
        self.extras["item_reward_info"] = self.item_reward_info
        obs = {"image": self.full_camera_array, "observation": self.obs_buf}
        return obs, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras
