"""Balloon task - host-side mirror of the reference's airgym/envs/task/balloon.py (class Balloon on
airgym/envs/base/customized.py): fly into a static red ball 2-3 m ahead; 18-dim noisy observation relative to the ball,
six reward terms, 8 s episodes, reset on collision.  `step` returns the plain observation tensor like balloon.py:141-143
(no onboard camera, balloon_config.py:52).  Everything per-env runs in custom_step_kernel<TASK_BALLOON, ...>
(airgym_amd/csrc/planning_kernel.hip, planning_math.hpp)."""
from airgym_amd.envs.base.hovering import Hovering

CUSTOMIZED_ACTION_LIMITS = {   # customized.py:93-123
    "pos": ([-3, -3, -3, -6.0], [3, 3, 3, 6.0]),
    "vel": ([-6, -6, -6, -6], [6, 6, 6, 6]),
    "atti": ([-1, -1, -1, -1, 0.], [1, 1, 1, 1, 1]),
    "rate": ([-1, -1, -1, 0], [1, 1, 1, 1]),
    "prop": ([0, 0, 0, 0], [1, 1, 1, 1]),
}


class Balloon(Hovering):
    TASK_NAME = "balloon"
    action_limits = CUSTOMIZED_ACTION_LIMITS

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.enable_onboard_cameras = False
        self.collisions = self.hip.collisions

    @property
    def balloon_positions(self):
        return self.hip.planning_get_state()["goal"]

    @property
    def pre_root_positions(self):
        return self.hip.planning_get_state()["extra"][:, 0:3]

    @property
    def privileged_obs(self):
        """env_asset_root_states of the one asset, the ball (customized.py:79-83): position + identity attitude."""
        import torch
        st = torch.zeros(self.num_envs, 1, 13, device=self.device)
        st[:, 0, 0:3] = self.balloon_positions
        st[:, 0, 6] = 1.0
        return st
