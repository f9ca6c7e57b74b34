"""TrackingCfg - field names and defaults of the reference's airgym/envs/task/tracking_config.py:7-70."""
import numpy as np

from airgym_amd.envs.base.base_config import BaseConfig
from airgym_amd.envs.base.hovering_config import HoveringCfg


class TrackingCfg(BaseConfig):
    seed = -1

    class env:
        target_state = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
        num_envs = 4
        num_observations = 18 + 30
        headless = True
        get_privileged_obs = True
        env_spacing = 10
        episode_length_s = 36
        num_control_steps_per_env_step = 1
        reset_on_collision = False
        create_ground_plane = True
        cam_dt = 0.04

    class viewer(HoveringCfg.viewer):
        pass

    class sim:
        dt = 0.01
        substeps = 1
        gravity = [0., 0., -9.81]
        up_axis = 1

        class physx(HoveringCfg.sim.physx):
            contact_collection = 1

    class asset_config(HoveringCfg.asset_config):
        pass
