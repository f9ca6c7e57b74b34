"""AvoidCfg - same field names and defaults as the reference's airgym/envs/task/avoid_config.py:7-90
(16 observations + 212x120 depth image, 6 s episodes, reset on collision, hover target (0, 0, 1), one thrown 1x1 cube)."""
import numpy as np

from airgym_amd.envs.base.base_config import Section, make_config_class
from airgym_amd.envs.base.hovering_config import ROBOT_X152B, common_sections

AVOID_TARGET = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0])     # avoid_config.py:11

_sections = common_sections(1, dict(ROBOT_X152B, enable_onboard_cameras=True, enable_tensors=True))
_sections["asset_config"].fields["include_single_asset"] = {
    "cubes/1x1": {"collision_mask": 0, "num_assets": 1, "density": 0.5, "fix_base_link": False},
    "balls/ball": {"disable_gravity": False, "color": [255, 102, 102], "collision_mask": 0, "num_assets": 0, "density": 1,
                   "fix_base_link": False},
}
_sections["asset_config"].fields["include_group_asset"] = {}
_sections["asset_config"].fields["include_boundary"] = {}

AvoidCfg = make_config_class("AvoidCfg", dict(
    seed=-1,
    env=Section(target_state=AVOID_TARGET, num_envs=4, num_observations=16, headless=True, get_privileged_obs=True,
                env_spacing=4, episode_length_s=6, num_control_steps_per_env_step=1, reset_on_collision=True,
                create_ground_plane=True, cam_dt=0.04),
    **_sections), __doc__)
