"""PlanningCfg - same field names and defaults as the reference's airgym/envs/task/planning_config.py:7-80
(16 observations + 212x120 depth image, 16 s episodes, cam_dt 0.04, goal ball + 40 'thin' obstacles)."""
from airgym_amd.envs.base.base_config import Section, make_config_class
from airgym_amd.envs.base.hovering_config import IDENTITY_TARGET, ROBOT_X152B, common_sections

_sections = common_sections(1, dict(ROBOT_X152B, enable_onboard_cameras=True, enable_tensors=True))
_sections["asset_config"].fields["include_single_asset"] = {"balls/ball": {"color": [255, 102, 102], "num_assets": 1}}
_sections["asset_config"].fields["include_group_asset"] = {"thin": {"num_assets": 40, "collision_mask": 1, "color": [139, 69, 0]}}
_sections["asset_config"].fields["include_boundary"] = {}

PlanningCfg = make_config_class("PlanningCfg", dict(
    seed=-1,
    env=Section(target_state=IDENTITY_TARGET, num_envs=4, num_observations=16, headless=True, get_privileged_obs=True,
                env_spacing=14, episode_length_s=16, num_control_steps_per_env_step=1, reset_on_collision=False,
                create_ground_plane=True, cam_dt=0.04),
    **_sections), __doc__)
