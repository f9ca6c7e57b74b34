"""BalloonCfg - same field names and defaults as the reference's airgym/envs/task/balloon_config.py:7-77
(18 observations, 8 s episodes, reset on collision, ground plane, one static target ball, no onboard camera)."""
from airgym_amd.envs.base.base_config import Section, make_config_class
from airgym_amd.envs.base.hovering_config import IDENTITY_TARGET, ROBOT_X152B, common_sections

_sections = common_sections(1, dict(ROBOT_X152B, enable_onboard_cameras=False, enable_tensors=True))
_sections["asset_config"].fields["include_single_asset"] = {"balls/ball": {"color": [255, 102, 102], "num_assets": 1}}
_sections["asset_config"].fields["include_group_asset"] = {}
_sections["asset_config"].fields["include_boundary"] = {}

BalloonCfg = make_config_class("BalloonCfg", dict(
    seed=-1,
    env=Section(target_state=IDENTITY_TARGET, num_envs=4, num_observations=18, headless=True, get_privileged_obs=True,
                env_spacing=10, episode_length_s=8, num_control_steps_per_env_step=1, reset_on_collision=True,
                create_ground_plane=True, cam_dt=0.04),
    **_sections), __doc__)
