"""HoveringCfg - same field names and defaults as the reference's airgym/envs/base/hovering_config.py:8-70,
declared as a spec (see base_config.make_config_class).  Only env.* and sim.dt reach the HIP kernel;
viewer / physx / asset_config are carried so that code reading them keeps working."""
import numpy as np

from .base_config import Section, make_config_class

IDENTITY_TARGET = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1] + [0] * 9)   # attitude I3, position/velocities zero

ROBOT_X152B = {
    "num_assets": 1, "enable_onboard_cameras": False, "cam_channel": 1, "enable_tensors": False,
    "width": 212, "height": 120, "far_plane": 5.0, "horizontal_fov": 87.0, "use_collision_geometry": True,
    "local_transform.p": (0.15, 0.00, 0.1), "local_transform.r": (0.0, 0.0, 0.0, 1.0), "collision_mask": 1,
}


def physx_section(contact_collection):
    return Section(num_threads=10, solver_type=1, num_position_iterations=4, num_velocity_iterations=0,
                   contact_offset=0.01, rest_offset=0.0, bounce_threshold_velocity=0.5,
                   max_depenetration_velocity=1.0, max_gpu_contact_pairs=2 ** 23, default_buffer_size_multiplier=5,
                   contact_collection=contact_collection)


def common_sections(contact_collection, robot):
    return dict(
        viewer=Section(ref_env=0, pos=[-5, -5, 4], lookat=[0, 0, 0]),
        sim=Section(dt=0.01, substeps=1, gravity=[0., 0., -9.81], up_axis=1, physx=physx_section(contact_collection)),
        asset_config=Section(include_robot={"X152b": robot}, include_single_asset={}, include_group_asset={},
                             include_boundary={}),
    )


HoveringCfg = make_config_class("HoveringCfg", dict(
    seed=-1,
    env=Section(target_state=IDENTITY_TARGET, num_envs=256, num_observations=18, get_privileged_obs=True,
                env_spacing=1, episode_length_s=24, num_control_steps_per_env_step=1, reset_on_collision=False,
                create_ground_plane=False),
    **common_sections(0, ROBOT_X152B)), __doc__)
