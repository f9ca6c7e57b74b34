"""ctypes binding of libairgym_hip.so (C ABI: include/airgym_hip.h).

The library is the ONLY execution path of the environments: if it is missing and cannot be
built, importing this module raises - there is no CPU / PyTorch fallback.
"""
import ctypes
import os

# torch must initialise its bundled HIP runtime BEFORE libairgym_hip.so is dlopen'ed: loading the
# library first binds it to /opt/rocm's libamdhip64 and the process ends up with two runtimes
# (symptom on the GPU box: hipGetDeviceCount() == 0 inside ag_create).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# AIRGYM_EXPERIMENTS=1 (tools/ only): the experiments build, which adds the ag_debug_* entry points of
# include/airgym_hip_debug.h (`python airgym_amd/csrc/build.py --experiments`); the product never sets it
EXPERIMENTS = os.environ.get("AIRGYM_EXPERIMENTS", "0") == "1"
LIB_PATH = os.path.join(_HERE, "libairgym_hip_exp.so" if EXPERIMENTS else "libairgym_hip.so")
if EXPERIMENTS and os.environ.get("AIRGYM_EXP_LIB"):      # a variant of the experiments build (build.py --experiments --tag)
    LIB_PATH = os.path.abspath(os.environ["AIRGYM_EXP_LIB"])

AG_TASKS = {"hovering": 0, "tracking": 1, "planning": 2, "balloon": 3, "avoid": 4}
AG_CTL_MODES = {"pos": 0, "vel": 1, "atti": 2, "rate": 3, "prop": 4}
AG_FLAG_REWARD_TERMS = 1 << 0
AG_FLAG_OBS_NOISE_OFF = 1 << 1
AG_FLAG_FIX_TIME_OUTS = 1 << 2
AG_FLAG_STAGGER_PHASE = 1 << 3
AG_NUM_REWARD_TERMS = 11

AG_ERR_UNKNOWN_TASK = -2
AG_ERR_UNKNOWN_CTL = -3
AG_ERR_UNSUPPORTED = -6

# order of ag_buffers.reward_terms (include/airgym_hip.h)
REWARD_TERM_NAMES = {
    "hovering": ["continous_action_reward", "effort_reward", "thrust_reward", "pos_reward", "vel_direction_reward",
                 "ups_reward", "spin_reward", "yaw_reward", "reward"],
    "tracking": ["dist_norm", "dist_reward", "yaw_reward", "spin_reward", "continous_action_reward", "thrust_reward",
                 "effort_reward", "ups_reward", "reward"],
    "planning": ["continous_action_reward", "heading_reward", "speed_reward", "forward_reward", "alive_reward",
                 "ups_reward", "z_reward", "esdf_reward", "thrust_reward", "reach_goal_reward", "reward"],
    "balloon": ["guidance_reward", "hit_reward", "action_smoothness_reward", "effort_reward", "ups_reward", "reward"],
    "avoid": ["pose_reward", "ups_reward", "spin_reward", "effort_reward", "action_smoothness_reward", "thrust_reward",
              "alive_reward", "reward"],
}


class AgConfig(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("task", ctypes.c_int32),
        ("ctl_mode", ctypes.c_int32),
        ("num_envs", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("seed", ctypes.c_uint64),
        ("env_id_offset", ctypes.c_uint32),
        ("dt", ctypes.c_double),
        ("max_episode_length", ctypes.c_int32),
        ("target_state", ctypes.c_float * 18),
    ]


class AgBuffers(ctypes.Structure):
    _fields_ = [
        ("num_envs", ctypes.c_int32),
        ("num_obs", ctypes.c_int32),
        ("num_actions", ctypes.c_int32),
        ("max_episode_length", ctypes.c_int32),
        ("obs_dev", ctypes.c_void_p),
        ("rew_dev", ctypes.c_void_p),
        ("reset_dev", ctypes.c_void_p),
        ("timeout_dev", ctypes.c_void_p),
        ("reset_mask_dev", ctypes.c_void_p),
        ("reset_ids_dev", ctypes.c_void_p),
        ("reset_count_dev", ctypes.c_void_p),
        ("reward_terms_dev", ctypes.c_void_p * AG_NUM_REWARD_TERMS),
        ("cmd_thrusts_dev", ctypes.c_void_p),
    ]


class AgPlanningBuffers(ctypes.Structure):
    _fields_ = [("image_dev", ctypes.c_void_p), ("collisions_dev", ctypes.c_void_p)]


class AgPlanningStateView(ctypes.Structure):
    _fields_ = [("obstacles_dev", ctypes.c_void_p), ("goal_dev", ctypes.c_void_p), ("extra_dev", ctypes.c_void_p),
                ("object_vel_dev", ctypes.c_void_p)]


class AgStateView(ctypes.Structure):
    _fields_ = [
        ("root_states_dev", ctypes.c_void_p),
        ("ctl_state_dev", ctypes.c_void_p),
        ("pre_actions_dev", ctypes.c_void_p),
        ("progress_dev", ctypes.c_void_p),
        ("was_reset_dev", ctypes.c_void_p),
    ]


class AgRolloutTail(ctypes.Structure):
    """ag_rollout_tail (include/airgym_hip.h)"""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("heads_dev", ctypes.c_void_p), ("logstd_dev", ctypes.c_void_p), ("vmean_dev", ctypes.c_void_p),
        ("vvar_dev", ctypes.c_void_p), ("veps", ctypes.c_float), ("seed", ctypes.c_ulonglong),
        ("counter_dev", ctypes.c_void_p), ("horizon", ctypes.c_int), ("slot", ctypes.c_int), ("id_offset", ctypes.c_longlong),
        ("actions_dev", ctypes.c_void_p), ("neglogp_dev", ctypes.c_void_p), ("values_dev", ctypes.c_void_p),
        ("mus_dev", ctypes.c_void_p), ("sigmas_dev", ctypes.c_void_p),
        ("scale", ctypes.c_float), ("shift", ctypes.c_float), ("min_val", ctypes.c_float), ("max_val", ctypes.c_float),
        ("log_val", ctypes.c_int), ("gamma", ctypes.c_float), ("bootstrap_timeouts", ctypes.c_int),
        ("shaped_dev", ctypes.c_void_p), ("cur_rew_dev", ctypes.c_void_p), ("cur_shaped_dev", ctypes.c_void_p),
        ("cur_len_dev", ctypes.c_void_p), ("partials_dev", ctypes.c_void_p),
    ]


class AgLossEpilogue(ctypes.Structure):
    """ag_loss_epilogue (include/airgym_hip.h)"""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("logstd_dev", ctypes.c_void_p), ("actions_dev", ctypes.c_void_p), ("old_neglogp_dev", ctypes.c_void_p),
        ("advantages_dev", ctypes.c_void_p), ("returns_dev", ctypes.c_void_p), ("old_values_dev", ctypes.c_void_p),
        ("old_mu_dev", ctypes.c_void_p), ("old_sigma_dev", ctypes.c_void_p), ("new_mu_dev", ctypes.c_void_p),
        ("new_sigma_dev", ctypes.c_void_p), ("heads_dev", ctypes.c_void_p), ("loss_partials_dev", ctypes.c_void_p),
        ("dwh_partials_dev", ctypes.c_void_p), ("db_partials_dev", ctypes.c_void_p),
        ("e_clip", ctypes.c_float), ("critic_coef", ctypes.c_float), ("bounds_loss_coef", ctypes.c_float),
        ("clip_value", ctypes.c_int), ("bound_type", ctypes.c_int), ("tile_rows", ctypes.c_int), ("partial_tiles", ctypes.c_int),
    ]


class AgInputLayerArgs(ctypes.Structure):
    """ag_input_layer_args (include/airgym_hip.h)"""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("D", ctypes.c_int),
        ("obs_dev", ctypes.c_void_p), ("mean_dev", ctypes.c_void_p), ("var_dev", ctypes.c_void_p),
        ("xn_dev", ctypes.c_void_p), ("h1_dev", ctypes.c_void_p),
        ("eps", ctypes.c_float), ("clip", ctypes.c_float),
    ]


# every symbol include/airgym_hip.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
class AgSumJob(ctypes.Structure):
    """ag_sum_job (include/airgym_hip.h)"""
    _fields_ = [("partials_dev", ctypes.c_void_p), ("out_dev", ctypes.c_void_p), ("rows", ctypes.c_int), ("n", ctypes.c_int)]


SYMBOLS = [
    ("ag_version", ctypes.c_int, []),
    ("ag_last_error", ctypes.c_char_p, []),
    ("ag_num_obs", ctypes.c_int, [ctypes.c_int]),
    ("ag_num_actions", ctypes.c_int, [ctypes.c_int]),
    ("ag_default_episode_length", ctypes.c_int, [ctypes.c_int, ctypes.c_double]),
    ("ag_arena_bytes", ctypes.c_size_t, [ctypes.POINTER(AgConfig)]),
    ("ag_create", ctypes.c_int, [ctypes.POINTER(AgConfig), _P, ctypes.POINTER(_P)]),
    ("ag_destroy", ctypes.c_int, [_P]),
    ("ag_reset_all", ctypes.c_int, [_P, _P]),
    ("ag_reset_envs", ctypes.c_int, [_P, _P, ctypes.c_int, _P]),
    ("ag_step", ctypes.c_int, [_P, _P, _P]),
    ("ag_step_into", ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    ("ag_step_with_inputs", ctypes.c_int, [_P, _P, _P, _P, _P]),
    ("ag_term_sum_tiles", ctypes.c_int, [ctypes.c_int]),
    ("ag_step_rollout", ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("ag_step_multi", ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, _P, _P, _P, _P]),
    ("ag_step_rollout_fused", ctypes.c_int, [_P, ctypes.POINTER(AgRolloutTail), _P, _P, _P, _P, _P]),
    ("ag_eval_obs_reward", ctypes.c_int, [_P, _P, _P, _P, _P]),
    ("ag_get_buffers", ctypes.c_int, [_P, ctypes.POINTER(AgBuffers)]),
    ("ag_get_state", ctypes.c_int, [_P, ctypes.POINTER(AgStateView), _P]),
    ("ag_set_state", ctypes.c_int, [_P, ctypes.POINTER(AgStateView), _P]),
    ("ag_compact_reset_ids", ctypes.c_int, [_P, _P]),
    ("ag_set_target_state", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_float)]),
    ("ag_get_tick", ctypes.c_uint64, [_P]),
    ("ag_set_tick", ctypes.c_int, [_P, ctypes.c_uint64]),
    ("ag_planning_set_obstacle_table", ctypes.c_int, [_P, _P, ctypes.c_int]),
    ("ag_planning_get_buffers", ctypes.c_int, [_P, ctypes.POINTER(AgPlanningBuffers)]),
    ("ag_planning_get_state", ctypes.c_int, [_P, ctypes.POINTER(AgPlanningStateView), _P]),
    ("ag_planning_set_state", ctypes.c_int, [_P, ctypes.POINTER(AgPlanningStateView), _P]),
    ("ag_planning_step_with_uniforms", ctypes.c_int, [_P, _P, _P, _P]),
    ("ag_planning_eval_post", ctypes.c_int, [_P, _P, _P, _P, _P]),
    ("ag_planning_render_now", ctypes.c_int, [_P]),
    ("ag_planning_last_step_rendered", ctypes.c_int, [_P]),
    ("ag_ppo_loss_finalize", ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P]),
    ("ag_rms_scratch_doubles", ctypes.c_longlong, [ctypes.c_int]),
    ("ag_rms_update", ctypes.c_int, [_P, ctypes.c_longlong, ctypes.c_int, _P, _P, _P, _P, _P]),
    ("ag_normalize_rows", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_longlong, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_float, _P]),
    ("ag_policy_sample", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_float, ctypes.c_ulonglong, _P, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_longlong, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_rollout_account_blocks", ctypes.c_int, [ctypes.c_int]),
    ("ag_rollout_account", ctypes.c_int, [_P, _P, _P, _P] + [ctypes.c_float] * 4 + [ctypes.c_int, ctypes.c_float,
                                          _P, _P, _P, _P, _P, ctypes.c_int, _P]),
    ("ag_gae", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_mlp_input_layer", ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, _P]),
    ("ag_elu_heads", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    ("ag_split_gemm_plane_bytes", ctypes.c_longlong, []),
    ("ag_split_gemm_prepare", ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_gemm_prepare_pair", ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_gemm", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_wgrad_slices", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_wgrad", ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_wgrad_input_supported", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_wgrad_input_slices", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_wgrad_input", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_gemm_input_wgrad_recompute_supported", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_gemm_input_wgrad_recompute", ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_gemm_pick_tile_rows", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_gemm_input_wgrad_rows", ctypes.c_int, []),
    ("ag_split_gemm_input_wgrad_supported", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_gemm_input_wgrad", ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, _P]),
    ("ag_split_gemm_elu_heads", ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, _P]),
    ("ag_split_gemm_loss_rows", ctypes.c_int, []),
    ("ag_split_gemm_loss_heads_bwd", ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.POINTER(AgLossEpilogue), ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_split_gemm_input_fwd_supported", ctypes.c_int, [ctypes.c_int]),
    ("ag_split_gemm_input_image_bytes", ctypes.c_longlong, []),
    ("ag_split_gemm_input_prepare", ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, _P]),
    ("ag_split_gemm_input_prepare_pair", ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, _P, _P]),
    ("ag_split_gemm_input_loss_heads_bwd", ctypes.c_int, [ctypes.POINTER(AgInputLayerArgs), _P, _P, _P, _P, _P,
                                                          ctypes.POINTER(AgLossEpilogue), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                          ctypes.c_int, _P]),
    ("ag_mlp_first_layer_supported", ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    ("ag_mlp_first_layer_image_bytes", ctypes.c_longlong, [ctypes.c_int]),
    ("ag_mlp_first_layer_prepare", ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P]),
    ("ag_mlp_first_layer", ctypes.c_int, [_P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_mlp_chain_supported", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("ag_mlp_chain_image_bytes", ctypes.c_longlong, [ctypes.c_int]),
    ("ag_mlp_chain_prepare", ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, ctypes.c_int, _P, _P]),
    ("ag_mlp_chain_forward", ctypes.c_int, [_P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_relu_bn_planes_per_block", ctypes.c_int, []),
    ("ag_relu_bn_stats", ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_relu_bn_apply", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_relu_bn_bwd_reduce", ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_relu_bn_bwd_dx", ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_relu_bn_stats_weighted", ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_relu_bn_bwd_dx_weighted", ctypes.c_int, [_P] * 8 + [ctypes.c_int] * 4 + [_P]),
    ("ag_relu_bn_bwd_dx_plane", ctypes.c_int, [_P] * 8 + [ctypes.c_int] * 4 + [_P]),
    ("ag_plane_border_sums", ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_bn_sums_from_conv", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P]),
    ("ag_bn_scratch_doubles", ctypes.c_longlong, []),
    ("ag_bn_finalize", ctypes.c_int, [_P, _P, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_double, _P, _P, _P, _P, _P, ctypes.c_float, ctypes.c_double, ctypes.c_int, _P, _P, _P, ctypes.c_int, _P, _P]),
    ("ag_bn_bwd_prep", ctypes.c_int, [_P, ctypes.c_longlong, ctypes.c_int, _P, _P, ctypes.c_double, ctypes.c_int, _P, _P, _P, _P, _P, _P]),
    ("ag_bn_pool_bwd_prep", ctypes.c_int, [_P, _P, ctypes.c_longlong, ctypes.c_int, _P, _P, ctypes.c_double, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P]),
    ("ag_cnn_conv_workspace_floats", ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    ("ag_weighted_moments_chunks", ctypes.c_int, []),
    ("ag_weighted_moments", ctypes.c_int, [_P, _P, _P, ctypes.c_longlong, ctypes.c_longlong, _P, _P]),
    ("ag_cnn_conv1_fwd", ctypes.c_int, [_P] * 8 + [ctypes.c_int, _P, _P]),
    ("ag_cnn_conv1_wgrad_partials", ctypes.c_int, [ctypes.c_int]),
    ("ag_cnn_conv1_wgrad", ctypes.c_int, [_P] * 9 + [ctypes.c_int, _P]),
    ("ag_cnn_conv_supported", ctypes.c_int, [ctypes.c_int] * 4),
    ("ag_cnn_conv_fwd_bands", ctypes.c_int, [ctypes.c_int] * 4),
    ("ag_cnn_conv_fwd", ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 5 + [_P, _P]),
    ("ag_cnn_conv_fwd_split_bands", ctypes.c_int, [ctypes.c_int] * 4),
    ("ag_cnn_conv_fwd_split", ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 5 + [_P, _P]),
    ("ag_cnn_conv_dgrad", ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P, _P]),
    ("ag_cnn_conv_dgrad_bn_rows", ctypes.c_int, [ctypes.c_int] * 5),
    ("ag_cnn_conv_dgrad_bn", ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 5 + [_P, _P]),
    ("ag_cnn_conv_dgrad_conv1_wgrad_partials", ctypes.c_int, [ctypes.c_int]),
    ("ag_cnn_conv_dgrad_conv1_wgrad", ctypes.c_int, [_P] * 10 + [ctypes.c_int, _P, _P]),
    ("ag_cnn_conv_wgrad_partials", ctypes.c_int, [ctypes.c_int] * 5),
    ("ag_cnn_conv_wgrad", ctypes.c_int, [_P] * 5 + [ctypes.c_int] * 6 + [_P]),
    ("ag_wgrad_rows_per_block", ctypes.c_int, [ctypes.c_int]),
    ("ag_input_wgrad_rows", ctypes.c_int, [ctypes.c_int]),
    ("ag_sum_rows_groups", ctypes.c_int, []),
    ("ag_sum_rows_multi", ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_longlong, _P]),
    ("ag_sum_rows_multi_finalize", ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_longlong, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P,
                                                  ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P]),
    ("ag_heads_bwd_elu_wgrad", ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    ("ag_elu_bwd_input_wgrad", ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_elu_bwd_bias_rows_per_block", ctypes.c_int, []),
    ("ag_elu_bwd_bias", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    ("ag_adam_clip_step", ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int] + [ctypes.c_float] * 8 + [_P]),
    ("ag_adam_state_bytes", ctypes.c_int, []),
    ("ag_ppo_loss_num_sums", ctypes.c_int, []),
    ("ag_ppo_loss_max_blocks", ctypes.c_int, []),
    ("ag_ppo_loss", ctypes.c_int, [_P] * 9 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, ctypes.POINTER(ctypes.c_int), _P]),
]

# include/airgym_hip_debug.h: present in the experiments build only
DEBUG_SYMBOLS = [
    ("ag_debug_touch", ctypes.c_int, [_P, _P, _P]),
    ("ag_debug_wave_placement", ctypes.c_int, [_P, _P, _P]),
    ("ag_debug_touch_variant", ctypes.c_int, [_P, _P, ctypes.c_int, _P]),
    ("ag_debug_planning_render_parts", ctypes.c_int, [_P, ctypes.c_int]),
    ("ag_debug_split_gemm_variant", ctypes.c_int, [ctypes.c_int]),
    ("ag_debug_split_wgrad_ordered", ctypes.c_int, [ctypes.c_int]),
    ("ag_debug_chain_skip", ctypes.c_int, [ctypes.c_int]),
]

_lib = None


def _build():
    from airgym_amd.csrc import build as _b
    return _b.build(verbose=False, experiments=EXPERIMENTS)


# mixed_precision twins (one bf16 MFMA per product; csrc/split_common.hpp AG_SPLIT_PLANES = 1): same signatures, suffix _bf16
BF16_TWINS = ("ag_split_gemm", "ag_split_wgrad", "ag_split_wgrad_input", "ag_split_gemm_input_wgrad_recompute",
              "ag_split_gemm_input_wgrad", "ag_split_gemm_elu_heads", "ag_split_gemm_loss_heads_bwd",
              "ag_split_gemm_input_loss_heads_bwd", "ag_mlp_chain_forward", "ag_mlp_first_layer")
SYMBOLS = SYMBOLS + [(n + "_bf16", r, a) for (n, r, a) in SYMBOLS if n in BF16_TWINS]


def load(rebuild_if_missing=True):
    """Load (building first if needed) libairgym_hip.so.  Raises RuntimeError when the HIP library is
    unavailable - the environments never fall back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not rebuild_if_missing:
            raise RuntimeError(f"{LIB_PATH} is missing; run `python airgym_amd/csrc/build.py`")
        try:
            _build()
        except Exception as e:  # hipcc missing or compile error
            raise RuntimeError(
                f"libairgym_hip.so is missing and could not be built with hipcc ({e}). "
                "airgym_amd has no CPU fallback: build it with `python airgym_amd/csrc/build.py`.") from e
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, res, args in SYMBOLS + (DEBUG_SYMBOLS if EXPERIMENTS else []):
        fn = getattr(lib, name)  # AttributeError = the .so is stale w.r.t. the header
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().ag_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    """Map ag_status to the exceptions the reference raises (task_registry.py:78-79 ValueError)."""
    if rc == 0:
        return
    msg = f"{what}: {last_error()} (ag_status {rc})"
    if rc in (AG_ERR_UNKNOWN_TASK, AG_ERR_UNKNOWN_CTL, -1):
        raise ValueError(msg)
    raise RuntimeError(msg)
