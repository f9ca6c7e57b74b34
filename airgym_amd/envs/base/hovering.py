"""Hovering task - host-side mirror of the reference's airgym/envs/base/hovering.py (class Hovering).

Every per-env computation of the reference's `pre_physics_step` (:203-281), `gym.simulate` (:290),
`compute_observations` (:337-358), `compute_quadcopter_reward` (:371-459) and `reset_idx` (:310-335)
happens inside ONE kernel launch (airgym_amd/csrc/step_kernel.hip); this class only keeps the
reference's Python surface: attributes, `step` return tuple, `extras` keys.
"""
import torch

from airgym_amd.envs.base.base_task import BaseTask
from airgym_amd.envs.base.hovering_config import HoveringCfg

CTL_MODES = ("pos", "vel", "atti", "rate", "prop")

ACTION_LIMITS = {   # hovering.py:93-121
    "pos": ([-3, -3, -3, -6.0], [3, 3, 3, 6.0]),
    "vel": ([-6, -6, -6, -6], [6, 6, 6, 6]),
    "atti": ([-1, -1, -1, -1, 0.], [1, 1, 1, 1, 1]),
    "rate": ([-6, -6, -6, 0], [6, 6, 6, 1]),
    "prop": ([0, 0, 0, 0], [1, 1, 1, 1]),
}


class Hovering(BaseTask):
    TASK_NAME = "hovering"
    action_limits = ACTION_LIMITS

    def __init__(self, cfg: HoveringCfg, sim_params, physics_engine, sim_device, headless):
        self.cfg = cfg
        assert cfg.env.ctl_mode is not None, "Please specify one control mode!"
        if cfg.env.ctl_mode not in CTL_MODES:
            # hovering.py:122-123 only prints "Mode Error!" and fails later; make it an error up front
            raise ValueError(f"unknown ctl_mode {cfg.env.ctl_mode!r}; options: {', '.join(CTL_MODES)}")
        self.ctl_mode = cfg.env.ctl_mode
        self.cfg.env.num_actions = 5 if cfg.env.ctl_mode == "atti" else 4
        self.max_episode_length = int(self.cfg.env.episode_length_s / self.cfg.sim.dt)
        self.debug_viz = False
        super().__init__(self.cfg, sim_params, physics_engine, sim_device, headless)

        self.privileged_obs_buf = None       # only the robot exists in Hovering/Tracking (hovering.py:79-83)
        self.counter = 0
        lo, hi = self.action_limits[self.ctl_mode]
        self.action_lower_limits = torch.tensor(lo, device=self.device, dtype=torch.float32)
        self.action_upper_limits = torch.tensor(hi, device=self.device, dtype=torch.float32)
        self.target_states = torch.tensor(self.cfg.env.target_state, device=self.device,
                                          dtype=torch.float32).repeat(self.num_envs, 1)
        self.actions = torch.zeros((self.num_envs, self.num_actions), device=self.device)
        self.item_reward_info = self.hip.reward_terms if self.hip.reward_terms is not None else {}

    # ---- state views.  The kernel keeps the state as float4 SoA; these materialise the reference's
    # AoS tensors on demand (hovering.py:73-77, :138, :164-165).
    @property
    def root_states(self):
        return self.hip.get_state()["root_states"]

    @property
    def root_positions(self):
        return self.root_states[..., 0:3]

    @property
    def root_quats(self):
        return self.root_states[..., 3:7]

    @property
    def root_linvels(self):
        return self.root_states[..., 7:10]

    @property
    def root_angvels(self):
        return self.root_states[..., 10:13]

    @property
    def progress_buf(self):
        return self.hip.get_state()["progress"].long()

    @property
    def pre_actions(self):
        return self.hip.get_state()["pre_actions"]

    @property
    def cmd_thrusts(self):
        return self.hip.cmd_thrusts

    def set_root_states(self, root_states):
        """gym.set_actor_root_state_tensor (hovering.py:331)."""
        self.hip.set_state(root_states=root_states)

    def callback(self, data):
        """hovering.py:154-156: new target state for every env."""
        ts = torch.as_tensor(getattr(data, "data", data), dtype=torch.float32)
        self.hip.set_target_state(ts.cpu().numpy())
        self.target_states = ts.to(self.device).repeat(self.num_envs, 1)

    # ---- stepping
    def step(self, actions):
        """Returns (obs_buf, privileged_obs_buf, rew_buf, reset_buf, extras) like hovering.py:286-308.
        The action tensor is read-only (the reference mutates it in place, quirk Q4)."""
        self.counter += 1
        self.actions = actions
        self.hip.step(actions)
        self.extras["time_outs"] = self.time_out_buf
        self.extras["item_reward_info"] = self.item_reward_info
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset_idx(self, env_ids):
        """hovering.py:310-335 as a host call.  The resets the reference issues from inside step() (hovering.py:211,302) happen
        in the step kernel; this entry serves BaseTask.reset() (all envs, base_task.py:109) and callers that reset a subset."""
        if len(env_ids) == self.num_envs:
            self.hip.reset_all()
        else:
            self.hip.reset_envs(env_ids)

    def compute_observations(self):
        return self.obs_buf

    def post_physics_step(self):
        return None
