"""BaseTask - the env object the reference's callers hold (airgym/envs/base/base_task.py:38-140), backed
by one libairgym_hip.so handle instead of an IsaacGym sim.

Same constructor signature, same public attributes (num_envs, num_obs, num_actions, device, obs_buf,
rew_buf, reset_buf, time_out_buf, extras) and the same `reset()` contract: reset_idx(all) followed by
one `step(zeros)` (base_task.py:107-111).  The buffers are persistent tensors mutated in place by the
kernel, exactly like the reference's env-owned tensors (base_task.py:72-76).
"""
import torch

from airgym_amd.hip_env import HipEnvHandle


def _parse_device_str(dev):
    """isaacgym.gymutil.parse_device_str: 'cuda:1' -> ('cuda', 1), 'cpu' -> ('cpu', 0)."""
    dev = str(dev)
    if dev.startswith("cuda"):
        return "cuda", int(dev.split(":")[1]) if ":" in dev else 0
    if dev == "cpu":
        return "cpu", 0
    raise ValueError(f"invalid device string {dev!r}")


class BaseTask:
    TASK_NAME = None   # 'hovering' | 'tracking', set by subclasses

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.cfg = cfg
        self.sim_params = sim_params
        self.dt = float(getattr(sim_params, "dt", None) or cfg.sim.dt)
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        sim_device_type, self.sim_device_id = _parse_device_str(sim_device)
        self.headless = headless
        if sim_device_type != "cuda":
            # the reference silently falls back to CPU tensors (base_task.py:53-56); this build has one
            # execution path, the gfx950 kernel
            raise RuntimeError("airgym_amd runs on a HIP device only: pass --sim_device cuda:<k>")
        self.device = f"cuda:{self.sim_device_id}"
        self.graphics_device_id = self.sim_device_id

        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.get_privileged_obs = cfg.env.get_privileged_obs
        self.num_actions = cfg.env.num_actions

        self.extras = {}
        self.create_sim()

        self.obs_buf = self.hip.obs_buf              # [N, num_obs] f32
        self.rew_buf = self.hip.rew_buf              # [N] f32
        self.reset_buf = self.hip.reset_buf          # [N] int64, ones after creation (base_task.py:75)
        self.time_out_buf = self.hip.time_out_buf    # [N] bool
        assert self.obs_buf.shape == (self.num_envs, self.num_obs)

        self.enable_viewer_sync = True
        self.viewer = None

    def create_sim(self):
        """Replaces gym.create_sim + the O(N) python loop of create_env/create_actor
        (hovering.py:158-201): one arena allocation + one reset kernel."""
        seed = getattr(self.cfg, "seed", 0)
        self.hip = HipEnvHandle(
            self.TASK_NAME, self.cfg.env.ctl_mode, self.num_envs, device=self.device,
            seed=0 if seed is None or seed < 0 else seed,
            env_id_offset=getattr(self.cfg.env, "env_id_offset", 0), dt=self.dt,
            max_episode_length=int(self.cfg.env.episode_length_s / self.dt),
            target_state=self.cfg.env.target_state,
            reward_terms=getattr(self.cfg.env, "emit_reward_terms", True),
            # opt-in: extras["time_outs"] flags the envs that hit the time limit (the reference's is never true, quirk Q3)
            fix_time_outs=bool(getattr(self.cfg.env, "fix_time_outs", False)),
            # opt-in (Hovering): a full reset starts every env at its own progress, so the 2 400-step time limit does not
            # end all episodes in the same rollout (the reference starts all at 0, hovering.py:310-335)
            stagger_episode_phase=bool(getattr(self.cfg.env, "stagger_episode_phase", False)))

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        """Reset all robots: reset_idx(all) then one step with zero actions (base_task.py:107-111)."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, privileged_obs, _, _, _ = self.step(
            torch.zeros(self.num_envs, self.num_actions, device=self.device, requires_grad=False))
        return obs, privileged_obs

    def step(self, actions):
        raise NotImplementedError

    def render(self, sync_frame_time=True):
        """No renderer in the HIP path (Hovering/Tracking have no cameras); kept so callers can invoke it."""
        return None

    def close(self):
        self.hip.close()
