"""Tracking task - mirror of the reference's airgym/envs/task/tracking.py (class Tracking(Hovering)):
48 observations (18 + 10 lemniscate look-ahead points), 36 s episodes, reset when > 1 m off the
reference (tracking.py:159-296).  Same kernel, TASK = tracking."""
import torch

from airgym_amd.envs.base.hovering import ACTION_LIMITS, Hovering

TRACKING_ACTION_LIMITS = dict(ACTION_LIMITS)
TRACKING_ACTION_LIMITS["pos"] = ([-6, -6, -6, -6.0], [6, 6, 6, 6.0])   # tracking.py:95-99


class Tracking(Hovering):
    TASK_NAME = "tracking"
    action_limits = TRACKING_ACTION_LIMITS

    def compute_traj_lemniscate(self, n_steps=10, step_size=5, scale=0.25):
        """tracking.py:194-200, evaluated on demand from the kernel's progress counters."""
        progress = self.progress_buf
        step = progress.unsqueeze(1).expand(-1, n_steps) \
            + torch.arange(n_steps, device=self.device).repeat(self.num_envs, 1) * step_size
        t = step * self.dt * scale
        ref_x = 3 * torch.sin(t) / (1 + torch.cos(t) ** 2)
        ref_y = 3 * torch.sin(t) * torch.cos(t) / (1 + torch.cos(t) ** 2)
        return torch.stack((ref_x, ref_y, torch.ones_like(ref_x)), dim=-1)

    @property
    def ref_positions(self):
        return self.compute_traj_lemniscate()
