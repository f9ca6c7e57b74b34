"""HipEnvHandle: owns one libairgym_hip.so handle and exposes its device buffers as torch tensors
(zero copy).  This is the only place where Python touches the C ABI; the task classes
(`airgym_amd/envs/...`) and the PPO runner build on it.

Memory: the arena is a torch uint8 CUDA tensor whose pointer is handed to `ag_create`, so every
buffer the kernel writes (obs_buf, rew_buf, reset_buf, ...) is a *view* of that tensor - the same
ownership rule as the reference (env-owned persistent tensors mutated in place, base_task.py:72-76).
"""
import ctypes

import numpy as np
import torch

from . import _native as N


class HipEnvHandle:
    def __init__(self, task, ctl_mode, num_envs, device="cuda:0", seed=0, env_id_offset=0, dt=0.01,
                 max_episode_length=0, target_state=None, reward_terms=True, obs_noise=True, fix_time_outs=False,
                 stagger_episode_phase=False):
        if task not in N.AG_TASKS:
            raise ValueError(f"Task with name: {task} was not registered")
        if ctl_mode not in N.AG_CTL_MODES:
            raise ValueError(f"unknown ctl_mode {ctl_mode!r}; options: pos, vel, atti, rate, prop")
        self.lib = N.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"airgym_amd environments run on a HIP device only (got device={device!r})")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: airgym_amd has no CPU fallback")
        self.task = task
        self.ctl_mode = ctl_mode
        self.num_envs = int(num_envs)
        cfg = N.AgConfig()
        cfg.struct_size = ctypes.sizeof(N.AgConfig)
        cfg.task = N.AG_TASKS[task]
        cfg.ctl_mode = N.AG_CTL_MODES[ctl_mode]
        cfg.num_envs = self.num_envs
        cfg.device = self.device.index or 0
        cfg.flags = ((N.AG_FLAG_REWARD_TERMS if reward_terms else 0) | (0 if obs_noise else N.AG_FLAG_OBS_NOISE_OFF)
                     | (N.AG_FLAG_FIX_TIME_OUTS if fix_time_outs else 0)
                     | (N.AG_FLAG_STAGGER_PHASE if stagger_episode_phase else 0))
        if stagger_episode_phase and task != "hovering":
            raise ValueError("stagger_episode_phase: Hovering only")
        self.stagger_episode_phase = bool(stagger_episode_phase)
        if fix_time_outs and task not in ("hovering", "tracking"):
            raise ValueError("fix_time_outs: Hovering / Tracking only")
        self.fix_time_outs = bool(fix_time_outs)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.env_id_offset = int(env_id_offset)
        cfg.dt = float(dt)
        cfg.max_episode_length = int(max_episode_length)
        if target_state is None:
            target_state = [1, 0, 0, 0, 1, 0, 0, 0, 1] + [0] * 9
            if task == "avoid":
                target_state[11] = 1            # avoid_config.py:11: hold (0, 0, 1)
        ts = np.asarray(target_state, dtype=np.float32).reshape(18)
        for i in range(18):
            cfg.target_state[i] = float(ts[i])
        self.cfg = cfg
        nbytes = self.lib.ag_arena_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            N.check(-1, "ag_arena_bytes")
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
            h = ctypes.c_void_p()
            N.check(self.lib.ag_create(ctypes.byref(cfg), self.arena.data_ptr(), ctypes.byref(h)), "ag_create")
        self.h = h
        b = N.AgBuffers()
        N.check(self.lib.ag_get_buffers(self.h, ctypes.byref(b)), "ag_get_buffers")
        self.num_obs, self.num_actions, self.max_episode_length = b.num_obs, b.num_actions, b.max_episode_length
        n = self.num_envs
        self.obs_buf = self._view(b.obs_dev, torch.float32, (n, self.num_obs))
        self.rew_buf = self._view(b.rew_dev, torch.float32, (n,))
        self.reset_buf = self._view(b.reset_dev, torch.int64, (n,))
        self.time_out_buf = self._view(b.timeout_dev, torch.uint8, (n,)).view(torch.bool)
        self.reset_mask = self._view(b.reset_mask_dev, torch.int64, ((n + 63) // 64,))
        self.reset_ids = self._view(b.reset_ids_dev, torch.int32, (n,))
        self.reset_count = self._view(b.reset_count_dev, torch.int32, (1,))
        self.image = None
        self.collisions = None
        if task == "planning":
            from airgym_amd.envs.task.planning_scene import load_variant_table
            table = load_variant_table()
            N.check(self.lib.ag_planning_set_obstacle_table(self.h, table.ctypes.data_as(ctypes.c_void_p), table.shape[0]),
                    "ag_planning_set_obstacle_table")
            with torch.cuda.device(self.device):
                N.check(self.lib.ag_reset_all(self.h, None), "ag_reset_all")
                torch.cuda.synchronize(self.device)
        if task in ("planning", "balloon", "avoid"):
            pb = N.AgPlanningBuffers()
            N.check(self.lib.ag_planning_get_buffers(self.h, ctypes.byref(pb)), "ag_planning_get_buffers")
            if pb.image_dev:
                self.image = self._view(pb.image_dev, torch.float32, (n, 1, 212, 120))
            self.collisions = self._view(pb.collisions_dev, torch.float32, (n,))
        self.reward_terms = None
        self.reward_terms_stacked = None
        self.cmd_thrusts = None
        if reward_terms:
            self.reward_terms = {
                name: self._view(b.reward_terms_dev[i], torch.float32, (n,))
                for i, name in enumerate(N.REWARD_TERM_NAMES[task])
            }
            if b.cmd_thrusts_dev:
                self.cmd_thrusts = self._view(b.cmd_thrusts_dev, torch.float32, (n, 4))
            # the term arrays are consecutive padded slices of the arena: one [T, n_pad] view lets callers reduce all
            # of them with a single kernel (padding lanes are never written and stay zero)
            names = N.REWARD_TERM_NAMES[task]
            p0, p1 = int(b.reward_terms_dev[0]), int(b.reward_terms_dev[1])
            n_pad = (p1 - p0) // 4
            if all(int(b.reward_terms_dev[i]) == p0 + i * n_pad * 4 for i in range(len(names))):
                self.reward_terms_stacked = self._view(b.reward_terms_dev[0], torch.float32, (len(names), n_pad))

    # ------------------------------------------------------------------ helpers
    def _view(self, ptr, dtype, shape):
        off = int(ptr) - self.arena.data_ptr()
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        assert 0 <= off and off + nbytes <= self.arena.numel(), "buffer outside the arena"
        return self.arena[off:off + nbytes].view(dtype).view(*shape)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_actions(self, actions):
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.shape != (self.num_envs, self.num_actions):
            raise ValueError(f"actions must be [{self.num_envs}, {self.num_actions}], got {tuple(actions.shape)}")
        return actions

    # ---------------------------------------------------------------------- API
    def reset_all(self):
        N.check(self.lib.ag_reset_all(self.h, self._stream()), "ag_reset_all")

    def reset_envs(self, env_ids):
        """reset_idx(env_ids) for a subset (ag_reset_envs): env_ids = integer tensor / sequence of local env ids."""
        ids = torch.as_tensor(env_ids).to(device=self.device, dtype=torch.int32).contiguous().view(-1)
        if ids.numel() == 0:
            return
        N.check(self.lib.ag_reset_envs(self.h, ids.data_ptr(), int(ids.numel()), self._stream()), "ag_reset_envs")
        torch.cuda.current_stream(self.device).synchronize()      # `ids` must outlive the kernel

    def step(self, actions):
        actions = self._check_actions(actions)
        N.check(self.lib.ag_step(self.h, actions.data_ptr(), self._stream()), "ag_step")

    def step_into(self, actions, obs_out=None, rew_out=None, reset_out=None):
        """Write obs / reward / done straight into caller tensors (e.g. slot t of a rollout buffer)."""
        actions = self._check_actions(actions)

        def ptr(t, dtype, numel):
            if t is None:
                return None
            assert t.is_contiguous() and t.dtype == dtype and t.numel() == numel and t.device == self.device
            return t.data_ptr()
        N.check(self.lib.ag_step_into(self.h, actions.data_ptr(),
                                      ptr(obs_out, torch.float32, self.num_envs * self.num_obs),
                                      ptr(rew_out, torch.float32, self.num_envs),
                                      ptr(reset_out, torch.int64, self.num_envs), self._stream()), "ag_step_into")

    def step_rollout(self, actions, obs_out, rew_out, done_out, term_sums=None):
        """Rollout form: obs / reward into rollout slots, done flags as uint8, per-tile reward-term sums
        [ceil(n/64), 12] instead of the per-env term arrays (ag_step_rollout)."""
        actions = self._check_actions(actions)
        n = self.num_envs
        assert obs_out.is_contiguous() and obs_out.dtype == torch.float32 and obs_out.numel() == n * self.num_obs
        assert rew_out.is_contiguous() and rew_out.dtype == torch.float32 and rew_out.numel() == n
        assert done_out.is_contiguous() and done_out.dtype == torch.uint8 and done_out.numel() == n
        assert obs_out.device == self.device and rew_out.device == self.device and done_out.device == self.device
        tp = None
        if term_sums is not None:
            assert term_sums.is_contiguous() and term_sums.dtype == torch.float32 and term_sums.device == self.device
            assert term_sums.numel() == self.lib.ag_term_sum_tiles(n) * 12
            tp = term_sums.data_ptr()
        N.check(self.lib.ag_step_rollout(self.h, actions.data_ptr(), obs_out.data_ptr(), rew_out.data_ptr(),
                                         done_out.data_ptr(), tp, self._stream()), "ag_step_rollout")

    def step_multi(self, actions, obs_out, rew_out, done_out, timeout_out=None, term_sums=None):
        """K = actions.shape[0] consecutive env steps in ONE launch (ag_step_multi): identical, bit for bit, to K calls of
        step_rollout(actions[t], obs_out[t], rew_out[t], done_out[t], term_sums[t]); the state stays in registers between
        the steps.  actions [K, n, A]; obs_out [K, n, num_obs]; rew_out [K, n]; done_out [K, n] u8; timeout_out [K, n] u8 or
        None; term_sums [K, ceil(n/64), 12] or None."""
        n, A = self.num_envs, self.num_actions
        assert actions.dim() == 3 and tuple(actions.shape[1:]) == (n, A), f"actions must be [K, {n}, {A}]"
        K = actions.shape[0]
        assert actions.is_contiguous() and actions.dtype == torch.float32 and actions.device == self.device

        def chk(t, dtype, numel):
            assert t.is_contiguous() and t.dtype == dtype and t.device == self.device and t.numel() == numel, \
                (tuple(t.shape), t.dtype, numel)
            return t.data_ptr()
        tiles = self.lib.ag_term_sum_tiles(n)
        N.check(self.lib.ag_step_multi(self.h, actions.data_ptr(), K, chk(obs_out, torch.float32, K * n * self.num_obs),
                                       chk(rew_out, torch.float32, K * n), chk(done_out, torch.uint8, K * n),
                                       chk(timeout_out, torch.uint8, K * n) if timeout_out is not None else None,
                                       chk(term_sums, torch.float32, K * tiles * 12) if term_sums is not None else None,
                                       self._stream()), "ag_step_multi")

    def step_rollout_fused(self, tail, obs_out, rew_out, done_out, term_sums=None):
        """One rollout step in ONE launch: policy sampling + env step + reward shaping / episode accounting
        (ag_step_rollout_fused).  `tail` is a filled N.AgRolloutTail; the other arguments as step_rollout."""
        n = self.num_envs
        assert obs_out.is_contiguous() and obs_out.dtype == torch.float32 and obs_out.numel() == n * self.num_obs
        assert rew_out.is_contiguous() and rew_out.dtype == torch.float32 and rew_out.numel() == n
        assert done_out.is_contiguous() and done_out.dtype == torch.uint8 and done_out.numel() == n
        assert obs_out.device == self.device and rew_out.device == self.device and done_out.device == self.device
        tp = None
        if term_sums is not None:
            assert term_sums.is_contiguous() and term_sums.dtype == torch.float32 and term_sums.device == self.device
            assert term_sums.numel() == self.lib.ag_term_sum_tiles(n) * 12
            tp = term_sums.data_ptr()
        N.check(self.lib.ag_step_rollout_fused(self.h, ctypes.byref(tail), obs_out.data_ptr(), rew_out.data_ptr(),
                                               done_out.data_ptr(), tp, self._stream()), "ag_step_rollout_fused")

    def eval_obs_reward(self, processed_actions, cmd_thrusts, noise=None):
        """compute_observations + compute_quadcopter_reward on the CURRENT state with the reference-recorded inputs
        (ag_eval_obs_reward; parity tests).  Results land in obs_buf / rew_buf / reset_buf / reward_terms."""
        n = self.num_envs
        a = processed_actions.to(device=self.device, dtype=torch.float32).contiguous()
        c = cmd_thrusts.to(device=self.device, dtype=torch.float32).contiguous()
        assert a.shape == (n, self.num_actions) and c.shape == (n, 4)
        z = None
        if noise is not None:
            z = noise.to(device=self.device, dtype=torch.float32).contiguous()
            assert z.shape == (n, 18)
        N.check(self.lib.ag_eval_obs_reward(self.h, a.data_ptr(), c.data_ptr(), z.data_ptr() if z is not None else None,
                                            self._stream()), "ag_eval_obs_reward")
        torch.cuda.current_stream(self.device).synchronize()

    def step_with_inputs(self, actions, noise, reset_uniforms):
        actions = self._check_actions(actions)
        noise = noise.to(device=self.device, dtype=torch.float32).contiguous()
        reset_uniforms = reset_uniforms.to(device=self.device, dtype=torch.float32).contiguous()
        assert noise.shape == (self.num_envs, 18) and reset_uniforms.shape == (self.num_envs, self.RESET_UNIFORMS.get(self.task, 12))
        N.check(self.lib.ag_step_with_inputs(self.h, actions.data_ptr(), noise.data_ptr(), reset_uniforms.data_ptr(),
                                             self._stream()), "ag_step_with_inputs")

    def get_state(self):
        n = self.num_envs
        out = {
            "root_states": torch.empty(n, 13, device=self.device),
            "ctl_state": torch.empty(n, 12, device=self.device),
            "pre_actions": torch.empty(n, self.num_actions, device=self.device),
            "progress": torch.empty(n, dtype=torch.int32, device=self.device),
            "was_reset": torch.empty(n, dtype=torch.int32, device=self.device),
        }
        v = N.AgStateView(out["root_states"].data_ptr(), out["ctl_state"].data_ptr(), out["pre_actions"].data_ptr(),
                          out["progress"].data_ptr(), out["was_reset"].data_ptr())
        N.check(self.lib.ag_get_state(self.h, ctypes.byref(v), self._stream()), "ag_get_state")
        return out

    def set_state(self, root_states=None, ctl_state=None, pre_actions=None, progress=None, was_reset=None):
        keep = []

        def p(t, dtype, shape):
            if t is None:
                return None
            t = t.to(device=self.device, dtype=dtype).contiguous()
            assert tuple(t.shape) == shape, (tuple(t.shape), shape)
            keep.append(t)
            return t.data_ptr()
        n = self.num_envs
        v = N.AgStateView(p(root_states, torch.float32, (n, 13)), p(ctl_state, torch.float32, (n, 12)),
                          p(pre_actions, torch.float32, (n, self.num_actions)), p(progress, torch.int32, (n,)),
                          p(was_reset, torch.int32, (n,)))
        N.check(self.lib.ag_set_state(self.h, ctypes.byref(v), self._stream()), "ag_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # `keep` must outlive the kernel

    # ---- planning extras
    RESET_UNIFORMS = {"planning": 121, "balloon": 15, "avoid": 11}

    def planning_step_with_uniforms(self, actions, reset_uniforms):
        actions = self._check_actions(actions)
        ru = reset_uniforms.to(device=self.device, dtype=torch.float32).contiguous()
        assert ru.shape == (self.num_envs, self.RESET_UNIFORMS[self.task])
        N.check(self.lib.ag_planning_step_with_uniforms(self.h, actions.data_ptr(), ru.data_ptr(), self._stream()),
                "ag_planning_step_with_uniforms")

    def planning_eval_post(self, actions, collisions, noise=None):
        """Post-physics half of Planning / Balloon / Avoid .step on the current state with supplied collision flags (and,
        for Balloon, supplied observation noise) - parity tests."""
        actions = self._check_actions(actions)
        c = collisions.to(device=self.device, dtype=torch.float32).contiguous()
        assert c.shape == (self.num_envs,)
        z = None
        if noise is not None:
            z = noise.to(device=self.device, dtype=torch.float32).contiguous()
            assert z.shape == (self.num_envs, 18)
        N.check(self.lib.ag_planning_eval_post(self.h, actions.data_ptr(), c.data_ptr(),
                                               z.data_ptr() if z is not None else None, self._stream()),
                "ag_planning_eval_post")
        torch.cuda.current_stream(self.device).synchronize()

    def last_step_rendered(self):
        """True if the most recent step wrote a new depth image (the camera runs every 4th step)."""
        r = self.lib.ag_planning_last_step_rendered(self.h)
        if r < 0:
            raise RuntimeError("this task has no camera")
        return bool(r)

    def planning_render_next_step(self, debug_skip=0):
        if debug_skip:      # experiments build only (AIRGYM_EXPERIMENTS=1)
            N.check(self.lib.ag_debug_planning_render_parts(self.h, int(debug_skip)), "ag_debug_planning_render_parts")
        else:
            N.check(self.lib.ag_planning_render_now(self.h), "ag_planning_render_now")

    def planning_get_state(self):
        n = self.num_envs
        out = {"goal": torch.empty(n, 3, device=self.device), "extra": torch.empty(n, 5, device=self.device)}
        if self.task == "planning":
            out["obstacles"] = torch.empty(n, 40, 4, device=self.device)
        if self.task == "avoid":
            out["object_vel"] = torch.empty(n, 3, device=self.device)
        v = N.AgPlanningStateView(out["obstacles"].data_ptr() if "obstacles" in out else None, out["goal"].data_ptr(),
                                  out["extra"].data_ptr(), out["object_vel"].data_ptr() if "object_vel" in out else None)
        N.check(self.lib.ag_planning_get_state(self.h, ctypes.byref(v), self._stream()), "ag_planning_get_state")
        return out

    def planning_set_state(self, obstacles=None, goal=None, extra=None, object_vel=None):
        keep = []

        def p(t, shape):
            if t is None:
                return None
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            assert tuple(t.shape) == shape
            keep.append(t)
            return t.data_ptr()
        n = self.num_envs
        v = N.AgPlanningStateView(p(obstacles, (n, 40, 4)), p(goal, (n, 3)), p(extra, (n, 5)), p(object_vel, (n, 3)))
        N.check(self.lib.ag_planning_set_state(self.h, ctypes.byref(v), self._stream()), "ag_planning_set_state")
        torch.cuda.current_stream(self.device).synchronize()

    def compact_reset_ids(self):
        """Ascending ids of the envs flagged done by the last step == reset_buf.nonzero().squeeze(-1)."""
        N.check(self.lib.ag_compact_reset_ids(self.h, self._stream()), "ag_compact_reset_ids")
        k = int(self.reset_count.item())
        return self.reset_ids[:k]

    def set_target_state(self, target_state):
        ts = np.asarray(target_state, dtype=np.float32).reshape(18)
        arr = (ctypes.c_float * 18)(*[float(x) for x in ts])
        N.check(self.lib.ag_set_target_state(self.h, arr), "ag_set_target_state")

    @property
    def tick(self):
        return int(self.lib.ag_get_tick(self.h))

    @tick.setter
    def tick(self, v):
        N.check(self.lib.ag_set_tick(self.h, int(v)), "ag_set_tick")

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.device)
            self.lib.ag_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
