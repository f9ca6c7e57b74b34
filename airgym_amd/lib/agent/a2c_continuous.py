"""Continuous-action PPO agent - the loop that drives the env hot path.

Mirrors the reference's `lib/agent/a2c_base.py::A2CBase` + `lib/agent/a2c_continuous.py::A2CAgent`
(same YAML keys, same math, same checkpoint layout) but is organised MI355X-first:

  * rollout tensors are allocated once; the env kernel writes obs / reward / done of step t straight
    into slot t of the rollout buffer (`ag_step_into`), no copy kernels (reference: experience buffer
    `update_data` copies, a2c_base.py:662-678);
  * no host synchronisation inside the rollout: episode statistics are reduced on the device per
    step and read back once per epoch (reference: `dones.nonzero()` every step, a2c_base.py:680-689);
  * all parameters live in ONE flat fp32 buffer and all gradients in another; the data-parallel
    exchange is a single RCCL all-reduce per optimizer step on that buffer with the KL scalar appended
    (reference: cat/all_reduce/copy-back + two scalar all-reduces + an LR broadcast per minibatch,
    a2c_base.py:293-309,348-352, a2c_continuous.py:111-123);
  * Adam, gradient clipping and the KL-adaptive LR rule run on device tensors, so one optimizer step
    is a fixed launch sequence that can be captured in a hipGraph (`use_hip_graph`).

Math restated from: play_steps a2c_base.py:651-711; discount_values (GAE) :463-478; prepare_dataset
a2c_continuous.py:140-177; calc_gradients :299-369; trancate_gradients_and_step a2c_base.py:293-316;
train_epoch a2c_continuous.py:78-138.  Pinned by tests/golden/ppo.npz, gae.npz (reference outputs).
"""
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist

from airgym_amd.lib.core import collectives, common_losses, schedulers, torch_ext
from airgym_amd.lib.core.datasets import PPODataset
from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
from airgym_amd.lib.utils import vecenv
from airgym_amd.lib.utils.tr_helpers import DefaultRewardsShaper


def swap_and_flatten01(arr):
    """[H, N, ...] -> [N*H, ...] env-major (a2c_base.py:26-33)."""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


def rescale_actions(low, high, action):
    d = (high - low) / 2.0
    m = (high + low) / 2.0
    return action * d + m


def discount_values(fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma, tau):
    """GAE(gamma, tau), a2c_base.py:463-478.  [H,N,1] tensors."""
    horizon = mb_rewards.shape[0]
    lastgaelam = 0
    mb_advs = torch.zeros_like(mb_rewards)
    for t in reversed(range(horizon)):
        if t == horizon - 1:
            nextnonterminal = 1.0 - fdones
            nextvalues = last_values
        else:
            nextnonterminal = 1.0 - mb_fdones[t + 1]
            nextvalues = mb_values[t + 1]
        nextnonterminal = nextnonterminal.unsqueeze(1)
        delta = mb_rewards[t] + gamma * nextvalues * nextnonterminal - mb_values[t]
        mb_advs[t] = lastgaelam = delta + gamma * tau * nextnonterminal * lastgaelam
    return mb_advs


class DedupFrames:
    """The images of a flattened rollout batch (env-major: sample b = env * H + t) without their duplicates.
    `frames` [S, N, ...] are the distinct images of the rollout, `frame_of_step[t]` says which one step t shows (the same for
    every env: the camera runs on a global cadence).  Slicing [b0:b1] - what PPODataset does for a minibatch - yields the
    distinct images of those samples, the sample -> image map and the multiplicities."""

    def __init__(self, frames, frame_of_step, horizon, in_place=False, cache=None):
        self.frames = frames
        self.in_place = in_place
        self.S, self.N = frames.shape[0], frames.shape[1]
        self.H = horizon
        # (start, stop) -> index tensors: the mini-epochs of an update ask for the same slices again.  `cache` (the agent's, keyed
        # by the camera's step pattern): the SAME tensors epoch after epoch - the camera runs on a global cadence, so the pattern
        # repeats - which is what lets a captured minibatch graph keep reading them
        pattern = tuple(int(f) for f in frame_of_step)
        if cache is not None:
            ent = cache.get(pattern)
            if ent is None:
                ent = cache[pattern] = {"frame_of_step": torch.tensor(pattern, dtype=torch.long, device=frames.device), "slices": {}}
            self.frame_of_step, self._slices = ent["frame_of_step"], ent["slices"]
        else:
            self.frame_of_step = torch.tensor(pattern, dtype=torch.long, device=frames.device)
            self._slices = {}

    def __len__(self):
        return self.N * self.H

    def __getitem__(self, sl):
        hit = self._slices.get((sl.start, sl.stop))
        if hit is not None:
            rows, inverse, counts = hit
            flat = self.frames.view((self.S * self.N,) + tuple(self.frames.shape[2:]))
            if self.in_place:
                return {"image": flat, "image_index": rows, "image_inverse": inverse, "image_counts": counts}
            return {"image": flat.index_select(0, rows), "image_inverse": inverse, "image_counts": counts}
        b = torch.arange(sl.start, sl.stop, device=self.frames.device)
        env = torch.div(b, self.H, rounding_mode="floor")
        keys = env * self.S + self.frame_of_step[b - env * self.H]          # non-decreasing within an env, envs ascending
        uniq, inverse, counts = torch.unique_consecutive(keys, return_inverse=True, return_counts=True)
        u_env = torch.div(uniq, self.S, rounding_mode="floor")
        flat = self.frames.view((self.S * self.N,) + tuple(self.frames.shape[2:]))
        rows = (uniq - u_env * self.S) * self.N + u_env
        counts = counts.to(torch.float32)
        counts.ag_sum = float(sl.stop - sl.start)      # sum of the multiplicities, known without a device read (fused_cnn._Trunk)
        self._slices[(sl.start, sl.stop)] = (rows, inverse, counts)
        if self.in_place:       # the model reads flat[rows] where it lies (cnn.forward(..., index)): no [U, 1, 212, 120] copy
            return {"image": flat, "image_index": rows, "image_inverse": inverse, "image_counts": counts}
        return {"image": flat.index_select(0, rows), "image_inverse": inverse, "image_counts": counts}


class FlatAdam:
    """Adam (eps 1e-8, no weight decay unless given) over ONE flat parameter / gradient buffer.
    All state and the learning rate are device tensors: the step is capturable and has no host sync.
    Same update rule as torch.optim.Adam (reference: optim.Adam(..., eps=1e-08), a2c_continuous.py:401)."""

    def __init__(self, flat_param, flat_grad, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.p, self.g = flat_param, flat_grad
        self.b1, self.b2 = betas
        self.eps, self.wd = eps, weight_decay
        dev = flat_param.device
        # {lr, step} + scratch for the fused HIP step (ag_adam_state_bytes = 4 doubles + 64 floats = 36 doubles)
        self.state = torch.zeros(36, dtype=torch.float64, device=dev)
        self.state[0] = float(lr)
        self.lr = self.state[0]
        self.step_t = self.state[1]
        self.exp_avg = torch.zeros_like(flat_param)
        self.exp_avg_sq = torch.zeros_like(flat_param)

    @torch.no_grad()
    def step(self):
        g = self.g
        if self.wd != 0.0:
            g = g.add(self.p, alpha=self.wd)
        self.step_t += 1.0
        self.exp_avg.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
        self.exp_avg_sq.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
        bc1 = 1.0 - torch.pow(self.b1, self.step_t)
        bc2 = 1.0 - torch.pow(self.b2, self.step_t)
        step_size = (self.lr / bc1).float()
        denom = (self.exp_avg_sq.sqrt() / bc2.sqrt().float()).add_(self.eps)
        self.p.sub_(step_size * (self.exp_avg / denom))

    @torch.no_grad()
    def fused_clip_step(self, grad_with_kl, max_grad_norm, kl_threshold, min_lr, max_lr):
        """clip-by-norm + Adam + KL-adaptive LR in one HIP launch (`ag_adam_clip_step`); grad_with_kl is the flat
        gradient buffer whose last element is the minibatch KL."""
        import ctypes

        from airgym_amd import _native as N
        lib = N.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.p.device).cuda_stream)
        N.check(lib.ag_adam_clip_step(self.p.data_ptr(), grad_with_kl.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.state.data_ptr(), self.p.numel(),
                                      self.b1, self.b2, self.eps, self.wd, float(max_grad_norm), float(kl_threshold),
                                      float(min_lr), float(max_lr), stream), "ag_adam_clip_step")

    def state_dict(self, layout=None):
        """torch.optim.Adam's state_dict layout (what the reference saves, a2c_base.py:528-542, and feeds back to
        optimizer.load_state_dict, :576-577): per-parameter exp_avg / exp_avg_sq / step in `model.parameters()` order.
        `layout` = [(offset, shape)] of every parameter in that order inside the flat buffer; without it the private flat
        form is returned (unit tests of the optimizer alone)."""
        if layout is None:
            return {"lr": self.lr.item(), "step": self.step_t.item(), "exp_avg": self.exp_avg.clone(),
                    "exp_avg_sq": self.exp_avg_sq.clone(), "betas": (self.b1, self.b2), "eps": self.eps}
        step = torch.tensor(float(self.step_t.item()))
        state = {}
        for idx, (off, shape) in enumerate(layout):
            n = int(np.prod(shape)) if len(shape) else 1
            state[idx] = {"step": step.clone(), "exp_avg": self.exp_avg[off:off + n].view(shape).clone(),
                          "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).clone()}
        group = {"lr": self.lr.item(), "betas": (self.b1, self.b2), "eps": self.eps, "weight_decay": self.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(layout)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, layout=None):
        if "param_groups" in sd:            # torch.optim.Adam layout (this build's checkpoints and the reference's)
            assert layout is not None, "the torch layout needs the parameter layout"
            self.lr.fill_(float(sd["param_groups"][0]["lr"]))
            steps = [float(st["step"]) for st in sd["state"].values() if "step" in st]
            self.step_t.fill_(max(steps) if steps else 0.0)
            for idx, (off, shape) in enumerate(layout):
                st = sd["state"].get(idx)
                if st is None:
                    continue
                n = int(np.prod(shape)) if len(shape) else 1
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            return
        self.lr.fill_(sd["lr"])
        self.step_t.fill_(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class A2CAgent:
    def __init__(self, base_name, params):
        self.name = base_name
        self.params = params
        self.network_config = params["network"]
        self.config = config = params["config"]
        self.experiment_name = config.get("full_experiment_name") or \
            config["name"] + datetime.now().strftime("_%d-%H-%M-%S")

        # ---- distributed setup: one process per GPU, torchrun-style env vars (a2c_base.py:102-123)
        self.multi_gpu = config.get("multi_gpu", False)
        self.local_rank, self.global_rank, self.world_size = 0, 0, 1
        if self.multi_gpu:
            self.local_rank = int(os.getenv("LOCAL_RANK", "0"))
            self.global_rank = int(os.getenv("RANK", "0"))
            self.world_size = int(os.getenv("WORLD_SIZE", "1"))
            backend = config.get("dist_backend", "nccl")      # "nccl" IS RCCL on ROCm; "gloo" for CPU tests
            if not dist.is_initialized():
                dist.init_process_group(backend, rank=self.global_rank, world_size=self.world_size)
            if backend == "nccl":
                config["device"] = "cuda:" + str(self.local_rank)
            if self.global_rank != 0:
                config["print_stats"] = False
        # `mixed_precision` (reference: torch.cuda.amp autocast + GradScaler around the model forward of the rollout and of
        # calc_gradients, lib/agent/a2c_base.py:236-237,566,582): honoured where EVERY matrix product of the configuration runs in
        # the hand-written kernels - they have one-bf16-MFMA-per-product twins (f32 accumulate, f32 master weights; no loss scaling:
        # bf16 has float32's exponent range).  A configuration that would leave part of its products with the float32 library
        # GEMMs (other widths, CNN / dict observations, CPU) is refused below rather than silently trained in another precision.
        self.mixed_precision = bool(config.get("mixed_precision", False))
        self.ppo_device = config.get("device", "cuda:0")
        if str(self.ppo_device).startswith("cuda"):
            torch.cuda.set_device(self.ppo_device)
            if config.get("tuned_gemms", True):
                from airgym_amd.utils.gemm_tuning import enable_tuned_gemms
                self.tuned_gemms = enable_tuned_gemms()

        # ---- environment (sharded by rank: env ids are global)
        self.num_actors = config["num_actors"]
        self.env_name = config["env_name"]
        self.env_config = dict(config.get("env_config", {}))
        if self.multi_gpu:
            self.env_config.setdefault("env_id_offset", self.global_rank * self.num_actors)
            if str(self.ppo_device).startswith("cuda"):
                self.env_config["sim_device"] = self.ppo_device   # env and learner share the rank's GPU
        self.vec_env = config.get("vec_env") or vecenv.create_vec_env(self.env_name, self.num_actors, **self.env_config)
        self.env_info = self.vec_env.get_env_info()
        self.value_size = self.env_info.get("value_size", 1)
        self.observation_space = self.env_info["observation_space"]
        if hasattr(self.observation_space, "spaces"):        # Dict{image, observation} (a2c_base.py:205-210)
            self.obs_shape = {k: v.shape for k, v in self.observation_space.spaces.items()}
        else:
            self.obs_shape = self.observation_space.shape
        action_space = self.env_info["action_space"]
        self.actions_num = action_space.shape[0]
        self.actions_low = torch.from_numpy(action_space.low.copy()).float().to(self.ppo_device)
        self.actions_high = torch.from_numpy(action_space.high.copy()).float().to(self.ppo_device)
        self.num_agents = self.env_info.get("agents", 1)

        # ---- hyper-parameters (same YAML keys as scripts/config/ppo_*.yaml)
        self.weight_decay = config.get("weight_decay", 0.0)
        self.truncate_grads = config.get("truncate_grads", False)
        self.grad_norm = config.get("grad_norm", 1.0)
        self.save_freq = config.get("save_frequency", 0)
        self.save_best_after = config.get("save_best_after", 100)
        self.print_stats = config.get("print_stats", True)
        self.ppo = config.get("ppo", True)
        self.max_epochs = config.get("max_epochs", -1)
        self.max_frames = config.get("max_frames", -1)
        self.is_adaptive_lr = config.get("lr_schedule") == "adaptive"
        self.linear_lr = config.get("lr_schedule") == "linear"
        self.schedule_type = config.get("schedule_type", "legacy")
        if self.is_adaptive_lr:
            self.kl_threshold = config["kl_threshold"]
            self.scheduler = schedulers.AdaptiveScheduler(self.kl_threshold, min_lr=config.get("min_lr", 1e-6),
                                                          max_lr=config.get("max_lr", 1e-2))
        elif self.linear_lr:
            self.scheduler = schedulers.LinearScheduler(float(config["learning_rate"]),
                                                        max_steps=self.max_epochs if self.max_epochs != -1 else self.max_frames,
                                                        use_epochs=self.max_epochs != -1)
        else:
            self.scheduler = schedulers.IdentityScheduler()
        self.e_clip = config["e_clip"]
        self.clip_value = config["clip_value"]
        # PpoDiagnostics (lib/core/dignostics.py:17-60; on in every shipped YAML): diagnostics/exp_var, clip_frac/<mini-epoch>,
        # rms_value/{mean,var}; rank 0 only like a2c_base.py:127-130
        self.use_diagnostics = bool(config.get("use_diagnostics", False)) and getattr(self, "global_rank", 0) == 0
        self.diag_dict = {}
        self._last_clip = None
        self.horizon_length = config["horizon_length"]
        self.normalize_advantage = config["normalize_advantage"]
        self.normalize_input = config.get("normalize_input", False)
        self.normalize_value = config.get("normalize_value", False)
        self.critic_coef = config["critic_coef"]
        self.gamma = config["gamma"]
        self.tau = config["tau"]
        self.entropy_coef = config["entropy_coef"]
        self.bounds_loss_coef = config.get("bounds_loss_coef", None)
        self.bound_loss_type = config.get("bound_loss_type", "bound")
        self.clip_actions = config.get("clip_actions", True)
        self.value_bootstrap = config.get("value_bootstrap", False)
        self.mini_epochs_num = config["mini_epochs"]
        self.batch_size_envs = self.horizon_length * self.num_actors
        self.batch_size = self.batch_size_envs * self.num_agents
        self.minibatch_size = config.get("minibatch_size", self.batch_size)
        self.num_minibatches = self.batch_size // self.minibatch_size
        assert self.batch_size % self.minibatch_size == 0
        self.last_lr = float(config["learning_rate"])
        self.use_hip_graph = config.get("use_hip_graph", False)
        self.sync_normalizers = config.get("sync_normalizers", True)
        self.sync_timers = config.get("sync_timers", bool(config.get("print_stats", True)))
        self._host_trace = [] if os.environ.get("AIRGYM_HOST_TRACE") else None
        self.frame = 0
        self.epoch_num = 0
        self.curr_frames = 0
        self.mean_rewards = self.last_mean_rewards = -100500
        self.games_to_track = config.get("games_to_track", 100)
        self.game_rewards = torch_ext.AverageMeter(self.value_size, self.games_to_track)
        self.game_shaped_rewards = torch_ext.AverageMeter(self.value_size, self.games_to_track)
        self.game_lengths = torch_ext.AverageMeter(1, self.games_to_track)
        rs = config.get("reward_shaper", {}) or {}
        self.rewards_shaper = rs if callable(rs) else DefaultRewardsShaper(**rs)
        self.algo_observer = (config.get("features") or {}).get("observer")

        # ---- model, flat parameter / gradient storage, optimizer
        keys = {"actions_num": self.actions_num, "input_shape": self.obs_shape,
                "num_seqs": self.num_actors * self.num_agents, "value_size": self.value_size,
                "normalize_value": self.normalize_value, "normalize_input": self.normalize_input}
        self.model = ModelA2CContinuousLogStd(params, keys).to(self.ppo_device)
        self._flatten_parameters()
        self.optimizer = FlatAdam(self.flat_param, self.flat_grad[:-1], self.last_lr, eps=1e-8,
                                  weight_decay=self.weight_decay)
        self.value_mean_std = self.model.value_mean_std if self.normalize_value else None
        self.dataset = PPODataset(self.batch_size, self.minibatch_size, False, self.ppo_device)
        self.has_value_loss = config.get("use_experimental_cv", True)
        self.group = dist.group.WORLD if self.multi_gpu else None

        # ---- logging (rank 0)
        self.train_dir = config.get("train_dir", "runs")
        self.experiment_dir = os.path.join(self.train_dir, self.experiment_name)
        self.nn_dir = os.path.join(self.experiment_dir, "nn")
        self.summaries_dir = os.path.join(self.experiment_dir, "summaries")
        self.writer = None
        if self.global_rank == 0 and config.get("write_summaries", True):
            os.makedirs(self.nn_dir, exist_ok=True)
            os.makedirs(self.summaries_dir, exist_ok=True)
            from airgym_amd.lib.utils.summary import make_writer
            self.writer = make_writer(self.summaries_dir)
        from airgym_amd.lib.agent.fused_update import FusedMLPStep
        self._fused_step = FusedMLPStep(self) if FusedMLPStep.supported(self) else None
        if self.mixed_precision and not (self._fused_step is not None and self._fused_step.covers_all_products):
            raise NotImplementedError(
                "config.mixed_precision: true needs the hand-written update path for every layer ([D -> 256 -> 256] trunk with "
                "D in {16, 18}, float32 observations, CUDA): this configuration would run part of its products in the float32 "
                "library GEMMs - set mixed_precision to false")
        self._graphs = {}
        # minibatch graphs: only where launches dominate (small minibatches).  With multi_gpu the gradient all-reduce is captured
        # inside the graph when RCCL allows it (`capture_gradient_allreduce`; train_actor_critic falls back to a capture split
        # at the all-reduce - graph A: forward / backward / reductions; the all-reduce eager; graph B: average + clip + Adam +
        # LR rule - if the capture throws); a minibatch whose forward all-reduces the normaliser moments (first mini-epoch,
        # sync_normalizers) runs eagerly
        self._capture_collective = bool(config.get("capture_gradient_allreduce", True))
        self.collective_capture_error = None
        self._graph_update = bool(self._fused_step is not None and self.use_hip_graph
                                  and config.get("use_hip_graph_update", self.minibatch_size <= 32768))
        self._upd_graphs = {}
        # generic update path (no FusedMLPStep: e.g. Planning's CNN + MLP(64,128,64)): the whole minibatch step - trunk forward,
        # fused PPO loss, autograd backward, clip + Adam + LR rule, ~150 small launches around the convolutions - as one hipGraph per
        # (minibatch index, statistics on / off, frame store, gamma-guard answer); opt-in (`use_hip_graph_update: true`): measured
        # slower than the eager step on Planning, see _generic_graph_ok
        self._graph_generic = False
        self._graph_generic_error = None
        self._upd_pool = None
        self._dedup_index_cache = {}
        self._ds_bufs = None
        self._rollouts_done = 0
        self.obs = None

    # ------------------------------------------------------------------ parameters
    def _flatten_parameters(self):
        """Re-home every parameter (and its .grad) as a view into one flat buffer.  The last element
        of the gradient buffer carries the KL scalar through the same all-reduce."""
        ps = [p for p in self.model.parameters() if p.requires_grad]
        m = self.model
        fused_heads = not m.separate and m.fixed_sigma
        if fused_heads:
            # mu and value heads adjacent (weights, then biases): the fused [A+1, H] head GEMM reads them as ONE view
            head = [m.mu.weight, m.value_head.weight, m.mu.bias, m.value_head.bias]
            ps = [p for p in ps if all(p is not q for q in head)] + head
        total = sum(p.numel() for p in ps)
        dev = self.ppo_device
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total + 1, dtype=torch.float32, device=dev)
        off = 0
        for p in ps:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            off += n
        self._params = ps
        # every backward of this agent starts from a zeroed gradient buffer (calc_gradients), so a module whose parameters have no
        # other consumer may write its gradients into these views itself (lib/network/fused_cnn.py: no AccumulateGrad launches)
        for mod in self.model.modules():
            if hasattr(mod, "direct_grads"):
                mod.direct_grads = True
        self.heads_w = self.heads_b = self.heads_w_grad = self.heads_b_grad = None
        if fused_heads:
            A1, Hd = m.mu.weight.shape[0] + 1, m.mu.weight.shape[1]
            o = total - A1 * Hd - A1
            self.heads_w, self.heads_w_grad = (t[o:o + A1 * Hd].view(A1, Hd) for t in (self.flat_param, self.flat_grad))
            self.heads_b, self.heads_b_grad = (t[o + A1 * Hd:o + A1 * Hd + A1] for t in (self.flat_param, self.flat_grad))
            m.fused_heads = (self.heads_w, self.heads_b)      # no-grad inference reads the views (no cat per call)

    def broadcast_parameters(self):
        """Initial parameter / normaliser sync from rank 0 (reference: broadcast_object_list of the
        pickled state dict, a2c_continuous.py:188-192) - one flat tensor broadcast instead."""
        if not self.multi_gpu:
            return
        collectives.broadcast(self.flat_param, 0, "initial_parameters")
        for b in self.model.buffers():
            collectives.broadcast(b, 0, "initial_buffers")

    # ------------------------------------------------------------------ rollout
    def init_tensors(self):
        H, N, dev = self.horizon_length, self.num_actors * self.num_agents, self.ppo_device
        f = dict(dtype=torch.float32, device=dev)
        self._hip_env = getattr(getattr(self.vec_env, "env", None), "hip", None)   # zero-copy path if available
        # Frozen image encoder (Planning with the depth VAE, lib/network/vae_image_encoder.py:34-53): its features are computed
        # ONCE per rendered image, during the rollout, and the rollout buffer keeps the [N, latent] features instead of the
        # [N, 1, 212, 120] images - the update then never runs the encoder (the reference re-encodes every minibatch of every
        # mini-epoch; same features up to the image normaliser having moved on meanwhile).
        self._cache_latents = (isinstance(self.obs_shape, dict) and getattr(self.model, "frozen_features_cacheable", False)
                               and bool(self.config.get("cache_frozen_features", True)) and self._hip_env is not None)
        if self._cache_latents:
            self.obs_buf = {"observation": torch.zeros((H + 1, N) + tuple(self.obs_shape["observation"]), **f),
                            "latent": torch.zeros(H + 1, N, self.model.feature_dim, **f)}
            ishape = tuple(self.obs_shape["image"])
            d = dict(dtype=torch.float64, device=dev)
            # moments of the images rendered during the rollout; merged into the image normaliser at the start of the update
            self._img_moments = [torch.zeros(ishape, **d), torch.zeros(ishape, **d), torch.zeros((), **d)]
        elif isinstance(self.obs_shape, dict) and self._dedup_ok():
            # Frame de-duplication (trainable CNN on a camera task): the depth camera renders every 4th env step
            # (planning.py:153-156, avoid.py:181-185), for every env on the same steps, so the H + 1 rollout slots hold at most
            # ceil((H + 1) / 4) + 1 DISTINCT images per env.  Only those are stored (`_frames`, with the slot -> frame map on the
            # host), the rollout runs the CNN once per rendered frame, and the update runs it on the distinct images of a
            # minibatch with BatchNorm / normaliser statistics weighted by their multiplicities: the same function of the
            # parameters as the reference's update over all samples (lib/agent/a2c_continuous.py:299-369) at ~1/4 of the
            # convolution work and 1/3 of the image memory.
            self._dedup = True
            smax = (H + 1 + 3) // 4 + 2
            self.obs_buf = {"observation": torch.zeros((H + 1, N) + tuple(self.obs_shape["observation"]), **f)}
            # two stores used in turn: the update of rollout k reads store k % 2 while rollout k + 1 fills the other
            self._frame_stores = [torch.zeros((smax, N) + tuple(self.obs_shape["image"]), **f) for _ in range(2)]
            self._frames = self._frame_stores[0]
            self._frame_feat = torch.zeros(N, self.model.feature_dim, **f)
            self._frame_of_slot = [0] * (H + 1)
            self._frame_top = 0
            self._graph_generic = self._generic_graph_ok()
        elif isinstance(self.obs_shape, dict):
            self.obs_buf = {k: torch.zeros((H + 1, N) + tuple(shp), **f) for k, shp in self.obs_shape.items()}
        else:
            self.obs_buf = torch.zeros((H + 1, N) + tuple(self.obs_shape), **f)
        self.actions_buf = torch.zeros(H, N, self.actions_num, **f)
        self.mus_buf = torch.zeros(H, N, self.actions_num, **f)
        self.sigmas_buf = torch.zeros(H, N, self.actions_num, **f)
        self.neglogpacs_buf = torch.zeros(H, N, **f)
        self.values_buf = torch.zeros(H, N, self.value_size, **f)
        self.rewards_buf = torch.zeros(H, N, self.value_size, **f)
        self.raw_rewards_buf = torch.zeros(H, N, **f)
        # dones[t] = done entering step t; uint8 like the reference's rollout buffer (lib/core/experience.py:329)
        self.dones_buf = torch.ones(H + 1, N, dtype=torch.uint8, device=dev)
        self.current_rewards = torch.zeros(N, self.value_size, **f)
        self.current_shaped_rewards = torch.zeros(N, self.value_size, **f)
        self.current_lengths = torch.zeros(N, **f)
        self.ep_stats = torch.zeros(H, 4, dtype=torch.float64, device=dev)    # count, sum rew, sum shaped, sum len
        # Episode/<reward term>: per-step means accumulated on device (reference: RLGPUAlgoObserver.process_infos
        # appends every step's item_reward_info and averages at print time, lib/utils/isaacgym_utils.py:66-99)
        terms = getattr(self._hip_env, "reward_terms", None) if self._hip_env is not None else None
        self._term_names = list(terms.keys()) if terms else []
        self._term_sums = torch.zeros(len(self._term_names), dtype=torch.float64, device=dev)
        self._term_steps = 0
        # Hovering / Tracking: the env kernel emits per-64-env-tile sums of the nine terms straight into this buffer
        # (ag_step_rollout) instead of nine f32[N] arrays per step; one reduction per rollout (end of play_steps)
        self._term_tiles = None
        if (self._hip_env is not None and self._term_names and self.config.get("log_reward_terms", True)
                and self._hip_env.task in ("hovering", "tracking")):
            self._term_tiles = torch.zeros(H, (N + 63) // 64, 12, **f)
        from airgym_amd.lib.agent.fused_update import FusedRolloutStep
        self._fused_rollout = FusedRolloutStep(self) if FusedRolloutStep.supported(self) else None
        if self.mixed_precision and not (self._fused_rollout is not None and self._fused_rollout.chain is not None):
            raise NotImplementedError("config.mixed_precision: true needs the one-launch policy forward (ag_mlp_chain_forward) in the "
                                      "rollout; this configuration would run it in float32 - set mixed_precision to false")
        if self._fused_rollout is not None and getattr(self, "_restored_noise_counter", None) is not None:
            self._fused_rollout.counter.fill_(int(self._restored_noise_counter))
        if self.config.get("print_paths", True) and getattr(self, "global_rank", 0) == 0:
            # which implementation THIS configuration got (the hand-written GEMMs cover 256-wide layers; everything else
            # runs the library): one line on stderr at start-up, and `config.paths` of bench.py's JSON line
            import json
            import sys
            print("[airgym_amd] compute paths: " + json.dumps(self.compute_paths()), file=sys.stderr, flush=True)

    def compute_paths(self):
        """{'rollout': ..., 'update': {layer: {forward, dW / backward, dX}}} - what runs for this network / config."""
        fr, fs = self._fused_rollout, self._fused_step
        if fr is None:
            rollout = "generic: model forward (torch) + env.step per rollout step"
        elif fr.chain is not None:
            rollout = ("ag_mlp_chain_forward (whole policy forward, one launch, activations in registers) + "
                       + ("ag_step_rollout_fused" if fr.fuse_tail else "ag_policy_sample + ag_step_rollout + ag_rollout_account"))
        else:
            gemm = "ag_split_gemm_elu_heads" if (fr.split and fr.fuse_gemm_heads) else "library f32 GEMMs (torch.addmm / torch.mm)"
            rollout = (("ag_mlp_input_layer + " if fr.fuse_input else "library first layer + ") + gemm + " + "
                       + ("ag_step_rollout_fused" if fr.fuse_tail else "ag_policy_sample + ag_step_rollout + ag_rollout_account"))
        if fs is None:
            update = "generic: autograd (torch) + fused PPO loss kernel where supported"
        else:
            update = fs.describe_paths()
        return {"rollout_step": rollout, "update": update,
                "optimizer": "ag_adam_clip_step (HIP)" if self.config.get("use_fused_adam", True) and str(self.ppo_device).startswith("cuda")
                else "FlatAdam (torch)",
                "minibatch_hip_graphs": bool(getattr(self, "_graph_update", False) or getattr(self, "_graph_generic", False))}

    def _dedup_ok(self):
        m = self.model
        return (bool(self.config.get("dedup_frames", True)) and self._hip_env is not None
                and getattr(self._hip_env, "image", None) is not None and getattr(m, "has_cnn", False) and not m.separate
                and "image" in self.obs_shape and str(self.ppo_device).startswith("cuda"))

    def _obs_at(self, n):
        if getattr(self, "_dedup", False):
            return {"observation": self.obs_buf["observation"][n], "image": self._frames[self._frame_of_slot[n]]}
        return {k: v[n] for k, v in self.obs_buf.items()} if isinstance(self.obs_buf, dict) else self.obs_buf[n]

    @torch.no_grad()
    def _new_frame(self, slot, first=False, encode=True):
        """The env's camera buffer holds a new image: keep it as the next distinct frame of this rollout and refresh the
        cached CNN features (policy weights and normalisers are fixed during a rollout).  encode=False only stores the
        frame: slot 0 of the NEXT rollout is stored before the update and encoded after it (_refresh_frame_features)."""
        self._frame_top = 0 if first else self._frame_top + 1
        if self._frame_top >= self._frames.shape[0]:
            raise RuntimeError(
                f"frame store overflow: {self._frame_top + 1} rendered frames in one rollout, the store holds "
                f"{self._frames.shape[0]} (sized for one render every 4 env steps, planning.py:153-156; a forced render or a "
                f"different camera cadence needs a larger store)")
        self._frames[self._frame_top].copy_(self._hip_env.image)
        self._frame_of_slot[slot] = self._frame_top
        if encode:
            self._refresh_frame_features()

    @torch.no_grad()
    def _refresh_frame_features(self):
        """CNN features of the newest stored frame with the CURRENT weights, BatchNorm statistics and image normaliser.
        Called for every rendered frame and at the start of every rollout: slot 0 carries the image the env held when the
        previous rollout ended, and the PPO update in between moved the CNN (the reference runs the current CNN on every
        step, lib/agent/a2c_base.py:357-369)."""
        self.model.eval()
        self._frame_feat.copy_(self.model.cnn_features(self._frames[self._frame_top]))

    @torch.no_grad()
    def _encode_current_image(self, n, count_moments=True):
        """latent slot n <- frozen-encoder features of the env's current depth image (normaliser state as of now)"""
        img = self._hip_env.image
        self.model.eval()
        self.obs_buf["latent"][n].copy_(self.model.encode_image(img))
        if count_moments and self.normalize_input:
            mean, var, cnt = self._img_moments
            bc = float(img.shape[0])
            # unbiased like RunningMeanStd.update (running_mean_std.py:31-62); one env has no variance estimate: 0, not NaN
            bm = img.mean(0).double()
            bv = img.var(0).double() if img.shape[0] > 1 else torch.zeros_like(bm)
            delta = bm - mean
            tot = cnt + bc
            m2 = var * cnt + bv * bc + delta ** 2 * cnt * bc / tot
            mean.add_(delta * bc / tot)
            var.copy_(m2 / tot)
            cnt.copy_(tot)

    def _obs_store(self, n, obs):
        if getattr(self, "_dedup", False):
            self.obs_buf["observation"][n].copy_(obs["observation"])
            if n == 0:        # slot 0 of a rollout: the image the env holds now (after a reset, or the previous rollout's last)
                self._new_frame(0, first=True)
            return
        if getattr(self, "_cache_latents", False):
            self.obs_buf["observation"][n].copy_(obs["observation"])
            if "latent" in obs:
                self.obs_buf["latent"][n].copy_(obs["latent"])
            else:
                self._encode_current_image(n, count_moments=False)
        elif isinstance(self.obs_buf, dict):
            for k, v in self.obs_buf.items():
                v[n].copy_(obs[k])
        else:
            self.obs_buf[n].copy_(obs)

    def preprocess_actions(self, actions):
        if self.clip_actions:
            clamped = torch.clamp(actions, -1.0, 1.0)
            return rescale_actions(self.actions_low, self.actions_high, clamped)
        return actions

    def env_reset(self):
        obs = self.vec_env.reset()
        self._obs_store(0, obs)
        self.dones_buf[0].fill_(1)          # a2c_base.py:404: dones start as ones
        return self._obs_at(0)

    @torch.no_grad()
    def get_action_values(self, obs):
        self.model.eval()
        return self.model({"is_train": False, "prev_actions": None, "obs": obs})

    @torch.no_grad()
    def _rollout_step(self, n):
        """One step of play_steps (a2c_base.py:651-695) with every tensor written in place."""
        if self._fused_rollout is not None:
            return self._fused_rollout.step(n)
        if getattr(self, "_dedup", False):      # the CNN ran when the frame was rendered
            res = self.get_action_values({"observation": self.obs_buf["observation"][n], "cnn_features": self._frame_feat})
        else:
            res = self.get_action_values(self._obs_at(n))
        self.actions_buf[n].copy_(res["actions"])
        self.neglogpacs_buf[n].copy_(res["neglogpacs"])
        self.values_buf[n].copy_(res["values"])
        self.mus_buf[n].copy_(res["mus"])
        self.sigmas_buf[n].copy_(res["sigmas"])
        env_actions = self.preprocess_actions(res["actions"])
        if self._hip_env is not None:
            if self._cache_latents:                # Planning, frozen encoder: features refreshed only when the camera ran
                self._hip_env.step_into(env_actions, self.obs_buf["observation"][n + 1], self.raw_rewards_buf[n], None)
                self.dones_buf[n + 1].copy_(self._hip_env.reset_buf)
                if self._hip_env.last_step_rendered():
                    self._encode_current_image(n + 1)
                else:
                    self.obs_buf["latent"][n + 1].copy_(self.obs_buf["latent"][n])
            elif getattr(self, "_dedup", False):   # camera task, trainable CNN: the image is kept only when the camera ran
                self._hip_env.step_into(env_actions, self.obs_buf["observation"][n + 1], self.raw_rewards_buf[n], None)
                self.dones_buf[n + 1].copy_(self._hip_env.reset_buf)
                if self._hip_env.last_step_rendered():
                    self._new_frame(n + 1)
                else:
                    self._frame_of_slot[n + 1] = self._frame_top
            elif isinstance(self.obs_buf, dict):   # Planning: state vector into the slot, image copied from the camera buffer
                self._hip_env.step_into(env_actions, self.obs_buf["observation"][n + 1], self.raw_rewards_buf[n], None)
                self.dones_buf[n + 1].copy_(self._hip_env.reset_buf)
                self.obs_buf["image"][n + 1].copy_(self._hip_env.image)
            elif self._hip_env.task in ("hovering", "tracking"):
                self._hip_env.step_rollout(env_actions, self.obs_buf[n + 1], self.raw_rewards_buf[n], self.dones_buf[n + 1],
                                           self._term_tiles[n] if self._term_tiles is not None else None)
            else:
                self._hip_env.step_into(env_actions, self.obs_buf[n + 1], self.raw_rewards_buf[n], None)
                self.dones_buf[n + 1].copy_(self._hip_env.reset_buf)
            time_outs = self._hip_env.time_out_buf
        else:
            obs, rewards, dones, infos = self.vec_env.step(env_actions)
            self._obs_store(n + 1, obs)
            self.raw_rewards_buf[n].copy_(rewards)
            self.dones_buf[n + 1].copy_(dones)
            time_outs = infos.get("time_outs") if isinstance(infos, dict) else None
        rewards = self.raw_rewards_buf[n].unsqueeze(1)
        shaped = self.rewards_shaper(rewards)
        if self.value_bootstrap and time_outs is not None:
            shaped = shaped + self.gamma * res["values"] * time_outs.unsqueeze(1).float()
        self.rewards_buf[n].copy_(shaped)
        # episode statistics, reduced on device (a2c_base.py:678-695)
        self.current_rewards += rewards
        self.current_shaped_rewards += shaped
        self.current_lengths += 1
        done_f = self.dones_buf[n + 1].float()
        # [count, sum reward, sum shaped reward, sum length] of the episodes that ended this step: one fused reduction
        self.ep_stats[n] = (torch.stack((torch.ones_like(done_f), self.current_rewards[:, 0],
                                         self.current_shaped_rewards[:, 0], self.current_lengths)) * done_f).sum(1)
        if self._term_tiles is not None:
            pass                              # per-tile sums were written by the env kernel; reduced once per rollout
        elif self._term_names and self.config.get("log_reward_terms", True):
            stacked = getattr(self._hip_env, "reward_terms_stacked", None)
            if stacked is not None:       # one reduction for all terms
                self._term_sums += stacked.sum(1).double() / self._hip_env.num_envs
            else:
                rt = self._hip_env.reward_terms
                self._term_sums += torch.stack([rt[k].mean() for k in self._term_names]).double()
        not_done = 1.0 - done_f
        self.current_rewards *= not_done.unsqueeze(1)
        self.current_shaped_rewards *= not_done.unsqueeze(1)
        self.current_lengths *= not_done

    @torch.no_grad()
    def play_steps(self):
        self._rollout_launch()
        return self._rollout_tail()

    @torch.no_grad()
    def _rollout_launch(self):
        """The H policy + env steps of one rollout (one hipGraph replay once captured)."""
        H = self.horizon_length
        # capture needs an even horizon (device tick ping-pong); the camera tasks (Planning, Avoid) decide on the HOST, at
        # capture time, which steps render (every 4th, planning.py:153-156, avoid.py:181-185), so the cadence survives replay
        # only if H % 4 == 0
        camera = self._hip_env is not None and getattr(self._hip_env, "image", None) is not None
        graphable = (self.use_hip_graph and str(self.ppo_device).startswith("cuda") and H % 2 == 0
                     and (not camera or H % 4 == 0)
                     and not self._cache_latents       # the frozen encoder runs MIOpen's find step: not inside a capture
                     and not getattr(self, "_dedup", False))
        fr = self._fused_rollout

        def rollout():
            if fr is not None:
                fr.begin_rollout()
            if self._cache_latents:
                # slot 0 carries the previous rollout's last features: re-encode with the normaliser as it is after the update
                self._encode_current_image(0, count_moments=False)
            if getattr(self, "_dedup", False):
                # likewise for the trainable CNN: the features of slot 0's frame were not computed yet (or, after a restore,
                # come from other weights) - encode it with the weights this rollout runs under
                self._refresh_frame_features()
            for n in range(H):
                self._rollout_step(n)
            if fr is not None:
                fr.end_rollout()
            if self._term_tiles is not None:
                self._term_sums += self._term_tiles.sum((0, 1), dtype=torch.float64)[:len(self._term_names)] / (
                    self.num_actors * self.num_agents)
        if "rollout" in self._graphs:
            self._graphs["rollout"].replay()
        else:
            rollout()
            if graphable:
                # the first rollout ran eagerly (lazy library initialisation); now capture the whole H-step
                # rollout - policy inference + env kernel, H launches of the device-tick ping-pong - for replay
                self._graphs["rollout"] = self._capture(rollout, warmup=False)
        self._rollouts_done += 1

    @torch.no_grad()
    def _rollout_tail(self):
        """Bootstrap value, GAE and the [H, N] -> [N * H] flattening of the rollout buffers (a2c_base.py:696-712)."""
        H = self.horizon_length
        fr = self._fused_rollout
        self.model.eval()
        if fr is not None:
            # bootstrap value of the last observation through the same fused forward as the rollout steps
            fr.refresh_weights()
            heads = fr.heads_of(self.obs_buf[H])
            last_values = self.model.denorm_value(heads[:, self.actions_num:self.actions_num + 1])
            mb_advs, mb_returns = fr.gae(last_values)
        else:
            if getattr(self, "_dedup", False):
                last_values = self.model({"is_train": False, "obs": {"observation": self.obs_buf["observation"][H],
                                                                      "cnn_features": self._frame_feat}})["values"]
            else:
                last_values = self.model({"is_train": False, "obs": self._obs_at(H)})["values"]
            fdones = self.dones_buf[H].float()
            mb_fdones = self.dones_buf[:H].float()
            mb_advs = self._gae(fdones, last_values, mb_fdones)
            mb_returns = mb_advs + self.values_buf
        if getattr(self, "_dedup", False):
            obses = {"observation": swap_and_flatten01(self.obs_buf["observation"][:H]),
                     "frames": DedupFrames(self._frames, self._frame_of_slot[:H], H, in_place=self._frames.is_cuda,
                                           cache=self._dedup_index_cache)}
        elif isinstance(self.obs_buf, dict):
            obses = {k: swap_and_flatten01(v[:H]) for k, v in self.obs_buf.items()}
        else:
            obses = swap_and_flatten01(self.obs_buf[:H])
        batch = {
            "obses": obses,
            "dones": swap_and_flatten01(self.dones_buf[:H]),
            "actions": swap_and_flatten01(self.actions_buf),
            "neglogpacs": swap_and_flatten01(self.neglogpacs_buf),
            "values": swap_and_flatten01(self.values_buf),
            "mus": swap_and_flatten01(self.mus_buf),
            "sigmas": swap_and_flatten01(self.sigmas_buf),
            "returns": swap_and_flatten01(mb_returns),
            "played_frames": self.batch_size,
        }
        # the last observation / done flags become slot 0 of the next rollout
        if getattr(self, "_dedup", False):
            # the update reads this rollout's frames: the next rollout (and its slot 0, the image the env holds now) uses the other store
            self._frames = self._frame_stores[1] if self._frames is self._frame_stores[0] else self._frame_stores[0]
            self.obs_buf["observation"][0].copy_(self.obs_buf["observation"][H])
            self._new_frame(0, first=True, encode=False)      # encoded at the start of the next rollout, after the update
        else:
            self._obs_store(0, self._obs_at(H))
        self.dones_buf[0].copy_(self.dones_buf[H])
        return batch

    def _gae(self, fdones, last_values, mb_fdones):
        return discount_values(fdones, last_values, mb_fdones, self.values_buf, self.rewards_buf, self.gamma, self.tau)

    # ------------------------------------------------------------------ update
    @torch.no_grad()
    def prepare_dataset(self, batch_dict):
        """a2c_continuous.py:140-177"""
        returns, values = batch_dict["returns"], batch_dict["values"]
        advantages = returns - values
        if self.normalize_value:
            grp = self.group if (self.multi_gpu and self.sync_normalizers) else None
            self.value_mean_std.eval()      # statistics are merged explicitly (and optionally across ranks)
            self.value_mean_std.update(values, grp)
            values = self.value_mean_std(values)
            self.value_mean_std.update(returns, grp)
            returns = self.value_mean_std(returns)
        advantages = torch.sum(advantages, axis=1)
        if self.normalize_advantage:
            advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
        values_dict = {
            "old_values": values, "old_logp_actions": batch_dict["neglogpacs"], "advantages": advantages,
            "returns": returns, "actions": batch_dict["actions"], "obs": batch_dict["obses"],
            "dones": batch_dict["dones"], "mu": batch_dict["mus"], "sigma": batch_dict["sigmas"],
        }
        if self._graph_update or self._graph_generic:
            # captured minibatch graphs read fixed addresses: keep the dataset in persistent buffers (dict observations: the
            # state vector is copied, the frame store and its cached index tensors are persistent already)
            def persist(v):
                return {kk: (torch.empty_like(vv, memory_format=torch.contiguous_format) if torch.is_tensor(vv) else None)
                        for kk, vv in v.items()} if isinstance(v, dict) else torch.empty_like(v, memory_format=torch.contiguous_format)
            if self._ds_bufs is None:
                self._ds_bufs = {k: persist(v) for k, v in values_dict.items()}
            for k, v in values_dict.items():
                if isinstance(v, dict):
                    for kk, vv in v.items():
                        if torch.is_tensor(vv):
                            self._ds_bufs[k][kk].copy_(vv)
                        else:
                            self._ds_bufs[k][kk] = vv
                else:
                    self._ds_bufs[k].copy_(v)
            values_dict = self._ds_bufs
        self.dataset.update_values_dict(values_dict)

    def _generic_graph_ok(self):
        """Minibatch hipGraphs on the generic path (OPT-IN, `use_hip_graph_update: true`): CUDA, the fused PPO loss (no host read
        between forward and backward), the de-duplicated frame store (persistent image addresses and index tensors), one GPU.
        Measured on Planning at 16 384 envs (round 6, one box, `tools/bench_planning_ppo.py --graph-update 0 | 1`): 96 graphs capture
        and replay correctly, and the epoch gets SLOWER - 1 059 ms against 1 025 ms eager, +4.7 GB - because the step is 93 % seven
        convolution kernels of ~1 ms each and the host already runs ahead of them (the one host read of the step, the
        multiplicity sum in the CNN trunk, is gone); a hipGraph node costs about what an eager launch costs.  Hence off by default."""
        return (self._fused_step is None and bool(self.config.get("use_hip_graph_update", False)) and getattr(self, "_dedup", False)
                and self._fused_loss_ok() and not self.multi_gpu and self._graph_generic_error is None)

    def _generic_graph_step(self, idx, mb):
        """One optimizer step of the generic path as a hipGraph replay (captured on first use from the third epoch on: by then the
        allocator, the autograd engine and the library workspaces have seen the step)."""
        cnn = getattr(self.model, "actor_cnn", None)
        stats_on = bool(self.model.update_stats)
        fw = bool(cnn.guard_decision(self.ppo_device)) if (cnn is not None and cnn.features[2].training) else False
        obs = mb["obs"]
        # (frame store in use, index tensors of this slice: a rollout whose camera pattern differs gets graphs of its own)
        key = (idx, stats_on, fw, obs["image"].data_ptr() if isinstance(obs, dict) and "image" in obs else 0,
               obs["image_index"].data_ptr() if isinstance(obs, dict) and "image_index" in obs else 0)
        entry = self._upd_graphs.get(key)
        if entry is None:
            out = torch.zeros(6, dtype=torch.float32, device=self.ppo_device)

            def body():
                a, c, e, b, mu, sigma = self._loss_and_backward(mb)
                kl = self._reduce_clip_step()
                out[0].copy_(a.reshape(())); out[1].copy_(c.reshape(())); out[2].copy_(e.reshape(())); out[3].copy_(b.reshape(()))
                out[4].copy_(kl.reshape(()))
                if self._last_clip is not None:
                    out[5].copy_(self._last_clip.reshape(()))
            if cnn is not None:
                cnn.capture_decision = fw
            try:
                if self._upd_pool is None:
                    self._upd_pool = torch.cuda.graph_pool_handle()
                graph = self._capture(body, warmup=False, pool=self._upd_pool)
            except Exception as e:      # a stack that refuses the capture: remember why, run eagerly from here on
                self._graph_generic_error = f"{type(e).__name__}: {e}"[:300]
                self._graph_generic = False
                torch.cuda.synchronize()
                print(f"[airgym_amd] minibatch hipGraph capture failed, eager update from here on: {self._graph_generic_error}",
                      file=__import__("sys").stderr, flush=True)
                return None
            finally:
                if cnn is not None:
                    cnn.capture_decision = None
            entry = (graph, out, mb)       # mb kept alive: the graph reads its views
            self._upd_graphs[key] = entry
            # (the capture itself does not execute: fall through to the replay)
        entry[0].replay()
        st = entry[1].clone()
        self._last_clip = st[5]
        return st[0], st[1], st[2], st[3], st[4].double()

    def _fused_loss_ok(self):
        m = self.model
        return (str(self.ppo_device).startswith("cuda") and self.config.get("use_fused_loss", True) and self.ppo
                and self.has_value_loss and not m.separate and m.fixed_sigma and self.value_size == 1)

    def _loss_and_backward_fused(self, mb):
        """Same quantities as _loss_and_backward, per-row math in one HIP kernel (ag_ppo_loss)."""
        from airgym_amd.lib.core.fused_loss import fused_ppo_loss
        self.model.trunk(mb["obs"], heads_only=True)
        loss, stats = fused_ppo_loss(
            self.model.last_heads, self.model.logstd, mb["actions"], mb["old_logp_actions"], mb["advantages"],
            mb["returns"], mb["old_values"], mb["mu"], mb["sigma"], e_clip=self.e_clip, critic_coef=self.critic_coef,
            entropy_coef=self.entropy_coef, bounds_loss_coef=self.bounds_loss_coef, clip_value=self.clip_value,
            bound_loss_type=self.bound_loss_type, write_back=True)
        self.flat_grad.zero_()
        loss.backward()
        with torch.no_grad():
            self.flat_grad[-1] = stats[4]
        self._last_clip = stats[5]      # diagnostics/clip_frac/<mini-epoch> on the dict-observation configs too
        return stats[0], stats[1], stats[2], stats[3], None, None

    def _loss_and_backward(self, mb):
        """calc_gradients up to backward(), a2c_continuous.py:299-350."""
        if self._fused_loss_ok():
            return self._loss_and_backward_fused(mb)
        res = self.model({"is_train": True, "prev_actions": mb["actions"], "obs": mb["obs"]})
        a_loss = common_losses.actor_loss(mb["old_logp_actions"], res["prev_neglogp"], mb["advantages"], self.ppo, self.e_clip)
        if self.use_diagnostics:
            with torch.no_grad():
                self._last_clip = torch_ext.policy_clip_fraction(res["prev_neglogp"].detach(), mb["old_logp_actions"], self.e_clip)
        if self.has_value_loss:
            c_loss = common_losses.critic_loss(mb["old_values"], res["values"], self.e_clip, mb["returns"], self.clip_value)
        else:
            c_loss = torch.zeros(1, device=self.ppo_device)
        mu, sigma = res["mus"], res["sigmas"]
        if self.bounds_loss_coef is None:
            b_loss = torch.zeros(1, device=self.ppo_device)
        elif self.bound_loss_type == "regularisation":
            b_loss = common_losses.reg_loss(mu)
        else:
            b_loss = common_losses.bound_loss(mu)
        a_loss, c_loss, entropy, b_loss = a_loss.mean(), c_loss.mean(), res["entropy"].mean(), b_loss.mean()
        loss = a_loss + 0.5 * c_loss * self.critic_coef - entropy * self.entropy_coef \
            + b_loss * (self.bounds_loss_coef or 0.0)
        self.flat_grad.zero_()
        loss.backward()
        with torch.no_grad():
            kl = torch_ext.policy_kl(mu.detach(), sigma.detach(), mb["mu"], mb["sigma"], True)
            self.flat_grad[-1] = kl
        return a_loss.detach(), c_loss.detach(), entropy.detach(), b_loss.detach(), mu.detach(), sigma.detach()

    @torch.no_grad()
    def _reduce_clip_step(self, need_kl=True, reduced=False):
        """trancate_gradients_and_step (a2c_base.py:293-316) + the legacy per-minibatch KL schedule
        (a2c_continuous.py:111-118): one all-reduce, clip-by-norm, Adam, LR update - all on device.
        reduced=True: the caller already all-reduced flat_grad (the captured tail of a multi-GPU minibatch graph)."""
        if self.multi_gpu:
            if not reduced:
                collectives.all_reduce(self.flat_grad, "gradient", group=self.group)
            self.flat_grad /= self.world_size
        adaptive = self.is_adaptive_lr and self.schedule_type == "legacy"
        fused_adam = self.flat_grad.is_cuda and self.config.get("use_fused_adam", True)
        kl = self.flat_grad[-1].double() if (need_kl or (adaptive and not fused_adam)) else None   # the torch schedule reads it
        if fused_adam:
            self.optimizer.fused_clip_step(
                self.flat_grad, self.grad_norm if self.truncate_grads else 0.0,
                self.kl_threshold if adaptive else 0.0, getattr(self.scheduler, "min_lr", 0.0),
                getattr(self.scheduler, "max_lr", 1.0))
            return kl
        g = self.flat_grad[:-1]
        if self.truncate_grads:
            total_norm = torch.linalg.vector_norm(g)
            g.mul_(torch.clamp(self.grad_norm / (total_norm + 1e-6), max=1.0))
        self.optimizer.step()
        if adaptive:
            lr = self.optimizer.lr
            thr = self.kl_threshold
            down = torch.clamp(lr / 1.5, min=self.scheduler.min_lr)
            up = torch.clamp(lr * 1.5, max=self.scheduler.max_lr)
            lr.copy_(torch.where(kl > 2.0 * thr, down, torch.where(kl < 0.5 * thr, up, lr)))
        return kl

    def train_actor_critic(self, idx):
        """One optimizer step on minibatch idx; returns device scalars (no sync)."""
        mb = self.dataset[idx]
        if self._fused_step is not None and mb["obs"].shape[0] == self.minibatch_size:
            stats_on = bool(self.model.update_stats)
            collective_in_forward = (self.multi_gpu and stats_on and getattr(self.model, "stats_group", None) is not None)
            if self._graph_update and self.epoch_num >= 2 and not collective_in_forward:
                # launch-bound regime (small minibatches): forward + backward + reductions + Adam of this minibatch as ONE
                # hipGraph, one graph per (minibatch index, statistics-on/off); captured on first use, then replayed.
                # multi_gpu (round 5): the gradient all-reduce is captured INSIDE that graph when RCCL allows it
                # (`capture_gradient_allreduce`, default on) - N > 1 then replays one graph per optimizer step exactly like N = 1.
                # If the capture throws, the round-4 structure is the fallback: the graph ends in front of the all-reduce, which
                # runs eagerly, and rank average + clip + Adam + LR rule are a second graph shared by every minibatch.
                key = (idx, stats_on)
                entry = self._upd_graphs.get(key)
                if entry is None:
                    row = torch.zeros(8, dtype=torch.float32, device=self.ppo_device)
                    if self.multi_gpu and self._capture_collective and dist.get_backend(self.group) != "nccl":
                        # gloo moves the buffer through the host: not capturable (tests run two ranks on one GPU over gloo)
                        self._capture_collective = False
                        self.collective_capture_error = f"backend {dist.get_backend(self.group)}: host-side collective, not capturable"
                    whole = (not self.multi_gpu) or self._capture_collective

                    def body(whole=whole):
                        self._fused_step.step(mb, stats_out=row)
                        if not self.multi_gpu:
                            self._reduce_clip_step(need_kl=False)
                        elif whole:
                            collectives.all_reduce(self.flat_grad, "gradient", group=self.group, counted=False)
                            self._reduce_clip_step(need_kl=False, reduced=True)
                    try:
                        graph = self._capture(body, warmup=False)
                    except Exception as e:      # RCCL refused the capture: remember why, fall back to the split graphs
                        if not (self.multi_gpu and whole):
                            raise
                        self._capture_collective = False
                        self.collective_capture_error = f"{type(e).__name__}: {e}"[:300]
                        torch.cuda.synchronize()
                        whole = False
                        graph = self._capture(lambda: body(False), warmup=False)
                    entry = (graph, row, mb, whole)      # mb kept alive: the graph reads its views
                    self._upd_graphs[key] = entry
                entry[0].replay()
                st = self._fused_step.next_stats_row()
                st.copy_(entry[1])
                self._last_clip = st[6]
                if not self.multi_gpu:
                    return st[0], st[1], st[2], st[3], st[4]
                if entry[3]:      # the all-reduce, the rank average, clip, Adam and the LR rule were part of the replay
                    collectives.count("gradient", self.flat_grad.numel() * self.flat_grad.element_size())
                    return st[0], st[1], st[2], st[3], self.flat_grad[-1].double()
                collectives.all_reduce(self.flat_grad, "gradient", group=self.group)
                tail = self._upd_graphs.get("tail")
                if tail is None:
                    tail = self._capture(lambda: self._reduce_clip_step(need_kl=False, reduced=True), warmup=False)
                    self._upd_graphs["tail"] = tail
                tail.replay()
                return st[0], st[1], st[2], st[3], self.flat_grad[-1].double()      # the rank-averaged KL
            st = self._fused_step.step(mb)
            self._last_clip = st[6]
            kl = self._reduce_clip_step(need_kl=self.multi_gpu)      # multi-GPU: the rank-averaged KL
            return st[0], st[1], st[2], st[3], (kl if self.multi_gpu else st[4])
        if self._graph_generic and self.epoch_num >= 3 and mb["actions"].shape[0] == self.minibatch_size:
            res = self._generic_graph_step(idx, mb)
            if res is not None:
                return res
        a, c, e, b, mu, sigma = self._loss_and_backward(mb)
        kl = self._reduce_clip_step()
        if mu is not None:      # the fused kernel already wrote the new rows back in place
            self.dataset.update_mu_sigma(mu, sigma)
        return a, c, e, b, kl

    def _capture(self, fn, warmup=True, pool=None):
        """Capture fn() into a hipGraph on a side stream (torch.cuda.graph); returns the graph.  pool: a shared graph memory pool
        (graphs that are replayed one after the other and keep nothing alive between replays may share their scratch memory)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if warmup:
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread may issue HIP calls while we capture (multi-GPU runs)
        with torch.cuda.graph(g, pool=pool, stream=s, capture_error_mode="thread_local"):
            out = fn()
        g.outputs = out
        return g

    def _ht(self, label):
        """Host-side timeline of the epoch (AIRGYM_HOST_TRACE=1; tools/host_timeline.py): where the enqueueing thread is while
        the GPU works - or waits."""
        if self._host_trace is not None:
            self._host_trace.append((label, time.perf_counter()))

    def train_epoch(self):
        """a2c_continuous.py:78-138"""
        play_time_start = time.time()
        self._ht("epoch_begin")
        self._rollout_launch()
        self._ht("rollout_enqueued")
        batch_dict = self._rollout_tail()
        self._ht("rollout_tail_enqueued")
        # without sync_timers the host keeps enqueueing the update while the rollout graph still runs; play_time is then
        # the host-side enqueue time only (the epoch total stays exact: one sync at the end)
        if self.sync_timers and str(self.ppo_device).startswith("cuda"):
            torch.cuda.synchronize()
        play_time_end = time.time()
        update_time_start = play_time_end
        self.model.train()
        self.curr_frames = batch_dict.pop("played_frames")
        # (capturing the rollout's tail + prepare_dataset - ~60 small eager launches, 0.45 ms of GPU idle per epoch at 65 536 envs -
        # in a hipGraph of their own was tried in round 5 and bought nothing: the graph's copy nodes wait ~80 us each,
        # `profiles/r05_epoch_gaps.md`)
        self.prepare_dataset(batch_dict)
        self._ht("dataset_enqueued")
        a_losses, c_losses, b_losses, entropies, kls = [], [], [], [], []
        if self.normalize_input:
            self.model.running_mean_std.eval()   # statistics are merged explicitly (model.update_stats), never by .train()
            if getattr(self, "_cache_latents", False):
                # the update sees features, not images: the image normaliser takes this rollout's rendered images here instead
                mean, var, cnt = self._img_moments
                if self.multi_gpu and self.sync_normalizers:
                    # the same statistics on every rank (like RunningMeanStd.update with a group): merge the ranks' moments
                    packed = torch.cat(((mean * cnt).reshape(-1), ((var + mean * mean) * cnt).reshape(-1), cnt.reshape(1)))
                    collectives.all_reduce(packed, "normaliser_moments", group=self.group)
                    npx = mean.numel()
                    tot = packed[-1].clamp_min(1.0)
                    gmean = (packed[:npx] / tot).view_as(mean)
                    gvar = ((packed[npx:2 * npx] / tot).view_as(var) - gmean * gmean).clamp_min(0.0)
                    mean, var, cnt = gmean, gvar, packed[-1]
                self.model.running_mean_std.running_mean_std["image"].merge_moments(mean, var, cnt)
                for t in self._img_moments:
                    t.zero_()
        self.model.stats_group = self.group if (self.multi_gpu and self.sync_normalizers) else None
        if self._fused_step is not None:
            self._fused_step.begin_epoch()
        if self.use_diagnostics:
            self._diag_epoch_begin()
        for mini_ep in range(self.mini_epochs_num):
            ep_kls = []
            clip_fracs = []
            # "don't need to update statistics more than one miniepoch", a2c_continuous.py:130-131
            self.model.update_stats = self.normalize_input and mini_ep == 0
            for i in range(len(self.dataset)):
                self._last_clip = None
                a, c, e, b, kl = self.train_actor_critic(i)
                a_losses.append(a); c_losses.append(c); entropies.append(e); ep_kls.append(kl)
                if self.use_diagnostics and self._last_clip is not None:
                    clip_fracs.append(self._last_clip)
                if self.bounds_loss_coef is not None:
                    b_losses.append(b)
            av_kls = torch_ext.mean_list(ep_kls)
            if self.is_adaptive_lr and self.schedule_type == "standard":
                # (the reference all-reduces av_kls here, a2c_continuous.py:120-122: its per-minibatch KL is rank-local.  Here every
                # minibatch's KL rode in the gradient all-reduce - flat_grad's last slot - and is the rank average already, so the
                # mean of them is identical on every rank: no collective.)
                self.last_lr, self.entropy_coef = self.scheduler.update(self.optimizer.lr.item(), self.entropy_coef,
                                                                        self.epoch_num, 0, av_kls.item())
                self.optimizer.lr.fill_(self.last_lr)
            kls.append(av_kls)
            if self.use_diagnostics and clip_fracs:     # PpoDiagnostics.mini_epoch, dignostics.py:43-46
                self.diag_dict[f"diagnostics/clip_frac/{mini_ep}"] = torch.stack([c.float().reshape(()) for c in clip_fracs]).mean()
        self.model.update_stats = False
        if self.linear_lr:
            self.last_lr, self.entropy_coef = self.scheduler.update(self.last_lr, self.entropy_coef, self.epoch_num,
                                                                    self.frame, 0.0)
            self.optimizer.lr.fill_(self.last_lr)
        # one device->host transfer for every logged scalar (it is also the epoch's only synchronisation point)
        dev_stats = torch.stack([torch_ext.mean_list(a_losses).float().reshape(()), torch_ext.mean_list(c_losses).float().reshape(()),
                                 torch_ext.mean_list(entropies).float().reshape(()), torch_ext.mean_list(kls).float().reshape(()),
                                 (torch_ext.mean_list(b_losses).float().reshape(()) if b_losses
                                  else torch.zeros((), device=self.ppo_device)),
                                 self.optimizer.lr.float().reshape(())])
        # ... and the episode accounting rides in the same transfer (float64 holds every float32 exactly): after this read the
        # host has nothing left to wait for, and the next rollout is enqueued one transfer - not three - later
        log_terms = bool(self._term_names) and self.config.get("log_reward_terms", True)
        parts = [dev_stats.double()]
        if log_terms:
            parts.append((self._term_sums / self.horizon_length).double().reshape(-1))
        parts.append(self.ep_stats.double().reshape(-1))
        self._ht("update_enqueued")
        host = torch.cat(parts).tolist()
        self._ht("stats_on_host")
        update_time_end = time.time()
        dev_stats = host[:6]
        self.last_lr = float(dev_stats[5])
        nt = len(self._term_names) if log_terms else 0
        self._flush_episode_stats(host[6:6 + nt] if log_terms else None, host[6 + nt:])
        self._ht("epoch_end")
        return {
            "play_time": play_time_end - play_time_start, "update_time": update_time_end - update_time_start,
            "total_time": update_time_end - play_time_start,
            "a_loss": dev_stats[0], "c_loss": dev_stats[1], "entropy": dev_stats[2], "kl": dev_stats[3],
            "b_loss": dev_stats[4], "last_lr": self.last_lr,
        }

    def _flush_episode_stats(self, term_sums, ep_stats):
        """Replays the reference's per-step AverageMeter updates from the epoch's one device->host read (train_epoch):
        term_sums = per-step means of the reward terms (or None), ep_stats = the flattened [steps, >= 4] episode accounting rows."""
        self.episode_term_means = {}
        if term_sums is not None:
            self._term_sums.zero_()
            self.episode_term_means = dict(zip(self._term_names, term_sums))
        w = self.ep_stats.shape[1]
        cnt = ep_stats[0::w]
        self.game_rewards.update_from_sums(zip(ep_stats[1::w], cnt))
        self.game_shaped_rewards.update_from_sums(zip(ep_stats[2::w], cnt))
        self.game_lengths.update_from_sums(zip(ep_stats[3::w], cnt))

    # ------------------------------------------------------------------ training loop
    def train(self):
        """a2c_continuous.py:179-294"""
        self.init_tensors()
        self.last_mean_rewards = -100500
        total_time = 0
        self.obs = self.env_reset()
        self.curr_frames = self.batch_size_envs
        self.broadcast_parameters()
        while True:
            self.epoch_num += 1
            epoch_num = self.epoch_num
            stats = self.train_epoch()
            total_time += stats["total_time"]
            should_exit = False
            curr_frames = self.curr_frames * self.world_size if self.multi_gpu else self.curr_frames
            self.frame += curr_frames
            frame = self.frame // self.num_agents
            if self.global_rank == 0:
                if self.print_stats:
                    fps_step_inference = curr_frames / max(stats["play_time"], 1e-9)
                    fps_total = curr_frames / max(stats["total_time"], 1e-9)
                    print(f"fps step and policy inference: {fps_step_inference:.0f} fps total: {fps_total:.0f} "
                          f"epoch: {epoch_num:.0f}/{self.max_epochs:.0f} frames: {frame:.0f}")
                self.write_stats(total_time, epoch_num, stats, frame, curr_frames)
                if self.game_rewards.current_size > 0:
                    mean_rewards = self.game_rewards.get_mean()
                    self.mean_rewards = mean_rewards[0]
                    checkpoint_name = self.config["name"] + "_ep_" + str(epoch_num) + "_rew_" + str(mean_rewards[0])
                    if self.save_freq > 0 and epoch_num % self.save_freq == 0:
                        self.save(os.path.join(self.nn_dir, "last_" + checkpoint_name))
                    if mean_rewards[0] > self.last_mean_rewards and epoch_num >= self.save_best_after:
                        print("saving next best rewards: ", mean_rewards)
                        self.last_mean_rewards = mean_rewards[0]
                        self.save(os.path.join(self.nn_dir, self.config["name"]))
                        if "score_to_win" in self.config and self.last_mean_rewards > self.config["score_to_win"]:
                            print("Maximum reward achieved. Network won!")
                            self.save(os.path.join(self.nn_dir, checkpoint_name))
                            should_exit = True
                if epoch_num >= self.max_epochs and self.max_epochs != -1:
                    mean_rewards = self.game_rewards.get_mean() if self.game_rewards.current_size > 0 else -np.inf
                    self.save(os.path.join(self.nn_dir, "last_" + self.config["name"] + "_ep_" + str(epoch_num)
                                           + "_rew_" + str(mean_rewards).replace("[", "_").replace("]", "_")))
                    print("MAX EPOCHS NUM!")
                    should_exit = True
                if self.frame >= self.max_frames and self.max_frames != -1:
                    print("MAX FRAMES NUM!")
                    should_exit = True
            if self.multi_gpu:
                # epoch and frame counters advance identically on every rank: those two exits need no collective.  Only
                # `score_to_win` depends on rank 0's episode meter (a2c_continuous.py:281-286 broadcasts should_exit every epoch);
                # one scalar per epoch, issued only when the YAML sets that key, outside the update's hot path
                local = ((epoch_num >= self.max_epochs and self.max_epochs != -1) or (self.frame >= self.max_frames and self.max_frames != -1))
                if "score_to_win" in self.config:
                    t = torch.tensor(float(should_exit), device=self.ppo_device)
                    collectives.broadcast(t, 0, "should_exit")
                    should_exit = bool(t.item())
                should_exit = should_exit or local
            if should_exit:
                return self.last_mean_rewards, epoch_num

    @torch.no_grad()
    def _diag_epoch_begin(self):
        """PpoDiagnostics.epoch + the exp_var part of .mini_batch (dignostics.py:28-41,49-57).  explained_variance is a
        function of the dataset's (old values, returns) only, so the per-minibatch values the reference averages over every
        minibatch of every mini-epoch are computed here once per epoch from the contiguous minibatch slices."""
        vd = self.dataset.values_dict
        v, r = vd["old_values"].reshape(-1), vd["returns"].reshape(-1)
        nmb = max(1, v.numel() // self.minibatch_size)
        v, r = v[:nmb * self.minibatch_size].view(nmb, -1), r[:nmb * self.minibatch_size].view(nmb, -1)
        self.diag_dict["diagnostics/exp_var"] = (1.0 - torch.var(r - v, dim=1) / torch.var(r, dim=1)).mean()
        if self.normalize_value:
            self.diag_dict["diagnostics/rms_value/mean"] = self.model.value_mean_std.running_mean.clone()
            self.diag_dict["diagnostics/rms_value/var"] = self.model.value_mean_std.running_var.clone()

    def write_stats(self, total_time, epoch_num, stats, frame, curr_frames):
        """Same TensorBoard tags as a2c_base.py:318-336 and a2c_continuous.py:225-242."""
        w = self.writer
        if w is None:
            return
        w.add_scalar("performance/step_inference_rl_update_fps", curr_frames / max(stats["total_time"], 1e-9), frame)
        w.add_scalar("performance/step_inference_fps", curr_frames / max(stats["play_time"], 1e-9), frame)
        # the env step is one kernel inside the rollout, not separately timed: the step_* tags the reference's dashboards
        # expect (a2c_base.py:323,326) carry the rollout figures
        w.add_scalar("performance/step_fps", curr_frames / max(stats["play_time"], 1e-9), frame)
        w.add_scalar("performance/rl_update_time", stats["update_time"], frame)
        w.add_scalar("performance/step_inference_time", stats["play_time"], frame)
        w.add_scalar("performance/step_time", stats["play_time"], frame)
        w.add_scalar("losses/a_loss", stats["a_loss"], frame)
        w.add_scalar("losses/c_loss", stats["c_loss"], frame)
        w.add_scalar("losses/entropy", stats["entropy"], frame)
        w.add_scalar("losses/bounds_loss", stats["b_loss"], frame)
        w.add_scalar("info/last_lr", stats["last_lr"], frame)
        w.add_scalar("info/lr_mul", 1.0, frame)
        w.add_scalar("info/e_clip", self.e_clip, frame)
        w.add_scalar("info/kl", stats["kl"], frame)
        w.add_scalar("info/epochs", epoch_num, frame)
        for k, v in self.diag_dict.items():      # PpoDiagnostics.send_info: x axis = epoch (dignostics.py:23-27)
            w.add_scalar(k, float(v.reshape(-1)[0]), epoch_num)
        for k, v in getattr(self, "episode_term_means", {}).items():
            w.add_scalar("Episode/" + k, v, epoch_num)
        if self.game_rewards.current_size > 0:
            mr, ms, ml = self.game_rewards.get_mean()[0], self.game_shaped_rewards.get_mean()[0], self.game_lengths.get_mean()[0]
            for tag, x in (("step", frame), ("iter", epoch_num), ("time", total_time)):
                w.add_scalar("rewards/" + tag, mr, x)
                w.add_scalar("shaped_rewards/" + tag, ms, x)
                w.add_scalar("episode_lengths/" + tag, ml, x)

    # ------------------------------------------------------------------ checkpoints (a2c_base.py:528-587)
    def _optimizer_layout(self):
        """(offset, shape) inside the flat buffers of every trainable parameter, in model.parameters() order = the index
        order of torch.optim.Adam(self.model.parameters()) in the reference (a2c_continuous.py:401)."""
        base = self.flat_param.data_ptr()
        return [((p.data_ptr() - base) // 4, tuple(p.shape)) for p in self.model.parameters() if p.requires_grad]

    def get_full_state_weights(self):
        """a2c_base.py:528-542 - same keys, `optimizer` in torch.optim.Adam's state_dict layout; `rollout_noise_counter` is
        this build's addition (the Philox counter of the fused rollout's action noise, so a resumed run does not replay
        the noise sequence of rollout 0)."""
        state = {"model": self.model.state_dict(), "epoch": self.epoch_num, "frame": self.frame,
                 "optimizer": self.optimizer.state_dict(self._optimizer_layout()), "last_mean_rewards": self.last_mean_rewards,
                 "env_state": self.vec_env.get_env_state()}
        fr = getattr(self, "_fused_rollout", None)
        if fr is not None:
            state["rollout_noise_counter"] = int(fr.counter.item())
        return state

    def set_full_state_weights(self, weights, set_epoch=True):
        sd = weights["model"]
        own = self.model.state_dict()
        # a frozen VAE encoder is not part of this model's state (it is loaded from vae_model.pth); the reference's
        # VAEImageEncoder is not an nn.Module either and is never saved - `actor_enc.*` keys can only come from checkpoints of
        # earlier versions of this build, which are accepted by dropping them
        extra = [k for k in sd if k.startswith("actor_enc.") and k not in own]
        if extra:
            sd = {k: v for k, v in sd.items() if k not in extra}
        self.model.load_state_dict(sd)
        if set_epoch:
            self.epoch_num = weights.get("epoch", 0)
            self.frame = weights.get("frame", 0)
        opt = weights.get("optimizer")
        if isinstance(opt, dict) and ("exp_avg" in opt or "param_groups" in opt):
            self.optimizer.load_state_dict(opt, self._optimizer_layout())
        self.last_mean_rewards = weights.get("last_mean_rewards", -100500)
        self.vec_env.set_env_state(weights.get("env_state"))
        self._restored_noise_counter = weights.get("rollout_noise_counter")
        fr = getattr(self, "_fused_rollout", None)
        if fr is not None and self._restored_noise_counter is not None:
            fr.counter.fill_(int(self._restored_noise_counter))

    def save(self, fn):
        torch_ext.save_checkpoint(fn, self.get_full_state_weights())

    def restore(self, fn, set_epoch=True):
        self.set_full_state_weights(torch_ext.load_checkpoint(fn), set_epoch=set_epoch)
