"""Inference ("play") loop - reference: lib/agent/players.py (BasePlayer.run :204-290, A2CPlayer :315-432).

Loads a checkpoint in the reference's layout (`model` state-dict with `actor_mlp.layers.*`, `mu.*`,
`logstd`, `value_head.*`, `running_mean_std.*`, `value_mean_std.*`), runs the deterministic policy (mu) - or
samples when `deterministic: False` - on the vectorised env and reports mean episodic reward / length.
Episode bookkeeping stays on the device; one host read per `print_every` steps instead of a nonzero()
per step.
"""
import torch

from airgym_amd.lib.agent.a2c_continuous import rescale_actions
from airgym_amd.lib.core import torch_ext
from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
from airgym_amd.lib.utils import vecenv


class A2CPlayer:
    def __init__(self, params):
        self.params = params
        self.config = config = params["config"]
        self.player_config = config.get("player", {}) or {}
        self.env_name = config["env_name"]
        self.env_config = dict(config.get("env_config", {}))
        self.device = config.get("device", "cuda:0")
        self.num_actors = self.player_config.get("num_actors", config["num_actors"])
        self.is_deterministic = self.player_config.get("deterministic", True)
        self.games_num = self.player_config.get("games_num", 2000)
        self.max_steps = self.player_config.get("max_steps", 108000 // 4)
        self.print_stats = self.player_config.get("print_stats", True)
        self.clip_actions = config.get("clip_actions", True)
        self.env = config.get("vec_env") or vecenv.create_vec_env(self.env_name, self.num_actors, **self.env_config)
        self.env_info = self.env.get_env_info()
        action_space = self.env_info["action_space"]
        self.actions_num = action_space.shape[0]
        self.actions_low = torch.from_numpy(action_space.low.copy()).float().to(self.device)
        self.actions_high = torch.from_numpy(action_space.high.copy()).float().to(self.device)
        space = self.env_info["observation_space"]
        if hasattr(space, "spaces"):        # Dict{image, observation} of the camera tasks (players.py:63-69)
            self.obs_shape = {k: v.shape for k, v in space.spaces.items()}
        else:
            self.obs_shape = space.shape
        keys = {"actions_num": self.actions_num, "input_shape": self.obs_shape, "num_seqs": self.num_actors,
                "value_size": 1, "normalize_value": config.get("normalize_value", False),
                "normalize_input": config.get("normalize_input", False)}
        self.model = ModelA2CContinuousLogStd(params, keys).to(self.device)
        self.model.eval()

    def restore(self, fn):
        checkpoint = torch_ext.load_checkpoint(fn)
        self.set_full_state_weights(checkpoint)
        if checkpoint.get("env_state") is not None:
            self.env.set_env_state(checkpoint["env_state"])

    def set_full_state_weights(self, checkpoint):
        """players.py:376-429: a full checkpoint loads strictly; an MLP-only checkpoint (pretrained without
        the image encoder) fills logstd, both normalisers, the MLP trunk, mu and value_head of a CNN model and
        leaves the encoder at its initialisation.  Unlike the reference's bare `except:` the fallback is taken
        only for the case it was written for (encoder keys missing); any other mismatch still raises."""
        weights = checkpoint["model"]
        try:
            self.model.load_state_dict(weights)
            return
        except RuntimeError as e:
            model_keys = set(self.model.state_dict().keys())
            missing = model_keys - set(weights.keys())
            encoder_only = missing and all(k.startswith(("actor_cnn.", "actor_enc.", "running_mean_std.running_mean_std.image"))
                                           or k.startswith("running_mean_std.running_mean_std.") for k in missing)
            if not (encoder_only and hasattr(self.model.running_mean_std, "running_mean_std")):
                raise e
        print("Missing CNN part. Loading Pretrained MLP Model......")
        with torch.no_grad():
            self.model.logstd.copy_(weights["logstd"])
        if self.model.normalize_input and "running_mean_std.running_mean" in weights:
            self.model.running_mean_std.running_mean_std["observation"].load_state_dict(
                {k: weights["running_mean_std." + k] for k in ("running_mean", "running_var", "count")})
        if self.model.normalize_value and "value_mean_std.running_mean" in weights:
            self.model.value_mean_std.load_state_dict(
                {k: weights["value_mean_std." + k] for k in ("running_mean", "running_var", "count")})
        mlp = {k[len("actor_mlp."):]: v for k, v in weights.items() if k.startswith("actor_mlp.")}
        own = self.model.actor_mlp.state_dict()
        # strict=False in the reference; shapes that do not fit (first layer: obs+features wide here) are skipped loudly
        fit = {k: v for k, v in mlp.items() if k in own and own[k].shape == v.shape}
        skipped = sorted(set(mlp) - set(fit))
        if skipped:
            print("  actor_mlp tensors not loaded (shape differs with the encoder features appended):", skipped)
        self.model.actor_mlp.load_state_dict(fit, strict=False)
        self.model.mu.load_state_dict({"weight": weights["mu.weight"], "bias": weights["mu.bias"]})
        self.model.value_head.load_state_dict({"weight": weights["value_head.weight"], "bias": weights["value_head.bias"]})

    @torch.no_grad()
    def get_action(self, obs, is_deterministic=True):
        res = self.model({"is_train": False, "prev_actions": None, "obs": obs})
        action = res["mus"] if is_deterministic else res["actions"]
        if self.clip_actions:
            return rescale_actions(self.actions_low, self.actions_high, torch.clamp(action, -1.0, 1.0))
        return action

    @torch.no_grad()
    def run(self, print_every=256):
        n = self.num_actors
        obs = self.env.reset()
        cr = torch.zeros(n, device=self.device)
        steps = torch.zeros(n, device=self.device)
        acc = torch.zeros(3, dtype=torch.float64, device=self.device)      # games, sum reward, sum steps
        games_played, sum_rewards, sum_steps = 0, 0.0, 0.0
        for it in range(self.max_steps):
            action = self.get_action(obs, self.is_deterministic)
            obs, r, done, info = self.env.step(action)
            cr += r
            steps += 1
            d = done.float()
            acc[0] += d.sum()
            acc[1] += (cr * d).sum()
            acc[2] += (steps * d).sum()
            cr *= 1.0 - d
            steps *= 1.0 - d
            if (it + 1) % print_every == 0 or it == self.max_steps - 1:
                g, sr, ss = acc.tolist()
                if g > games_played and self.print_stats:
                    print(f"reward: {(sr - sum_rewards) / (g - games_played):.2f} "
                          f"steps: {(ss - sum_steps) / (g - games_played):.1f}")
                games_played, sum_rewards, sum_steps = g, sr, ss
                if games_played >= self.games_num:
                    break
        if games_played > 0:
            print("av reward:", sum_rewards / games_played, "av steps:", sum_steps / games_played)
        return {"games": games_played, "av_reward": sum_rewards / max(games_played, 1),
                "av_steps": sum_steps / max(games_played, 1)}
