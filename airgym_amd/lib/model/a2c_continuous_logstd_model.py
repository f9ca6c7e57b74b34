"""Actor-critic with state-independent log-std (reference: lib/model/a2c_continuous_logstd_model.py:14-227,
lib/model/base_model.py:5-35).

  vector obs :  obs -> RunningMeanStd (clamp +-5) -> MLP -> heads
  dict obs   :  image -> RunningMeanStd['image'] -> CNN -> cat(observation, features) ->
                RunningMeanStd['observation'] (over the 16+feature_dim concat, a2c_continuous_logstd_model.py:71-75,
                157-161) -> MLP -> heads
  heads      :  mu Linear (x0.1 init), value_head Linear (x0.1 init), logstd parameter (init 0)

State-dict keys are the reference's (`actor_mlp.layers.*`, `actor_cnn.features.*`, `actor_cnn.fc.*`, `mu.*`, `logstd`,
`value_head.*`, `running_mean_std.*`, `value_mean_std.*`).  The GEMMs/convolutions are PyTorch-ROCm
(hipBLASLt / MIOpen -> MFMA); this is the only MFMA-eligible work on the hot path.

Normaliser statistics are never updated implicitly inside forward(): the agent sets `update_stats` for the first
mini-epoch (a2c_continuous.py:130-131) and the moments are merged in place (optionally all-reduced across ranks).
"""
import math

import torch
import torch.nn as nn

from airgym_amd.lib.core.running_mean_std import RunningMeanStd, RunningMeanStdObs
from airgym_amd.lib.network.mlp import MLP
from airgym_amd.lib.network.splitk_linear import linear


class ModelA2CContinuousLogStd(nn.Module):
    def __init__(self, params, keys):
        super().__init__()
        actions_num = keys.get("actions_num")
        input_shape = keys.get("input_shape")
        self.normalize_value = params["config"].get("normalize_value", False)
        self.normalize_input = params["config"].get("normalize_input", False)
        self.value_size = params["config"].get("value_size", 1)
        self.load(params["network"])
        if self.has_resnet:
            raise NotImplementedError("the ResNet18 encoder (torchvision weights download) is not part of this build; "
                                      "CNN, frozen-VAE and MLP are")
        self.dict_obs = isinstance(input_shape, dict)
        if (self.has_cnn or self.has_vae) != self.dict_obs:
            raise ValueError("a 'cnn' / 'vae' network block needs dict observations {image, observation} and vice versa")
        self._frozen = []            # frozen encoders: NOT registered, so they stay out of parameters() / state_dict()
        if self.has_vae:
            # a2c_continuous_logstd_model.py:32-35,45-48,114-126: frozen VAE means take the CNN features' place
            from airgym_amd.lib.network.vae import FrozenVAEEncoder
            input_shape = dict(input_shape)
            self._frozen.append(FrozenVAEEncoder(self.vae_cfg, device="cpu"))
            mlp_in = input_shape["observation"][0] + self.feature_dim
            self.actor_mlp = MLP(mlp_in, self.mlp_cfg["units"], self.mlp_cfg["activation"])
            if self.separate:
                self.critic_mlp = MLP(mlp_in, self.mlp_cfg["units"], self.mlp_cfg["activation"])
        elif self.has_cnn:
            from airgym_amd.lib.network.cnn import CNNFeatureExtractor
            input_shape = dict(input_shape)
            self.actor_cnn = CNNFeatureExtractor(feature_dim=self.feature_dim)
            mlp_in = input_shape["observation"][0] + self.feature_dim
            self.actor_mlp = MLP(mlp_in, self.mlp_cfg["units"], self.mlp_cfg["activation"])
            if self.separate:
                self.critic_cnn = CNNFeatureExtractor(feature_dim=self.feature_dim)
                self.critic_mlp = MLP(mlp_in, self.mlp_cfg["units"], self.mlp_cfg["activation"])
        else:
            self.actor_mlp = MLP(input_shape[0], self.mlp_cfg["units"], self.mlp_cfg["activation"])
            if self.separate:
                self.critic_mlp = MLP(input_shape[0], self.mlp_cfg["units"], self.mlp_cfg["activation"])
        out_size = self.mlp_cfg["units"][-1]
        self.mu = nn.Linear(out_size, actions_num)
        self.mu.weight.data.mul_(0.1)
        self.mu.bias.data.mul_(0.0)
        if self.fixed_sigma:
            self.logstd = nn.Parameter(torch.zeros(actions_num, dtype=torch.float32), requires_grad=True)
        else:
            self.logstd = nn.Linear(out_size, actions_num)
            nn.init.constant_(self.logstd.weight, 0.0)
        self.value_head = nn.Linear(out_size, 1)
        self.value_head.weight.data.mul_(0.1)
        self.value_head.bias.data.mul_(0.0)
        if self.normalize_value:
            self.value_mean_std = RunningMeanStd((self.value_size,))
        if self.normalize_input:
            if self.dict_obs:
                input_shape["observation"] = (input_shape["observation"][0] + self.feature_dim,)
                self.running_mean_std = RunningMeanStdObs(input_shape)
            else:
                self.running_mean_std = RunningMeanStd(input_shape)
        self.update_stats = False     # set by the agent during the first mini-epoch
        self.stats_group = None       # torch.distributed group for synchronised moments (or None)
        self.last_heads = None
        self.fused_heads = None       # (weight [A+1,H], bias [A+1]) views of the agent's flat parameter buffer

    def load(self, params):
        """Parse the YAML `network` block (a2c_continuous_logstd_model.py:200-227)."""
        self.separate = params.get("separate", False)
        self.mlp_cfg = params["mlp"]
        self.has_cnn = "cnn" in params
        self.has_resnet = "resnet" in params
        self.has_vae = "vae" in params
        if self.has_cnn:
            self.feature_dim = params["cnn"]["output_dim"]
        if self.has_vae:
            self.vae_cfg = dict(params["vae"])
            self.feature_dim = int(self.vae_cfg.get("latent_dims", 64))
        space = params.get("space", {}).get("continuous", {})
        self.fixed_sigma = space.get("fixed_sigma", True)

    # ---- normalisers (base_model.py:21-35)
    def _norm(self, rms, x, weights=None):
        if not self.normalize_input:
            return x
        with torch.no_grad():
            if self.update_stats:
                rms.update(x.detach(), self.stats_group, weights) if weights is not None else rms.update(x.detach(), self.stats_group)
            mean = rms.running_mean.float()
            std = torch.sqrt(rms.running_var.float() + rms.epsilon)
        return torch.clamp((x - mean) / std, min=-5.0, max=5.0)      # running_mean_std.py:78-79

    def norm_obs(self, observation):
        return self._norm(self.running_mean_std, observation) if self.normalize_input else observation

    def norm_image(self, image, weights=None):
        return self._norm(self.running_mean_std.running_mean_std["image"], image, weights) if self.normalize_input else image

    def image_norm(self, image, weights=None, index=None):
        """The image normaliser as (mean, std) per pixel instead of a normalised copy of `image` (None when inputs are not
        normalised): the CNN's first convolution applies clamp((x - mean) / std, -5, 5) while it reads the raw image
        (lib/network/cnn.py forward(..., norm)).  The statistics are updated exactly where norm_image would update them."""
        if not self.normalize_input:
            return None
        rms = self.running_mean_std.running_mean_std["image"]
        with torch.no_grad():
            if self.update_stats:
                if weights is not None or index is not None:
                    rms.update(image.detach(), self.stats_group, weights, index)
                else:
                    rms.update(image.detach(), self.stats_group)
            return rms.running_mean.float(), torch.sqrt(rms.running_var.float() + rms.epsilon)

    def norm_observation(self, observation):
        return self._norm(self.running_mean_std.running_mean_std["observation"], observation) if self.normalize_input else observation

    @torch.no_grad()
    def cnn_features(self, image):
        """Inference-time CNN features of a batch of images (BatchNorm on its running statistics, the image normaliser as it
        is now): what the rollout caches per rendered frame.  Shared-trunk models only."""
        assert self.has_cnn and not self.separate
        return self.actor_cnn(image, None, self.image_norm(image))

    def encode_image(self, image):
        """Frozen-VAE features of a batch of depth images, normalised with the image statistics as they are NOW."""
        return self._frozen[0].encode(self.norm_image(image))

    @property
    def frozen_features_cacheable(self):
        """True when the image features are a deterministic function of the image and the normaliser state (frozen VAE,
        posterior means): they can then be computed once per rendered image instead of once per forward."""
        return bool(self.has_vae and self._frozen and not self._frozen[0].return_sampled_latent)

    def denorm_value(self, value):
        return self.value_mean_std(value, denorm=True) if self.normalize_value else value

    def _apply(self, fn, *a, **k):
        for enc in self._frozen:      # follow .to(device) / .float() although the encoder is not a registered child
            enc.encoder._apply(fn, *a, **k)
        return super()._apply(fn, *a, **k)

    # ---- forward
    def trunk(self, obs, heads_only=False):
        if self.dict_obs and self.has_vae:
            if "latent" in obs:       # features of the frozen encoder computed once, when the image was rendered (agent cache)
                feat = obs["latent"]
            else:
                feat = self.encode_image(obs["image"])        # one frozen encoder serves actor and critic (same weights)
            a_in = self.norm_observation(torch.cat((obs["observation"], feat), dim=-1))
            a_out = self.actor_mlp(a_in)
            c_out = self.critic_mlp(a_in) if self.separate else a_out
        elif self.dict_obs:
            # Frame de-duplication (agent: dedup_frames): obs["image"] holds the U DISTINCT images of the minibatch,
            # obs["image_inverse"] [B] maps every sample to its image and obs["image_counts"] [U] is how often each occurs (the
            # depth camera renders every 4th env step, planning.py:153-156).  The CNN runs on the distinct images with the
            # batch statistics weighted by the counts and the features are gathered back per sample: the same function of the
            # parameters as the reference's forward over all B images, at about a quarter of the convolution work.
            inverse, counts = obs.get("image_inverse"), obs.get("image_counts")
            if "cnn_features" in obs:       # rollout: features computed when the image was rendered (weights are fixed there)
                a_feat = c_feat = obs["cnn_features"]
            else:
                index = obs.get("image_index")      # the distinct frames are obs["image"][index] (read in place, not gathered)
                norm = self.image_norm(obs["image"], counts, index)
                a_feat = self.actor_cnn(obs["image"], counts, norm, index)
                c_feat = self.critic_cnn(obs["image"], counts, norm, index) if self.separate else None
                if inverse is not None:
                    a_feat = a_feat.index_select(0, inverse)
                    c_feat = c_feat.index_select(0, inverse) if c_feat is not None else None
            a_in = self.norm_observation(torch.cat((obs["observation"], a_feat), dim=-1))
            a_out = self.actor_mlp(a_in)
            if self.separate:
                c_out = self.critic_mlp(self.norm_observation(torch.cat((obs["observation"], c_feat), dim=-1)))
            else:
                c_out = a_out
        else:
            norm_out = self.norm_obs(obs)
            a_out = self.actor_mlp(norm_out)
            c_out = self.critic_mlp(norm_out) if self.separate else a_out
        if not self.separate:
            # mu and value heads read the same trunk output: one [*,H]x[H,A+1] GEMM instead of two
            # (parameters stay separate modules so the state-dict keys are the reference's)
            n_act = self.mu.weight.shape[0]
            if self.fused_heads is not None and not torch.is_grad_enabled():
                heads = torch.addmm(self.fused_heads[1], a_out, self.fused_heads[0].t())
            else:
                heads = linear(a_out, torch.cat((self.mu.weight, self.value_head.weight), 0),
                               torch.cat((self.mu.bias, self.value_head.bias), 0))
            self.last_heads = heads      # [*, A+1] GEMM output, consumed by the fused PPO-loss kernel
            if heads_only:
                return heads
            mu, value = heads[:, :n_act], heads[:, n_act:]
        else:
            mu = self.mu(a_out)
            value = self.value_head(c_out)
        if self.fixed_sigma:
            logstd = mu * 0.0 + self.logstd
        else:
            logstd = self.logstd(a_out)
        return mu, logstd, value

    def forward(self, input_dict):
        is_train = input_dict.get("is_train", True)
        prev_actions = input_dict.get("prev_actions", None)
        mu, logstd, value = self.trunk(input_dict["obs"])
        sigma = torch.exp(logstd)
        if is_train:
            # Normal(mu, sigma).entropy() = 0.5 + 0.5 log(2 pi) + log sigma
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(dim=-1)
            prev_neglogp = self.neglogp(prev_actions, mu, sigma, logstd)
            return {"prev_neglogp": torch.squeeze(prev_neglogp), "values": value, "entropy": entropy,
                    "mus": mu, "sigmas": sigma}
        selected_action = mu + sigma * torch.randn_like(mu)      # Normal(mu, sigma).sample()
        neglogp = self.neglogp(selected_action, mu, sigma, logstd)
        return {"neglogpacs": torch.squeeze(neglogp), "values": self.denorm_value(value),
                "actions": selected_action, "mus": mu, "sigmas": sigma}

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) \
            + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] + logstd.sum(dim=-1)
