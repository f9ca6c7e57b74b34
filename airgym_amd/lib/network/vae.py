"""Frozen depth-image VAE encoder for the Planning policy (reference: lib/network/VAE.py:52-148 ImgEncoder,
:161-254 VAE.encode; lib/network/vae_image_encoder.py:7-53 VAEImageEncoder).

Only the ENCODER half is on the policy path (the decoder is used for offline reconstruction plots), so only it is built:

    image [N,1,H,W] --bilinear resize to cfg.image_res--> residual conv stack (ELU) --> [N, 4*7*128] --> 512 --> 2*latent
    features = the first `latent` columns (means), or means + N(0,1) * exp(0.5 * logvar) with return_sampled_latent

The layer table below is the architecture's specification (kernel / stride / padding of every convolution and which
tensors the two skip convolutions connect); sub-module names equal the reference's so that `trained/vae_model.pth`
(absent from the reference checkout) loads with its `encoder.*` keys.  The encoder is frozen and is NOT part
of the policy's state dict, exactly as in the reference where VAEImageEncoder is a plain object, not an nn.Module member.
Convolutions run on MIOpen (MFMA implicit GEMM); nothing here is hand-written HIP.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# name: (in_ch, out_ch, kernel, stride, padding)
_CONVS = {
    "conv0": (None, 32, 5, 2, 2),
    "conv0_1": (32, 32, 3, 2, 2),
    "conv1_0": (32, 32, 5, 2, 1),
    "conv1_1": (32, 64, 3, 1, 1),
    "conv2_0": (64, 64, 5, 2, 2),
    "conv2_1": (64, 128, 3, 2, 1),
    "conv3_0": (128, 128, 3, 1, 1),
    "conv0_jump_2": (32, 64, 4, 2, 1),          # skip: stage-0 output -> stage-1 output
    "conv1_jump_3": (64, 128, 5, 4, (2, 1)),    # skip: stage-1 output -> stage-2 output
}
_XAVIER = ("conv0_1", "conv1_1", "conv2_1")     # reference re-initialises these (xavier, zero bias)
FLAT_FEATURES = 4 * 7 * 128                     # conv3_0 output for a 120x212 input


def _center_crop_like(t, ref):
    dh, dw = (t.shape[2] - ref.shape[2]) // 2, (t.shape[3] - ref.shape[3]) // 2
    return t[:, :, dh:dh + ref.shape[2], dw:dw + ref.shape[3]]


class DepthEncoder(nn.Module):
    def __init__(self, input_dim=1, latent_dim=64):
        super().__init__()
        self.latent_dim = latent_dim
        for name, (cin, cout, k, s, p) in _CONVS.items():
            conv = nn.Conv2d(input_dim if cin is None else cin, cout, kernel_size=k, stride=s, padding=p)
            if name in _XAVIER:
                nn.init.xavier_uniform_(conv.weight, gain=nn.init.calculate_gain("linear"))
                nn.init.zeros_(conv.bias)
            setattr(self, name, conv)
        self.dense0 = nn.Linear(FLAT_FEATURES, 512)
        self.dense1 = nn.Linear(512, 2 * latent_dim)

    def forward(self, img):
        s0 = F.elu(self.conv0_1(self.conv0(img)))
        s1 = self.conv1_1(self.conv1_0(s0))
        s1 = F.elu(s1 + _center_crop_like(self.conv0_jump_2(s0), s1))
        s2 = self.conv2_1(self.conv2_0(s1))
        s2 = F.elu(s2 + _center_crop_like(self.conv1_jump_3(s1), s2))
        flat = self.conv3_0(s2).flatten(1)
        return self.dense1(F.elu(self.dense0(flat)))


def _clean_keys(state_dict):
    """'module.' (DataParallel) and 'dronet.' -> 'encoder.' prefixes, vae_image_encoder.py:7-15."""
    out = {}
    for k, v in state_dict.items():
        k = k.replace("module.", "").replace("dronet.", "encoder.")
        out[k] = v
    return out


class FrozenVAEEncoder:
    """Inference-only wrapper; mirrors VAEImageEncoder's interface (encode / get_latent_dims_size)."""

    def __init__(self, config, device="cuda:0"):
        get = (lambda k, d=None: config.get(k, d)) if isinstance(config, dict) else (lambda k, d=None: getattr(config, k, d))
        self.latent_dim = int(get("latent_dims", 64))
        self.image_res = tuple(get("image_res", (120, 212)))
        self.interpolation_mode = get("interpolation_mode", "bilinear")
        self.return_sampled_latent = bool(get("return_sampled_latent", False))
        self.encode_chunk = int(get("encode_chunk", 0) or 0)      # > 0: encode at most this many images per convolution call
        # MIOpen picks its convolution kernels per shape.  Its immediate mode (PyTorch's default) falls back to solvers that
        # need no workspace: 21 us per image for this encoder at 16 384 images; with the find step (run once per shape,
        # forward-only here, a few seconds) it is 0.2 us per image.  Scoped to the encoder's own calls.
        self.miopen_find = bool(get("miopen_find", True))
        self.encoder = DepthEncoder(1, self.latent_dim).to(device)
        path = os.path.join(get("model_folder", "") or "", get("model_file", "") or "")
        if os.path.isfile(path):
            sd = _clean_keys(torch.load(path, map_location="cpu"))
            enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
            self.encoder.load_state_dict(enc)           # decoder keys (img_decoder.*) are not needed on this path
            self.pretrained = True
        elif get("allow_random_init", False):
            self.pretrained = False                     # benchmarks / tests: architecture with random weights
        else:
            raise FileNotFoundError(
                f"VAE weights not found at '{path}' (trained/vae_model.pth is not shipped with AirGym, SURVEY 5.4); "
                "set network.vae.model_folder/model_file, or allow_random_init: True for a random-weight encoder")
        self.encoder.eval()
        for p in self.encoder.parameters():
            p.requires_grad_(False)

    def to(self, device):
        self.encoder.to(device)
        return self

    @torch.no_grad()
    def encode(self, images):
        if tuple(images.shape[-2:]) != self.image_res:
            images = F.interpolate(images, self.image_res, mode=self.interpolation_mode)   # a resize, not a transpose (Q14)
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = bool(self.miopen_find and images.is_cuda) or prev
        try:
            if self.encode_chunk and images.shape[0] > self.encode_chunk:
                z = torch.cat([self.encoder(part) for part in images.split(self.encode_chunk)], 0)
            else:
                z = self.encoder(images)
        finally:
            torch.backends.cudnn.benchmark = prev
        means, logvars = z[:, :self.latent_dim], z[:, self.latent_dim:]
        if self.return_sampled_latent:
            return means + torch.randn_like(logvars) * torch.exp(0.5 * logvars)
        return means

    def get_latent_dims_size(self):
        return self.latent_dim
