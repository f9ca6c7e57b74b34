"""Depth-image feature extractor (reference: lib/network/cnn.py:3-33): three stride-2 convolutions, each
followed by ReLU and BatchNorm, global average pooling, one Linear to `feature_dim`.  Module names
(`features.<i>`, `fc`) are the reference's so that `trained/planning_cnn_rate.pth` style checkpoints load."""
import torch
import torch.nn as nn


class CNNFeatureExtractor(nn.Module):
    def __init__(self, feature_dim=12):
        super().__init__()
        layers = []
        for cin, cout, k, pad in ((1, 16, 5, 2), (16, 32, 3, 1), (32, 64, 3, 1)):   # (1,212,120) -> (64,27,15)
            layers += [nn.Conv2d(cin, cout, kernel_size=k, stride=2, padding=pad), nn.ReLU(), nn.BatchNorm2d(cout)]
        layers.append(nn.AdaptiveAvgPool2d((1, 1)))
        self.features = nn.Sequential(*layers)
        self.fc = nn.Linear(64, feature_dim)
        self.fused_relu_bn = True       # False: the plain torch modules (MIOpen batch norm), e.g. for A/B timing
        self.hip_convs = True           # False: torch's conv2d (MIOpen: NHWC kernels between NCHW<->NHWC transposes)
        self.fused_trunk = True         # False: layer by layer (every ReLU + BatchNorm output written and read)
        self.bn_sums_from_weights = True   # the trunk's BatchNorm backward reductions from (w, dw) of the next convolution (exact
                                           # identity, fused_cnn.bn_sums_from_conv; needs BatchNorm weights != 0); False: reduction kernel
        # ... guarded: the identity recovers dgamma as (sum w dw - beta sum dy) / gamma, which amplifies the float32 rounding of
        # dw by 1 / |gamma|.  A step whose smallest |gamma_c| (first two BatchNorm layers) is below bn_gamma_guard x the layer's
        # largest takes the reduction kernels instead (exact for any gamma, 0 included).  The ratio is read from a pinned host
        # copy that every training forward refreshes asynchronously - no host synchronisation after the first call, the decision
        # is one step stale (gamma moves by <= lr per step; the threshold has a 2x margin over the 1e-3 the error bound needs).
        self.bn_gamma_guard = 2e-3
        self._gamma_ratio_host = None      # pinned [1] float32: min_c |gamma_c| / max_c |gamma_c| over the guarded layers
        self._gamma_ratio_event = None
        self.bn_fallback_steps = 0         # training forwards that took the reduction kernels because of the guard
        self.capture_decision = None       # set by an owner that captures training steps in hipGraphs (see guard_decision)
        self.dgrad_epilogue = True      # (with bn_sums_from_weights) the second layer's ReLU + BatchNorm backward in the epilogue of the
                                        # third convolution's input gradient; False: as a pass of its own (ag_relu_bn_bwd_dx)
        self.conv1_wgrad_fused = True   # (with bn_sums_from_weights) conv2's input gradient + layer 1's backward + conv1's weight gradient
                                        # as one kernel; False: two kernels with the 1.9 GB gradient between them
        self.direct_grads = False       # True (set by an owner that zeroes .grad before every backward): the trunk's backward writes
                                        # the parameter gradients into the existing .grad tensors itself (no accumulation launches)

    def reset_gamma_guard(self):
        """Forget the cached gamma ratio: the next training forward synchronises once and decides on the CURRENT weights (call
        after writing the BatchNorm weights from outside the optimizer; load_state_dict does it itself)."""
        self._gamma_ratio_host = None
        self._gamma_ratio_event = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.reset_gamma_guard()
        return super()._load_from_state_dict(*args, **kwargs)

    # the guard's pinned buffer and HIP event are per-process scratch, not model state: copy.deepcopy / pickle of the module must
    # work after a training forward (a torch.cuda.Event cannot be pickled), and the copy decides afresh on its first step
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_gamma_ratio_host"] = None
        state["_gamma_ratio_event"] = None
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_gamma_ratio_host", "_gamma_ratio_event") else copy.deepcopy(v, memo)
        return new

    @torch.no_grad()
    def _sums_from_weights_ok(self, device):
        """True when the (w, dw) identity is well conditioned for this step - see bn_gamma_guard."""
        if not self.bn_sums_from_weights:
            return False
        if torch.cuda.is_current_stream_capturing():
            # event.synchronize() + a pinned host copy cannot be part of a hipGraph.  A caller that captures the step decides
            # OUTSIDE the capture (guard_decision(), every step, as the eager path does) and keys its graphs by the answer;
            # without such a caller the captured step takes the reduction kernels (exact for any gamma)
            return bool(self.capture_decision) if self.capture_decision is not None else False
        g1, g2 = self.features[2].weight, self.features[5].weight
        ratio = torch.minimum(g1.abs().min() / g1.abs().max().clamp_min(1e-30),
                              g2.abs().min() / g2.abs().max().clamp_min(1e-30)).float().reshape(1)
        first = self._gamma_ratio_host is None
        if first:
            self._gamma_ratio_host = torch.empty(1, dtype=torch.float32).pin_memory()
            self._gamma_ratio_event = torch.cuda.Event()
        else:
            # the value an EARLIER step left (its copy completed long ago; wait only if a caller runs two steps back to back
            # faster than one tiny copy - then the event wait is microseconds)
            self._gamma_ratio_event.synchronize()
            ok = float(self._gamma_ratio_host[0]) >= self.bn_gamma_guard
        self._gamma_ratio_host.copy_(ratio, non_blocking=True)
        self._gamma_ratio_event.record(torch.cuda.current_stream(device))
        if first:      # nothing to go by yet (fresh model, or a restored checkpoint): one synchronisation
            self._gamma_ratio_event.synchronize()
            ok = float(self._gamma_ratio_host[0]) >= self.bn_gamma_guard
        if not ok:
            self.bn_fallback_steps += 1
        return ok

    def guard_decision(self, device):
        """The gamma guard's answer for the NEXT training forward, taken outside any capture (refreshes the pinned ratio exactly as an
        eager step would).  An owner that replays captured steps calls this before every replay, keeps one graph per answer and
        sets `capture_decision` while it captures."""
        if not self.bn_sums_from_weights:
            return False
        return self._sums_from_weights_ok(device)

    def forward(self, x, weights=None, norm=None, index=None):
        """weights [N] (optional, training): image i stands for weights[i] identical images of the minibatch (frame
        de-duplication, see fused_relu_bn.relu_batchnorm): BatchNorm statistics are those of the full minibatch.
        norm = (mean, std) (optional, broadcastable to an image): x is the RAW image, normalised here as the policy's input
        normaliser does (clamp((x - mean) / std, -5, 5), running_mean_std.py:78-79) - inside the first convolution on the GPU.
        index (optional, int64 [N]): the batch is x[index] (the distinct frames of a minibatch inside the rollout's frame store);
        the GPU trunk reads them in place."""
        if x.is_cuda and self.fused_trunk and self.hip_convs and self.fused_relu_bn and not x.requires_grad:
            # the whole trunk as one autograd node (lib/network/fused_cnn.py): the ReLU + BatchNorm outputs are never written
            from airgym_amd.lib.network import fused_cnn
            if fused_cnn.usable(x, self.features) and (self.features[2].training or not torch.is_grad_enabled()):
                fw = self.bn_sums_from_weights
                if fw and self.features[2].training and torch.is_grad_enabled():
                    fw = self._sums_from_weights_ok(x.device)
                return self.fc(fused_cnn.trunk(x, self.features, weights, norm, index, self.direct_grads, fw,
                                                 self.dgrad_epilogue, self.conv1_wgrad_fused))
        if index is not None:
            x = x.index_select(0, index)
        if norm is not None:
            x = torch.clamp((x - norm[0].view(1, *x.shape[1:])) / norm[1].view(1, *x.shape[1:]), min=-5.0, max=5.0)
        if (x.is_cuda and (self.fused_relu_bn or self.hip_convs)) or weights is not None:
            # ReLU + BatchNorm2d pairs run as one node on csrc/cnn_kernels.hip, the convolutions on csrc/conv_kernels.hip (the
            # modules stay for the state dict)
            from airgym_amd.lib.network import hip_conv
            from airgym_amd.lib.network.fused_relu_bn import relu_batchnorm, relu_batchnorm_torch, usable
            layers = list(self.features)
            i = 0
            while i < len(layers):
                if isinstance(layers[i], nn.ReLU) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.BatchNorm2d):
                    bn = layers[i + 1]
                    if (x.is_cuda and self.fused_relu_bn and usable(x, bn) and (bn.training or not torch.is_grad_enabled())):
                        x = relu_batchnorm(x, bn, weights if bn.training else None)
                    else:
                        x = relu_batchnorm_torch(x, bn, weights)
                    i += 2
                elif isinstance(layers[i], nn.Conv2d) and self.hip_convs and hip_conv.supported(x, layers[i]):
                    x = hip_conv.conv2d(x, layers[i])
                    i += 1
                else:
                    x = layers[i](x)
                    i += 1
        else:
            x = self.features(x)
        return self.fc(x.view(x.size(0), -1))
