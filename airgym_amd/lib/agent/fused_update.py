"""Hand-scheduled forward + backward of ONE PPO minibatch for the shared-trunk ELU-MLP Gaussian policy
(reference math: lib/agent/a2c_continuous.py:299-369 calc_gradients under autograd; lib/network/mlp.py:36-39;
lib/model/a2c_continuous_logstd_model.py:130-198).

Why not autograd: at the bench configuration (M = 196 608 rows, MLP 18-256-256-(4+1)) the minibatch is ~1.0 ms of
GEMM / ELU / loss work that already runs at the fp32-MFMA or HBM rate, plus ~0.5 ms of tiny kernels autograd adds around
it (gradient accumulation adds, zero fills, cat/split of the fused head, a [M,5] column sum, seven eager ops for the
input normaliser, scalar bookkeeping).  Here every gradient is written straight into its slice of the flat gradient
buffer and the schedule is exactly:

    ag_mlp_input_layer (normalise + Linear + ELU) -> middle layers [GEMM+bias, ELU in place] -> last layer GEMM+bias ->
    ag_elu_heads (ELU + head product) -> ag_ppo_loss -> ag_ppo_loss_finalize -> head wgrad (split-K bmm + sum) ->
    ag_heads_bwd_elu (dX of the head + ELU' + bias sums) -> per layer reversed [bias sum, split-K wgrad, dX GEMM,
    ag_elu_bwd_bias]

The GEMMs stay hipBLASLt (MFMA); everything else is the HIP kernels of csrc/ppo_kernels.hip.  All buffers are
preallocated once, so the step is allocation-free and capturable.
"""
import ctypes

import torch
import torch.nn.functional as F

from airgym_amd import _native as N
from airgym_amd.lib.core.fused_loss import BOUND_TYPES
from airgym_amd.lib.network.splitk_linear import SPLIT_K


class SplitGemm256:
    """ag_split_gemm for one [256, 256] weight: float32-accurate products on the bf16 matrix cores (csrc/split_gemm.hip).
    `planes` hold the exact three-way bf16 split of W (forward, C = X W^T) and of W^T (backward, dX = dZ W); they are
    refreshed by `prepare()` whenever W changed (once per optimizer step / once per rollout)."""

    @staticmethod
    def applies(weight, config):
        return (bool(config.get("use_split_gemm", True)) and weight.is_cuda and tuple(weight.shape) == (256, 256)
                and weight.dtype == torch.float32)

    def __init__(self, weight, backward=True, bf16=False):
        self.lib = N.load()
        self.sfx = "_bf16" if bf16 else ""      # mixed_precision: the one-MFMA-per-product twins of the compute entry points
        self.w = weight
        nbytes = self.lib.ag_split_gemm_plane_bytes()
        self.fwd = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        self.bwd = torch.empty(nbytes, dtype=torch.uint8, device=weight.device) if backward else None
        self.in_image = None      # (prepare_input_image: the fused first layer's weights + chain-ordered forward planes)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.w.device).cuda_stream)

    def prepare(self):
        st = self._stream()
        if self.bwd is not None:
            N.check(self.lib.ag_split_gemm_prepare_pair(self.w.data_ptr(), self.fwd.data_ptr(), self.bwd.data_ptr(), 256, 256, st),
                    "ag_split_gemm_prepare_pair")
        else:
            N.check(self.lib.ag_split_gemm_prepare(self.w.data_ptr(), self.fwd.data_ptr(), 256, 256, 0, st), "ag_split_gemm_prepare")

    def forward(self, x, out, bias=None):
        """out [M, 256] = x [M, 256] W^T (+ bias)"""
        N.check(getattr(self.lib, "ag_split_gemm" + self.sfx)(x.data_ptr(), self.fwd.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       out.data_ptr(), x.shape[0], 256, 256, self._stream()), "ag_split_gemm")

    def forward_elu_heads(self, x, z, bias, heads_w, heads_b, heads):
        """z [M, 256] = x W^T (bias-free pre-activation), heads [M, A1] = ELU(z + bias) heads_w^T + heads_b: the GEMM and
        what used to be the ag_elu_heads pass over z, in one launch (ag_split_gemm_elu_heads)."""
        N.check(getattr(self.lib, "ag_split_gemm_elu_heads" + self.sfx)(x.data_ptr(), self.fwd.data_ptr(), bias.data_ptr(), heads_w.data_ptr(),
                                                 heads_b.data_ptr(), z.data_ptr(), heads.data_ptr(), x.shape[0], 256, 256,
                                                 heads_w.shape[0], self._stream()), "ag_split_gemm_elu_heads")

    def forward_loss_heads_bwd(self, x, dz, bias, heads_w, heads_b, loss):
        """The GEMM of forward_elu_heads with the PPO loss and the head layer's backward in its epilogue
        (ag_split_gemm_loss_heads_bwd): dz [M, 256] = (d_heads heads_w) * ELU'(h) and the per-tile partials named in `loss`
        (an AgLossEpilogue) - heads, d_heads and the pre-activation never go through HBM."""
        N.check(getattr(self.lib, "ag_split_gemm_loss_heads_bwd" + self.sfx)(x.data_ptr(), self.fwd.data_ptr(), bias.data_ptr(), heads_w.data_ptr(),
                                                      heads_b.data_ptr(), dz.data_ptr(), ctypes.byref(loss), x.shape[0], 256, 256,
                                                      heads_w.shape[0], self._stream()), "ag_split_gemm_loss_heads_bwd")

    def prepare_input_image(self, w1, b1):
        """One image per optimizer step for forward_input_loss_heads_bwd: the first layer's weights + bias as MFMA fragments and this
        layer's forward planes in the K order the fused launch produces its A operand in (ag_split_gemm_input_prepare)."""
        if self.in_image is None:
            self.in_image = torch.empty(self.lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device=self.w.device)
        if self.bwd is not None:      # the backward planes of this layer ride in the same launch (no separate prepare())
            N.check(self.lib.ag_split_gemm_input_prepare_pair(w1.data_ptr(), b1.data_ptr(), w1.shape[1], self.w.data_ptr(),
                                                              self.in_image.data_ptr(), self.bwd.data_ptr(), self._stream()),
                    "ag_split_gemm_input_prepare_pair")
            return
        N.check(self.lib.ag_split_gemm_input_prepare(w1.data_ptr(), b1.data_ptr(), w1.shape[1], self.w.data_ptr(),
                                                     self.in_image.data_ptr(), self._stream()), "ag_split_gemm_input_prepare")

    def forward_input_loss_heads_bwd(self, inp, M, dz, bias, heads_w, heads_b, loss):
        """forward_loss_heads_bwd with the FIRST layer formed inside the launch (ag_split_gemm_input_loss_heads_bwd): `inp` is an
        AgInputLayerArgs (observations, normaliser statistics; xn and h1 are written as by-products); weights: prepare_input_image."""
        N.check(getattr(self.lib, "ag_split_gemm_input_loss_heads_bwd" + self.sfx)(ctypes.byref(inp), self.in_image.data_ptr(), bias.data_ptr(),
                                                            heads_w.data_ptr(), heads_b.data_ptr(), dz.data_ptr(), ctypes.byref(loss),
                                                            M, 256, 256, heads_w.shape[0], self._stream()),
                "ag_split_gemm_input_loss_heads_bwd")

    def backward_input_wgrad(self, dz, h_prev, x_prev, dw_partials, db_partials):
        """dX = dz W of this layer, consumed in the epilogue by the PREVIOUS (first) layer's backward: ELU'(h_prev), then
        dw_partials [tiles, 256, D] / db_partials [tiles, 256] against its inputs x_prev [M, D] (ag_split_gemm_input_wgrad)."""
        N.check(getattr(self.lib, "ag_split_gemm_input_wgrad" + self.sfx)(dz.data_ptr(), self.bwd.data_ptr(), h_prev.data_ptr(), x_prev.data_ptr(),
                                                   dw_partials.data_ptr(), db_partials.data_ptr(), dz.shape[0], 256, 256,
                                                   x_prev.shape[1], self._stream()), "ag_split_gemm_input_wgrad")

    def backward_input(self, dz, out):
        """out [M, 256] = dz [M, 256] W"""
        N.check(getattr(self.lib, "ag_split_gemm" + self.sfx)(dz.data_ptr(), self.bwd.data_ptr(), None, out.data_ptr(), dz.shape[0], 256, 256,
                                       self._stream()), "ag_split_gemm")


def obs_float32_cuda(agent):
    return str(agent.ppo_device).startswith("cuda")


class FusedMLPStep:
    @staticmethod
    def supported(agent):
        m = agent.model
        return (str(agent.ppo_device).startswith("cuda") and agent.config.get("use_fused_update", True)
                and agent._fused_loss_ok() and not m.dict_obs and m.actor_mlp.activation_name == "elu"
                and agent.minibatch_size % SPLIT_K == 0 and getattr(agent, "heads_w", None) is not None
                and all(l.weight.shape[0] % 4 == 0 and l.weight.shape[0] <= 1024 and 256 % (l.weight.shape[0] // 4) == 0
                        for l in m.actor_mlp.layers))

    def __init__(self, agent):
        self.agent = agent
        self.lib = N.load()
        m, dev = agent.model, agent.ppo_device
        M = self.M = agent.minibatch_size
        self.A = agent.actions_num
        self.layers = [(l.weight, l.bias, l.weight.grad, l.bias.grad) for l in m.actor_mlp.layers]
        f = dict(dtype=torch.float32, device=dev)
        D = self.layers[0][0].shape[1]
        self.xn = torch.empty(M, D, **f)
        self.h = [torch.empty(M, w.shape[0], **f) for w, _, _, _ in self.layers]
        widest = max(w.shape[0] for w, _, _, _ in self.layers)
        self.dh = torch.empty(M * widest, **f)          # d loss / d h_l   (reused by every layer)
        self.dz = torch.empty(M * widest, **f)          # d loss / d z_l
        self.heads = torch.empty(M, self.A + 1, **f)
        self.d_heads = torch.empty(M, self.A + 1, **f)
        self.nsums = self.lib.ag_ppo_loss_num_sums()
        self.loss_partials = torch.empty(max(self.lib.ag_ppo_loss_max_blocks(), (M + 63) // 64), self.nsums, **f)
        L = len(self.layers)
        erows = self.lib.ag_elu_bwd_bias_rows_per_block()
        wrows, irows = self.lib.ag_wgrad_rows_per_block(0), max(1, self.lib.ag_input_wgrad_rows(D))
        self.wg_blocks = (M + wrows - 1) // wrows
        # ... or, with the loss and the head layer's backward in the last GEMM's epilogue (ag_split_gemm_loss_heads_bwd), one
        # partial per GEMM row tile
        # row tile of the loss / recompute launches: 256 rows (8-wave workgroups) unless the minibatch has fewer such tiles than
        # the device has CUs - then 128 rows (4-wave workgroups), so that e.g. the reference's 32 768-sample minibatches at 65 536
        # envs use the whole chip (ag_split_gemm_pick_tile_rows); `split_tile_rows: 128 | 256` overrides
        cfg_rows = int(agent.config.get("split_tile_rows", 0) or 0)
        if cfg_rows not in (0, 128, 256):
            raise ValueError(f"split_tile_rows: {cfg_rows} is not a tile the split GEMM has - use 128, 256, or 0 / absent for the "
                             f"automatic choice (ag_split_gemm_pick_tile_rows)")
        self.tile_rows = cfg_rows or self.lib.ag_split_gemm_pick_tile_rows(M)
        if (not agent.config.get("split_tile_rows") and self.tile_rows == 256 and M % 128 == 0
                and not self.lib.ag_split_gemm_input_fwd_supported(D)):
            # input widths whose first layer is NOT formed inside the forward launch (Tracking's 48): without the 48 KB first-layer
            # image two 4-wave workgroups fit a CU, one's epilogue runs beside the other's main loop: 29.37 -> 29.09 ms per Tracking
            # epoch in an interleaved A/B.  (With the image only one fits: the headline at 128-row tiles is 25.4 -> 30.3 ms.)
            self.tile_rows = 128
        lrows = self.tile_rows
        self.fuse_gemm_loss = (L >= 2 and bool(agent.config.get("fuse_gemm_loss", True))
                               and bool(agent.config.get("fuse_gemm_heads", True)) and self.A + 1 == 5
                               and SplitGemm256.applies(self.layers[-1][0], agent.config) and M % lrows == 0
                               and 64 <= self.layers[-1][0].shape[0] <= 256)
        if self.fuse_gemm_loss:
            self.wg_blocks = M // lrows
        # ... and, for a [D -> 256 -> 256] trunk, with the first layer formed inside that launch as well (no ag_mlp_input_layer)
        self.fuse_gemm_input = (self.fuse_gemm_loss and L == 2 and bool(agent.config.get("fuse_gemm_input", True))
                                and self.layers[0][0].shape[0] == 256 and M % self.tile_rows == 0
                                and bool(self.lib.ag_split_gemm_input_fwd_supported(D)))
        # small weight gradients folded into the ELU' passes (head: always; first layer: D in {16,18,20}, >= 2 layers)
        # ... and for a [D -> 256 -> 256] trunk the first layer's whole backward rides in the epilogue of the second layer's
        # dX GEMM (ag_split_gemm_input_wgrad): dh1 / dz1 are never written, one partial per row tile (D = 18: Hovering, 48: Tracking)
        self.fuse_gemm_input_wgrad = (L == 2 and bool(agent.config.get("fuse_gemm_input_wgrad", True))
                                      and SplitGemm256.applies(self.layers[1][0], agent.config)
                                      and self.layers[0][0].shape[0] == 256
                                      and bool(self.lib.ag_split_gemm_input_wgrad_supported(D)))
        # ... and (round 5) with h1 never stored: the forward launch skips the 201 MB write, the dX epilogue and the weight gradient
        # RECOMPUTE it from the D-wide inputs on the matrix cores (ag_split_gemm_input_wgrad_recompute, ag_split_wgrad_input)
        self.recompute_h1 = (self.fuse_gemm_input and self.fuse_gemm_input_wgrad and bool(agent.config.get("recompute_h1", True))
                             and self._split_wgrad_ok(self.layers[1][0], agent.config) and M % 32 == 0
                             and bool(self.lib.ag_split_gemm_input_wgrad_recompute_supported(D))
                             and bool(self.lib.ag_split_wgrad_input_supported(D)))
        self.fuse_input_wgrad = L >= 2 and (self.lib.ag_input_wgrad_rows(D) > 0 or self.fuse_gemm_input_wgrad)
        if self.fuse_gemm_input_wgrad:
            irows = self.tile_rows if self.recompute_h1 else self.lib.ag_split_gemm_input_wgrad_rows()
        self.in_wg_blocks = (M + irows - 1) // irows
        self.head_wg_partials = torch.empty(self.wg_blocks, self.A + 1, self.layers[-1][0].shape[0], **f)
        self.bias_partials, self.wgrad_partials = [], []
        self.split_wgrad = set()      # layers whose weight gradient runs on the bf16 matrix cores (256 x 256, li >= 1)
        for li, (w, _, _, _) in enumerate(self.layers):
            C, K = w.shape
            if li == L - 1:
                nb = self.wg_blocks
            elif li == 0 and self.fuse_input_wgrad:
                nb = self.in_wg_blocks
            else:
                nb = (M + erows - 1) // erows
            self.bias_partials.append(torch.empty(nb, C, **f))
            if li == 0 and self.fuse_input_wgrad:
                self.wgrad_partials.append(torch.empty(self.in_wg_blocks, C, K, **f))
            elif self._split_wgrad_ok(w, agent.config):
                # ag_split_wgrad: one partial per row slice (= per CU), each operand read once (csrc/split_wgrad.hip)
                ns = self.lib.ag_split_wgrad_input_slices(M) if (li == 1 and self.recompute_h1) else self.lib.ag_split_wgrad_slices(M)
                self.wgrad_partials.append(torch.empty(ns, C, K, **f))
                self.split_wgrad.add(li)
            else:
                self.wgrad_partials.append(torch.empty(SPLIT_K, C, K, **f))
        # every partial-sum reduction of the step runs in ONE ag_sum_rows_multi call (two launches) after the backward
        jobs = [(self.head_wg_partials, agent.heads_w_grad)]
        for li, (_, _, gw, gb) in enumerate(self.layers):
            jobs.append((self.bias_partials[li], gb))
            jobs.append((self.wgrad_partials[li], gw))
        self._jobs = (N.AgSumJob * len(jobs))()
        tot = 0
        for j, (src, dst) in enumerate(jobs):
            n = dst.numel()
            assert src.numel() % n == 0 and dst.is_contiguous()
            self._jobs[j] = N.AgSumJob(src.data_ptr(), dst.data_ptr(), src.numel() // n, n)
            tot += n
        self.use_sum_multi = len(jobs) <= 12 and all(d.numel() % 4 == 0 for _, d in jobs)
        self._sum_pairs = jobs
        self.fuse_finalize = bool(agent.config.get("fuse_loss_finalize", True))
        self.sum_scratch = torch.empty(self.lib.ag_sum_rows_groups() * tot, **f)
        C0, Cl = self.layers[0][0].shape[0], self.layers[-1][0].shape[0]
        self.fuse_input = (D * C0 + 64 * D) * 4 <= 64 * 1024
        # first layer as a launch of its own on the matrix cores (csrc/first_layer.hip) where the forward GEMM cannot produce it
        # itself (Tracking's 48 inputs): replaces ag_mlp_input_layer's vector-ALU kernel (131 -> ~65 us at D = 48, M = 196 608)
        self.mfma_input = (not self.fuse_gemm_input and bool(agent.config.get("use_mfma_input_layer", True)) and obs_float32_cuda(agent)
                           and bool(self.lib.ag_mlp_first_layer_supported(D, C0)) and bool(agent.config.get("use_split_gemm", True)))
        self.in_image = (torch.empty(self.lib.ag_mlp_first_layer_image_bytes(D), dtype=torch.uint8, device=dev) if self.mfma_input else None)
        self.fuse_heads = len(self.layers) >= 2 and 64 <= Cl <= 256 and (Cl & (Cl - 1)) == 0 and self.A + 1 in (5, 6)
        self.wt_last = torch.empty(self.layers[-1][0].shape[1], Cl, **f)      # W_last^T, refreshed before every forward
        # 256 x 256 layers: float32-accurate GEMMs on the bf16 matrix cores (forward and dX); other widths stay with the library
        self.bf16 = bool(getattr(agent, "mixed_precision", False))
        self.sfx = "_bf16" if self.bf16 else ""
        self.split = {li: SplitGemm256(w, bf16=self.bf16) for li, (w, _, _, _) in enumerate(self.layers)
                      if li >= 1 and SplitGemm256.applies(w, agent.config)}
        # ... and for the last of them the ELU + head product rides in the GEMM epilogue (ag_split_gemm_elu_heads)
        self.fuse_gemm_heads = bool(agent.config.get("fuse_gemm_heads", True)) and self.A + 1 in (5, 6)
        self.stats_ring = torch.zeros(max(1, agent.mini_epochs_num * agent.num_minibatches), 8, **f)
        self.k = 0
        self.last_launches = {}       # the step's dominant launches as (entry point, replayable closure) - for bench.py

    @property
    def covers_all_products(self):
        """True when every matrix product of the update runs in the hand-written kernels (what `mixed_precision` needs: the
        library GEMMs and the vector-ALU first layer have no bf16 twin)."""
        return bool(self.fuse_gemm_input and self.recompute_h1 and self.fuse_gemm_loss and self.split_wgrad == {1})

    def describe_paths(self):
        """Which implementation each layer of the update took (bench.py `config.paths`; printed once at agent start-up)."""
        L = len(self.layers)
        lib_gemm = "library f32 GEMM (torch.addmm / torch.mm: hipBLASLt / rocBLAS)"
        out = {}
        for li, (w, _, _, _) in enumerate(self.layers):
            C, K = w.shape
            key = f"layer{li} [{K}->{C}]"
            if li == 0:
                if self.fuse_gemm_input:
                    fwd = "inside the next layer's forward launch, on the matrix cores (ag_split_gemm_input_loss_heads_bwd)"
                elif self.mfma_input:
                    fwd = "ag_mlp_first_layer (normalise + Linear + ELU as a launch of its own on the matrix cores, split-bf16)"
                elif self.fuse_input:
                    fwd = "ag_mlp_input_layer (normalise + Linear + ELU, HIP, vector ALU)"
                else:
                    fwd = lib_gemm
                if self.recompute_h1:
                    bwd = "in the dX launch's epilogue with h1 recomputed (ag_split_gemm_input_wgrad_recompute); h1 never stored"
                elif self.fuse_gemm_input_wgrad:
                    bwd = "in the dX launch's epilogue (ag_split_gemm_input_wgrad), h1 read back"
                elif self.fuse_input_wgrad:
                    bwd = "ag_elu_bwd_input_wgrad (HIP, vector ALU)"
                else:
                    bwd = lib_gemm + " split-K bmm"
                out[key] = {"forward": fwd, "backward": bwd}
                continue
            sg = li in self.split
            if li == L - 1 and sg and self.fuse_heads and self.fuse_gemm_loss:
                fwd = "split-bf16 GEMM + ELU + heads + PPO loss + head backward in one launch (ag_split_gemm_*loss_heads_bwd)"
            elif li == L - 1 and sg and self.fuse_heads and self.fuse_gemm_heads:
                fwd = "split-bf16 GEMM + ELU + heads (ag_split_gemm_elu_heads); loss and head backward as separate HIP launches"
            elif sg:
                fwd = "split-bf16 GEMM (ag_split_gemm)"
            else:
                fwd = lib_gemm
            if li == 1 and self.recompute_h1:
                dw = "split-bf16, X operand produced on chip (ag_split_wgrad_input)"
            elif li in self.split_wgrad:
                dw = "split-bf16 (ag_split_wgrad)"
            else:
                dw = lib_gemm + " split-K bmm"
            dx = ("split-bf16 GEMM (ag_split_gemm / ..._input_wgrad*)" if sg else lib_gemm)
            out[key] = {"forward": fwd, "dW": dw, "dX": dx}
        if self.fuse_gemm_loss:
            out["forward_row_tile"] = (f"{self.tile_rows} rows per workgroup (" + ("8 waves, one workgroup per CU" if self.tile_rows == 256
                                       else "4 waves; two workgroups per CU where the first layer is not formed in this launch") + ")")
        out["product_arithmetic"] = ("mixed_precision: one bf16 MFMA per product, f32 accumulate (_bf16 entry points)" if self.bf16
                                     else "float32-accurate: exact 3-way bf16 split, six MFMAs per product (library GEMM layers: f32 MFMA)")
        return out

    @staticmethod
    def _split_wgrad_ok(weight, config):
        return SplitGemm256.applies(weight, config) and bool(config.get("use_split_wgrad", True))

    @property
    def gemm_description(self):
        if not self.split:
            return "library f32 GEMM"
        return ("ag_split_gemm: exact 3-way bf16 split of every f32 operand, 6 bf16 MFMAs per product, f32 accumulate "
                "(float32-accurate; tests/test_gpu_split_gemm.py) for forward, dX"
                + (" and dW (ag_split_wgrad: no library GEMM left in the update)" if self.split_wgrad else "; dW = library f32"))

    def begin_epoch(self):
        self.k = 0

    def next_stats_row(self):
        row = self.stats_ring[self.k % self.stats_ring.shape[0]]
        self.k += 1
        return row

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.agent.ppo_device).cuda_stream)

    @torch.no_grad()
    def step(self, mb, stats_out=None):
        """Forward, loss, backward of minibatch `mb`; gradients (and the KL in the appended slot) are left in
        agent.flat_grad.  Returns the stats row [a_loss, c_loss, entropy, b_loss, kl, loss, clip_frac, 0] (device, no sync);
        `stats_out` overrides where that row is written (hipGraph capture needs a fixed address)."""
        ag, lib, m = self.agent, self.lib, self.agent.model
        M, A, S = self.M, self.A, SPLIT_K
        obs = mb["obs"]
        assert obs.shape[0] == M and obs.is_contiguous()
        st = self._stream()
        # ---- forward.  Statistics are merged during the first mini-epoch only (a2c_continuous.py:130-131).
        rms = m.running_mean_std if m.normalize_input else None
        if rms is not None and m.update_stats:
            rms.update(obs, m.stats_group)
        w0, b0 = self.layers[0][0], self.layers[0][1]
        D, C0 = obs.shape[1], w0.shape[0]
        inputs = []
        in_args = None
        if self.fuse_gemm_input:      # the first layer is formed inside the last GEMM's launch (which writes xn and h[0])
            in_args = N.AgInputLayerArgs()
            in_args.struct_size = ctypes.sizeof(N.AgInputLayerArgs)
            in_args.D = D
            in_args.obs_dev = obs.data_ptr()
            in_args.mean_dev = rms.running_mean.data_ptr() if rms is not None else None
            in_args.var_dev = rms.running_var.data_ptr() if rms is not None else None
            in_args.xn_dev = self.xn.data_ptr() if rms is not None else None
            in_args.h1_dev = None if self.recompute_h1 else self.h[0].data_ptr()
            in_args.eps, in_args.clip = (float(rms.epsilon) if rms is not None else 0.0), 5.0
            x = self.xn if rms is not None else obs
        elif self.mfma_input:       # normalise + Linear(D -> 256) + ELU on the matrix cores (exact 3-way split, float32-accurate)
            N.check(lib.ag_mlp_first_layer_prepare(w0.data_ptr(), b0.data_ptr(), D, self.in_image.data_ptr(), st), "ag_mlp_first_layer_prepare")
            N.check(getattr(lib, "ag_mlp_first_layer" + self.sfx)(
                obs.data_ptr(), rms.running_mean.data_ptr() if rms is not None else None,
                rms.running_var.data_ptr() if rms is not None else None, float(rms.epsilon) if rms is not None else 0.0, 5.0,
                self.in_image.data_ptr(), self.xn.data_ptr() if rms is not None else None, self.h[0].data_ptr(), M, D, st),
                "ag_mlp_first_layer")
            x = self.xn if rms is not None else obs
        elif self.fuse_input:       # normalise + Linear(D -> C0) + ELU in one pass
            mean_p = rms.running_mean.data_ptr() if rms is not None else None
            var_p = rms.running_var.data_ptr() if rms is not None else None
            N.check(lib.ag_mlp_input_layer(obs.data_ptr(), mean_p, var_p, w0.data_ptr(), b0.data_ptr(),
                                           self.xn.data_ptr() if rms is not None else None, self.h[0].data_ptr(), M, D, C0,
                                           float(rms.epsilon) if rms is not None else 0.0, 5.0, st), "ag_mlp_input_layer")
            x = self.xn if rms is not None else obs
        else:
            if rms is not None:
                N.check(lib.ag_normalize_rows(obs.data_ptr(), rms.running_mean.data_ptr(), rms.running_var.data_ptr(),
                                              self.xn.data_ptr(), M, D, float(rms.epsilon), 5.0, st), "ag_normalize_rows")
                x = self.xn
            else:
                x = obs
            torch.addmm(b0, x, w0.t(), out=self.h[0])
            F.elu_(self.h[0])
        inputs.append(x)
        x = self.h[0]
        last = len(self.layers) - 1
        heads_done = loss_done = False
        for li_, sg in self.split.items():      # the weights moved in the previous optimizer step
            if in_args is None or li_ != len(self.layers) - 1:
                sg.prepare()
        if in_args is not None:                 # one launch: first-layer image + chain-ordered forward planes + backward planes
            self.split[len(self.layers) - 1].prepare_input_image(w0, b0)
        for li in range(1, len(self.layers)):
            w, b = self.layers[li][0], self.layers[li][1]
            inputs.append(x)
            h = self.h[li]
            sg = self.split.get(li)
            if li == last and self.fuse_heads and sg is not None and self.fuse_gemm_loss:
                # GEMM + heads + PPO loss + the head layer's backward in one launch: dz of this layer comes out directly
                Lp = N.AgLossEpilogue()
                Lp.struct_size = ctypes.sizeof(N.AgLossEpilogue)
                Lp.logstd_dev = m.logstd.data_ptr()
                Lp.actions_dev = mb["actions"].data_ptr()
                Lp.old_neglogp_dev = mb["old_logp_actions"].data_ptr()
                Lp.advantages_dev = mb["advantages"].data_ptr()
                Lp.returns_dev = mb["returns"].data_ptr()
                Lp.old_values_dev = mb["old_values"].data_ptr()
                Lp.old_mu_dev = Lp.new_mu_dev = mb["mu"].data_ptr()
                Lp.old_sigma_dev = Lp.new_sigma_dev = mb["sigma"].data_ptr()
                Lp.heads_dev = None
                Lp.loss_partials_dev = self.loss_partials.data_ptr()
                Lp.dwh_partials_dev = self.head_wg_partials.data_ptr()
                Lp.db_partials_dev = self.bias_partials[li].data_ptr()
                Lp.e_clip, Lp.critic_coef = float(ag.e_clip), float(ag.critic_coef)
                Lp.bounds_loss_coef = float(ag.bounds_loss_coef or 0.0)
                Lp.clip_value = int(bool(ag.clip_value))
                Lp.bound_type = int(BOUND_TYPES[ag.bound_loss_type] if ag.bounds_loss_coef is not None else 0)
                Lp.tile_rows = self.tile_rows
                # (what the three partial tables were sized for at construction: a launch that would write more tiles is refused)
                Lp.partial_tiles = min(self.head_wg_partials.shape[0], self.bias_partials[li].shape[0],
                                       self.loss_partials.shape[0] if self.loss_partials.dim() == 2 else self.wg_blocks)
                dz_out = self.dz[:M * w.shape[0]].view(M, w.shape[0])
                if in_args is not None:
                    fwd = ("ag_split_gemm_input_loss_heads_bwd" + (" (h1 not stored)" if self.recompute_h1 else ""),
                           lambda sg=sg, a=in_args, d=dz_out, b=b, L=Lp: sg.forward_input_loss_heads_bwd(a, M, d, b, ag.heads_w, ag.heads_b, L))
                else:
                    fwd = ("ag_split_gemm_loss_heads_bwd",
                           lambda sg=sg, x=x, d=dz_out, b=b, L=Lp: sg.forward_loss_heads_bwd(x, d, b, ag.heads_w, ag.heads_b, L))
                fwd[1]()
                self.last_launches["forward"] = fwd      # (entry point, replayable launch): bench.py times what the step ran
                heads_done = loss_done = True
            elif li == last and self.fuse_heads and sg is not None:
                if self.fuse_gemm_heads:            # heads formed in the GEMM epilogue; h keeps the bias-free pre-activation
                    sg.forward_elu_heads(x, h, b, ag.heads_w, ag.heads_b, self.heads)
                else:
                    sg.forward(x, h)                # bias-free pre-activation; ag_elu_heads adds the bias
                    N.check(lib.ag_elu_heads(h.data_ptr(), ag.heads_w.data_ptr(), ag.heads_b.data_ptr(),
                                             self.heads.data_ptr(), M, w.shape[0], A + 1, 0, b.data_ptr(), st), "ag_elu_heads")
                heads_done = True
            elif sg is not None:
                sg.forward(x, h, b)
                F.elu_(h)
            elif li == last and self.fuse_heads:      # ELU + the [M,C]x[C,A+1] head product in one pass over z
                # x W^T as an NN product against a transposed copy of W (the libraries' NN kernel for this shape is 13 %
                # faster than TN + bias epilogue); the bias is added by the consumers.  No write-back: the buffer keeps the
                # bias-free pre-activation, the backward pass rebuilds ELU(z + b) on the fly.
                self.wt_last.copy_(w.t())
                torch.mm(x, self.wt_last, out=h)
                N.check(lib.ag_elu_heads(h.data_ptr(), ag.heads_w.data_ptr(), ag.heads_b.data_ptr(), self.heads.data_ptr(),
                                         M, w.shape[0], A + 1, 0, b.data_ptr(), st), "ag_elu_heads")
                heads_done = True
            else:
                torch.addmm(b, x, w.t(), out=h)
                F.elu_(h)
            x = h
        if not heads_done:
            torch.addmm(ag.heads_b, x, ag.heads_w.t(), out=self.heads)
        # ---- loss, d loss / d heads, per-block partial sums
        nb = ctypes.c_int(0)
        bcoef = float(ag.bounds_loss_coef or 0.0)
        bt = BOUND_TYPES[ag.bound_loss_type] if ag.bounds_loss_coef is not None else 0
        logstd = m.logstd
        if loss_done:
            nb.value = self.wg_blocks      # one row of loss sums per GEMM row tile
        else:
            N.check(lib.ag_ppo_loss(self.heads.data_ptr(), logstd.data_ptr(), mb["actions"].data_ptr(),
                                    mb["old_logp_actions"].data_ptr(), mb["advantages"].data_ptr(), mb["returns"].data_ptr(),
                                    mb["old_values"].data_ptr(), mb["mu"].data_ptr(), mb["sigma"].data_ptr(), M, A,
                                    float(ag.e_clip), float(ag.critic_coef), bcoef, int(bool(ag.clip_value)), int(bt),
                                    self.d_heads.data_ptr(), mb["mu"].data_ptr(), mb["sigma"].data_ptr(),
                                    self.loss_partials.data_ptr(), ctypes.byref(nb), st), "ag_ppo_loss")
        if stats_out is not None:
            stats = stats_out
        else:
            stats = self.stats_ring[self.k % self.stats_ring.shape[0]]
            self.k += 1
        # (the loss partials -> d loss / d logstd, head bias gradient, KL slot, logged scalars: one workgroup's work; it rides in
        # the first launch of the partial-sum reductions at the end of the step - `fuse_loss_finalize`, default on - instead of being
        # a launch of its own here)
        fin_args = (self.loss_partials.data_ptr(), nb.value, M, A, logstd.data_ptr(), float(ag.entropy_coef), float(ag.critic_coef),
                    bcoef, logstd.grad.data_ptr(), ag.heads_b_grad.data_ptr(), ag.flat_grad[-1:].data_ptr(), stats.data_ptr())
        fin_late = self.use_sum_multi and self.fuse_finalize
        if not fin_late:
            N.check(lib.ag_ppo_loss_finalize(*fin_args, st), "ag_ppo_loss_finalize")
        # ---- backward.  The head's dX = d_heads Wh and its weight gradient are formed inside the last layer's ELU' pass;
        # the first layer's weight/bias gradients are formed inside ITS ELU' pass (dz of layer 0 is never stored).
        dh = None
        for li in range(last, -1, -1):
            w = self.layers[li][0]
            h, xin = self.h[li], inputs[li]
            C, K = w.shape
            dz = self.dz[:M * C].view(M, C)
            parts = self.bias_partials[li]
            if li == last and loss_done:
                pass      # dz, the head weight-gradient partials and this layer's bias partials came out of the forward launch
            elif li == last:
                N.check(lib.ag_heads_bwd_elu_wgrad(self.d_heads.data_ptr(), ag.heads_w.data_ptr(), h.data_ptr(), dz.data_ptr(),
                                                   parts.data_ptr(), self.head_wg_partials.data_ptr(), M, C, A + 1,
                                                   int(heads_done), self.layers[li][1].data_ptr() if heads_done else None, st),
                        "ag_heads_bwd_elu_wgrad")
            elif li == 0 and self.fuse_input_wgrad:
                N.check(lib.ag_elu_bwd_input_wgrad(dh.data_ptr(), h.data_ptr(), xin.data_ptr(),
                                                   self.wgrad_partials[0].data_ptr(), parts.data_ptr(), M, C, K, st),
                        "ag_elu_bwd_input_wgrad")
                break
            else:
                N.check(lib.ag_elu_bwd_bias(dh.data_ptr(), h.data_ptr(), dz.data_ptr(), parts.data_ptr(), M, C, st),
                        "ag_elu_bwd_bias")
            if li == 1 and self.recompute_h1:
                # X operand (h1) and ELU'(h1) are produced from the network inputs + the first-layer image of the forward launch
                wp, img, x0 = self.wgrad_partials[1], self.split[1].in_image, inputs[0]
                wg = ("ag_split_wgrad_input", lambda dz=dz, wp=wp, img=img, x0=x0: N.check(getattr(lib, "ag_split_wgrad_input" + self.sfx)(
                    dz.data_ptr(), x0.data_ptr(), img.data_ptr(), wp.data_ptr(), M, C, K, x0.shape[1], wp.shape[0], st),
                    "ag_split_wgrad_input"))
                dx = ("ag_split_gemm_input_wgrad_recompute", lambda dz=dz, img=img, x0=x0: N.check(
                    getattr(lib, "ag_split_gemm_input_wgrad_recompute" + self.sfx)(dz.data_ptr(), self.split[1].bwd.data_ptr(), img.data_ptr(), x0.data_ptr(),
                                                            self.wgrad_partials[0].data_ptr(), self.bias_partials[0].data_ptr(), M, 256,
                                                            256, x0.shape[1], self.tile_rows, st), "ag_split_gemm_input_wgrad_recompute"))
                wg[1]()
                dx[1]()
                self.last_launches["wgrad"], self.last_launches["dx"] = wg, dx
                break
            if li in self.split_wgrad:
                wp = self.wgrad_partials[li]
                wg = ("ag_split_wgrad", lambda dz=dz, xin=xin, wp=wp: N.check(getattr(lib, "ag_split_wgrad" + self.sfx)(
                    dz.data_ptr(), xin.data_ptr(), wp.data_ptr(), M, C, K, wp.shape[0], st), "ag_split_wgrad"))
                wg[1]()
                if li == last:
                    self.last_launches["wgrad"] = wg
            else:
                torch.bmm(dz.view(S, M // S, C).transpose(1, 2), xin.view(S, M // S, K), out=self.wgrad_partials[li])
            if li == 1 and self.fuse_gemm_input_wgrad:
                dx = ("ag_split_gemm_input_wgrad", lambda dz=dz, x0=inputs[0]: self.split[1].backward_input_wgrad(
                    dz, self.h[0], x0, self.wgrad_partials[0], self.bias_partials[0]))
                dx[1]()
                self.last_launches["dx"] = dx
                break
            if li > 0:
                dh = self.dh[:M * K].view(M, K)
                if li in self.split:
                    self.split[li].backward_input(dz, dh)
                else:
                    torch.mm(dz, w, out=dh)
        # ---- every partial-sum reduction (bias / weight gradients of all layers + the head) in two launches
        if fin_late:
            N.check(lib.ag_sum_rows_multi_finalize(self._jobs, len(self._jobs), self.sum_scratch.data_ptr(), self.sum_scratch.numel(),
                                                   *fin_args, st), "ag_sum_rows_multi_finalize")
        elif self.use_sum_multi:
            N.check(lib.ag_sum_rows_multi(self._jobs, len(self._jobs), self.sum_scratch.data_ptr(), self.sum_scratch.numel(), st),
                    "ag_sum_rows_multi")
        else:
            for src, dst in self._sum_pairs:
                torch.sum(src.view(-1, dst.numel()), 0, out=dst.view(-1))
        return stats


class FusedRolloutStep:
    """One step of A2CBase.play_steps (lib/agent/a2c_base.py:651-695) as three launches:
    ag_mlp_input_layer -> ag_split_gemm_elu_heads -> ag_step_rollout_fused (policy sampling + env step + reward / episode
    accounting; `fuse_rollout_tail: false` = the separate ag_policy_sample -> ag_step_rollout -> ag_rollout_account launches).  The action noise is Philox-based and counter-keyed, so the
    captured hipGraph of the whole rollout draws fresh noise on every replay (begin_rollout bumps the device counter)."""

    @staticmethod
    def supported(agent):
        from airgym_amd.lib.utils.tr_helpers import DefaultRewardsShaper
        m = agent.model
        sh = agent.rewards_shaper
        return (str(agent.ppo_device).startswith("cuda") and agent.config.get("use_fused_rollout", True)
                and agent._hip_env is not None and agent._hip_env.task in ("hovering", "tracking")
                and agent._fused_loss_ok() and not m.dict_obs
                and m.actor_mlp.activation_name == "elu" and getattr(agent, "heads_w", None) is not None
                and agent.actions_num in (4, 5) and agent.clip_actions
                and bool((agent.actions_low == -1).all()) and bool((agent.actions_high == 1).all())
                and isinstance(sh, DefaultRewardsShaper)
                and all(l.weight.shape[0] % 4 == 0 and l.weight.shape[0] <= 1024 and 256 % (l.weight.shape[0] // 4) == 0
                        for l in m.actor_mlp.layers))

    def __init__(self, agent):
        self.agent = agent
        self.lib = N.load()
        m, dev = agent.model, agent.ppo_device
        n = self.n = agent.num_actors * agent.num_agents
        self.A = agent.actions_num
        self.layers = [(l.weight, l.bias) for l in m.actor_mlp.layers]
        f = dict(dtype=torch.float32, device=dev)
        D = self.layers[0][0].shape[1]
        self.xn = torch.empty(n, D, **f)
        self.h = [torch.empty(n, w.shape[0], **f) for w, _ in self.layers]
        self.heads = torch.empty(n, self.A + 1, **f)
        self.env_actions = torch.empty(n, self.A, **f)
        C0, Cl = self.layers[0][0].shape[0], self.layers[-1][0].shape[0]
        self.fuse_input = (D * C0 + 64 * D) * 4 <= 64 * 1024
        self.fuse_heads = len(self.layers) >= 2 and 64 <= Cl <= 256 and (Cl & (Cl - 1)) == 0
        self.wt_last = torch.empty(self.layers[-1][0].shape[1], Cl, **f)
        self.bf16 = bool(getattr(agent, "mixed_precision", False))
        self.split = {li: SplitGemm256(w, backward=False, bf16=self.bf16) for li, (w, _) in enumerate(self.layers)
                      if li >= 1 and SplitGemm256.applies(w, agent.config)}
        self.fuse_gemm_heads = bool(agent.config.get("fuse_gemm_heads", True)) and self.A + 1 in (5, 6)
        # the whole policy forward as ONE launch with the activations in registers (csrc/mlp_chain.hip): [D -> 256 -> 256] trunks
        self.chain = None
        if (bool(agent.config.get("use_mlp_chain", True)) and len(self.layers) == 2 and self.split
                and tuple(self.layers[0][0].shape) == (256, D) and self.lib.ag_mlp_chain_supported(D, 256, self.A + 1)):
            self.chain = torch.empty(self.lib.ag_mlp_chain_image_bytes(D), dtype=torch.uint8, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        # per-step, per-block episode sums; reduced over blocks ONCE per rollout (end_rollout)
        self.acct_partials = torch.zeros(agent.horizon_length, self.lib.ag_rollout_account_blocks(n), 4,
                                         dtype=torch.float64, device=dev)
        self.seed = (int(agent.params.get("seed", 0) or 0) * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) & 0xFFFFFFFFFFFFFFFF
        self.id_offset = agent.global_rank * n
        # policy sampling + env step + reward / episode accounting in ONE launch (ag_step_rollout_fused): the rollout step is
        # input layer -> GEMM (+heads) -> that launch
        self.fuse_tail = bool(agent.config.get("fuse_rollout_tail", True))
        if self.fuse_tail:
            self.acct_partials = torch.zeros(agent.horizon_length, self.lib.ag_term_sum_tiles(n), 4, dtype=torch.float64, device=dev)
        self._tails = {}

    @property
    def launches_per_step(self):
        if self.chain is not None:
            return 1 + (1 if self.fuse_tail else 3)
        gemms = len(self.layers) - 1
        return 1 + gemms + (0 if (self.fuse_gemm_heads and self.split) else 1) + (1 if self.fuse_tail else 3)

    def _tail(self, slot):
        """ag_rollout_tail of rollout slot `slot` (pointers are fixed for the life of the agent: built once per slot)"""
        t = self._tails.get(slot)
        if t is not None:
            return t
        ag, m = self.agent, self.agent.model
        vms = m.value_mean_std if m.normalize_value else None
        sh = ag.rewards_shaper
        t = N.AgRolloutTail()
        t.struct_size = ctypes.sizeof(N.AgRolloutTail)
        t.heads_dev = self.heads.data_ptr()
        t.logstd_dev = m.logstd.data_ptr()
        t.vmean_dev = vms.running_mean.data_ptr() if vms is not None else None
        t.vvar_dev = vms.running_var.data_ptr() if vms is not None else None
        t.veps = float(vms.epsilon) if vms is not None else 0.0
        t.seed = self.seed
        t.counter_dev = self.counter.data_ptr()
        t.horizon, t.slot, t.id_offset = ag.horizon_length, slot, self.id_offset
        t.actions_dev = ag.actions_buf[slot].data_ptr()
        t.neglogp_dev = ag.neglogpacs_buf[slot].data_ptr()
        t.values_dev = ag.values_buf[slot].data_ptr()
        t.mus_dev = ag.mus_buf[slot].data_ptr()
        t.sigmas_dev = ag.sigmas_buf[slot].data_ptr()
        t.scale, t.shift = float(sh.scale_value), float(sh.shift_value)
        t.min_val, t.max_val = float(sh.min_val), float(sh.max_val)
        t.log_val, t.gamma = int(bool(sh.log_val)), float(ag.gamma)
        t.bootstrap_timeouts = int(bool(ag.value_bootstrap))
        t.shaped_dev = ag.rewards_buf[slot].data_ptr()
        t.cur_rew_dev = ag.current_rewards.data_ptr()
        t.cur_shaped_dev = ag.current_shaped_rewards.data_ptr()
        t.cur_len_dev = ag.current_lengths.data_ptr()
        t.partials_dev = self.acct_partials[slot].data_ptr()
        self._tails[slot] = t
        return t

    def begin_rollout(self):
        self.counter.add_(1)      # captured with the rollout graph: every replay advances the noise counter
        self.refresh_weights()

    def refresh_weights(self):
        """Transposed copy of the last hidden layer's weight for the NN-form GEMM (parameters are constant in a rollout)."""
        if self.chain is not None:
            (w0, b0), (w1, _) = self.layers
            N.check(self.lib.ag_mlp_chain_prepare(w0.data_ptr(), b0.data_ptr(), w0.shape[1], w1.data_ptr(),
                                                  self.agent.heads_w.data_ptr(), self.A + 1, self.chain.data_ptr(), self._stream()),
                    "ag_mlp_chain_prepare")
            return
        if self.fuse_heads:
            self.wt_last.copy_(self.layers[-1][0].t())
        for sg in self.split.values():
            sg.prepare()

    def end_rollout(self):
        torch.sum(self.acct_partials, 1, out=self.agent.ep_stats)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.agent.ppo_device).cuda_stream)

    @torch.no_grad()
    def heads_of(self, obs):
        """Policy forward on [n, D] observations -> heads [n, A+1] (mu | normalised value)."""
        ag, lib, m = self.agent, self.lib, self.agent.model
        n, A, st = self.n, self.A, self._stream()
        rms = m.running_mean_std if m.normalize_input else None
        w0, b0 = self.layers[0]
        D, C0 = obs.shape[1], w0.shape[0]
        if self.chain is not None:
            N.check(getattr(lib, "ag_mlp_chain_forward" + ("_bf16" if self.bf16 else ""))(
                obs.data_ptr(), rms.running_mean.data_ptr() if rms is not None else None,
                                             rms.running_var.data_ptr() if rms is not None else None,
                                             float(rms.epsilon) if rms is not None else 0.0, 5.0, self.chain.data_ptr(),
                                             self.layers[1][1].data_ptr(), ag.heads_b.data_ptr(), self.heads.data_ptr(), None, None,
                                             None, n, D, A + 1, st), "ag_mlp_chain_forward")
            return self.heads
        if self.fuse_input:
            N.check(lib.ag_mlp_input_layer(obs.data_ptr(), rms.running_mean.data_ptr() if rms is not None else None,
                                           rms.running_var.data_ptr() if rms is not None else None, w0.data_ptr(),
                                           b0.data_ptr(), self.xn.data_ptr() if rms is not None else None,
                                           self.h[0].data_ptr(), n, D, C0, float(rms.epsilon) if rms is not None else 0.0,
                                           5.0, st), "ag_mlp_input_layer")
        else:
            x = obs
            if rms is not None:
                N.check(lib.ag_normalize_rows(obs.data_ptr(), rms.running_mean.data_ptr(), rms.running_var.data_ptr(),
                                              self.xn.data_ptr(), n, D, float(rms.epsilon), 5.0, st), "ag_normalize_rows")
                x = self.xn
            torch.addmm(b0, x, w0.t(), out=self.h[0])
            F.elu_(self.h[0])
        x = self.h[0]
        last = len(self.layers) - 1
        heads_done = False
        for li in range(1, len(self.layers)):
            w, b = self.layers[li]
            h = self.h[li]
            sg = self.split.get(li)
            if li == last and self.fuse_heads and sg is not None:
                if self.fuse_gemm_heads:                  # planes refreshed once per rollout (refresh_weights)
                    sg.forward_elu_heads(x, h, b, ag.heads_w, ag.heads_b, self.heads)
                else:
                    sg.forward(x, h)
                    N.check(lib.ag_elu_heads(h.data_ptr(), ag.heads_w.data_ptr(), ag.heads_b.data_ptr(),
                                             self.heads.data_ptr(), n, w.shape[0], A + 1, 0, b.data_ptr(), st), "ag_elu_heads")
                heads_done = True
            elif sg is not None:
                sg.forward(x, h, b)
                F.elu_(h)
            elif li == last and self.fuse_heads:
                torch.mm(x, self.wt_last, out=h)          # wt_last refreshed once per rollout (begin_rollout)
                N.check(lib.ag_elu_heads(h.data_ptr(), ag.heads_w.data_ptr(), ag.heads_b.data_ptr(), self.heads.data_ptr(),
                                         n, w.shape[0], A + 1, 0, b.data_ptr(), st), "ag_elu_heads")
                heads_done = True
            else:
                torch.addmm(b, x, w.t(), out=h)
                F.elu_(h)
            x = h
        if not heads_done:
            torch.addmm(ag.heads_b, x, ag.heads_w.t(), out=self.heads)
        return self.heads

    @torch.no_grad()
    def step(self, slot):
        ag, lib, m = self.agent, self.lib, self.agent.model
        n, A = self.n, self.A
        self.heads_of(ag.obs_buf[slot])
        if self.fuse_tail:
            ag._hip_env.step_rollout_fused(self._tail(slot), ag.obs_buf[slot + 1], ag.raw_rewards_buf[slot], ag.dones_buf[slot + 1],
                                           ag._term_tiles[slot] if ag._term_tiles is not None else None)
            return
        st = self._stream()
        vms = m.value_mean_std if m.normalize_value else None
        N.check(lib.ag_policy_sample(self.heads.data_ptr(), m.logstd.data_ptr(),
                                     vms.running_mean.data_ptr() if vms is not None else None,
                                     vms.running_var.data_ptr() if vms is not None else None,
                                     float(vms.epsilon) if vms is not None else 0.0, self.seed, self.counter.data_ptr(),
                                     ag.horizon_length, slot, self.id_offset, ag.actions_buf[slot].data_ptr(),
                                     ag.neglogpacs_buf[slot].data_ptr(), ag.values_buf[slot].data_ptr(),
                                     ag.mus_buf[slot].data_ptr(), ag.sigmas_buf[slot].data_ptr(),
                                     self.env_actions.data_ptr(), n, A, st), "ag_policy_sample")
        env = ag._hip_env
        # env kernel in its rollout form: u8 done flags and per-tile reward-term sums straight into the rollout buffers
        env.step_rollout(self.env_actions, ag.obs_buf[slot + 1], ag.raw_rewards_buf[slot], ag.dones_buf[slot + 1],
                         ag._term_tiles[slot] if ag._term_tiles is not None else None)
        sh = ag.rewards_shaper
        tmo = env.time_out_buf if ag.value_bootstrap else None
        N.check(lib.ag_rollout_account(ag.raw_rewards_buf[slot].data_ptr(), ag.dones_buf[slot + 1].data_ptr(),
                                       tmo.data_ptr() if tmo is not None else None,
                                       ag.values_buf[slot].data_ptr() if tmo is not None else None,
                                       float(sh.scale_value), float(sh.shift_value), float(sh.min_val), float(sh.max_val),
                                       int(bool(sh.log_val)), float(ag.gamma), ag.rewards_buf[slot].data_ptr(),
                                       ag.current_rewards.data_ptr(), ag.current_shaped_rewards.data_ptr(),
                                       ag.current_lengths.data_ptr(), self.acct_partials[slot].data_ptr(), n, st),
                "ag_rollout_account")

    @torch.no_grad()
    def gae(self, last_values):
        """advantages, returns [H, n, 1] from the rollout buffers (a2c_base.py:463-478)."""
        ag = self.agent
        H = ag.horizon_length
        advs = torch.empty_like(ag.values_buf)
        rets = torch.empty_like(ag.values_buf)
        lv = last_values.contiguous()
        N.check(self.lib.ag_gae(ag.rewards_buf.data_ptr(), ag.values_buf.data_ptr(), ag.dones_buf.data_ptr(), lv.data_ptr(),
                                float(ag.gamma), float(ag.tau), advs.data_ptr(), rets.data_ptr(), H, self.n, self._stream()),
                "ag_gae")
        return advs, rets
