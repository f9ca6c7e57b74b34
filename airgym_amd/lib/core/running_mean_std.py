"""RunningMeanStd - input / value normaliser (reference: lib/core/running_mean_std.py:8-81).

Float64 statistics, parallel-variance merge of batch moments, clamp to +-5 after normalisation.
State-dict keys (`running_mean`, `running_var`, `count`) match the reference so checkpoints are
interchangeable.  The statistics are updated IN PLACE so that a captured hipGraph keeps reading the
same device addresses."""
import torch
import torch.nn as nn

from airgym_amd.lib.core import collectives


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        self.insize = insize
        self.epsilon = epsilon
        self.norm_only = norm_only
        if per_channel:
            raise NotImplementedError("per_channel statistics are only used by the image pipelines")
        self.axis = [0]
        self.register_buffer("running_mean", torch.zeros(insize, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(insize, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))

    @torch.no_grad()
    def update(self, x, group=None, weights=None, index=None):
        """Merge the moments of batch x ([B, ...]).  With `group` (torch.distributed) the batch moments
        are first combined across ranks so that every replica keeps identical statistics (the reference
        lets them drift).  weights [B] (optional): row i stands for weights[i] identical rows
        (frame de-duplication of the depth images): the moments are those of the expanded batch.
        index [B] (optional, int64): the batch is x[index] (read in place)."""
        if weights is not None or index is not None:
            if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous():
                n, mean, var = self._weighted_moments_hip(x, weights, index)
            else:
                if index is not None:
                    x = x.index_select(0, index)
                if weights is None:
                    weights = torch.ones(x.shape[0], device=x.device)
                wv = weights.to(device=x.device, dtype=torch.float64).view(-1, *([1] * (x.dim() - 1)))
                n = wv.sum()
                mean = (wv * x).sum(0) / n
                var = (wv * (x - mean) ** 2).sum(0) / torch.clamp(n - 1.0, min=1.0)      # unbiased, like x.var(0) of the expansion
            if group is not None:
                import torch.distributed as dist
                if dist.get_world_size(group) > 1:
                    packed = torch.cat(((mean * n).reshape(-1), ((var * (n - 1.0)) + n * mean * mean).reshape(-1), n.reshape(1)))
                    collectives.all_reduce(packed, "normaliser_moments", group=group)
                    k = mean.numel()
                    n = packed[-1]
                    mean = (packed[:k] / n).view_as(mean)
                    var = ((packed[k:2 * k].view_as(var) - n * mean * mean) / torch.clamp(n - 1.0, min=1.0))
            return self.merge_moments(mean, var, n)
        if group is None and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[0] >= 2 \
                and 0 < x[0].numel() <= 256:
            return self._update_hip(x)
        batch_count = float(x.size()[0])
        mean = x.mean(self.axis).double()
        var = x.var(self.axis).double()
        if group is not None:
            import torch.distributed as dist
            ws = dist.get_world_size(group)
            if ws > 1:
                # equal batch sizes per rank: pooled mean / unbiased variance from per-rank moments
                stats = torch.stack((mean, var * (batch_count - 1) + batch_count * mean * mean))
                collectives.all_reduce(stats, "normaliser_moments", group=group)
                tot = batch_count * ws
                mean = stats[0] / ws
                var = (stats[1] - tot * mean * mean) / (tot - 1)
                batch_count = tot
        delta = mean - self.running_mean
        tot_count = self.count + batch_count
        new_mean = self.running_mean + delta * batch_count / tot_count
        m2 = self.running_var * self.count + var * batch_count + delta ** 2 * self.count * batch_count / tot_count
        self.running_mean.copy_(new_mean)
        self.running_var.copy_(m2 / tot_count)
        self.count.copy_(tot_count)

    @torch.no_grad()
    def merge_moments(self, mean, var, batch_count):
        """Merge pre-computed batch moments (float64 tensors `mean`, `var` of shape insize and a 0-d `batch_count`) with the
        same parallel-variance formula as update(); a zero count leaves the statistics unchanged."""
        delta = mean - self.running_mean
        tot_count = self.count + batch_count
        new_mean = self.running_mean + delta * batch_count / tot_count
        m2 = self.running_var * self.count + var * batch_count + delta ** 2 * self.count * batch_count / tot_count
        self.running_mean.copy_(new_mean)
        self.running_var.copy_(m2 / tot_count)
        self.count.copy_(tot_count)

    def _weighted_moments_hip(self, x, weights, index):
        """(n, mean, unbiased var) of the weighted / indexed batch in ONE pass over the rows (`ag_weighted_moments`: float64 sums of
        w x and w x^2 per element; images are 25 440 elements wide, so the torch formulation above costs two passes over float64
        temporaries of the whole batch)."""
        import ctypes

        from airgym_amd import _native as N
        lib = N.load()
        rows = x.shape[0] if index is None else index.shape[0]
        D = x[0].numel()
        if weights is not None:
            weights = weights.to(device=x.device, dtype=torch.float32).contiguous()
        if index is not None:
            index = index.to(device=x.device, dtype=torch.long).contiguous()
        partial = torch.empty(lib.ag_weighted_moments_chunks(), 2, D, dtype=torch.float64, device=x.device)
        N.check(lib.ag_weighted_moments(x.data_ptr(), index.data_ptr() if index is not None else None,
                                        weights.data_ptr() if weights is not None else None, rows, D, partial.data_ptr(),
                                        ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "ag_weighted_moments")
        s = partial.sum(0)
        n = weights.double().sum() if weights is not None else torch.tensor(float(rows), dtype=torch.float64, device=x.device)
        mean = s[0] / n
        var = torch.clamp(s[1] - n * mean * mean, min=0.0) / torch.clamp(n - 1.0, min=1.0)
        return n, mean.view(x.shape[1:]), var.view(x.shape[1:])

    def _update_hip(self, x):
        """Same merge in two HIP launches (`ag_rms_update`): float64 column moments, then the in-place merge."""
        import ctypes

        from airgym_amd import _native as N
        lib = N.load()
        D = x[0].numel()
        if getattr(self, "_scratch", None) is None or self._scratch.device != x.device:
            self._scratch = torch.empty(lib.ag_rms_scratch_doubles(D), dtype=torch.float64, device=x.device)
        count = self.count.view(1)
        N.check(lib.ag_rms_update(x.data_ptr(), x.shape[0], D, self.running_mean.data_ptr(), self.running_var.data_ptr(),
                                  count.data_ptr(), self._scratch.data_ptr(),
                                  ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "ag_rms_update")

    def forward(self, input, denorm=False, mask=None):
        if self.training:
            self.update(input)
        mean = self.running_mean.float()
        var = self.running_var.float()
        if denorm:
            y = torch.clamp(input, min=-5.0, max=5.0)
            return torch.sqrt(var + self.epsilon) * y + mean
        if self.norm_only:
            return input / torch.sqrt(var + self.epsilon)
        y = (input - mean) / torch.sqrt(var + self.epsilon)
        return torch.clamp(y, min=-5.0, max=5.0)


class RunningMeanStdObs(nn.Module):
    """One RunningMeanStd per key of a dict observation (reference: running_mean_std.py:83-92)."""

    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        assert isinstance(insize, dict)
        super().__init__()
        self.running_mean_std = nn.ModuleDict({k: RunningMeanStd(v, epsilon, per_channel, norm_only) for k, v in insize.items()})

    def forward(self, input, denorm=False):
        return {k: self.running_mean_std[k](v, denorm) for k, v in input.items()}
