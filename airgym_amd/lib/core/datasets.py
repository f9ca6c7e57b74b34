"""PPODataset - contiguous, UNSHUFFLED minibatch slices of the flattened rollout (reference:
lib/core/datasets.py:6-47, quirk Q12) including the mu/sigma write-back after every minibatch."""


class PPODataset:
    def __init__(self, batch_size, minibatch_size, is_discrete, device):
        self.batch_size = batch_size
        self.minibatch_size = minibatch_size
        self.device = device
        self.length = self.batch_size // self.minibatch_size
        self.is_discrete = is_discrete
        self.values_dict = None
        self.last_range = (0, 0)

    def update_values_dict(self, values_dict):
        self.values_dict = values_dict

    def update_mu_sigma(self, mu, sigma):
        start, end = self.last_range
        self.values_dict["mu"][start:end] = mu
        self.values_dict["sigma"][start:end] = sigma

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        start = idx * self.minibatch_size
        end = (idx + 1) * self.minibatch_size
        self.last_range = (start, end)
        out = {}
        for k, v in self.values_dict.items():
            if v is None:
                continue
            if isinstance(v, dict):
                out[k] = {}
                for kd, vd in v.items():
                    piece = vd[start:end]
                    if isinstance(piece, dict):      # de-duplicated frames: {image, image_inverse, image_counts} of this slice
                        out[k].update(piece)
                    else:
                        out[k][kd] = piece
            else:
                out[k] = v[start:end]
        return out
