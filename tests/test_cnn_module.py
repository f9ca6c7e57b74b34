"""CNNFeatureExtractor (reference: lib/network/cnn.py:3-33) as a module: state that is not model state must not leak into copies."""
import copy
import pickle

import torch

from airgym_amd.lib.network.cnn import CNNFeatureExtractor


class _Unpicklable:
    """stands in for the torch.cuda.Event the gamma guard holds after a training forward on the GPU"""

    def __reduce__(self):
        raise TypeError("cannot pickle Event")


def test_module_with_a_live_gamma_guard_can_be_copied_and_pickled():
    m = CNNFeatureExtractor(30)
    m._gamma_ratio_event, m._gamma_ratio_host = _Unpicklable(), torch.zeros(1)
    c = copy.deepcopy(m)
    p = pickle.loads(pickle.dumps(m))
    for other in (c, p):
        assert other._gamma_ratio_event is None and other._gamma_ratio_host is None       # the copy decides afresh
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), other.state_dict().values()))
    assert m._gamma_ratio_event is not None                                                # the original keeps its guard
    x = torch.randn(2, 1, 212, 120)
    m.eval(); c.eval()
    assert torch.equal(m(x), c(x))
