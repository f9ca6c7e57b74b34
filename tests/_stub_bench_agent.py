"""Test double for bench.py's launcher test: the PRODUCT agent (A2CAgent, its data-parallel path and its one all-reduce per
optimizer step) on the oracle-backed CPU vec-env of tests/_stub_env.py, so that `python bench.py --gpus 2 --device cpu`
exercises the self-launcher, the barriers, the rccl probe and the JSON line under gloo without a GPU."""
from airgym_amd.lib.agent.a2c_continuous import A2CAgent
from tests import _stub_env


class StubAgent(A2CAgent):
    def __init__(self, name, params):
        _stub_env.register()
        c = params["config"]
        c["env_name"] = "oracle_hovering"
        c["env_config"] = {"ctl_mode": "rate", "seed": 3}
        c["device"] = "cpu"
        c["use_hip_graph"] = False
        c["train_dir"] = __import__("tempfile").mkdtemp(prefix="airgym_bench_stub_")
        params["network"]["mlp"]["units"] = [32, 32]
        super().__init__(name, params)


class BrokenCollectiveAgent(StubAgent):
    """A run whose first collective fails on every rank (what an RCCL / IPC failure looks like to bench.py): the launcher must
    retry once (on a GPU box: with the other IPC setting) and, when nothing works, still print ONE parseable line."""

    def broadcast_parameters(self):
        raise RuntimeError("hipIpcGetMemHandle: invalid argument (simulated)")
