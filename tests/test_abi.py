"""The C-ABI library loads without a GPU and exports every symbol include/airgym_hip.h declares.
No compute entry point is called here (that needs a device: tests/test_gpu_*.py)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "airgym_hip.h")


@pytest.fixture(scope="module")
def lib():
    from airgym_amd import _native
    return _native.load()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ag_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/airgym_hip.h but not exported"


def test_binding_table_covers_header():
    from airgym_amd import _native
    assert sorted(n for n, _, _ in _native.SYMBOLS) == declared_symbols()


def test_struct_sizes_match_header(lib):
    from airgym_amd import _native
    # ag_arena_bytes validates struct_size: a wrong ctypes layout returns 0
    cfg = _native.AgConfig()
    cfg.struct_size = ctypes.sizeof(_native.AgConfig)
    cfg.task, cfg.ctl_mode, cfg.num_envs, cfg.dt = 0, 3, 1000, 0.01
    n = lib.ag_arena_bytes(ctypes.byref(cfg))
    assert n > 1000 * (13 * 4 + 18 * 4)
    cfg.struct_size -= 4
    assert lib.ag_arena_bytes(ctypes.byref(cfg)) == 0
    assert b"struct_size" in lib.ag_last_error()


def test_pure_host_queries(lib):
    assert lib.ag_version() == 100
    assert lib.ag_num_obs(0) == 18 and lib.ag_num_obs(1) == 48 and lib.ag_num_obs(7) < 0
    assert lib.ag_num_actions(2) == 5 and lib.ag_num_actions(3) == 4 and lib.ag_num_actions(9) < 0
    assert lib.ag_default_episode_length(0, 0.01) == 2400   # int(24 / 0.01), hovering.py:48
    assert lib.ag_default_episode_length(1, 0.01) == 3600


def test_invalid_configs_are_errors_not_prints(lib):
    """hovering.py:122-123 only prints 'Mode Error!'; the library must refuse."""
    from airgym_amd import _native
    cfg = _native.AgConfig()
    cfg.struct_size = ctypes.sizeof(_native.AgConfig)
    cfg.task, cfg.ctl_mode, cfg.num_envs, cfg.dt = 0, 11, 64, 0.01
    h = ctypes.c_void_p()
    assert lib.ag_create(ctypes.byref(cfg), None, ctypes.byref(h)) == _native.AG_ERR_UNKNOWN_CTL
    cfg.ctl_mode, cfg.task = 3, 5
    assert lib.ag_create(ctypes.byref(cfg), None, ctypes.byref(h)) == _native.AG_ERR_UNKNOWN_TASK
    cfg.task, cfg.num_envs = 0, 0
    assert lib.ag_create(ctypes.byref(cfg), None, ctypes.byref(h)) == -1
    assert h.value is None


def test_product_has_no_cpu_fallback():
    """HipEnvHandle refuses to run without a HIP device instead of silently computing elsewhere."""
    import torch
    from airgym_amd.hip_env import HipEnvHandle
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        HipEnvHandle("hovering", "rate", 64, device="cuda:0")
    with pytest.raises(RuntimeError):
        HipEnvHandle("hovering", "rate", 64, device="cpu")


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; import airgym_amd, airgym_amd.hip_env, airgym_amd._native; "
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; assert not bad, bad")
    subprocess.check_call([sys.executable, "-c", code], cwd=REPO)
    pkg = os.path.join(REPO, "airgym_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(root, f)
