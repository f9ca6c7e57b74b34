"""The C-ABI library loads without a GPU and exports every symbol include/airgym_hip.h declares.
No compute entry point is called here (that needs a device: tests/test_gpu_*.py)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "airgym_hip.h")
DEBUG_HEADER = os.path.join(REPO, "include", "airgym_hip_debug.h")      # experiments build only


@pytest.fixture(scope="module")
def lib():
    from airgym_amd import _native
    return _native.load()


def declared_symbols(header=HEADER):
    src = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ag_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/airgym_hip.h but not exported"


def test_binding_table_covers_header():
    from airgym_amd import _native
    assert sorted(n for n, _, _ in _native.SYMBOLS) == declared_symbols()
    assert sorted(n for n, _, _ in _native.DEBUG_SYMBOLS) == declared_symbols(DEBUG_HEADER)


def test_shipped_library_has_no_experiment_entry_points(lib):
    """include/airgym_hip_debug.h (launch-floor probes, scheduling variants, render-phase skipping) lives in the experiments
    build only: the shipped library exports none of it, and nothing named ag_debug_* / ag_set_launch_params at all."""
    for name in declared_symbols(DEBUG_HEADER) + ["ag_set_launch_params"]:
        assert not hasattr(lib, name), f"{name} leaked into the shipped library"
    exp = os.path.join(REPO, "airgym_amd", "_native", "libairgym_hip_exp.so")
    shipped = os.path.join(REPO, "airgym_amd", "_native", "libairgym_hip.so")
    if os.path.exists(exp) and os.path.getmtime(exp) >= os.path.getmtime(shipped) - 600:      # a stale tools build proves nothing
        e = ctypes.CDLL(exp)
        for name in declared_symbols(DEBUG_HEADER) + declared_symbols():
            assert hasattr(e, name), f"{name} missing from the experiments build"


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


def test_ppo_kernel_entry_points_validate_arguments_before_any_launch(lib):
    """Argument validation of the PPO-loop entry points happens on the host, before anything is launched: callable here.
    (error codes: include/airgym_hip.h - AG_ERR_INVALID_ARG = -1, AG_ERR_UNSUPPORTED = -2 ...)"""
    from airgym_amd import _native as N
    inval = lib.ag_ppo_loss_finalize(None, 1, 1, 4, None, 0.0, 0.0, 0.0, None, None, None, None, None)
    assert inval != 0
    unsupported = None
    buf = (ctypes.c_float * 64)()
    dbl = (ctypes.c_double * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    pd = ctypes.cast(dbl, ctypes.c_void_p)
    # host-side pure queries
    assert lib.ag_ppo_loss_num_sums() == 4 + 5 + 6 + 1 and lib.ag_ppo_loss_max_blocks() >= 256
    assert lib.ag_wgrad_rows_per_block(0) > 0 and lib.ag_input_wgrad_rows(18) > 0 and lib.ag_input_wgrad_rows(48) == 0
    assert lib.ag_sum_rows_groups() >= 16 and lib.ag_rollout_account_blocks(65536) == 256
    assert lib.ag_rms_scratch_doubles(18) > 0 and lib.ag_adam_state_bytes() >= 32
    # shape / pointer errors are return codes, not crashes
    assert lib.ag_normalize_rows(None, pd, pd, p, 4, 18, 1e-5, 5.0, None) == inval
    assert lib.ag_normalize_rows(p, pd, pd, p, 4, 100000, 1e-5, 5.0, None) not in (0, inval)          # D too large
    unsupported = lib.ag_normalize_rows(p, pd, pd, p, 4, 100000, 1e-5, 5.0, None)
    assert lib.ag_elu_heads(p, p, p, p, 8, 48, 5, 0, None, None) == unsupported                     # C not a power of two >= 64
    assert lib.ag_elu_heads(p, p, p, p, 8, 256, 7, 0, None, None) == unsupported                    # A1 not in {5, 6}
    assert lib.ag_elu_bwd_input_wgrad(p, p, p, p, p, 8, 256, 48, None) == unsupported               # D = 48 keeps the unfused path
    assert lib.ag_mlp_input_layer(p, pd, None, p, p, p, p, 8, 18, 256, 1e-5, 5.0, None) == inval    # mean without var
    assert lib.ag_policy_sample(p, p, pd, None, 1e-5, 0, p, 24, 0, 0, p, p, p, p, p, None, 8, 4, None) == inval
    assert lib.ag_gae(p, p, p, p, 0.99, 0.95, p, p, 0, 8, None) == inval
    # round 3: weight gradient on the matrix cores, fused rollout step
    assert lib.ag_split_wgrad_slices(0) == 0 and lib.ag_term_sum_tiles(65) == 2
    assert lib.ag_split_wgrad(None, p, p, 64, 256, 256, 4, None) == inval
    assert lib.ag_split_wgrad(p, p, p, 64, 128, 256, 4, None) == unsupported                         # 256 x 256 layers only
    assert lib.ag_step_rollout_fused(None, None, p, p, p, None, None) == inval                       # no handle
    jobs = (N.AgSumJob * 13)()
    for j in range(13):
        jobs[j] = N.AgSumJob(p.value, p.value, 4, 8)
    assert lib.ag_sum_rows_multi(jobs, 13, p, 1 << 20, None) == unsupported                          # more than AG_MAX_SUM_JOBS
    jobs[0] = N.AgSumJob(p.value, p.value, 4, 6)
    assert lib.ag_sum_rows_multi(jobs, 1, p, 1 << 20, None) == unsupported                           # n % 4 != 0
    jobs[0] = N.AgSumJob(p.value, p.value, 4, 8)
    assert lib.ag_sum_rows_multi(jobs, 1, p, 4, None) == inval                                       # scratch too small


def test_ctypes_mirrors_of_the_argument_structs_match_the_header(tmp_path):
    """The structs passed BY POINTER (ag_loss_epilogue, ag_input_layer_args, ag_sum_job, ag_config): size and every field offset of
    the ctypes mirrors in airgym_amd/_native against what a C compiler makes of include/airgym_hip.h - without a GPU."""
    import shutil
    import subprocess
    from airgym_amd import _native
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    mirrors = {"ag_loss_epilogue": _native.AgLossEpilogue, "ag_input_layer_args": _native.AgInputLayerArgs,
               "ag_sum_job": _native.AgSumJob, "ag_config": _native.AgConfig}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, mirror in mirrors.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-std=c99", "-o", str(exe), str(src)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.strip().splitlines():
        cname, what, val = line.split()
        mirror = mirrors[cname]
        if what == "size":
            assert ctypes.sizeof(mirror) == int(val), (cname, ctypes.sizeof(mirror), val)
        else:
            assert getattr(mirror, what).offset == int(val), (cname, what, getattr(mirror, what).offset, val)
        seen += 1
    assert seen == sum(len(m._fields_) + 1 for m in mirrors.values())
