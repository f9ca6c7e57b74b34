"""airgym_amd - MI355X-native hot path of the AirGym vectorised quadrotor environments.

Layout
  csrc/        HIP kernels for gfx950 + the C ABI (include/airgym_hip.h)
  _native/     ctypes binding; libairgym_hip.so is built in-tree here
  hip_env.py   HipEnvHandle: zero-copy torch views of the library's device buffers
  envs/, utils/ host-side mirror of the reference's task API (task_registry.make_env, Hovering, Tracking)
  lib/         PPO loop mirror of the reference's lib/ (PyTorch-ROCm policy, RCCL grad all-reduce)
"""
import os

AIRGYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))

__version__ = "0.1.0"

import os as _os

# MIOpen's default find mode benchmarks EVERY convolution solver - including its naive reference kernels, seconds per call at
# Planning's 4096-image minibatches (~100 s of kernel time before the first epoch, profiles/r01_planning_ppo_kernel_trace.md)
# - the first time a shape is seen.  Immediate mode picks from the heuristic and costs 6 % of steady-state update time.
# Set MIOPEN_FIND_MODE yourself (e.g. NORMAL) to get the exhaustive search back.
_os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
