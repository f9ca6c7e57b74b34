#!/usr/bin/env python3
"""Build libairgym_hip.so for gfx950 with hipcc (no cmake; 24 translation units compiled in parallel).

    python airgym_amd/csrc/build.py [--force] [--jobs N] [--experiments]

The library lands in-tree at airgym_amd/_native/libairgym_hip.so (git-ignored, shipped to the GPU box by gpurun).
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB_DIR = os.path.join(PKG, "_native")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(LIB_DIR, "libairgym_hip.so")
LIB_EXP = os.path.join(LIB_DIR, "libairgym_hip_exp.so")      # --experiments: + include/airgym_hip_debug.h (tools/ only)
ARCH = "gfx950"

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]

HEADERS = ["env_math.hpp", "planning_math.hpp", "kernel_args.hpp", "rollout_math.hpp", "handle.hpp", "split_common.hpp", "ppo_loss_math.hpp",
           os.path.join("..", "..", "include", "airgym_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def units(experiments=False):
    """(object, source, defines) of every translation unit.  The experiments build compiles the same sources with
    -DAG_EXPERIMENTS into its own object directory and adds experiments.hip."""
    od = os.path.join(OBJ_DIR, "exp") if experiments else OBJ_DIR
    x = ["-DAG_EXPERIMENTS"] if experiments else []
    out = []
    for task in (0, 1):
        for ctl in range(5):
            # -ffp-contract=on: a*b+c fuses only inside ONE source expression (decided by the front end), never across statements
            # (the default `fast` lets the back end fuse whatever ends up adjacent after inlining, which differs between the
            # kernels that share env_math.hpp): every step kernel then runs the same arithmetic, bit for bit
            out.append((os.path.join(od, f"step_{task}_{ctl}.o"), "step_kernel.hip",
                        [f"-DAG_TASK={task}", f"-DAG_CTL={ctl}", "-ffp-contract=on"] + x))
    for name in ("airgym_hip", "ppo_kernels", "planning_kernel", "rollout_kernels", "split_gemm", "split_wgrad", "cnn_kernels", "conv_kernels", "mlp_chain",
                 "first_layer"):
        # rollout_kernels.hip shares rollout_math.hpp with the fused step kernel (policy sampling, reward shaping): same rule
        # mlp_chain.hip: the K-step loop of the chain kernel must be unrolled completely (its accumulator tiles are indexed by the
        # step); at 20 steps (Tracking's 48 inputs) the body exceeds LLVM's default budget for `#pragma unroll`, the loop stays
        # rolled and the tiles go to scratch (576 B per lane, 186 us instead of ~75 at M = 65 536) - hence the raised threshold
        flags = ["-ffp-contract=on"] if name == "rollout_kernels" else (["-mllvm", "-pragma-unroll-threshold=100000"] if name == "mlp_chain" else [])
        out.append((os.path.join(od, name + ".o"), name + ".hip", flags + list(x)))
    # mixed_precision: the three matrix-core sources once more with ONE bf16 plane per operand (one MFMA per product, f32
    # accumulate); their compute entry points are exported with the suffix _bf16 (split_common.hpp)
    for name in ("split_gemm", "split_wgrad", "mlp_chain", "first_layer"):
        out.append((os.path.join(od, name + "_bf16.o"), name + ".hip", ["-DAG_SPLIT_PLANES=1"]
                    + (["-mllvm", "-pragma-unroll-threshold=100000"] if name == "mlp_chain" else []) + list(x)))
    if experiments:
        out.append((os.path.join(od, "experiments.o"), "experiments.hip", list(x)))
    return out


def compile_one(obj, src, defs, extra):
    cmd = COMMON + extra + defs + ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return obj, r.returncode, (r.stdout + r.stderr).strip(), " ".join(cmd)


def build(force=False, jobs=None, extra=(), verbose=True, experiments=False, tag=None):
    """tag (experiments only): build a VARIANT of the experiments library with the extra -D flags in `extra` into
    libairgym_hip_exp_<tag>.so (always a full rebuild into its own object directory); load it with AIRGYM_EXP_LIB=<path>."""
    if tag:
        assert experiments, "--tag is for experiment variants"
        return _build_variant(tag, list(extra), jobs, verbose)
    os.makedirs(os.path.join(OBJ_DIR, "exp") if experiments else OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    lib = LIB_EXP if experiments else LIB
    hdrs = [os.path.join(HERE, h) for h in HEADERS if os.path.exists(os.path.join(HERE, h))] + [os.path.abspath(__file__)]
    if experiments:
        hdrs.append(os.path.join(HERE, "..", "..", "include", "airgym_hip_debug.h"))
    todo = [(o, s, d) for (o, s, d) in units(experiments) if force or _newer(o, [os.path.join(HERE, s)] + hdrs)]
    t0 = time.time()
    if todo:
        jobs = jobs or min(len(todo), os.cpu_count() or 4)
        with cf.ThreadPoolExecutor(jobs) as ex:
            for obj, rc, log, cmd in ex.map(lambda u: compile_one(*u, list(extra)), todo):
                if log and verbose:
                    print(log)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed ({rc}): {cmd}\n{log}")
    objs = [o for (o, _, _) in units(experiments)]
    if todo or _newer(lib, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed: {' '.join(cmd)}\n{r.stdout}{r.stderr}")
    if verbose:
        print(f"[airgym_amd] {lib} ready ({len(todo)} units rebuilt, {time.time() - t0:.1f}s)")
    return lib


def _build_variant(tag, extra, jobs, verbose):
    od = os.path.join(OBJ_DIR, "exp_" + tag)
    os.makedirs(od, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    lib = os.path.join(LIB_DIR, f"libairgym_hip_exp_{tag}.so")
    todo = [(os.path.join(od, os.path.basename(o)), s, d) for (o, s, d) in units(True)]
    with cf.ThreadPoolExecutor(jobs or min(len(todo), os.cpu_count() or 4)) as ex:
        for obj, rc, log, cmd in ex.map(lambda u: compile_one(*u, extra), todo):
            if rc != 0:
                raise RuntimeError(f"hipcc failed ({rc}): {cmd}\n{log}")
    r = subprocess.run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + [o for (o, _, _) in todo],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {r.stdout}{r.stderr}")
    if verbose:
        print(f"[airgym_amd] {lib} ready (variant {tag}: {' '.join(extra)})")
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--experiments", action="store_true",
                    help="build libairgym_hip_exp.so: the library + the ag_debug_* entry points (include/airgym_hip_debug.h)")
    ap.add_argument("--tag", default=None, help="with --experiments: build a variant library libairgym_hip_exp_<tag>.so")
    ap.add_argument("extra", nargs="*", help="extra hipcc flags, e.g. -Rpass-analysis=kernel-resource-usage")
    a = ap.parse_args()
    try:
        build(a.force, a.jobs, a.extra, experiments=a.experiments, tag=a.tag)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
