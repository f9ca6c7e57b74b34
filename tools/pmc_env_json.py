#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc_summary.txt (tools/gpu_pmc_env.sh) -> the JSON bench.py reads `roofline.traffic` from
(profiles/<tag>_env_kernel_pmc.json): per entry point FETCH_SIZE x 2 (gfx950 counts a 128-byte request of a wide coalesced read
as 64, MI355X_MICROARCH.md) + WRITE_SIZE, in bytes per launch, with the hash of the kernel sources it was measured on."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airgym_amd.utils.kernel_bench import (ALGO_BYTES_PER_ENV_STEP, env_kernel_source_sha, fused_algo_bytes, kernel_name,  # noqa: E402
                                           multi_own_bytes)

path, tag = sys.argv[1], sys.argv[2]
vals = {}
for line in open(path):
    m = re.match(r"form=(\w+) (\w+) \S+\s+n=\s*(\d+) mean=\s*([\d.]+)", line)
    if m:
        vals[(m.group(1), m.group(2))] = (float(m.group(4)), int(m.group(3)))
out, envs, sha = {}, 65536, env_kernel_source_sha()
MULTI_K = 24
keys = {"rollout": ("hovering_rate", "ag_step_rollout", False), "api": ("hovering_rate_ag_step", "ag_step", False),
        "fused": ("hovering_rate_fused", "ag_step_rollout_fused", True),
        "multi": (f"hovering_rate_multi{MULTI_K}", f"ag_step_multi ({MULTI_K} env steps per launch)", False)}
for form, (key, entry, fused) in keys.items():
    if (form, "FETCH_SIZE") not in vals or (form, "WRITE_SIZE") not in vals:
        continue
    (f, nf), (w, nw) = vals[(form, "FETCH_SIZE")], vals[(form, "WRITE_SIZE")]
    algo = fused_algo_bytes("hovering", "rate", 18, 4) if fused else ALGO_BYTES_PER_ENV_STEP[("hovering", "rate")]
    out[key] = {"kernel": kernel_name("hovering", "rate", fused, single=form in ("rollout", "api")), "entry_point": entry, "envs": envs,
                "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "fetch_correction": 2.0,
                "traffic_bytes_per_launch": int(round((2.0 * f + w) * 1024)), "algorithmic_bytes_per_launch": algo * envs,
                "source_sha": sha,
                **({"steps_per_launch": MULTI_K, "algorithmic_bytes_per_launch": algo * envs * MULTI_K,
                    "own_bytes_per_launch": int(round(multi_own_bytes("hovering", "rate", 18, 4, MULTI_K) * envs * MULTI_K))}
                   if form == "multi" else {}),
                "source": f"profiles/{tag}_env_kernel_pmc.md (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, {nf} / {nw} dispatches)"}
print(json.dumps(out, indent=1))
