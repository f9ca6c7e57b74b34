#!/usr/bin/env python3
"""Where does the MI355X put the two waves of each env-step workgroup?  (dev tool, GPU box)

Decodes HW_REG_HW_ID (gfx9: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]) and HW_REG_XCC_ID per
wave of the step kernel's launch geometry and prints how the role-0 ("physics") waves spread over the SIMDs of each CU."""
import collections
import ctypes
import json
import os
import sys
os.environ.setdefault("AIRGYM_EXPERIMENTS", "1")      # ag_debug_* live in the experiments build

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd import _native as N  # noqa: E402
from airgym_amd.hip_env import HipEnvHandle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = HipEnvHandle("hovering", "rate", n, seed=0)
out = torch.zeros((n + 63) // 64 * 2, 2, dtype=torch.int32, device="cuda")
res = {}
for rep in range(3):
    N.check(env.lib.ag_debug_wave_placement(env.h, out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "placement")
    torch.cuda.synchronize()
    hw, xcc = out[:, 0].cpu().numpy().astype("int64") & 0xFFFFFFFF, out[:, 1].cpu().numpy().astype("int64") & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = xcc * 4096 + se * 64 + sh * 16 + cu
    role = (torch.arange(out.shape[0]) % 2).numpy()             # wave 0 / wave 1 of each workgroup
    per_simd = collections.Counter()
    phys_per_simd = collections.Counter()
    for k, s_, r in zip(cu_key, simd, role):
        per_simd[(k, s_)] += 1
        if r == 0:
            phys_per_simd[(k, s_)] += 1
    hist_all = collections.Counter(per_simd.values())
    hist_phys = collections.Counter(phys_per_simd.get(key, 0) for key in per_simd)
    same_simd = sum(1 for w in range(0, out.shape[0], 2) if simd[w] == simd[w + 1] and cu_key[w] == cu_key[w + 1])
    same_cu = sum(1 for w in range(0, out.shape[0], 2) if cu_key[w] == cu_key[w + 1])
    res = {"envs": n, "waves": int(out.shape[0]), "distinct_cus": len(set(cu_key)), "distinct_simds": len(per_simd),
           "waves_per_simd_hist": dict(sorted(hist_all.items())), "wave0_per_simd_hist": dict(sorted(hist_phys.items())),
           "wg_waves_on_same_simd": same_simd, "wg_waves_on_same_cu": same_cu, "xcc_hist": dict(collections.Counter(xcc.tolist()))}
    print(json.dumps(res))
env.close()
