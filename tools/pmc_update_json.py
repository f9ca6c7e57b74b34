#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc.txt (tools/gpu_pmc_update.sh: one `COUNTER kernel-substring n= mean=` line per counter and kernel) -> the JSON
bench.py's top-level `roofline` reads `traffic` and `mfma_busy_pct` from (profiles/<round>_update_kernels_pmc.json), keyed by the
entry point of each of the optimizer step's three large launches, with the hash of the kernel sources it was measured on.
FETCH_SIZE x 2 (gfx950 counts a 128-byte request of a wide coalesced read as 64, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes;
matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).

    python tools/pmc_update_json.py gpurun_out/r06_update_pmc.txt r06 [kernel-trace.md] > profiles/r06_update_kernels_pmc.json
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airgym_amd.utils.kernel_bench import update_source_sha  # noqa: E402

path, tag = sys.argv[1], sys.argv[2]
trace = sys.argv[3] if len(sys.argv) > 3 else None
vals = {}
for line in open(path):
    m = re.match(r"(\w+) (\S+)\s+n=\s*(\d+) mean=\s*([\d.]+)", line)
    if m:
        vals[(m.group(2), m.group(1))] = (float(m.group(4)), int(m.group(3)))
symbols = {}
if trace and os.path.exists(trace):      # full symbols (template arguments) from the kernel trace of the same build
    for line in open(trace):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| [\d.]+ \| ([\d.]+) ", line)
        k = re.search(r"(split_\w+<[^>]*>)", m.group(1)) if m else None
        if k:       # rows are sorted by total time: the first row of a kernel family is the instantiation the step launches
            symbols.setdefault(k.group(1).replace(" ", ""), float(m.group(3)))
launches = {"ag_split_gemm_input_loss_heads_bwd": "split_gemm_kernel<true", "ag_split_wgrad_input": "split_wgrad_fin_kernel",
            "ag_split_gemm_input_wgrad_recompute": "split_gemm_kernel<false"}
out, sha = {}, update_source_sha()
for entry, sub in launches.items():
    if (sub, "FETCH_SIZE") not in vals or (sub, "WRITE_SIZE") not in vals:
        continue
    (f, nf), (w, nw) = vals[(sub, "FETCH_SIZE")], vals[(sub, "WRITE_SIZE")]
    rec = {"kernel": next((s for s in symbols if sub in s), sub + ",...>"),
           "entry_point": entry, "rows": 196608,
           "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "fetch_correction": 2.0,
           "traffic_bytes_per_launch": int(round((2.0 * f + w) * 1024)), "source_sha": sha,
           "source": f"profiles/{tag}_update_kernels_pmc.md (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, {nf} / {nw} dispatches)"}
    if (sub, "SQ_VALU_MFMA_BUSY_CYCLES") in vals and (sub, "GRBM_GUI_ACTIVE") in vals:
        busy, gui = vals[(sub, "SQ_VALU_MFMA_BUSY_CYCLES")][0], vals[(sub, "GRBM_GUI_ACTIVE")][0]
        rec["mfma_busy_pct"] = round(100.0 * busy / 1024.0 / (gui / 8.0), 1)
        rec["kcycles"] = round(gui / 8.0 / 1e3, 1)
    if rec["kernel"] in symbols:
        rec["trace_avg_us"] = symbols[rec["kernel"]]
    out[entry] = rec
print(json.dumps(out, indent=1))
