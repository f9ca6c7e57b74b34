#!/bin/bash
# HBM traffic + VALU occupancy of the Planning render kernel from PMC counters (separate rocprofv3 passes, kernel-trace only) ->
# gpurun_out/<tag>_planning_render_pmc.json with the kernel-source hash bench.py checks before quoting `side.planning.roofline.traffic`.
#   bash tools/gpu_pmc_planning.sh <tag> [envs]
set -u
TAG=${1:-r06}; ENVS=${2:-16384}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -f $OUT/${TAG}_planning_render_pmc.txt
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  D=/tmp/pmcpl_${TAG}_$(echo $SET | tr ' ' '_' | cut -c1-20); rm -rf $D
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o f -- python $REPO/tools/bench_planning.py --envs $ENVS --steps 16 > /dev/null 2>> $OUT/${TAG}_planning_render_pmc.err
  for C in $SET; do python $REPO/tools/pmc_summary.py $D $C planning_render_kernel >> $OUT/${TAG}_planning_render_pmc.txt 2>&1; done
done
cat $OUT/${TAG}_planning_render_pmc.txt
python - <<PY > $OUT/${TAG}_planning_render_pmc.json
import json, re, sys
sys.path.insert(0, "$REPO")
from airgym_amd.utils.kernel_bench import planning_source_sha
v = {}
for line in open("$OUT/${TAG}_planning_render_pmc.txt"):
    m = re.match(r"(\w+) planning_render_kernel\s+n=\s*(\d+) mean=\s*([\d.]+)", line)
    if m:
        v[m.group(1)] = (float(m.group(3)), int(m.group(2)))
rec = {"kernel": "ag::planning_render_kernel<0>", "envs": $ENVS, "FETCH_SIZE_KB_mean": v["FETCH_SIZE"][0], "WRITE_SIZE_KB_mean": v["WRITE_SIZE"][0],
       "fetch_correction": 2.0, "traffic_bytes_per_launch": int(round((2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024)),
       "algorithmic_bytes_per_launch": $ENVS * 212 * 120 * 4, "source_sha": planning_source_sha(),
       "source": "profiles/${TAG}_planning_render_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, %d / %d dispatches)" % (v["FETCH_SIZE"][1], v["WRITE_SIZE"][1])}
if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v:
    # one 1 024-thread workgroup (16 waves) per CU (103 KB of LDS): four waves resident on every SIMD for the whole launch, so
    # SIMD-resident time = SQ_WAVE_CYCLES / 4 and the VALU pipe's share of it:
    rec["valu_busy_pct"] = round(100.0 * v["SQ_ACTIVE_INST_VALU"][0] / (v["SQ_WAVE_CYCLES"][0] / 4.0), 1)
    rec["sq"] = {k: x[0] for k, x in sorted(v.items())}
print(json.dumps(rec, indent=1))
PY
cat $OUT/${TAG}_planning_render_pmc.json
