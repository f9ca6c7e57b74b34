#!/bin/bash
# Rebuild only split_gemm.o with extra -D flags and relink (experiments).  Usage: bash tools/build_with.sh -DAG_IW_PREFETCH=4
python - "$@" <<'PY'
import subprocess, os, sys
sys.path.insert(0, os.getcwd())
from airgym_amd.csrc import build as B
obj = [o for (o, s, _) in B.units() if s == "split_gemm.hip"][0]
r = subprocess.run(B.COMMON + sys.argv[1:] + ["-c", os.path.join(B.HERE, "split_gemm.hip"), "-o", obj], capture_output=True, text=True)
print(r.returncode, r.stderr[-400:])
objs = [o for (o, _, _) in B.units()]
r = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", B.LIB] + objs, capture_output=True, text=True)
print(r.returncode, r.stderr[-200:])
PY
