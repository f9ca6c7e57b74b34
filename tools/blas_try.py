import sys, os, json, subprocess
sys.path.insert(0, "/root/repo")
import torch
lib = sys.argv[1]
torch.backends.cuda.preferred_blas_library(lib)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
import runpy
runpy.run_path("/root/repo/bench.py", run_name="__main__")
