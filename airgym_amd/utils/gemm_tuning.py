"""Library-GEMM selection for the PPO update on gfx950.

hipBLASLt's default heuristic is within a few percent of the best library kernel for the forward GEMMs, but rocBLAS has
faster solutions for the [196608x256]x[256x256] dX product and the 64-slice split-K weight-gradient bmm (184 vs 219 us,
176 vs 190 us; table produced by `PYTORCH_TUNABLEOP_ENABLED=1 python bench.py` on an MI355X, see profiles/).  The table
ships in airgym_amd/assets/tunableop_gfx950.csv and is applied through torch's TunableOp with tuning OFF: shapes that
are not in the table keep the default heuristic, nothing is tuned or written at run time.  TunableOp ignores the table
when its validator lines (torch / HIP / hipBLASLt / rocBLAS versions, gfx arch) do not match the running stack.
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "tunableop_gfx950.csv")
_enabled = False


def enable_tuned_gemms(path=TABLE):
    """Idempotent; returns True when the table was handed to TunableOp."""
    global _enabled
    if _enabled:
        return True
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    if os.getenv("PYTORCH_TUNABLEOP_ENABLED") is not None:      # the user drives TunableOp explicitly: leave it alone
        return False
    try:
        import shutil
        import sys
        import tempfile

        import torch.cuda.tunable as tunable
        # TunableOp may (re)write the file it is pointed at: hand it a private copy, never the in-tree asset
        private = os.path.join(tempfile.mkdtemp(prefix="airgym_tunableop_"), "table.csv")
        shutil.copyfile(path, private)
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.set_filename(private, insert_device_ordinal=False)
        _enabled = bool(tunable.read_file(private))
        return _enabled
    except Exception as e:      # older torch without the API: keep the default heuristic
        print(f"[airgym_amd] TunableOp table not applied: {e}", file=sys.stderr)
        return False
