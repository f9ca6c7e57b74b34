"""CPU oracle for the AirGym hot path (TEST INFRASTRUCTURE ONLY).

This package is a PyTorch-CPU / numpy restatement of the reference's per-env
step (emNavi/AirGym, `airgym/envs/base/hovering.py`, `airgym/envs/task/tracking.py`)
and of the PPO numerics in `lib/`.  It is the *checker* for the HIP path in
`airgym_amd/`; it is never the thing shipped or measured.

Who may import it:  `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py`.  Nothing under `airgym_amd/` imports it, and
the product path raises if the HIP library is missing rather than falling
back to this code.

Parity status (see DESIGN.md section "Oracle"):

* Reference-owned arithmetic (action pre-processing, wrench assembly
  constants, reset distributions, observation layout + noise sigmas, reward
  and termination, lemniscate reference, PPO losses / GAE / running-mean-std /
  KL scheduler) is PINNED: `tests/golden/*.npz` were produced by calling the
  reference's own functions (stub-import harness `tests/golden/make_golden.py`,
  run once in the build container) and `tests/test_oracle_golden.py` checks
  the oracle against them.
* The rigid-body integrator (IsaacGym/PhysX, closed source) and the control
  cascades (rlPx4Controller, un-vendored, unpinned) are NOT in the reference
  tree.  For those two pieces the oracle is the build's own written spec
  (`rigid_body.py`, `px4_cascade.py`):  **parity unpinned**.
* `pytorch3d.transforms` (un-vendored) is restated in `rotations.py` from its
  published algorithm and cross-checked against scipy: parity unpinned w.r.t.
  the third-party package, pinned against scipy.
"""
