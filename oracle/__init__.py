"""CPU oracle for the CEM safe-MPC hot path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy float64 restatement of the reference algorithm (oscarkey/safe-exploration) for the one
path this repository accelerates, plus a plain-C restatement of the rollout (``oracle/csrc``, OpenMP over particles;
``oracle.c_oracle``) that serves as a second check and as the CPU baseline of ``bench.py``.  It exists to CHECK the HIP
path, never to serve it:

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it;
* nothing under ``safe_exploration_amd/`` imports it, and the product path raises if the HIP library is missing.

Pinning (see DESIGN.md "Oracle"):

* ``oracle.reachability`` / ``oracle.ellipsoid``  -- PINNED against the reference's own
  ``gp_reachability_pytorch`` / ``gp_reachability`` / ``utils`` / ``utils_ellipsoid`` run in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the known answers in the reference tests.
* ``oracle.gp``  -- the exact-GP posterior is computed by gpytorch 0.3.2 in the reference, which is not vendored and
  not installable here.  Restated from the closed form; numeric VALUES are "parity unpinned", structural
  properties (independent outputs, likelihood noise included, Jacobian layout) are tested.
* ``oracle.cem``  -- the optimiser loop lives in the un-vendored ``constrained-cem-mpc`` submodule.  The loop is
  specified by this repository (DESIGN.md "CEM specification"): "parity unpinned"; the pieces the reference's
  tests do pin (ActionConstraint cost, get_actions contract, PQ layout) are tested.
"""
