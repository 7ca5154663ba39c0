"""Randomised check of the fused rollout (sx_cem_rollout) and the GP posterior (sx_gp_predict) against the oracle over the
training-set size: every residue of N modulo the 16-row blocks (the mean / Jacobian rows behind the training rows may
straddle two row blocks), tiny N, the single-launch mode switches.  python tools/rollout_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cem as ocem  # noqa: E402  (checker only)
from oracle.gp import ExactGP  # noqa: E402
from safe_exploration_amd import problems  # noqa: E402
from safe_exploration_amd.cem_mpc import cem_rollout  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
bad = 0
for case in range(cases):
    which = 'cartpole' if rng.random() < 0.3 else 'pendulum'
    n_train = int(rng.choice([rng.integers(1, 20), rng.integers(20, 260), rng.integers(260, 700)], p=[.2, .6, .2]))
    spec = getattr(problems, which)(n_train=n_train, seed=int(rng.integers(0, 1000)))
    ssm, env = problems.build(spec, dev)
    gp = ExactGP(spec.X, spec.Y, spec.lengthscale, spec.outputscale, spec.noise)
    n_s, n_u = spec.n_s, spec.n_u
    z = rng.uniform(-0.4, 0.4, size=(int(rng.integers(1, 40)), n_s + n_u))
    m, v, j = ssm.predict_with_jacobians(T(z[:, :n_s]), T(z[:, n_s:]))
    mo, vo, jo = gp.predict(z)
    ok = (np.allclose(m.cpu().numpy(), mo, rtol=1e-8, atol=1e-11) and np.allclose(v.cpu().numpy(), vo, rtol=1e-7, atol=1e-11)
          and np.allclose(j.cpu().numpy(), jo, rtol=1e-8, atol=1e-10))
    P, H = int(rng.integers(1, 60)), int(rng.integers(1, 5))
    acts = rng.normal(0, 0.2, size=(P, H, n_u))
    x0 = rng.normal(0, 0.02, size=n_s)
    r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
    ref = ocem.rollout(problems.oracle_problem(spec, ocem), gp, x0, acts)
    traj = r['traj'][0].cpu().numpy()
    ok = ok and np.allclose(traj[:, :, :n_s], ref.traj_p, rtol=1e-7, atol=1e-10)
    ok = ok and np.allclose(traj[:, :, n_s:].reshape(P, H, n_s, n_s), ref.traj_q, rtol=1e-6, atol=1e-10)
    ok = ok and np.allclose(r['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-7, atol=1e-10)
    ok = ok and np.array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost) and int(r['status'].item()) == ref.status
    if not ok:
        bad += 1
        print(f'MISMATCH case {case}: {which} N={n_train} P={P} H={H} status {int(r["status"].item())} / {ref.status}', flush=True)
print(f'{cases} cases, {bad} mismatches')
sys.exit(1 if bad else 0)
