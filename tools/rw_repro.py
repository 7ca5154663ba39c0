"""One fused rollout per (n_s, n_u, N) on a synthetic problem, checked against the oracle, each in a child process with its
stderr kept, stopping at the first failure.  SX_ROLLOUT=rh|rw|stream (+ SX_ROLLOUT_STRICT=1) picks the kernel form: this is
how the suite covers the forms that are not the default.   python tools/rw_repro.py [n_s,n_u[,N] ...]
TIME=1 also times one rollout of 4096 particles x 15 steps per shape (A/B of the forms on shapes no BASELINE config has)."""
import os
import subprocess
import sys

CHILD = r'''
import ctypes, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from safe_exploration_amd import _lib, problems
from safe_exploration_amd.cem_mpc import cem_rollout
from safe_exploration_amd.utils import dlqr
n_s, n_u, n_train = %(ns)d, %(nu)d, %(n)d
rng = np.random.default_rng(100 * n_s + n_u)
d_in = n_s + n_u
a = np.eye(n_s) + 0.05 * rng.normal(size=(n_s, n_s))
b = 0.3 * rng.normal(size=(n_s, n_u))
k_fb = -dlqr(a, b, np.eye(n_s), 5.0 * np.eye(n_u))[0]
X, Y = problems.synthetic_training_set(n_train, n_s, n_u, seed=n_s * 7 + n_u, scale=0.6)
ls = rng.uniform(0.6, 1.4, size=(n_s, d_in))
s, nz = rng.uniform(0.01, 0.03, size=n_s), rng.uniform(1e-5, 5e-5, size=n_s)
h_mat = np.vstack((np.eye(n_s), -np.eye(n_s)))
h_vec = np.full((2 * n_s, 1), 0.5 if n_s <= 2 else 1.2)
spec = problems.ProblemSpec('synthetic', n_s, n_u, X, Y, ls, s, nz, a, b, k_fb, rng.uniform(0.01, 0.05, size=n_s),
                            rng.uniform(0.01, 0.05, size=n_s), 2.5, h_mat, h_vec, np.full(n_u, -0.4),
                            np.full(n_u, 0.4), obj_mode=_lib.SX_OBJ_AFFINE_ABS, obj_w_abs=rng.uniform(0, 1, size=n_s),
                            obj_target=rng.normal(0, 0.1, size=n_s), obj_w_lin=rng.normal(0, 0.2, size=n_s))
ssm, env = problems.build(spec, 'cuda:0')
print('built', flush=True)
P, H = 53, 5
acts = rng.normal(0, 0.25, size=(P, H, n_u))
x0 = rng.normal(0, 0.02, size=n_s)
T = lambda v: torch.tensor(v, dtype=torch.float64, device='cuda:0')
r = cem_rollout(ssm, env, T(x0[None]), H, actions=T(acts[None]), want_traj=True, want_sigma=True)
torch.cuda.synchronize()
print('rollout ok', float(r['obj_cost'].sum()), int(r['status'].item()), flush=True)
# against the oracle (the checker): trajectory centres and shapes, variances, costs
from oracle import cem as ocem
from oracle.gp import ExactGP
ref = ocem.rollout(problems.oracle_problem(spec, ocem), ExactGP(X, Y, ls, s, nz), x0, acts)
traj = r['traj'][0].cpu().numpy()
np.testing.assert_allclose(traj[:, :, :n_s], ref.traj_p, rtol=1e-8, atol=1e-11)
np.testing.assert_allclose(traj[:, :, n_s:].reshape(P, H, n_s, n_s), ref.traj_q, rtol=1e-7, atol=1e-11)
np.testing.assert_allclose(r['sigma'][0].cpu().numpy(), ref.sigma, rtol=1e-8, atol=1e-12)
np.testing.assert_allclose(r['obj_cost'][0].cpu().numpy(), ref.obj_cost, rtol=1e-8, atol=1e-11)
np.testing.assert_array_equal(r['con_cost'][0].cpu().numpy(), ref.con_cost)
assert int(r['status'].item()) == ref.status
form = _lib.lib().sx_cem_rollout_form(ctypes.byref(ssm.device_model), H)
print('matches the oracle; form', int(form), flush=True)
if %(time)d:
    # the same model at config-2 scale (4096 particles, H = 15): one launch, timed over 20 repeats
    import time
    Pt, Ht = 4096, 15
    big = T(rng.normal(0, 0.25, size=(1, Pt, Ht, n_u)))
    for _ in range(3):
        cem_rollout(ssm, env, T(x0[None]), Ht, actions=big)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        cem_rollout(ssm, env, T(x0[None]), Ht, actions=big)
    torch.cuda.synchronize()
    print('timed: %%.1f us per rollout of 4096 particles x 15 steps; form %%d' %% ((time.perf_counter() - t0) / 20 * 1e6,
          int(_lib.lib().sx_cem_rollout_form(ctypes.byref(ssm.device_model), Ht))), flush=True)
'''


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(2, 1, 77), (1, 1, 77)]
    for shape in shapes:
        ns, nu = shape[0], shape[1]
        n = shape[2] if len(shape) > 2 else 77
        env = dict(os.environ, SX_DEBUG_SYNC='1', AMD_LOG_LEVEL=os.environ.get('AMD_LOG_LEVEL', '1'))
        p = subprocess.run([sys.executable, '-c', CHILD % dict(root=root, ns=ns, nu=nu, n=n, time=int(os.environ.get('TIME', '0')))], capture_output=True, text=True,
                           env=env, timeout=120)
        print(f'== n_s={ns} n_u={nu} N={n}: rc={p.returncode}')
        print(p.stdout[-600:])
        print(p.stderr[-2500:])
        if p.returncode != 0:
            sys.exit(1)


if __name__ == '__main__':
    main()
