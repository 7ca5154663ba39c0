"""Times the config-4 shape (cart-pole, N_train=2000, P=16384, H=20) through the large-training-set path: one rollout
= 20 x (kstar_big_kernel, trmm_reduce_kernel, step_big_kernel).  Refuses to report a throughput on a non-zero status.
Environment: N, P, H override the shape; SX_TRMM_VARIANT / SX_TRMM_ORDER select trmm_reduce_kernel variants (csrc/sx_kernels.hip)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import _lib, problems
from safe_exploration_amd.cem_mpc import FusedCemMpc, cem_rollout
N, P, H = int(os.environ.get('N', 2000)), int(os.environ.get('P', 16384)), int(os.environ.get('H', 20))
dev = torch.device('cuda:0')
w = problems.baseline_workload(4, n_train=N)
spec = w.spec
t0 = time.perf_counter(); ssm, env = problems.build(spec, dev); torch.cuda.synchronize(); print(f'fit+pack N={N}: {time.perf_counter()-t0:.2f} s', flush=True)
x0 = torch.tensor(w.x0[:1], dtype=torch.float64, device=dev)
mpc = FusedCemMpc(ssm, env, H, P, 16, 1, device=dev, init_std=w.init_std[:H], warm_start='safe_policy')
mean = mpc.safe_policy_plan(x0).contiguous()
std = torch.tensor(w.init_std[:H], dtype=torch.float64, device=dev).reshape(1, H, 1).contiguous()
noise = torch.randn((1, P, H, 1), dtype=torch.float64, device=dev)
lib = _lib.lib()
for i in range(3):
    _lib.check(lib.sx_profile_enable(4096), 'enable')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = _lib.profile_collect(); lib.sx_profile_disable()
    st = int(r['status'].item())
    if st != 0 and not (int(os.environ.get("SX_TRMM_ORDER", "0")) & 6):
        raise SystemExit(f'device status {st}: no throughput reported')
    feas = float((r['con_cost'] == 0).double().mean())
    print(f'variant {os.environ.get("SX_TRMM_VARIANT", "13")} order {os.environ.get("SX_TRMM_ORDER", "0")} rollout P={P} H={H} N={N}: {dt*1e3:.1f} ms  {P*H/dt:.3e} particle-steps/s  status {st} feasible {feas:.3f}  '
          + '  '.join(f'{n} {ms/c*1e3:.0f}us' for n, (ms, c) in k.items()), flush=True)
