"""BASELINE config 5 on one GPU: 8 independent pendulum episodes x cfg-2 solves (4096 particles, H=15, N=200, 8 CEM
iterations) in ONE fused solve, against 8 single-episode solves."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import problems
from safe_exploration_amd.cem_mpc import FusedCemMpc
dev = torch.device('cuda:0')
spec = problems.pendulum(n_train=200)
ssm, env = problems.build(spec, dev)
E, P, H, it = int(os.environ.get('E', 8)), 4096, 15, 8
mpc = FusedCemMpc(ssm, env, H, P, 409, it, device=dev, seed=1, init_std=0.1)
g = torch.Generator(device=dev); g.manual_seed(3)
x0 = 0.05 * torch.randn((E, 2), dtype=torch.float64, device=dev, generator=g)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


tb = timed(lambda: mpc.solve(x0))
ts = timed(lambda: [mpc.solve(x0[e:e + 1]) for e in range(E)])
ps = E * P * H * it
print(f'{E} episodes batched: {1e3 * tb:.3f} ms ({ps / tb:.3e} particle-steps/s, {E / tb:.0f} solves/s); '
      f'one by one: {1e3 * ts:.3f} ms ({ps / ts:.3e} particle-steps/s)')
