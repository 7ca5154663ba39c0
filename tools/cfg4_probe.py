"""Times the config-4 shape (cart-pole, N_train=2000, P=16384, H=20) through the large-training-set path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import problems
from safe_exploration_amd.cem_mpc import cem_rollout
N, P, H = int(os.environ.get('N', 2000)), int(os.environ.get('P', 16384)), int(os.environ.get('H', 20))
dev = torch.device('cuda:0')
spec = problems.cartpole(n_train=N)
t0 = time.perf_counter(); ssm, env = problems.build(spec, dev); torch.cuda.synchronize(); print(f'fit+pack N={N}: {time.perf_counter()-t0:.2f} s', flush=True)
x0 = torch.zeros((1, 4), dtype=torch.float64, device=dev)
mean = torch.zeros((1, H, 1), dtype=torch.float64, device=dev); std = torch.full((1, H, 1), 0.3, dtype=torch.float64, device=dev)
noise = torch.randn((1, P, H, 1), dtype=torch.float64, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flops = 4 * (2 * N * N) * P * H
    print(f'rollout P={P} H={H}: {dt*1e3:.1f} ms  {P*H/dt:.3e} particle-steps/s  algorithmic {flops/dt/1e12:.1f} TFLOP/s  status {int(r["status"].item())}', flush=True)
