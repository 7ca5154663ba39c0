"""How long does the host take to ENQUEUE one solve, against how long the GPU takes to run it?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import problems
from safe_exploration_amd.cem_mpc import FusedCemMpc
dev = torch.device('cuda:0')
spec = problems.pendulum(n_train=200)
ssm, env = problems.build(spec, dev)
mpc = FusedCemMpc(ssm, env, 15, 4096, 409, 8, device=dev, seed=1, init_std=0.1)
x0 = torch.tensor([[0.02, -0.03]], dtype=torch.float64, device=dev)
for _ in range(3): mpc.solve(x0)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): mpc.solve(x0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3*(t1-t0)/n:.3f} ms/solve, total {1e3*(t2-t0)/n:.3f} ms/solve')

# the latency a control loop sees: get_actions = flat state in, solve, ONE device->host hand-off (status + flags), actions out
flat = torch.zeros((1, 6), dtype=torch.float64, device=dev)
flat[0, :2] = x0[0]
for _ in range(3): mpc.get_actions(flat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): mpc.get_actions(flat)
t1 = time.perf_counter()
print(f'get_actions (synchronous, one solve at a time): {1e3*(t1-t0)/n:.3f} ms per call')
