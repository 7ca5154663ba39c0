"""Does capturing one MPC solve (8 x [rollout, rank/refit]) in a HIP graph shorten it?  Eager vs graph replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import problems
from safe_exploration_amd.cem_mpc import FusedCemMpc
dev = torch.device('cuda:0')
spec = problems.pendulum(n_train=200)
ssm, env = problems.build(spec, dev)
P, H, it = 4096, 15, 8
mpc = FusedCemMpc(ssm, env, H, P, 409, it, device=dev, seed=1, init_std=0.1)
x0 = torch.tensor([[0.02, -0.03]], dtype=torch.float64, device=dev)
noise = torch.randn((it, 1, P, H, 1), dtype=torch.float64, device=dev)


def timed(fn, n=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


eager = timed(lambda: mpc.solve(x0, noise=noise))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        mpc.solve(x0, noise=noise)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = mpc.solve(x0, noise=noise)
ref = mpc.solve(x0, noise=noise)
g.replay()
torch.cuda.synchronize()
print('same result:', torch.equal(out[0], ref[0]), int(out[1][0]), int(ref[1][0]))
graph = timed(g.replay)
print(f'eager {eager:.3f} ms/solve   graph replay {graph:.3f} ms/solve (noise generation excluded in both)')
