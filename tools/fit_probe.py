"""Times sx_gp_fit + sx_gp_mll_grad (the training-loop body) and the full update_model."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import problems
dev = torch.device('cuda:0')
for which, n in (('pendulum', 200), ('pendulum', 590), ('cartpole', 2000)):
    spec = getattr(problems, which)(n_train=n)
    ssm, env = problems.build(spec, dev)
    x, y = ssm.x_train, ssm.y_train
    ssm._update_model(x, y); ssm.mll_and_grad(x, y)   # warm-up: code objects load on first launch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ssm._update_model(x, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(3): ssm.mll_and_grad(x, y)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{which} N={n}: update_model {(t1-t0)/3*1e3:.2f} ms, fit+mll_grad {(t2-t1)/3*1e3:.2f} ms', flush=True)
