"""Times the CEM rollout over a degenerate-kernel GP ('nn' kernel, LAYERS=8,16 by default; 'linear' with KERNEL=linear) at
config 2's shape (pendulum, P=4096 particles, H = 15): cem_rollout_feat_kernel, one particle per lane."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import problems
from safe_exploration_amd.cem_mpc import cem_rollout
from safe_exploration_amd.gp_reachability_pytorch import make_env
from safe_exploration_amd.ssm_cem.gp_ssm_cem import GpCemSSM

kernel = os.environ.get('KERNEL', 'nn')
layers = [int(v) for v in os.environ.get('LAYERS', '8,16').split(',')]


class Conf:
    exact_gp_kernel, nn_kernel_layers, device, exact_gp_training_iterations = kernel, layers, 'cuda:0', 0


dev = torch.device('cuda:0')
spec = problems.pendulum(200, model_error=0.02)
T = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
ssm = GpCemSSM(Conf(), 2, 1)
rng = np.random.default_rng(0)
if kernel == 'nn':
    net, prev = [], 3
    for w in layers:
        net.append((rng.normal(size=(w, prev)) / np.sqrt(prev), rng.normal(size=w) * 0.3))
        prev = w
    ssm.set_network(net, prelu=0.25)
ssm.set_hyperparameters(kernel_scale=np.full(2, 0.5), noise=np.full(2, 2e-3))
ssm.update_model(T(spec.X), T(spec.Y), replace_old=True)
env = make_env(2, 1, a=spec.a, b=spec.b, k_fb=spec.k_fb, l_mu=spec.l_mu, l_sigma=spec.l_sigma, beta=spec.beta, h_mat=spec.h_mat,
               h_vec=spec.h_vec, u_min=spec.u_min, u_max=spec.u_max)
P, H = int(os.environ.get('P', 4096)), 15
x0 = T([[0.02, -0.03]])
mean, std = torch.zeros((1, H, 1), dtype=torch.float64, device=dev), torch.full((1, H, 1), 0.1, dtype=torch.float64, device=dev)
noise = torch.randn((1, P, H, 1), dtype=torch.float64, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{kernel} {layers if kernel == "nn" else ""} P={P} H={H}: {dt*1e3:.3f} ms  {P*H/dt:.3e} particle-steps/s  '
          f'status {int(r["status"].item())}', flush=True)
