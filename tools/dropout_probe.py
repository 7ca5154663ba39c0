"""Times the CEM rollout over an MC-dropout ensemble (the reference's default network: 64 x 64 hidden units, 30 members) at
config 2's shape (pendulum, 4096 particles, H = 15): the matrix-core kernel (cem_rollout_mlp_mfma_kernel), and with
PATHS=mfma,valu also the one-particle-per-lane kernel (cem_rollout_mlp_kernel).  HIDDEN=64,64  P=4096 override the shape."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import _lib, problems
from safe_exploration_amd.cem_mpc import cem_rollout
from safe_exploration_amd.gp_reachability_pytorch import make_env
from safe_exploration_amd.ssm_cem.dropout_ssm_cem import McDropoutSSM


class Conf:
    mc_dropout_training_iterations, mc_dropout_num_samples, mc_dropout_predict_std, mc_dropout_reinitialize = int(os.environ.get("ITERS", 200)), 30, False, False
    mc_dropout_hidden_features = [int(v) for v in os.environ.get('HIDDEN', '64,64').split(',')]
    mc_dropout_type, mc_dropout_fixed_probability, mc_dropout_on_input, device = 'fixed', 0.1, False, 'cuda:0'


dev = torch.device('cuda:0')
spec = problems.pendulum(200, model_error=0.02)
ssm = McDropoutSSM(Conf(), 2, 1)
T = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
ssm.update_model(T(spec.X), T(spec.Y), replace_old=True)
env = make_env(2, 1, a=spec.a, b=spec.b, k_fb=spec.k_fb, l_mu=spec.l_mu, l_sigma=spec.l_sigma, beta=spec.beta, h_mat=spec.h_mat,
               h_vec=spec.h_vec, u_min=spec.u_min, u_max=spec.u_max)
P, H = int(os.environ.get('P', 4096)), 15
x0 = T([[0.02, -0.03]])
mean, std = torch.zeros((1, H, 1), dtype=torch.float64, device=dev), torch.full((1, H, 1), 0.1, dtype=torch.float64, device=dev)
noise = torch.randn((1, P, H, 1), dtype=torch.float64, device=dev)
for path in os.environ.get('PATHS', 'mfma').split(','):
    os.environ.pop('SX_MLP_PATH', None)
    if path == 'valu':
        os.environ['SX_MLP_PATH'] = 'valu'
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        w = [3] + Conf.mc_dropout_hidden_features
        macs = 30 * (sum(w[i] * w[i + 1] for i in range(len(w) - 1)) + w[-1] * 2) * 3     # forward + 2 reverse sweeps
        print(f'{path}: hidden {Conf.mc_dropout_hidden_features} S=30 P={P} H={H}: {dt*1e3:.2f} ms  {P*H/dt:.3e} particle-steps/s  '
              f'~{2*macs*P*H/dt/1e12:.2f} TFLOP/s algorithmic  status {int(r["status"].item())}', flush=True)
