import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from safe_exploration_amd import problems
from safe_exploration_amd.cem_mpc import FusedCemMpc
dev = torch.device('cuda:0')
cands = {
 'e': (np.array([[0.8,1.0,0.7],[0.9,0.75,1.2]]), np.array([0.006,0.004]), np.array([1e-5,2e-5])),
 'f': (np.array([[0.9,1.1,0.8],[1.0,0.85,1.3]]), np.array([0.008,0.005]), np.array([1e-5,1.5e-5])),
 'g': (np.array([[0.75,0.7,0.7],[0.7,0.72,0.75]]), np.array([0.009,0.0085]), np.array([1e-5,1.1e-5])),
 'h': (np.array([[0.7,0.7,0.7],[0.7,0.7,0.7]]), np.array([0.01,0.01]), np.array([1e-5,1e-5])),
 'i': (np.array([[1.0,1.2,0.9],[1.1,0.95,1.4]]), np.array([0.005,0.003]), np.array([1e-5,2e-5])),
}
w = problems.baseline_workload(2)
for name,(ls,os_,nz) in cands.items():
    spec = w.spec
    spec.lengthscale, spec.outputscale, spec.noise = ls, os_, nz
    ssm, env = problems.build(spec, dev)
    mpc = FusedCemMpc(ssm, env, w.horizon, w.particles, w.elites, w.iterations, device=dev, seed=1, init_std=w.init_std)
    x0 = torch.tensor(w.x0[:1], dtype=torch.float64, device=dev)
    oks=[]
    for _ in range(6):
        best, ok, _, status = mpc.solve(x0)
        oks.append(int(ok[0].item()))
    print(name, oks, int(status.item()))
