"""Diagnostic (SX_STAMPS build): cycle shares of the rank/refit kernel's sections."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import _lib
from safe_exploration_amd.cem_mpc import cem_rank_refit
dev = torch.device('cuda:0')
P, k, L = int(os.environ.get('P', 4096)), int(os.environ.get('K', 409)), int(os.environ.get('L', 15))
NCON = int(os.environ.get('NCON', 2))   # distinct constraint-cost values (H=30 workloads have dozens)
buf = torch.zeros(32, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.sx_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.sx_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device=dev); g.manual_seed(0)
for feas in (0.5, 0.02):
    con = (torch.rand((1, P), device=dev, generator=g, dtype=torch.float64) > feas).double() * 10 * torch.randint(1, NCON, (1, P), device=dev, generator=g).double()
    obj = torch.randn((1, P), device=dev, generator=g, dtype=torch.float64)
    act = torch.randn((1, P, L), device=dev, generator=g, dtype=torch.float64)
    for _ in range(3):
        cem_rank_refit(con, obj, act, k)
    torch.cuda.synchronize()
    print(f'feasible fraction {feas}: select/compact/sort/output+refit cycles =', buf[:4].tolist(),
          ' select = best/walk/passes', buf[4:7].tolist(), 'in', int(buf[7]), 'radix passes; first pass = count/barrier/scan', buf[8:11].tolist())
    if os.environ.get('SX_RANK_PATH', '')[:1] != 's':
        cem_rank_refit(con, obj, act, k, want_rows=True)
        torch.cuda.synchronize()
        print('  counting kernel (workgroup 0): keys->LDS / barrier / count / reduce / rows cycles =', buf[16:21].tolist(),
              '; last workgroup: ticket', int(buf[21]), 'refit', int(buf[22]), 'whole life', int(buf[23]))
