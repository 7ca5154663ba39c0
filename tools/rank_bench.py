"""Times sx_cem_rank_refit on the shapes of the BASELINE configs: python tools/rank_bench.py (path chosen by the launcher)
or SX_RANK_PATH=select|count python tools/rank_bench.py (forced; the variable is read once per process)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cem as ocem  # noqa: E402  (checker only)
from safe_exploration_amd.cem_mpc import cem_rank_refit  # noqa: E402

dev = torch.device('cuda:0')
REFIT = '--norefit' not in sys.argv
shapes = [(1, 4096, 409, 15), (1, 8192, 819, 30), (1, 3272, 409, 15), (1, 6552, 819, 30), (2, 4096, 409, 15),
          (4, 4096, 409, 15), (8, 4096, 409, 15), (1, 64, 8, 5), (1, 1024, 100, 15), (1, 2048, 204, 15)]
print('SX_RANK_PATH =', os.environ.get('SX_RANK_PATH', '(launcher)'))
for E, P, k, L in shapes:
    rng = np.random.default_rng(P + k)
    con = rng.choice([0., 0., 3., 10., 13., 20.], size=(E, P))
    obj = rng.normal(size=(E, P))
    act = rng.normal(size=(E, P, L))
    tc, to, ta = (torch.tensor(x, device=dev) for x in (con, obj, act))
    out = cem_rank_refit(tc, to, ta, k, want_rows=True)
    torch.cuda.synchronize()
    ok = True
    for e in range(E):
        want = ocem.rank(con[e], obj[e], k)
        got = out['elite_idx'][e].cpu().numpy()
        m, s = ocem.refit(act[e][want])
        ok &= got[0] == want[0] and set(got.tolist()) == set(want.tolist())
        ok &= np.allclose(out['mean'][e].cpu().numpy(), m, rtol=1e-12, atol=1e-14)
        ok &= np.allclose(out['std'][e].cpu().numpy(), s, rtol=1e-12, atol=1e-14)
    n = 200
    for _ in range(20):
        cem_rank_refit(tc, to, ta, k, want_rows=True, want_refit=REFIT)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        cem_rank_refit(tc, to, ta, k, want_rows=True, want_refit=REFIT)
    b.record()
    torch.cuda.synchronize()
    print(f'E={E} P={P} k={k} L={L}: {a.elapsed_time(b) / n * 1e3:7.1f} us per call (back to back, host-bound below ~10 us)  '
          f'{"ok" if ok else "MISMATCH"}', flush=True)
