"""Randomised check of sx_cem_rank_refit (both kernels, chosen by shape) against the oracle's ranking: random shapes, few
distinct constraint costs, exact ties, NaNs, +-inf, strided candidate rows.  python tools/rank_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cem as ocem  # noqa: E402  (checker only)
from safe_exploration_amd import _lib  # noqa: E402
from safe_exploration_amd.cem_mpc import cem_rank_refit  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
bad = counted = 0
for case in range(cases):
    E = int(rng.choice([1, 1, 1, 2, 3, 5]))
    P = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), rng.integers(600, 8193), rng.integers(8193, 16385)],
                       p=[.2, .3, .4, .1]))
    k = int(rng.integers(1, min(P, 2048) + 1)) if rng.random() < .3 else max(1, min(P, 2048) // int(rng.integers(2, 20)))
    L = int(rng.choice([1, 3, 15, 30, 33, 70]))
    con = rng.choice([0., 0., 0., 3., 10., 13., 20., 23.], size=(E, P))
    if rng.random() < .3:
        con = rng.normal(size=(E, P)).round(int(rng.integers(0, 3)))         # arbitrary doubles, many ties
    obj = rng.normal(size=(E, P))
    if rng.random() < .5:
        obj = obj.round(int(rng.integers(0, 3)))                             # exact ties in the objective too
    for arr in (con, obj):
        if rng.random() < .3:
            arr[rng.integers(0, E), rng.integers(0, P)] = rng.choice([np.nan, np.inf, -np.inf])
    act = rng.normal(size=(E, P, L))
    strided = rng.random() < .3
    if strided:       # candidate-row layout: [con, obj, actions...] rows, as after the multi-GPU exchange
        rows = np.concatenate((con[..., None], obj[..., None], act), axis=2)
        flat = T(rows).reshape(-1)
        out = cem_rank_refit(flat, flat[1:], flat[2:], k, cost_stride=2 + L, act_stride=2 + L, row_len=L, num_candidates=P,
                             num_problems=E, want_rows=True)
    else:
        out = cem_rank_refit(T(con), T(obj), T(act), k, want_rows=True)
    counted += int(_lib.lib().sx_cem_rank_counts(E, P))
    for e in range(E):
        want = ocem.rank(con[e], obj[e], k)
        got = out['elite_idx'][e].cpu().numpy()
        rows_out = out['elite_rows'][e].cpu().numpy()
        ok = len(set(got.tolist())) == k and (got >= 0).all() and (got < P).all()
        ok = ok and set(got.tolist()) == set(want.tolist()) and got[0] == want[0]
        ok = ok and np.array_equal(rows_out[:, 2:], act[e][got]) and np.array_equal(rows_out[:, 0], con[e][got], equal_nan=True) and np.array_equal(rows_out[:, 1], obj[e][got], equal_nan=True)
        m, sd = ocem.refit(act[e][got])
        ok = ok and np.allclose(out['mean'][e].cpu().numpy(), m, rtol=1e-11, atol=1e-13)
        ok = ok and np.allclose(out['std'][e].cpu().numpy(), sd, rtol=1e-11, atol=1e-13)
        if not ok:
            bad += 1
            print(f'MISMATCH case {case}: E={E} P={P} k={k} L={L} strided={strided} problem {e}', flush=True)
print(f'{cases} cases ({counted} by the counting kernel), {bad} mismatches')
sys.exit(1 if bad else 0)
