"""Diagnostic: where does a rollout step spend its cycles?  Needs the -DSX_STAMPS build:

    SX_OUT=libsxamd_stamps.so SX_EXTRA_FLAGS=-DSX_STAMPS safe_exploration_amd/csrc/build.sh
    SX_LIB=$PWD/safe_exploration_amd/csrc/libsxamd_stamps.so python tools/phase_stamps.py

Reads SHARES only (the stamp build forbids overlaps the product build has; never quote its run time).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import _lib, problems  # noqa: E402
from safe_exploration_amd.cem_mpc import cem_rollout  # noqa: E402

P, H = int(os.environ.get('P', 4096)), int(os.environ.get('H', 15))
spec = getattr(problems, os.environ.get('WHICH', 'pendulum'))(n_train=int(os.environ.get('N', 200)))
ssm, env = problems.build(spec, 'cuda:0')
dev = torch.device('cuda:0')
nwg, nw = (P + 15) // 16, int(os.environ.get('NW', 4))
buf = torch.zeros((nwg * nw, 8), dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.sx_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.sx_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
x0 = torch.tensor([[0.02, -0.03, 0.01, 0.02][:spec.n_s]], dtype=torch.float64, device=dev)
mean = torch.zeros((1, H, 1), dtype=torch.float64, device=dev)
std = torch.full((1, H, 1), 0.1, dtype=torch.float64, device=dev)
noise = torch.randn((1, P, H, 1), dtype=torch.float64, device=dev)
for _ in range(3):
    cem_rollout(ssm, env, x0, H, mean=mean, std=std, noise=noise)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(nwg, nw, 8).astype(np.float64)
s = raw[:, :, :4] / H
clk = np.median(raw[:, 0, 6] / (raw[:, 0, 7] * 10.0))
print(f'shader clock while the kernel runs: {clk:.2f} GHz (s_memtime cycles per s_memrealtime tick; all {nwg} workgroups resident)')
names = ['kstar', 'kstar->barrier', 'mfma', 'mfma->barrier']
print(f'cycles per step (median over {nwg} workgroups), per wave:')
for w in range(nw):
    med = np.median(s[:, w, :], axis=0)
    print(f'  wave {w}: ' + '  '.join(f'{n}={v:8.0f}' for n, v in zip(names, med)) + f'   total={med.sum():8.0f}')
if os.environ.get('SX_ROLLOUT', 'rh') == 'rh' and raw[:, :, 4].any():
    # the 8-wave form packs the launch's fixed parts into slots 4 and 5: prologue | first step's Kstar phase, epilogue | first
    # step's wait before its first barrier (the resident W arriving)
    i4, i5 = buf.cpu().numpy().reshape(nwg, nw, 8)[:, :, 4], buf.cpu().numpy().reshape(nwg, nw, 8)[:, :, 5]
    lo, hi = (lambda a: np.median(a & 0xffffffff, axis=0)), (lambda a: np.median(a >> 32, axis=0))
    print('fixed parts of the launch (cycles, median over workgroups), per wave:')
    for w in range(nw):
        print(f'  wave {w}: prologue={lo(i4)[w]:8.0f}  first kstar={hi(i4)[w]:8.0f}  first kstar->barrier={hi(i5)[w]:8.0f}  '
              f'epilogue={lo(i5)[w]:8.0f}  step loop={np.median(raw[:, w, 6]):9.0f}')
if os.environ.get('SX_ROLLOUT', 'rh') == 'rw' and raw[:, 0, 4].any():
    # the 4-wave form stamps two sections of finish() on wave 0 (slots 4, 5): posterior assembly | reachability step
    print('finish() on wave 0, cycles per step: collect = %.0f   reach_ellipsoid = %.0f   (the rest: costs, stores, next query point)'
          % (np.median(raw[:, 0, 4]) / H, np.median(raw[:, 0, 5]) / H))
