"""Small host helpers shared by the boundary classes (same contracts as the reference's utils)."""
from typing import Any, Optional

import numpy as np
import scipy.linalg as sla
import torch


def assert_shape(x, shape: tuple, ignore_if_none: bool = False) -> None:
    """ValueError on a shape mismatch -- the reference's error behaviour (safe_exploration/utils.py:640-648)."""
    if x is None:
        if ignore_if_none:
            return
        raise ValueError(f'Wanted shape {shape}, got None')
    if tuple(x.shape) != tuple(shape):
        raise ValueError(f'Wanted shape {shape}, got {tuple(x.shape)}')


def get_device(force_device: Optional[Any] = None) -> str:
    """Device rule of the reference (utils.py:693-709): explicit string, else conf.device, else cuda:0 if present."""
    if isinstance(force_device, str):
        return force_device
    if force_device is not None:
        dev = getattr(force_device, 'device', None)
        if dev is not None:
            return dev
    return 'cuda:0' if torch.cuda.is_available() else 'cpu'


def dlqr(a, b, q, r):
    """Infinite-horizon discrete LQR gain for x+ = a x + b u, u = -k x (reference utils.py:23-38).

    Returns (k, x, closed-loop eigenvalues).
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    r = np.atleast_2d(np.asarray(r, dtype=np.float64))
    x = sla.solve_discrete_are(a, b, q, r)
    k = np.linalg.solve(b.T @ x @ b + r, b.T @ x @ a)
    return k, x, np.linalg.eigvals(a - b @ k)
