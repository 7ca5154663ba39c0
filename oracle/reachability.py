"""Oracle: ellipsoidal one-step reachability + polytope test (numpy float64, batched).
TEST INFRASTRUCTURE - see oracle/__init__.py.

PINNED against the reference run in the build container (tests/golden/make_golden.py) and the reference's
known-answer tests.  Every function names the reference lines it restates (paths relative to the reference root).
"""
import numpy as np

STATUS_NAN = 1        # a NaN reached one of the three checks  -> reference raises ValueError
STATUS_ZERO_FIX = 2   # an exact zero was clamped to 1e-5       -> reference prints a warning and carries on
STATUS_UB_NONPOS = 4  # ellipsoid_from_rectangle got u_b <= 0   -> reference assertion fails


def fix_zeros_nans(x):
    """safe_exploration/gp_reachability_pytorch.py:234-243.

    Whole-batch semantics: any NaN -> failure; any exact zero -> every element <= 0 becomes 1e-5.
    Returns (x, ok, zero_found).
    """
    if np.isnan(x).any():
        return np.empty_like(x), False, False
    if (x == 0).any():
        x = x.copy()
        x[x <= 0] = 1e-5
        return x, True, True
    return x, True, False


def ellipsoid_from_rectangle(u_b):
    """safe_exploration/utils_ellipsoid.py:282-309: Q = diag(n * u_b^2); u_b must be > 0.   [P x n] -> [P x n x n]"""
    if not np.all(u_b > 0):
        raise AssertionError('All elements of u_b must be >0')
    n = u_b.shape[1]
    d = n * u_b ** 2
    q = np.zeros((u_b.shape[0], n, n))
    idx = np.arange(n)
    q[:, idx, idx] = d
    return q


def sum_two_ellipsoids(p_1, q_1, p_2, q_2):
    """safe_exploration/utils_ellipsoid.py:102-140 (trace_batch: utils.py:668-679).

    c = sqrt(tr q_1 / tr q_2);  p = p_1 + p_2;  q = (1 + 1/c) q_1 + (1 + c) q_2
    """
    c = np.sqrt(np.trace(q_1, axis1=1, axis2=2) / np.trace(q_2, axis1=1, axis2=2))[:, None, None]
    return p_1 + p_2, (1 + (1. / c)) * q_1 + (1 + c) * q_2


def compute_remainder_overapproximations(q, k_fb, l_mu, l_sigma):
    """safe_exploration/utils.py:152-194 (eigenvalues_batch: utils.py:651-665).

    q [P x n_s x n_s], k_fb [n_u x n_s] shared, l_mu/l_sigma [n_s] shared -> u_mu, u_sigma [P x n_s].
    Uses the general (non-symmetric) eigen-solver on Q.B exactly like the reference and, like it, refuses
    complex eigenvalues.
    """
    n_u, n_s = k_fb.shape
    s = np.hstack((np.eye(n_s), k_fb.T))
    b = s @ s.T
    qb = q @ b
    evals = np.linalg.eigvals(qb)
    if np.iscomplexobj(evals):
        assert not (np.abs(evals.imag) > 1e-12 * (1.0 + np.abs(evals.real))).any(), 'All imaginary parts should be 0'
        evals = evals.real
    r_sqr = evals.max(axis=1)
    u_mu = l_mu[None, :] * r_sqr[:, None]
    u_sigma = l_sigma[None, :] * np.sqrt(r_sqr)[:, None]
    return u_mu, u_sigma


def onestep_reachability(p_center, gp, k_ff, l_mu, l_sigma, q_shape=None, k_fb=None, c_safety=1., a=None, b=None):
    """safe_exploration/gp_reachability_pytorch.py:18-181.

    gp: object with predict(z, jacobians) -> (mean [P x n_s], var [P x n_s], jac [P x n_s x D]).
    Returns (p_1 [P x n_s], q_1 [P x n_s x n_s], sigma [P x n_s], status bits).
    Raises ValueError where the reference does (NaN after the zero/NaN fix-up).
    """
    P, n_s = p_center.shape
    n_u = k_ff.shape[1]
    status = 0
    if a is None:  # :60-62
        a = np.eye(n_s)
        b = np.zeros((n_s, n_u))
    z = np.concatenate((p_center, k_ff), axis=1)

    if q_shape is None:  # point branch :65-99
        mu_0, sigm_0, _ = gp.predict(z, jacobians=False)
        sigm_0, ok, zf = fix_zeros_nans(sigm_0)
        status |= STATUS_ZERO_FIX if zf else 0
        if not ok:
            raise ValueError('nan in sigm_0')
        rkhs_bounds = c_safety * np.sqrt(sigm_0)
        rkhs_bounds, ok, zf = fix_zeros_nans(rkhs_bounds)
        status |= STATUS_ZERO_FIX if zf else 0
        if not ok:
            raise ValueError('nan/zero in rkhs_bounds')
        q_1 = ellipsoid_from_rectangle(rkhs_bounds)
        p_1 = p_center @ a.T + k_ff @ b.T + mu_0
        return p_1, q_1, sigm_0, status

    # ellipsoid branch :100-181
    mu_0, sigm_0, jac_mu = gp.predict(z, jacobians=True)
    sigm_0, ok, zf = fix_zeros_nans(sigm_0)
    status |= STATUS_ZERO_FIX if zf else 0
    if not ok:
        raise ValueError('nan in sigm_0')
    a_mu = jac_mu[:, :, :n_s]
    b_mu = jac_mu[:, :, n_s:]
    H = a[None] + a_mu + (b_mu + b[None]) @ k_fb                      # :131
    p_0 = mu_0 + p_center @ a.T + k_ff @ b.T                           # :132
    Q_0 = H @ q_shape @ H.transpose(0, 2, 1)                           # :134
    ub_mean, ub_sigma = compute_remainder_overapproximations(q_shape, k_fb, l_mu, l_sigma)  # :145
    b_sigma_eps = c_safety * (np.sqrt(sigm_0) + ub_sigma)              # :146
    b_sigma_eps, ok, zf = fix_zeros_nans(b_sigma_eps)
    status |= STATUS_ZERO_FIX if zf else 0
    if not ok:
        raise ValueError('nan in b_sigma_eps')
    Q_lagrange_sigm = ellipsoid_from_rectangle(b_sigma_eps)            # :155
    Q_lagrange_mu = ellipsoid_from_rectangle(ub_mean)                  # :162
    zeros = np.zeros((P, n_s))
    p_sum, Q_sum = sum_two_ellipsoids(zeros, Q_lagrange_sigm, zeros, Q_lagrange_mu)  # :169
    p_1, q_1 = sum_two_ellipsoids(p_sum, Q_sum, p_0, Q_0)              # :172
    return p_1, q_1, sigm_0, status


def lin_ellipsoid_safety_distance(p_center, q_shape, h_mat, h_vec, c_safety=1.0):
    """safe_exploration/gp_reachability_pytorch.py:184-215: d[:, j] = h_j.p + c sqrt(h_j^T Q h_j) - b_j.

    p [P x n_s], q [P x n_s x n_s], h_mat [m x n_s], h_vec [m x 1] -> [P x m]
    """
    d_center = p_center @ h_mat.T
    d_shape = c_safety * np.sqrt(np.einsum('mi,pij,mj->pm', h_mat, q_shape, h_mat))
    return d_center + d_shape - h_vec.reshape(1, -1)


def is_ellipsoid_inside_polytope(p_center, q_shape, h_mat, h_vec):
    """safe_exploration/gp_reachability_pytorch.py:218-231: inside <=> no d >= 0 (a NaN distance counts as inside)."""
    d = lin_ellipsoid_safety_distance(p_center, q_shape, h_mat, h_vec)
    return (d >= 0).sum(axis=1) == 0


def pq_flatten(p, q):
    """safe_exploration/safempc_cem.py:40-54: [p | vec_rowmajor(Q)], q=None -> zeros."""
    P, n_s = p.shape
    if q is None:
        q = np.zeros((P, n_s, n_s))
    return np.concatenate((p.reshape(P, -1), q.reshape(P, -1)), axis=1)


def pq_unflatten(flat, n_s):
    """safe_exploration/safempc_cem.py:56-73: an all-zero Q block over the WHOLE batch means q=None."""
    P = flat.shape[0]
    p = flat[:, :n_s]
    q = flat[:, n_s:].reshape(P, n_s, n_s)
    if np.count_nonzero(q) == 0:
        q = None
    return p, q
