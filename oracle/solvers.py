"""Oracle: linear-response solves.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates:
  ConjugateGradientSolver.get_hinv_vec / get_hinv_vec_subsets   LRVB/ConjugateGradient.py:63-105
      (scipy.sparse.linalg.cg with tol=1e-8; on scipy < 1.0 `tol` is relative to ||b||, i.e.
      today's rtol=tol, atol=0)
  get_masks / split_vector / recursive_split                     LRVB/ConjugateGradient.py:19-57
  ParametricSensitivityLinearApproximation.set_base_values       LRVB/ModelSensitivity.py:585-612
      S = -cho_solve(cho_factor(H), cross)
  LRVB covariance  M H^-1 M^T                                    Example.ipynb:398-415,
                                                                 LRVB/SparseObjectives.py:541-558
  get_sym_matrix_inv_sqrt                                        LRVB/OptimizationUtils.py:6-20
"""
import numpy as np
import scipy.linalg


def cg_solve(matvec, b, x0=None, tol=1e-8, maxiter=None, Minv=None):
    """Textbook preconditioned CG with scipy's stopping rule: stop when ||r|| < tol * ||b||
    (checked at the top of each iteration); returns (x, info) with info = 0 on convergence,
    maxiter otherwise."""
    b = np.asarray(b, dtype=np.float64)
    n = b.size
    if maxiter is None:
        maxiter = 10 * n
    bnorm = np.linalg.norm(b)
    if bnorm == 0.0:
        return np.zeros(n), 0, 0
    atol = tol * bnorm
    if x0 is None:
        x = np.zeros(n)
        r = b.copy()
    else:
        x = np.array(x0, dtype=np.float64)
        r = b - matvec(x)
    rho_prev = 0.0
    p = None
    for it in range(maxiter):
        if np.linalg.norm(r) < atol:
            return x, 0, it
        z = r if Minv is None else Minv @ r
        rho = float(r @ z)
        p = z.copy() if it == 0 else z + (rho / rho_prev) * p
        q = matvec(p)
        alpha = rho / float(p @ q)
        x = x + alpha * p
        r = r - alpha * q
        rho_prev = rho
    return x, maxiter, maxiter


def get_masks(full_len, min_mask_len):
    """Consecutive boolean masks of at most min_mask_len True entries covering range(full_len)."""
    assert min_mask_len > 0 and min_mask_len < full_len
    masks = []
    for start in range(0, full_len, min_mask_len):
        m = np.zeros(full_len, dtype=bool)
        m[start:min(start + min_mask_len, full_len)] = True
        masks.append(m)
    return masks


def split_vector(vec):
    """Two masks holding the first floor(t/2) and the remaining True entries of `vec`."""
    vec = np.asarray(vec, dtype=bool)
    idx = np.flatnonzero(vec)
    half = len(idx) // 2
    a = np.zeros(len(vec), dtype=bool)
    b = np.zeros(len(vec), dtype=bool)
    a[idx[:half]] = True
    b[idx[half:]] = True
    return a, b


def recursive_split(mask, terminate_len=10):
    out = []
    if np.sum(mask) > terminate_len:
        a, b = split_vector(mask)
        out += recursive_split(a, terminate_len)
        out += recursive_split(b, terminate_len)
    else:
        out.append(np.asarray(mask, dtype=bool))
    return out


def linear_response(hess, cross):
    """d theta_hat / d eps^T = -H^-1 cross  (LRVB/ModelSensitivity.py:594-602)."""
    chol = scipy.linalg.cho_factor(np.asarray(hess))
    return -scipy.linalg.cho_solve(chol, np.asarray(cross))


def lrvb_covariance(hess, moment_jac):
    """M H^-1 M^T for a moment Jacobian M (Q x D)."""
    chol = scipy.linalg.cho_factor(np.asarray(hess))
    return np.asarray(moment_jac) @ scipy.linalg.cho_solve(chol, np.asarray(moment_jac).T)


def sym_matrix_inv_sqrt(hessian, ev_min=None, ev_max=None):
    """(H^-1/2, H_corrected) with eigenvalues clamped to [ev_min, ev_max].
    LRVB/OptimizationUtils.py:6-20."""
    hs = 0.5 * (hessian + hessian.T)
    w, U = np.linalg.eigh(hs)
    if ev_min is not None:
        w = np.where(w <= ev_min, ev_min, w)
    if ev_max is not None:
        w = np.where(w >= ev_max, ev_max, w)
    corrected = (U * w) @ U.T
    inv_sqrt = (U / np.sqrt(w)) @ U.T
    return inv_sqrt, corrected
