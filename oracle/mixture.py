"""CPU restatement of the per-observation simplex rows of a mixture model -- TEST INFRASTRUCTURE.

Row n of a SimplexParam (LRVB/SimplexParams.py:69-175) enters the objective through
l_n = -w_n z_n . s_n + w_n z_n . log z_n with s_n = x~_n Lam.  The reference obtains the free
Hessian of such a row from `constrain_grad_from_moment` / `constrain_hess_from_moment`
(SimplexParams.py:33-63) through `convert_vector_to_free_hessian` (Parameters.py:57-77); this module
restates exactly that composition with plain per-row Python loops (small N only).
"""
import numpy as np

from . import packing


def mixture_rows(theta_z, X, w, Lam):
    """Returns val2 = [-sum w z.s, sum w z log z], the free local gradient (N, K-1), the local
    free Hessian blocks (N, K-1, K-1), S64 = [x~|z]^T diag(w) [x~|z] with x~ padded to 32 columns and
    z to 32, and R[(j, j'), (k, k')] = sum_n w_n^2 x~_nj x~_nj' (J_n H_nn^-1 J_n^T)[k, k']."""
    X = np.asarray(X, dtype=np.float64)
    N, V = X.shape
    q, K = Lam.shape
    assert q == V + 1
    Z = packing.simplex_constrain(theta_z, N, K)
    Xt = np.hstack([np.ones((N, 1)), X])
    S = Xt @ Lam
    val2 = np.array([-np.sum(w[:, None] * Z * S), np.sum(w[:, None] * Z * np.log(Z))])
    gfree = np.zeros((N, K - 1))
    Hloc = np.zeros((N, K - 1, K - 1))
    R = np.zeros((q * q, K * K))
    for n in range(N):
        p = Z[n]
        g = -w[n] * (S[n] - np.log(p) - 1.0)                      # d l_n / d z_n
        J = packing.simplex_row_jac(p)
        T = packing.simplex_row_hess(p)
        gfree[n] = J.T @ g
        Hloc[n] = J.T @ np.diag(w[n] / p) @ J + np.einsum('k,kij->ij', g, T)
        A = J @ np.linalg.solve(Hloc[n], J.T)
        R += w[n] ** 2 * np.outer(np.outer(Xt[n], Xt[n]).ravel(), A.ravel())
    U = np.zeros((N, 64))
    U[:, :q] = Xt
    U[:, 32:32 + K] = Z
    S64 = U.T @ (w[:, None] * U)
    return val2, gfree, Hloc, S64, R
