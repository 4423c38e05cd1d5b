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
    blk = 256                                                      # rows whose outer products are added to R in one product
    XX, AA = np.zeros((blk, q * q)), np.zeros((blk, K * K))
    for n in range(N):
        p = Z[n]
        g = -w[n] * (S[n] - np.log(p) - 1.0)                      # d l_n / d z_n
        J = packing.simplex_row_jac(p)
        T = packing.simplex_row_hess(p)
        gfree[n] = J.T @ g
        Hloc[n] = J.T @ np.diag(w[n] / p) @ J + np.einsum('k,kij->ij', g, T)
        A = J @ np.linalg.solve(Hloc[n], J.T)
        # R += w_n^2 vec(x~ x~^T) vec(A)^T, the rows of a block summed by one matrix product (N = 1e4 rows of K = 32 would
        # otherwise write 1e10 doubles one outer product at a time)
        XX[n % blk] = w[n] ** 2 * np.outer(Xt[n], Xt[n]).ravel()
        AA[n % blk] = A.ravel()
        if n % blk == blk - 1 or n == N - 1:
            used = n % blk + 1
            R += XX[:used].T @ AA[:used]
    U = np.zeros((N, 64))
    U[:, :q] = Xt
    U[:, 32:32 + K] = Z
    S64 = U.T @ (w[:, None] * U)
    return val2, gfree, Hloc, S64, R


def mixture_schur(K, q, R, jlam, hgg, scale=None, diag_add=None):
    """Schur complement of the global block: diag(scale) hgg diag(scale) + diag(diag_add) - sym(jlam^T Rm jlam)
    with Rm[(j K + k), (j' K + k')] = R[(j q + j'), (k K + k')].  This is the reference's
    H_free = J^T H_vec J + sum_k (df/d eta_k) d2 eta_k (LRVB/Parameters.py:397-424) applied to the
    block that remains after eliminating the per-observation simplex rows (doc/sensitivity.lyx)."""
    n = K * q
    R = np.asarray(R, dtype=np.float64).reshape(q, q, K, K)
    Rm = R.transpose(0, 2, 1, 3).reshape(n, n)
    jlam = np.asarray(jlam, dtype=np.float64)
    S = jlam.T @ Rm @ jlam
    H = np.array(hgg, dtype=np.float64)
    if scale is not None:
        H = H * np.asarray(scale)[:, None] * np.asarray(scale)[None, :]
    if diag_add is not None:
        H = H + np.diag(np.asarray(diag_add))
    return H - 0.5 * (S + S.T)
