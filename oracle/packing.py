"""Oracle: free <-> vector packing maps with analytic first and second derivatives.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, in closed form:
  box      LRVB/Parameters.py:31-61 (constrain / unconstrain), :15-28 (bounds checks)
  psd      LRVB/MatrixParameters.py:16-23 (SymIndex), :38-41, :59-68, :83-112 (pack / unpack)
  simplex  LRVB/SimplexParams.py:11-23 (constrain / unconstrain), :33-63 (jac / hess from moments)
  layout   LRVB/ParameterDictionary.py:39-46, 55-99 (concatenation in push order)
  convert_vector_to_free_hessian   LRVB/Parameters.py:397-424
The reference obtains box and psd derivatives from autograd; the closed forms here are derived
in DESIGN.md and checked against torch-fp64 AD in tests/test_oracle_vs_torch_ad.py.
"""
import numpy as np

BOX, PSD, SIMPLEX = 0, 1, 2


# ---------------------------------------------------------------------------------- box
def box_constrain(f, lb, ub):
    """eta(f) elementwise; returns (eta, eta', eta'').  LRVB/Parameters.py:47-61."""
    f = np.asarray(f, dtype=np.float64)
    if not ub > lb:
        raise ValueError('Upper bound must be greater than lower bound')
    has_lb, has_ub = np.isfinite(lb), np.isfinite(ub)
    if not has_lb and not has_ub:
        return f.copy(), np.ones_like(f), np.zeros_like(f)
    if has_lb and not has_ub:
        e = np.exp(f)
        return e + lb, e, e.copy()
    if has_ub and not has_lb:
        e = np.exp(-f)
        return ub - e, e, -e
    # two-sided: (ub - lb) * logistic(f) + lb, evaluated in the overflow-free form
    ef = np.exp(-np.abs(f))
    s = np.where(f >= 0, 1.0 / (1.0 + ef), ef / (1.0 + ef))
    r = ub - lb
    sp = s * (1.0 - s)
    return r * s + lb, r * sp, r * sp * (1.0 - 2.0 * s)


def box_unconstrain(v, lb, ub):
    """LRVB/Parameters.py:15-44, including the out-of-bounds ValueErrors."""
    v = np.asarray(v, dtype=np.float64)
    if not ub > lb:
        raise ValueError('Upper bound must be greater than lower bound')
    if not np.all(v <= ub):
        raise ValueError('Elements larger than the upper bound')
    if not np.all(v >= lb):
        raise ValueError('Elements smaller than the lower bound')
    has_lb, has_ub = np.isfinite(lb), np.isfinite(ub)
    with np.errstate(divide='ignore'):
        if not has_lb and not has_ub:
            return v.copy()
        if has_lb and not has_ub:
            return np.log(v - lb)
        if has_ub and not has_lb:
            return -np.log(ub - v)
        return np.log(v - lb) - np.log(ub - v)


# ---------------------------------------------------------------------------------- psd
def ld_index(a, b):
    """Row-major lower-triangle index, a >= b.  LRVB/MatrixParameters.py:16-23."""
    return b + a * (a + 1) // 2


def psd_size(k):
    return k * (k + 1) // 2


def _tril_pairs(k):
    rows, cols = np.tril_indices(k)      # row-major lower triangle == reference order
    return rows, cols


def psd_chol_from_free(f, k):
    L = np.zeros((k, k))
    r, c = _tril_pairs(k)
    L[r, c] = f
    d = np.arange(k)
    L[d, d] = np.exp(L[d, d])
    return L


def psd_constrain(f, k, diag_lb=0.0):
    """free (k(k+1)/2) -> lower triangle of A = L L^T + diag_lb I.  MatrixParameters.py:108-112."""
    L = psd_chol_from_free(np.asarray(f, dtype=np.float64), k)
    A = L @ L.T + diag_lb * np.eye(k)
    r, c = _tril_pairs(k)
    return A[r, c]


def psd_matrix_from_vector(v, k):
    """Symmetric matrix from its lower-triangle vector.  MatrixParameters.py:117-125."""
    A = np.zeros((k, k))
    r, c = _tril_pairs(k)
    A[r, c] = v
    A[c, r] = v
    return A


def psd_unconstrain(v, k, diag_lb=0.0):
    """MatrixParameters.py:101-105: Cholesky of (A - diag_lb I), log of its diagonal."""
    A = psd_matrix_from_vector(np.asarray(v, dtype=np.float64), k)
    L = np.linalg.cholesky(A - diag_lb * np.eye(k))
    d = np.arange(k)
    L[d, d] = np.log(L[d, d])
    r, c = _tril_pairs(k)
    return L[r, c]


def psd_jac(f, k):
    """d vec(A)_(i,j) / d f_(a,b) = dL_ab (d_ia L_jb + d_ja L_ib), dL_ab = L_aa if a == b else 1."""
    L = psd_chol_from_free(np.asarray(f, dtype=np.float64), k)
    m = psd_size(k)
    r, c = _tril_pairs(k)
    J = np.zeros((m, m))
    for col in range(m):
        a, b = r[col], c[col]
        dL = L[a, a] if a == b else 1.0
        # rows (i, j) with i == a: L[j, b];  rows with j == a: L[i, b]
        for row in range(m):
            i, j = r[row], c[row]
            v = 0.0
            if i == a:
                v += L[j, b]
            if j == a:
                v += L[i, b]
            J[row, col] = dL * v
    return J


def psd_third(f, k, g):
    """sum_(i>=j) g_(ij) d2 A_ij / d f d f^T  as an (m x m) matrix."""
    f = np.asarray(f, dtype=np.float64)
    L = psd_chol_from_free(f, k)
    m = psd_size(k)
    r, c = _tril_pairs(k)
    Gs = psd_matrix_from_vector(np.asarray(g, dtype=np.float64), k)   # symmetric lookup of g
    T = np.zeros((m, m))
    for row in range(m):
        a, b = r[row], c[row]
        dLab = L[a, a] if a == b else 1.0
        for col in range(m):
            cc, d = r[col], c[col]
            if b == d:
                dLcd = L[cc, cc] if cc == d else 1.0
                T[row, col] += dLab * dLcd * Gs[a, cc] * (2.0 if a == cc else 1.0)
        if a == b:
            s = 2.0 * Gs[a, a] * L[a, a] + float(np.dot(Gs[a + 1:, a], L[a + 1:, a]))
            T[row, row] += L[a, a] * s
    return T


# ---------------------------------------------------------------------------------- simplex
def simplex_constrain(f, rows, K):
    """(rows, K-1) free -> (rows, K) probabilities; category 0 is the reference.
    LRVB/SimplexParams.py:11-18."""
    F = np.asarray(f, dtype=np.float64).reshape(rows, K - 1)
    Faug = np.hstack([np.zeros((rows, 1)), F])
    mx = Faug.max(axis=1, keepdims=True)
    E = np.exp(Faug - mx)
    return E / E.sum(axis=1, keepdims=True)


def simplex_unconstrain(p, rows, K):
    """LRVB/SimplexParams.py:21-23."""
    Pm = np.asarray(p, dtype=np.float64).reshape(rows, K)
    return (np.log(Pm[:, 1:]) - np.log(Pm[:, :1])).ravel()


def simplex_row_jac(p):
    """J[k, j] = p_k (d_{k,j+1} - p_{j+1});  equals LRVB/SimplexParams.py:33-38."""
    K = p.shape[0]
    J = -np.outer(p, p[1:])
    J[np.arange(1, K), np.arange(K - 1)] += p[1:]
    return J


def simplex_row_hess(p):
    """H[k, i, j] = p_k [ (d_{k,i+1} - p_{i+1})(d_{k,j+1} - p_{j+1}) - p_{i+1}(d_ij - p_{j+1}) ];
    equals the construction at LRVB/SimplexParams.py:42-63 (doc/simplex_derivatives.lyx)."""
    K = p.shape[0]
    q = p[1:]
    E = np.zeros((K, K - 1))
    E[np.arange(1, K), np.arange(K - 1)] = 1.0
    Dm = E - q[None, :]                                   # (d_{k,i+1} - p_{i+1})
    common = np.diag(q) - np.outer(q, q)                  # p_{i+1}(d_ij - p_{j+1})
    return p[:, None, None] * (Dm[:, :, None] * Dm[:, None, :] - common[None, :, :])


# ---------------------------------------------------------------------------------- layout
class Block(object):
    def __init__(self, kind, free_size, vec_size, dim0=0, dim1=0, lb=-np.inf, ub=np.inf, name=''):
        self.kind, self.free_size, self.vec_size = kind, int(free_size), int(vec_size)
        self.dim0, self.dim1, self.lb, self.ub, self.name = int(dim0), int(dim1), float(lb), float(ub), name
        self.free_off = 0
        self.vec_off = 0


def box_block(n, lb=-np.inf, ub=np.inf, name=''):
    return Block(BOX, n, n, dim0=n, lb=lb, ub=ub, name=name)


def psd_block(k, diag_lb=0.0, name=''):
    return Block(PSD, psd_size(k), psd_size(k), dim0=k, lb=diag_lb, name=name)


def simplex_block(rows, K, name=''):
    return Block(SIMPLEX, rows * (K - 1), rows * K, dim0=rows, dim1=K, name=name)


class Layout(object):
    """Concatenation of blocks in push order (LRVB/ParameterDictionary.py:39-46)."""

    def __init__(self, blocks):
        self.blocks = []
        self.D = 0
        self.V = 0
        for b in blocks:
            b.free_off, b.vec_off = self.D, self.V
            self.D += b.free_size
            self.V += b.vec_size
            self.blocks.append(b)

    def _check_free(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.size != self.D:
            raise ValueError('Wrong size for parameter.  Expected {}, got {}'.format(self.D, theta.size))
        return theta.ravel()

    def constrain(self, theta):
        theta = self._check_free(theta)
        eta = np.empty(self.V)
        for b in self.blocks:
            f = theta[b.free_off:b.free_off + b.free_size]
            if b.kind == BOX:
                v = box_constrain(f, b.lb, b.ub)[0]
            elif b.kind == PSD:
                v = psd_constrain(f, b.dim0, b.lb)
            else:
                v = simplex_constrain(f, b.dim0, b.dim1).ravel()
            eta[b.vec_off:b.vec_off + b.vec_size] = v
        return eta

    def unconstrain(self, eta):
        eta = np.asarray(eta, dtype=np.float64).ravel()
        if eta.size != self.V:
            raise ValueError('Wrong size for parameter.  Expected {}, got {}'.format(self.V, eta.size))
        theta = np.empty(self.D)
        for b in self.blocks:
            v = eta[b.vec_off:b.vec_off + b.vec_size]
            if b.kind == BOX:
                f = box_unconstrain(v, b.lb, b.ub)
            elif b.kind == PSD:
                f = psd_unconstrain(v, b.dim0, b.lb)
            else:
                f = simplex_unconstrain(v, b.dim0, b.dim1)
            theta[b.free_off:b.free_off + b.free_size] = f
        return theta

    def jac(self, theta):
        """Dense d eta / d theta (V x D), block diagonal (ParameterDictionary.py:70-78)."""
        theta = self._check_free(theta)
        J = np.zeros((self.V, self.D))
        for b in self.blocks:
            f = theta[b.free_off:b.free_off + b.free_size]
            fs, vs = slice(b.free_off, b.free_off + b.free_size), slice(b.vec_off, b.vec_off + b.vec_size)
            if b.kind == BOX:
                J[vs, fs] = np.diag(box_constrain(f, b.lb, b.ub)[1])
            elif b.kind == PSD:
                J[vs, fs] = psd_jac(f, b.dim0)
            else:
                P = simplex_constrain(f, b.dim0, b.dim1)
                K = b.dim1
                for r in range(b.dim0):
                    J[b.vec_off + r * K: b.vec_off + (r + 1) * K,
                      b.free_off + r * (K - 1): b.free_off + (r + 1) * (K - 1)] = simplex_row_jac(P[r])
        return J

    def third_order(self, theta, g_eta):
        """sum_k g_k d2 eta_k / d theta d theta^T (D x D); the sparse-list contraction of
        LRVB/Parameters.py:405-417 written densely."""
        theta = self._check_free(theta)
        g_eta = np.asarray(g_eta, dtype=np.float64).ravel()
        T = np.zeros((self.D, self.D))
        for b in self.blocks:
            f = theta[b.free_off:b.free_off + b.free_size]
            g = g_eta[b.vec_off:b.vec_off + b.vec_size]
            fs = slice(b.free_off, b.free_off + b.free_size)
            if b.kind == BOX:
                T[fs, fs] = np.diag(g * box_constrain(f, b.lb, b.ub)[2])
            elif b.kind == PSD:
                T[fs, fs] = psd_third(f, b.dim0, g)
            else:
                P = simplex_constrain(f, b.dim0, b.dim1)
                K = b.dim1
                for r in range(b.dim0):
                    Hr = simplex_row_hess(P[r])
                    o = b.free_off + r * (K - 1)
                    T[o:o + K - 1, o:o + K - 1] = np.tensordot(g[r * K:(r + 1) * K], Hr, axes=(0, 0))
        return T


def convert_vector_to_free_hessian(layout, theta, vector_grad, vector_hess):
    """H_free = J^T H_vec J + sum_k g_k d2 eta_k.  LRVB/Parameters.py:397-424."""
    J = layout.jac(theta)
    return J.T @ np.asarray(vector_hess) @ J + layout.third_order(theta, vector_grad)
