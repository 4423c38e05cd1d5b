"""Objectives that are quadratic in the data: the conjugate-normal family of models whose
per-observation term is  l_n(eta) = 1/2 z_n^T Q(eta) z_n + c(eta),  so that

    f(eta) = sum_n w_n l_n(eta) + R(eta) = 1/2 tr(Q(eta) S(w)) + W c(eta) + R(eta),
    S(w) = sum_n w_n z_n z_n^T   (device: weighted-SYRK kernel),   W = sum_n w_n.

The O(N) arithmetic -- the weighted sufficient statistics S(w) for every new weight vector, and the
N x V per-observation gradient matrix for weight sensitivities -- runs on the GPU
(`lrvb_weighted_gram`, `lrvb_obs_quadform`); the packing Jacobians / free-coordinate conversion run
on the GPU too (`lrvb_free_to_vector_jac`, `lrvb_free_hessian_from_vector`).  What stays on the
host is the N-independent closed form of Q, c, R and their first two derivatives in the
constrained vector eta (a few Kronecker products of small matrices).

`NormalRegressionObjective` is the model of the reference's Example.ipynb:247-274
(`-loglik = 1/2 sum_n w_n r_n^T Lambda r_n - 1/2 (sum w) log|Lambda|`, r_n = y_n - beta^T x_n,
beta an ArrayParam with lb = 0, Lambda a PosDefMatrixParam) -- BASELINE.json config 1.
"""
import numpy as np

from . import _hip
from .models import DeviceContext
from .packing import VectorParam


def duplication_matrix(k):
    """Dup (k^2 x k(k+1)/2): vec_rowmajor(A) = Dup @ tril_vector(A) for symmetric A, with the
    lower-triangle vector in the reference's row-major order (LRVB/MatrixParameters.py:16-23)."""
    m = k * (k + 1) // 2
    D = np.zeros((k * k, m))
    for i in range(k):
        for j in range(k):
            a, b = (i, j) if i >= j else (j, i)
            D[i * k + j, b + a * (a + 1) // 2] = 1.0
    return D


class QuadraticDataObjective(object):
    """Base functor.  Subclasses provide the N-independent closed forms:

      _terms(eta, S, W)   -> (value, grad (V,), hess (V, V)) in vector coordinates
      _obs_terms(eta)     -> (M (V, q, q) symmetric, c (V,)):  d l_n / d eta_k = 1/2 z_n^T M_k z_n + c_k
    """
    _lrvb_device_functor = True

    def __init__(self, par, z, weights=None, device=0):
        self.par = par
        z = _hip.as_f64(z)
        self.n_obs, self.q = z.shape
        self.ctx = DeviceContext(par.layout_blocks(), loss='data_only', n_obs=self.n_obs, n_cols=self.q, device=device)
        if self.ctx.D != par.free_size() or self.ctx.V != par.vector_size():
            raise ValueError('layout_blocks() of the parameter disagrees with its free/vector sizes')
        self.ctx.set_data(_hip.SLOT_X, z)
        w0 = np.ones(self.n_obs) if weights is None else _hip.as_f64(weights).ravel().copy()
        self.weights_par = VectorParam('weights', self.n_obs, val=w0)
        self.tilt_par = None
        self._w_cache = None
        self._S = None
        self._W = None

    # ---- device state ----------------------------------------------------------------------
    def _push_state(self):
        w = np.asarray(self.weights_par.get_vector(), dtype=np.float64)
        if self._w_cache is None or not np.array_equal(w, self._w_cache):
            self.ctx.set_weights(w)
            self._w_cache = w.copy()
            self._S = None

    def _stats(self):
        self._push_state()
        if self._S is None:
            self._S = self.ctx.weighted_gram()           # GPU: sum_n w_n z_n z_n^T
            self._W = float(np.sum(self._w_cache))
        return self._S, self._W

    def _eta(self, x, is_free):
        x = _hip.as_f64(x).ravel()
        return self.ctx.constrain(x) if is_free else x

    # ---- functor protocol ------------------------------------------------------------------
    def __call__(self):
        return self.value(np.asarray(self.par.get_free(), dtype=np.float64), True)

    def value(self, x, is_free):
        S, W = self._stats()
        return float(self._terms(self._eta(x, is_free), S, W)[0])

    def grad(self, x, is_free):
        S, W = self._stats()
        g = self._terms(self._eta(x, is_free), S, W)[1]
        if not is_free:
            return g
        return self.ctx.free_to_vector_jac(x).T @ g

    jacobian = grad

    def hessian(self, x, is_free):
        S, W = self._stats()
        _, g, H = self._terms(self._eta(x, is_free), S, W)
        if not is_free:
            return H
        return self.ctx.free_hessian_from_vector(x, g, H)      # J^T H J + sum_k g_k d2 eta_k, on device

    def hvp(self, x, v, is_free):
        return self.hessian(x, is_free) @ _hip.as_f64(v).ravel()

    def hyper_kind(self, hyper_par):
        if hyper_par is self.weights_par:
            return 'weights'
        raise NotImplementedError('the second parameter must be this objective\'s `weights_par`')

    def cross_hessian(self, hyper_par, val1, val1_is_free):
        """d2 f / d par1 d w^T: (n1, N).  Rows of G come from the device, the chain rule through
        the packing Jacobian is one small product."""
        self.hyper_kind(hyper_par)
        self._push_state()
        M, c = self._obs_terms(self._eta(val1, val1_is_free))
        G = self.ctx.obs_quadform(M, c)                          # N x V
        if val1_is_free:
            G = G @ self.ctx.free_to_vector_jac(val1)            # N x D
        return np.ascontiguousarray(G.T)


class NormalRegressionObjective(QuadraticDataObjective):
    """Example.ipynb:247-274.  `par` must contain an array parameter `beta_name` of shape
    (dx, dy) and a positive-definite matrix parameter `lambda_name` of size dy, in that order or
    any other (offsets come from the dictionary)."""

    def __init__(self, par, x, y, beta_name='beta', lambda_name='lambda', weights=None, device=0):
        x, y = _hip.as_f64(x), _hip.as_f64(y)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError('x must be (N, dx) and y (N, dy)')
        self.dx, self.dy = x.shape[1], y.shape[1]
        self._bs = par.vector_indices_dict[beta_name]
        self._ls = par.vector_indices_dict[lambda_name]
        if len(self._bs) != self.dx * self.dy:
            raise ValueError('Wrong size for {}.  Expected {}, got {}'.format(beta_name, self.dx * self.dy, len(self._bs)))
        if len(self._ls) != self.dy * (self.dy + 1) // 2:
            raise ValueError('Wrong size for {}'.format(lambda_name))
        self._dup = duplication_matrix(self.dy)
        super().__init__(par, np.hstack([x, y]), weights=weights, device=device)

    def _unpack(self, eta):
        beta = eta[self._bs.start:self._bs.stop].reshape(self.dx, self.dy)
        lam = (self._dup @ eta[self._ls.start:self._ls.stop]).reshape(self.dy, self.dy)
        return beta, lam

    def _terms(self, eta, S, W):
        dx, dy = self.dx, self.dy
        beta, lam = self._unpack(eta)
        Sxx, Sxy, Syy = S[:dx, :dx], S[:dx, dx:], S[dx:, dx:]
        E = Sxx @ beta - Sxy                                   # dx x dy
        Mres = Syy - beta.T @ Sxy - Sxy.T @ beta + beta.T @ Sxx @ beta     # sum_n w_n r_n r_n^T
        sign, logdet = np.linalg.slogdet(lam)
        if sign <= 0:
            raise ValueError('Matrix is not positive definite')
        P = np.linalg.inv(lam)
        value = 0.5 * np.sum(lam * Mres) - 0.5 * W * logdet
        V = eta.size
        g = np.zeros(V)
        g[self._bs.start:self._bs.stop] = (E @ lam).ravel()
        g[self._ls.start:self._ls.stop] = self._dup.T @ (0.5 * Mres - 0.5 * W * P).ravel()
        H = np.zeros((V, V))
        bs, ls = slice(self._bs.start, self._bs.stop), slice(self._ls.start, self._ls.stop)
        H[bs, bs] = np.kron(Sxx, lam)
        Hbl = np.kron(E, np.eye(dy)) @ self._dup
        H[bs, ls] = Hbl
        H[ls, bs] = Hbl.T
        H[ls, ls] = self._dup.T @ (0.5 * W * np.kron(P, P)) @ self._dup
        return value, g, H

    def _obs_terms(self, eta):
        dx, dy, q = self.dx, self.dy, self.q
        beta, lam = self._unpack(eta)
        B = np.hstack([-beta.T, np.eye(dy)])                  # r_n = B z_n
        P = np.linalg.inv(lam)
        V = eta.size
        M = np.zeros((V, q, q))
        c = np.zeros(V)
        LB = lam @ B
        for a in range(dx):
            for b in range(dy):
                dB = np.zeros((dy, q))
                dB[b, a] = -1.0
                t = dB.T @ LB
                M[self._bs.start + a * dy + b] = t + t.T
        for i in range(dy):
            for j in range(i + 1):
                Eij = np.zeros((dy, dy))
                Eij[i, j] = 1.0
                Eij[j, i] = 1.0
                k = self._ls.start + j + i * (i + 1) // 2
                M[k] = B.T @ Eij @ B
                c[k] = -0.5 * (P[i, j] if i == j else 2.0 * P[i, j])
        return M, c
