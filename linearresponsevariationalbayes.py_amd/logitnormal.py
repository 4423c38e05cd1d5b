"""Logistic regression with a mean-field Gaussian variational posterior: the model the non-conjugate logistic term of
LRVB/Modeling.py:16-52 is written for.

    q(beta_j) = N(mean_j, 1 / info_j)          -- a `UVNParamVector` (LRVB/NormalParams.py:51-76)
    z_n = x_n . beta ~ N(x_n . mean, x_n^2 . (1 / info))
    KL(theta) = sum_n w_n ( E log(1 + e^{z_n}) - y_n x_n . mean )                 (Gauss-Hermite, Modeling.py:36-52)
              + 1/2 prior_info sum_j (mean_j^2 + 1 / info_j)                      (beta_j ~ N(0, 1 / prior_info))
              + 1/2 sum_j log info_j                                              (minus the entropy, up to a constant)

The reference would hand this expression to autograd; here the O(N) work runs on the GPU in the coordinates (mean, var):
two row products, one element-wise quadrature kernel, and for the Hessian three weighted MFMA products X^T D X,
X^T D (X o X), (X o X)^T D (X o X) (`lrvb_logitnormal_terms`).  The chain var = 1 / info, the N-independent terms and the
free-coordinate conversion (device) follow.  Same functor protocol as the other model classes, so `Objective`,
`ParametricSensitivityLinearApproximation` (hyper-parameter = the observation weights) and the optimiser wrappers apply.
"""
import numpy as np

from . import _hip
from .models import DeviceContext
from .packing import VectorParam, HyperVectorParam, ResidentVector


class LogitNormalRegressionObjective(object):
    _lrvb_device_functor = True

    def __init__(self, par, x, y, beta_name='beta', prior_info=1.0, gh_deg=20, weights=None, device=0):
        self.par = par
        x = _hip.as_f64(x)
        self.n_obs, self.P = x.shape
        beta = par[beta_name]
        if beta['mean'].free_size() != self.P or beta['info'].free_size() != self.P:
            raise ValueError('`{}` must be a UVNParamVector of length {}'.format(beta_name, self.P))
        if par.vector_size() != 2 * self.P:
            raise ValueError('the parameter must hold the UVNParamVector and nothing else')
        self.prior_info = float(prior_info)
        self.gh_x, self.gh_w = np.polynomial.hermite.hermgauss(int(gh_deg))
        self.ctx = DeviceContext(par.layout_blocks(), loss='logistic', n_obs=self.n_obs, n_cols=self.P, device=device)
        if self.ctx.D != par.free_size() or self.ctx.V != par.vector_size():
            raise ValueError('layout_blocks() of the parameter disagrees with its free/vector sizes')
        self.ctx.set_data(_hip.SLOT_X, x)
        self._y = _hip.as_f64(y).ravel().copy()
        self.ctx.set_data(_hip.SLOT_Y, self._y)
        w0 = np.ones(self.n_obs) if weights is None else _hip.as_f64(weights).ravel().copy()
        self.weights_par = HyperVectorParam('weights', self.n_obs, val=w0)
        self.tilt_par = None
        self._w_res = ResidentVector()
        self._x = x

    def _push_state(self):
        w = self._w_res.changed(self.weights_par)              # O(1) for the objective's own HyperVectorParam
        if w is not None:
            self.ctx.set_weights(w)
            self._h_key = None

    def _eta(self, x, is_free):
        x = _hip.as_f64(x).ravel()
        return self.ctx.constrain(x) if is_free else x

    # ---- observations sharded over GPUs: the data term is a sum over rows --------------------------------------
    def local_stats(self, eta):
        """[value | gradient (2 P) | H_mm, H_mv, H_vv (3 P^2)] of THIS process's rows in the coordinates (mean, var) at the
        vector-coordinate point eta = [mean | info]: the buffer of the one sum all-reduce per evaluation (SURVEY.md section
        8(e)); prior, entropy, the chain var = 1 / info and the free conversion are replicated afterwards."""
        eta = _hip.as_f64(eta).ravel()
        self._push_state()
        val, g, Hb = self.ctx.logitnormal_terms(eta[:self.P], 1.0 / eta[self.P:], self.gh_x, self.gh_w)
        return np.concatenate([[val], g, Hb[0].ravel(), Hb[1].ravel(), Hb[2].ravel()])

    def set_reduced_stats(self, flat, eta=None):
        """Install statistics summed over all shards for the point eta (None = use this process's own rows again)."""
        if flat is None:
            self._external = None
            return
        flat = np.asarray(flat, dtype=np.float64).ravel()
        P = self.P
        if flat.size != 1 + 2 * P + 3 * P * P or eta is None:
            raise ValueError('expected {} statistics and the point they were formed at'.format(1 + 2 * P + 3 * P * P))
        self._external = (np.asarray(eta, dtype=np.float64).copy(), flat.copy())
        self._h_key = None

    def _data_terms(self, eta, want_grad, want_hess):
        P = self.P
        ext = getattr(self, '_external', None)
        if ext is not None:
            if not np.array_equal(ext[0], eta):
                raise ValueError('the installed statistics were formed at another point')
            f = ext[1]
            Hb = f[1 + 2 * P:].reshape(3, P, P)
            return float(f[0]), f[1:1 + 2 * P], (Hb[0], Hb[1], Hb[2])
        self._push_state()
        return self.ctx.logitnormal_terms(eta[:P], 1.0 / eta[P:], self.gh_x, self.gh_w, want_grad=want_grad, want_hess=want_hess)

    # ---- vector coordinates (mean, info) -------------------------------------------------------------------
    def _terms(self, eta, want_grad=True, want_hess=True):
        P = self.P
        mean, info = eta[:P], eta[P:]
        var = 1.0 / info
        val, g_mv, Hb = self._data_terms(eta, want_grad or want_hess, want_hess)
        tau = self.prior_info
        val += 0.5 * tau * (np.sum(mean ** 2) + np.sum(var)) + 0.5 * np.sum(np.log(info))
        if not (want_grad or want_hess):
            return val, None, None
        d1 = -var * var                                      # d var / d info
        g_var = g_mv[P:] + 0.5 * tau
        g = np.concatenate([g_mv[:P] + tau * mean, g_var * d1 + 0.5 / info])
        if not want_hess:
            return val, g, None
        H = np.empty((2 * P, 2 * P))
        H[:P, :P] = Hb[0] + tau * np.eye(P)
        H[:P, P:] = Hb[1] * d1[None, :]
        H[P:, :P] = H[:P, P:].T
        H[P:, P:] = Hb[2] * d1[:, None] * d1[None, :] + np.diag(g_var * 2.0 * var ** 3 - 0.5 / info ** 2)
        return val, g, H

    # ---- functor protocol ------------------------------------------------------------------------------------
    def __call__(self):
        return self.value(np.asarray(self.par.get_free(), dtype=np.float64), True)

    def value(self, x, is_free=True):
        return float(self._terms(self._eta(x, is_free), False, False)[0])

    def grad(self, x, is_free=True):
        g = self._terms(self._eta(x, is_free), True, False)[1]
        return self.ctx.free_to_vector_jac(x).T @ g if is_free else g

    jacobian = grad

    def hessian(self, x, is_free=True):
        _, g, H = self._terms(self._eta(x, is_free))
        return self.ctx.free_hessian_from_vector(x, g, H) if is_free else H

    def _hessian_cached(self, x, is_free):
        self._push_state()
        key = (bool(is_free), np.asarray(x, dtype=np.float64).tobytes(), self._w_res.key)
        if getattr(self, '_h_key', None) != key:
            self._h_val = self.hessian(x, is_free)
            self._h_key = key
        return self._h_val

    def hvp(self, x, v, is_free=True):
        return self._hessian_cached(x, is_free) @ _hip.as_f64(v).ravel()

    def cg_solve(self, free_val, b, x0=None, Minv=None, tol=1e-8, maxiter=0):
        H = self._hessian_cached(free_val, True)
        return self.ctx.cg_solve_matrix(H, b, x0=x0, Minv=Minv, tol=tol, maxiter=maxiter)

    # ---- weight sensitivity ------------------------------------------------------------------------------------
    def hyper_kind(self, hyper_par):
        if hyper_par is self.weights_par:
            return 'weights'
        raise NotImplementedError('the second parameter must be this objective\'s `weights_par`')

    def cross_hessian(self, hyper_par, val1, val1_is_free):
        """d2 f / d par1 d w^T (n1 x N): row n of the per-observation gradient matrix is
        [psi_mu x_n | psi_var x_n^2] chained to (mean, info) and, if asked, to free coordinates."""
        self.hyper_kind(hyper_par)
        eta = self._eta(val1, val1_is_free)
        P = self.P
        mean, info = eta[:P], eta[P:]
        var = 1.0 / info
        x = self._x
        mu, v = x @ mean, (x * x) @ var
        _, d1 = self.ctx.gh_logistic(mu, np.sqrt(v), self.gh_x, self.gh_w, order=1)
        y = self._y
        G = np.hstack([(d1[:, 0] - y)[:, None] * x, (d1[:, 1] * 0.5 / np.sqrt(v))[:, None] * (x * x) * (-var * var)[None, :]])
        if val1_is_free:
            G = G @ self.ctx.free_to_vector_jac(val1)
        return np.ascontiguousarray(G.T)

