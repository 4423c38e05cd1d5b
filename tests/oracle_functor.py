"""TEST DOUBLE: a functor with the product's device-functor protocol whose arithmetic is the CPU
oracle.  It exists so that the host logic of the package (Objective plumbing, preconditioned
family, TwoParameterObjective, sensitivity classes, optimiser wrappers, the sharded build) can be
exercised by `-m "not gpu"` tests in a container without a GPU.  It lives under tests/ and is never
imported by the product."""
import numpy as np
import scipy.linalg

from oracle import models as om
from oracle import solvers as osv


class OracleCtx(object):
    """Stands in for DeviceContext in the solve calls of sensitivity.py / cg.py."""

    def __init__(self, functor):
        self.f = functor
        self.D = functor.model.layout.D
        self.V = functor.model.layout.V
        self._chol = None
        self.chol_token = 0

    def chol_factor(self, H):
        self._chol = scipy.linalg.cho_factor(np.asarray(H))
        self.chol_token += 1

    def chol_solve(self, B):
        return scipy.linalg.cho_solve(self._chol, np.asarray(B))

    def lrvb_cov(self, M):
        return np.asarray(M) @ scipy.linalg.cho_solve(self._chol, np.asarray(M).T)

    def dk_grad_vec(self, eta, U=None, w_override=None, include_quad=True):
        return self.f.model.dk_grad_vec(eta, U, w_override, include_quad)

    def cg_solve(self, free, b, x0=None, Minv=None, tol=1e-8, maxiter=0):
        self.f._sync()
        x, info, iters = osv.cg_solve(lambda v: self.f.model.hvp(free, v), b, x0=x0, tol=tol,
                                      maxiter=None if maxiter <= 0 else maxiter, Minv=Minv)
        return x, info, iters


class OracleFunctor(object):
    _lrvb_device_functor = True

    OTHER_HYPERS = ('tilt', 'prior_mean', 'prior_info', 'quad_scale', 'lik_info')

    def __init__(self, par, model, weights_par=None, tilt_par=None, scale_fun=None, **other_hyper_pars):
        """other_hyper_pars: prior_mean_par=..., prior_info_par=..., quad_scale_par=..., lik_info_par=... (parameter objects
        whose vector value is pushed into the oracle model before every evaluation, like the device functor does)."""
        self.par = par
        self.model = model
        self.weights_par = weights_par
        self.tilt_par = tilt_par
        for name in self.OTHER_HYPERS[1:]:
            setattr(self, name + '_par', other_hyper_pars.pop(name + '_par', None))
        assert not other_hyper_pars, 'unknown arguments {}'.format(sorted(other_hyper_pars))
        self.scale_fun = scale_fun
        self.ctx = OracleCtx(self)

    def _sync(self, argv=(), argk=None):
        if self.weights_par is not None:
            self.model.w = np.asarray(self.weights_par.get_vector(), dtype=np.float64)
        for name in self.OTHER_HYPERS:
            p = getattr(self, name + '_par')
            if p is not None and not (name == 'quad_scale' and self.scale_fun is not None):
                self.model.set_hyper(name, p.get_vector())
        if self.scale_fun is not None:
            s = self.scale_fun(*argv, **(argk or {}))
            if self.quad_scale_par is not None:
                s = s * float(np.ravel(self.quad_scale_par.get_vector())[0])
            self.model.quad_scale = s

    _push_state = _sync

    def __call__(self, *argv, **argk):
        self._sync(argv, argk)
        return self.model.value(np.asarray(self.par.get_free(), dtype=np.float64))

    def value(self, x, is_free, *argv, **argk):
        self._sync(argv, argk)
        return self.model.value(x) if is_free else self.model.value_vec(x)

    def grad(self, x, is_free, *argv, **argk):
        self._sync(argv, argk)
        return self.model.grad(x) if is_free else self.model.grad_vec(x)

    jacobian = grad

    def hessian(self, x, is_free, *argv, **argk):
        self._sync(argv, argk)
        return self.model.hessian(x) if is_free else self.model.hessian_vec(x)

    def hvp(self, x, v, is_free, *argv, **argk):
        self._sync(argv, argk)
        return self.model.hvp(x, v) if is_free else self.model.hvp_vec(x, v)

    def hyper_kind(self, hyper_par):
        if hyper_par is self.weights_par:
            return 'weights'
        for name in self.OTHER_HYPERS:
            if hyper_par is getattr(self, name + '_par'):
                return name
        raise NotImplementedError('unknown hyper-parameter')

    def hyper_direction_vec(self, hyper_par, eta, U, eps_dir):
        """D_eta^r [d g_eta / d eps [eps_dir]] [rows of U]: by central differences in eps of the oracle's `dk_grad_vec`
        (exact up to rounding: the gradient is linear in every hyper-parameter's vector form) -- deliberately NOT the
        closed forms the device functor uses."""
        self._sync()
        kind, m = self.hyper_kind(hyper_par), self.model
        eps_dir = np.asarray(eps_dir, dtype=np.float64).ravel()
        if kind == 'weights':
            return m.dk_grad_vec(eta, U, eps_dir, False)
        h0 = m.hyper_value(kind)
        try:
            m.set_hyper(kind, h0 + eps_dir)
            gp = m.dk_grad_vec(eta, U, None, True)
            m.set_hyper(kind, h0 - eps_dir)
            gm = m.dk_grad_vec(eta, U, None, True)
        finally:
            m.set_hyper(kind, h0)
        return 0.5 * (gp - gm)

    def hyper_grad(self, hyper_par, val1, val1_is_free, *argv, **argk):
        self._sync(argv, argk)
        m = self.model
        eta = m.layout.constrain(val1) if val1_is_free else np.asarray(val1, dtype=np.float64)
        if hyper_par is self.weights_par:
            z = m.x @ eta[m.glm_off:m.glm_off + m.P]
            return om.loss_terms(m.loss, m.y, z, m.lik_info)[0]
        return m.hyper_grad_vec(self.hyper_kind(hyper_par), eta)

    def cross_hessian(self, hyper_par, val1, val1_is_free, *argv, **argk):
        self._sync(argv, argk)
        if hyper_par is self.weights_par:
            if val1_is_free:
                return self.model.obs_grad(val1).T
            eta = np.asarray(val1, dtype=np.float64)
            z = self.model.x @ eta[self.model.glm_off:self.model.glm_off + self.model.P]
            l1 = om.loss_terms(self.model.loss, self.model.y, z, self.model.lik_info)[1]
            out = np.zeros((self.model.layout.V, self.model.N))
            out[self.model.glm_off:self.model.glm_off + self.model.P, :] = (l1[:, None] * self.model.x).T
            return out
        kind = self.hyper_kind(hyper_par)
        return self.model.cross_hessian_hyper(kind, val1) if val1_is_free else self.model.cross_hessian_hyper_vec(kind, val1)
