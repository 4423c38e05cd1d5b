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

    def __init__(self, par, model, weights_par=None, tilt_par=None, scale_fun=None):
        self.par = par
        self.model = model
        self.weights_par = weights_par
        self.tilt_par = tilt_par
        self.scale_fun = scale_fun
        self.ctx = OracleCtx(self)

    def _sync(self, argv=(), argk=None):
        if self.weights_par is not None:
            self.model.w = np.asarray(self.weights_par.get_vector(), dtype=np.float64)
        if self.tilt_par is not None:
            self.model.quad_b = np.asarray(self.tilt_par.get_vector(), dtype=np.float64)
        if self.scale_fun is not None:
            self.model.quad_scale = self.scale_fun(*argv, **(argk or {}))

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
        if hyper_par is self.tilt_par:
            return 'tilt'
        raise NotImplementedError('unknown hyper-parameter')

    def hyper_grad(self, hyper_par, val1, val1_is_free, *argv, **argk):
        self._sync(argv, argk)
        m = self.model
        eta = m.layout.constrain(val1) if val1_is_free else np.asarray(val1, dtype=np.float64)
        if hyper_par is self.weights_par:
            z = m.x @ eta[m.glm_off:m.glm_off + m.P]
            return om.loss_terms(m.loss, m.y, z, m.lik_info)[0]
        if hyper_par is self.tilt_par:
            return m.quad_scale * eta
        raise NotImplementedError('unknown hyper-parameter')

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
        if hyper_par is self.tilt_par:
            if val1_is_free:
                return self.model.cross_hessian_tilt(val1)
            return self.model.quad_scale * np.eye(self.model.layout.V)
        raise NotImplementedError('unknown hyper-parameter')
