"""Objectives written as closures in torch operations (extension; host plumbing around the HIP path).

The reference differentiates ANY closure of the parameter object with autograd (LRVB/SparseObjectives.py:95-116).  This
package's objectives are declared models with derivatives in closed form on the device; an opaque zero-argument closure
gets Richardson differences up to `objectives.NUMERIC_FALLBACK_MAX_D` parameters and is refused above.  `TorchObjective`
is the route for everything else that can be written down: a closure of the VECTOR-coordinate tensor in torch operations.

    fun = TorchObjective(par, lambda eta: 0.5 * eta @ (A @ eta) + torch.logsumexp(B @ eta, 0))
    objective = Objective(par, fun)           # fun_free_grad / fun_free_hessian / fun_free_hvp / fun_vector_* as usual

    # with a hyper-parameter (any parameter object; the closure takes its vector form as second argument):
    fun = TorchObjective(par, lambda eta, eps: loss(eta) + 0.5 * (eta - eps) @ (A @ (eta - eps)), hyper_par=prior_mean)
    sens = ParametricSensitivityLinearApproximation(fun, par, prior_mean, theta_hat, prior_mean.get_vector())

What runs where: torch.func (in place of autograd) forms the value, gradient, Hessian and Jacobian in VECTOR coordinates on
the context's GPU; the conversion to free coordinates -- J^T g, J^T H J + sum_k g_k d2 eta_k of `convert_vector_to_free_hessian`
(LRVB/Parameters.py:397-424) -- is the library's (`lrvb_jac_t_matmul`, `lrvb_free_hessian_from_vector`), and the result lives in
a device context like any declared objective's, so `ParametricSensitivityLinearApproximation`, the Cholesky and the CG solver
take it unchanged.  Nothing here is on the benchmarked path.
"""
import numpy as np

from . import _hip
from .models import DeviceContext


class TorchObjective(object):
    """`fun(eta, *args, **kwargs)`: eta a float64 torch tensor of the parameter's vector coordinates (on `cuda:device`),
    the return value a scalar tensor (or, for `jacobian`, a tensor of any shape).  Extra positional / keyword arguments
    of the `Objective` methods are passed through as the reference passes them to its closure."""

    _lrvb_device_functor = True

    def __init__(self, par, fun, device=0, hyper_par=None):
        import torch
        self._torch = torch
        self.par, self.fun = par, fun
        # one hyper-parameter (LRVB/SparseObjectives.py:321-449 takes any second parameter object): the closure is then
        # fun(eta, eps, ...) with eps the VECTOR form of hyper_par, read from the object at every evaluation
        self.hyper_par = hyper_par
        self.hyper_pars = {} if hyper_par is None else {getattr(hyper_par, 'name', 'hyper'): hyper_par}
        self._dev = torch.device('cuda', int(device))
        # a context for the packing maps and the solves: the layout of `par`, a zero quadratic term, no data
        self.ctx = DeviceContext(par.layout_blocks(), loss=None, quad_kind=_hip.QUAD_DIAG, device=int(device))
        self.ctx.set_data(_hip.SLOT_QUAD_A, np.zeros(self.ctx.V))
        self._memo = (None, None)                     # (key, vector-coordinate (g, H)) of the last Hessian: serves hvp at the same point

    # ---- helpers ---------------------------------------------------------------------------------------------------------
    def _eta(self, x, is_free):
        x = np.asarray(x, dtype=np.float64).ravel()
        return self.ctx.constrain(x) if is_free else x

    def _eps(self):
        return self._tensor(np.asarray(self.hyper_par.get_vector(), dtype=np.float64).ravel())

    def _closure(self, argv, argk):
        if self.hyper_par is None:
            return lambda e: self.fun(e, *argv, **argk)
        eps = self._eps()
        return lambda e: self.fun(e, eps, *argv, **argk)

    # ---- the hyper-parameter protocol of TwoParameterObjective / the sensitivity classes -----------------------------------------
    def hyper_kind(self, par):
        if self.hyper_par is None or par is not self.hyper_par:
            raise NotImplementedError('this objective was declared with another (or no) hyper-parameter')
        return 'torch'

    def hyper_grad(self, par, x, is_free, *argv, **argk):
        """d f / d vec(hyper_par) at the point x of the input parameter."""
        self.hyper_kind(par)
        self._set(x, is_free)
        eta = self._tensor(self._eta(x, is_free))
        return self._torch.func.grad(lambda eps: self.fun(eta, eps, *argv, **argk))(self._eps()).cpu().numpy()

    def cross_hessian(self, par, x, is_free, *argv, **argk):
        """d2 f / d x d vec(hyper_par)^T: forward-over-reverse in vector coordinates, J^T applied by the library for a free x."""
        self.hyper_kind(par)
        self._set(x, is_free)
        torch = self._torch
        eta = self._tensor(self._eta(x, is_free))
        g_of_eps = lambda eps: torch.func.grad(lambda e: self.fun(e, eps, *argv, **argk))(eta)
        C = torch.func.jacfwd(g_of_eps)(self._eps()).cpu().numpy()                     # (V, V_hyper)
        return self.ctx.jac_t_matmul(np.asarray(x, dtype=np.float64).ravel(), C) if is_free else C

    def _tensor(self, eta):
        return self._torch.tensor(eta, dtype=self._torch.float64, device=self._dev)

    def _set(self, x, is_free):
        (self.par.set_free if is_free else self.par.set_vector)(np.asarray(x, dtype=np.float64).ravel())

    # ---- the functor protocol of objectives.Objective -----------------------------------------------------------------------
    def value(self, x, is_free, *argv, **argk):
        self._set(x, is_free)
        out = self._closure(argv, argk)(self._tensor(self._eta(x, is_free)))
        return float(out) if out.ndim == 0 else out.cpu().numpy()

    def __call__(self, *argv, **argk):
        return self.value(self.par.get_vector(), False, *argv, **argk)

    def _grad_vec(self, eta, argv, argk):
        return self._torch.func.grad(self._closure(argv, argk))(self._tensor(eta)).cpu().numpy()

    def grad(self, x, is_free, *argv, **argk):
        self._set(x, is_free)
        g = self._grad_vec(self._eta(x, is_free), argv, argk)
        if not is_free:
            return g
        return self.ctx.jac_t_matmul(x, g[:, None])[:, 0]

    def jacobian(self, x, is_free, *argv, **argk):
        """d fun / d x for a tensor-valued closure: shape fun(...).shape + (len(x),) (autograd.jacobian's convention)."""
        self._set(x, is_free)
        eta = self._tensor(self._eta(x, is_free))
        Jv = self._torch.func.jacrev(self._closure(argv, argk))(eta).cpu().numpy()          # ans.shape + (V,)
        if not is_free:
            return Jv
        shape = Jv.shape[:-1]
        out = self.ctx.jac_t_matmul(x, np.ascontiguousarray(Jv.reshape(-1, Jv.shape[-1]).T))     # (D, ans.size)
        return out.T.reshape(shape + (out.shape[0],))

    def _vec_derivs(self, x, is_free, argv, argk):
        eta = self._eta(x, is_free)
        key = (eta.tobytes(), repr(argv), repr(sorted(argk.items())),
               None if self.hyper_par is None else np.asarray(self.hyper_par.get_vector(), dtype=np.float64).tobytes())
        if self._memo[0] != key:
            f, t = self._closure(argv, argk), self._tensor(eta)
            g = self._torch.func.grad(f)(t).cpu().numpy()
            H = self._torch.func.hessian(f)(t).cpu().numpy()
            self._memo = (key, (g, 0.5 * (H + H.T)))
        return self._memo[1]

    def hessian(self, x, is_free, *argv, **argk):
        self._set(x, is_free)
        g, H = self._vec_derivs(x, is_free, argv, argk)
        if not is_free:
            return H.copy()
        return self.ctx.free_hessian_from_vector(np.asarray(x, dtype=np.float64).ravel(), g, H)

    def hvp(self, x, v, is_free, *argv, **argk):
        """H v.  Vector coordinates: one forward-over-reverse pass of torch.func.  Free coordinates: through the free
        Hessian of the point (formed once per point, see `hessian`) -- the chain rule needs J v and the second-order term
        of the packing map, which the library applies to whole matrices."""
        v = np.asarray(v, dtype=np.float64).ravel()
        if not is_free:
            self._set(x, False)
            f, t = self._closure(argv, argk), self._tensor(self._eta(x, False))
            return self._torch.func.jvp(self._torch.func.grad(f), (t,), (self._tensor(v),))[1].cpu().numpy()
        return self.hessian(x, True, *argv, **argk) @ v
