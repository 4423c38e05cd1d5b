"""Linear-response sensitivity of an optimum to hyper-parameters.

Drop-in for `ParametricSensitivityLinearApproximation` and the module-level `set_par` of
LRVB/ModelSensitivity.py:13-17, 555-612:

    d theta_hat / d eps^T = -H^-1  d2 f / d theta d eps^T        (doc/sensitivity.lyx:137-155)

The Hessian (Objective.fun_free_hessian), the cross Hessian
(TwoParameterObjective.fun_hessian_free1_vector2), the Cholesky factorisation and the solve all
run on the device; the factor stays resident in the objective's context.  `hyper_par` is any
hyper-parameter the declared objective lists in `hyper_pars` (observation weights, tilt, prior mean /
information / scale, likelihood information, the priors of the model families), in vector or free
coordinates; a plain closure of a few parameters is differentiated by the host fallback of
objectives.py and solved in a scratch device context.  The higher-order class lives in taylor.py.
"""
from copy import deepcopy

import numpy as np

from . import objectives as obj_lib


def set_par(par, val, is_free):
    if is_free:
        par.set_free(val)
    else:
        par.set_vector(val)


class DeviceCholesky(object):
    """Handle on a Cholesky factor held inside a device context (the counterpart of the
    `(c, lower)` tuple scipy.linalg.cho_factor returns at LRVB/ModelSensitivity.py:594)."""

    def __init__(self, ctx, hess):
        self.ctx = ctx
        self.dim = int(np.shape(hess)[0])
        self._hess = hess
        ctx.chol_factor(hess)
        self._token = ctx.chol_token

    def ensure_resident(self, ctx=None):
        """Re-factor if another factorisation has replaced this one in the context since."""
        ctx = self.ctx if ctx is None else ctx
        if ctx is not self.ctx or ctx.chol_token != self._token:
            ctx.chol_factor(self._hess)
            self.ctx, self._token = ctx, ctx.chol_token

    def solve(self, rhs):
        self.ensure_resident()
        return self.ctx.chol_solve(rhs)

    def lrvb_cov(self, moment_jac):
        self.ensure_resident()
        return self.ctx.lrvb_cov(moment_jac)


_scratch = {}


def _solve_context(functor):
    """The device context a factorisation for `functor` lives in: the declared objective's own, or -- for a plain
    closure, whose derivatives came from the host fallback of objectives.py -- one scratch context per process (the
    Cholesky and the solves still run on the device: there is no host solver in this package)."""
    ctx = getattr(functor, 'ctx', None)
    if ctx is not None:
        return ctx
    if getattr(functor, '_lrvb_device_functor', False) or not callable(functor):
        raise NotImplementedError('the objective functor exposes no device context for the solve')
    if 'ctx' not in _scratch:
        from .models import scratch_context
        _scratch['ctx'] = scratch_context()
    return _scratch['ctx']


def _factor_and_solve(functor, hess, rhs):
    chol = DeviceCholesky(_solve_context(functor), hess)
    return chol, chol.solve(np.asarray(rhs, dtype=np.float64))


class ParametricSensitivityLinearApproximation(object):
    def __init__(self, objective_functor, input_par, hyper_par, input_val0, hyper_val0,
                 input_is_free=True, hyper_is_free=False, hess0=None, hyper_par_objective_functor=None,
                 stream_hyper=False):
        # stream_hyper=True (extension): do not form the D x P cross Hessian and the D x P sensitivity
        # at construction -- with hyper_par = N observation weights they are D x N -- but keep the
        # factor resident and serve `get_doutput_dhyper_rows` from the device, row block by row block
        self.stream_hyper = bool(stream_hyper)
        self.objective_functor = objective_functor
        self.input_par = input_par
        self.hyper_par = hyper_par
        self.input_is_free = input_is_free
        self.hyper_is_free = hyper_is_free
        if hyper_par_objective_functor is None:
            self.hyper_par_objective_functor = objective_functor
        else:
            self.hyper_par_objective_functor = hyper_par_objective_functor
        self.objective = obj_lib.Objective(self.input_par, self.objective_functor)
        self.joint_objective = obj_lib.TwoParameterObjective(
            self.input_par, self.hyper_par, self.hyper_par_objective_functor)
        self.set_base_values(input_val0, hyper_val0, hess0=hess0)

    def set_par_to_base_values(self):
        set_par(self.input_par, self.input_val0, self.input_is_free)
        set_par(self.hyper_par, self.hyper_val0, self.hyper_is_free)

    def set_base_values(self, input_val0, hyper_val0, hess0=None):
        self.input_val0 = deepcopy(input_val0)
        self.hyper_val0 = deepcopy(hyper_val0)
        self.set_par_to_base_values()
        if hess0 is None:
            if self.input_is_free:
                self.hess0 = self.objective.fun_free_hessian(self.input_val0)
            else:
                self.hess0 = self.objective.fun_vector_hessian(self.input_val0)
        else:
            self.hess0 = hess0
        if self.stream_hyper:
            self.hess0_chol = DeviceCholesky(_solve_context(self.objective_functor), self.hess0)
            self.hyper_par_cross_hessian0 = None
            self.hyper_par_sensitivity = None
            return
        self.hyper_par_cross_hessian0 = self.joint_objective._cross12(
            self.input_val0, self.hyper_val0, self.input_is_free, self.hyper_is_free)
        self.hess0_chol, solved = _factor_and_solve(
            self.objective_functor, self.hess0, self.hyper_par_cross_hessian0)
        self.hyper_par_sensitivity = -1 * solved

    def get_dinput_dhyper(self):
        if self.hyper_par_sensitivity is None:
            raise RuntimeError('constructed with stream_hyper=True: use get_doutput_dhyper_rows')
        return self.hyper_par_sensitivity

    def get_doutput_dhyper_rows(self, moment_jac, n0=0, n1=None):
        """(moment_jac @ get_dinput_dhyper())[:, n0:n1] TRANSPOSED, shape (n1 - n0, Q), for
        hyper_par = the observation weights (Example.ipynb:425-441): the sensitivity of Q moments
        to each observation's weight.  Streams over the observations on the device from the
        resident factor; nothing of size D x N is formed."""
        fun = self.hyper_par_objective_functor
        kind = fun.hyper_kind(self.hyper_par) if hasattr(fun, 'hyper_kind') else None
        if kind != 'weights' or self.hyper_is_free or not hasattr(fun, 'ctx'):
            raise NotImplementedError('row streaming is defined for hyper_par = observation weights of a declared objective')
        self.set_par_to_base_values()
        fun._push_state()
        self.hess0_chol.ensure_resident(fun.ctx)
        return fun.ctx.obs_influence(self.input_val0, moment_jac, n0=n0, n1=n1, is_free=self.input_is_free)

    def predict_input_par_from_hyperparameters(self, new_hyper_par_value):
        hyper_par_diff = new_hyper_par_value - self.hyper_val0
        return self.input_val0 + self.hyper_par_sensitivity @ hyper_par_diff

    def get_lrvb_cov(self, moment_jac):
        """LRVB covariance M H^-1 M^T of moments with Jacobian M (Example.ipynb:398-415), reusing
        the resident factor."""
        return self.hess0_chol.lrvb_cov(np.asarray(moment_jac, dtype=np.float64))


# ---- the names BASELINE.json's north_star uses for the path (they belong to the author's later
# code, not to this reference checkout; each is the reference object named beside it) ---------------------------------
HyperparameterSensitivityLinearApproximation = ParametricSensitivityLinearApproximation   # LRVB/ModelSensitivity.py:555-612


def get_kl_hessian(objective, free_val, *argv, **argk):
    """Dense Hessian of the KL / -ELBO objective in free coordinates: `Objective.fun_free_hessian`
    (LRVB/SparseObjectives.py:156-158) -- one device build."""
    return objective.fun_free_hessian(free_val, *argv, **argk)


def get_lrvb_cov(objective, free_val, moment_jac, kl_hessian=None):
    """Linear-response covariance M H^-1 M^T of the moments with free-coordinate Jacobian M at the optimum `free_val`
    (what Example.ipynb:398-415 computes inline with cho_factor / cho_solve): Hessian build (unless given), Cholesky
    and the solve all on the device; the factor stays resident in the objective's context."""
    hess = get_kl_hessian(objective, free_val) if kl_hessian is None else kl_hessian
    return DeviceCholesky(_solve_context(objective.fun), hess).lrvb_cov(np.asarray(moment_jac, dtype=np.float64))


# the k-th order class and its term algebra live in taylor.py; the reference keeps them in this module
from .taylor import (ParametricSensitivityTaylorExpansion, DerivativeTerm, get_taylor_base_terms,   # noqa: E402,F401
                     consolidate_terms, differentiate_terms)


# LRVB/ModelSensitivity.py holds the Taylor-expansion machinery too: same names under this module
from .taylor import (DerivativeTerm, ParametricSensitivityTaylorExpansion, append_jvp, generate_two_term_derivative_array,  # noqa: E402,F401
                     consolidate_terms, differentiate_terms, evaluate_terms, evaluate_dketa_depsk, get_taylor_base_terms)
