"""Objective wrappers: functions of the flat free / vector parameter with gradient, Hessian,
Jacobian and Hessian-vector product, evaluated on the device.

Drop-in for LRVB/SparseObjectives.py: same class names, method names, argument orders and side
effects --
  Objective                 :95-240   (fun_free, fun_vector, *_grad, *_hessian, *_jacobian, *_hvp,
                                      the preconditioned `_cond` family, get_conditioned_x,
                                      uncondition_x; attributes par, fun, preconditioner, logger)
  ParameterConverter        :245-308
  TwoParameterObjective     :321-449
  ParametricSensitivity     :487-573  (deprecated there too)
  Timer / Logger / safe_matmul / make_index_param / get_sparse_sub_matrix / CSR packing
-- but every derivative is one call into liblrvb_hip.so instead of D+1 autograd tape walks.
`fun` is a declared objective (models.DeviceObjective or another object exposing the same functor
protocol).  An opaque Python closure is still accepted, exactly as the reference calls it: the
value-only methods run it as is, and for up to NUMERIC_FALLBACK_MAX_D parameters its derivatives come
from Richardson-extrapolated differences on the host (`_NumericFunctor`: plumbing for the reference's
small closed-form models, not a compute path); above that a derivative request raises
NotImplementedError.

Side-effect contract kept from the reference (comment at :131-140): after every call `par` holds
the numeric value of the evaluation point.
"""
import time
import warnings
from copy import deepcopy

import numpy as np
import scipy as sp
from scipy import sparse

_NO_DERIV = ('derivatives of an opaque Python closure need a tracing AD engine on the host; this '
             'framework differentiates DECLARED objectives on the GPU -- build `fun` with '
             'models.DeviceObjective / GLMObjective / QuadraticObjective / LinearMoments')


NUMERIC_FALLBACK_MAX_D = 64


class _NumericFunctor(object):
    """Small-D stand-in for autograd on an OPAQUE zero-argument closure (`Objective(par, lambda: ...)`, as
    LRVB/SparseObjectives.py:96-116 accepts and BASELINE.json's "plumbing" config words it): gradient, Jacobian, Hessian
    and Hessian-vector product by Richardson-extrapolated central differences of the closure itself (steps h, h/2, h/4:
    error O(h^6) in the first, O(h^6) in the mixed second differences; exact up to rounding for the reference's quadratic
    test models).  Host plumbing for a handful of parameters -- refused above NUMERIC_FALLBACK_MAX_D, where the declared
    objectives on the device are the path."""

    def __init__(self, par, fun):
        self.par, self.fun = par, fun

    def _f(self, is_free, argv, argk):
        def f(x):
            set_par(self.par, x, is_free)
            return np.asarray(self.fun(*argv, **argk), dtype=np.float64)
        return f

    def _check(self, x):
        if np.size(x) > NUMERIC_FALLBACK_MAX_D:
            raise NotImplementedError('{} parameters: '.format(np.size(x)) + _NO_DERIV)
        return np.asarray(x, dtype=np.float64).ravel()

    def value(self, x, is_free, *argv, **argk):
        return self._f(is_free, argv, argk)(x)

    def jacobian(self, x, is_free, *argv, **argk):
        f, x = self._f(is_free, argv, argk), self._check(x)
        shape = np.shape(f(x))
        J = numeric_jacobian(lambda z: np.ravel(f(z)), x)       # (ans.size, n): flatten first, whatever ans.ndim is
        return J.reshape(shape + (x.size,))                     # autograd.jacobian: ans.shape + x.shape

    def grad(self, x, is_free, *argv, **argk):
        return self.jacobian(x, is_free, *argv, **argk).reshape(np.size(x))

    def hessian(self, x, is_free, *argv, **argk):
        x, f = self._check(x), self._f(is_free, argv, argk)
        n = x.size
        h0 = 1e-2 * np.maximum(1.0, np.abs(x))
        est = []
        for h in (h0, h0 / 2, h0 / 4):
            H = np.empty((n, n))
            for i in range(n):
                for j in range(i + 1):
                    ei, ej = np.zeros(n), np.zeros(n)
                    ei[i], ej[j] = h[i], h[j]
                    H[i, j] = H[j, i] = (f(x + ei + ej) - f(x + ei - ej) - f(x - ei + ej) + f(x - ei - ej)) / (4 * h[i] * h[j])
            est.append(H)
        r1 = [(4 * est[1] - est[0]) / 3, (4 * est[2] - est[1]) / 3]
        return (16 * r1[1] - r1[0]) / 15

    def hvp(self, x, v, is_free, *argv, **argk):
        x, v = self._check(x), np.asarray(v, dtype=np.float64).ravel()
        scale = max(np.max(np.abs(v)), 1e-300)
        g = lambda t: self.grad(x + t[0] * v / scale, is_free, *argv, **argk)
        return numeric_jacobian(g, np.zeros(1), rel_step=1e-2)[:, 0] * scale


class _NumericTwoParFunctor(object):
    """Small-D stand-in for autograd on an opaque closure of TWO parameters (`TwoParameterObjective(par1, par2, fun)` as
    LRVB/SparseObjectives.py:321-449 accepts; the reference's tests at LRVB/test_objectives.py:220-243, 294-381 and the
    `QuadraticModel` of LRVB/test_model_sensitivity.py:36-88 are written this way): gradients by Richardson-extrapolated
    central differences, the cross Hessian by Richardson-extrapolated mixed second differences, directly in the
    coordinates (free or vector) each argument is given in.  Refused above NUMERIC_FALLBACK_MAX_D entries per parameter."""

    def __init__(self, par1, par2, fun):
        self.par1, self.par2, self.fun = par1, par2, fun

    @staticmethod
    def _check(x):
        if np.size(x) > NUMERIC_FALLBACK_MAX_D:
            raise NotImplementedError('{} parameters: '.format(np.size(x)) + _NO_DERIV)
        return np.array(x, dtype=np.float64).ravel()

    def _f(self, free1, free2, argv, argk):
        def f(a, b):
            set_par(self.par1, a, free1)
            set_par(self.par2, b, free2)
            return float(self.fun(*argv, **argk))
        return f

    def grad1(self, val1, val2, free1, free2, *argv, **argk):
        a, b, f = self._check(val1), self._check(val2), self._f(free1, free2, argv, argk)
        return numeric_jacobian(lambda z: f(z, b), a).reshape(a.size)

    def grad2(self, val1, val2, free1, free2, *argv, **argk):
        a, b, f = self._check(val1), self._check(val2), self._f(free1, free2, argv, argk)
        return numeric_jacobian(lambda z: f(a, z), b).reshape(b.size)

    def cross12(self, val1, val2, free1, free2, *argv, **argk):
        a, b, f = self._check(val1), self._check(val2), self._f(free1, free2, argv, argk)
        ha0, hb0 = 1e-2 * np.maximum(1.0, np.abs(a)), 1e-2 * np.maximum(1.0, np.abs(b))
        est = []
        for div in (1.0, 2.0, 4.0):
            ha, hb = ha0 / div, hb0 / div
            C = np.empty((a.size, b.size))
            for i in range(a.size):
                ap, am = a.copy(), a.copy()
                ap[i] += ha[i]
                am[i] -= ha[i]
                for j in range(b.size):
                    bp, bm = b.copy(), b.copy()
                    bp[j] += hb[j]
                    bm[j] -= hb[j]
                    C[i, j] = (f(ap, bp) - f(ap, bm) - f(am, bp) + f(am, bm)) / (4.0 * ha[i] * hb[j])
            est.append(C)
        r1 = [(4 * est[1] - est[0]) / 3, (4 * est[2] - est[1]) / 3]
        return (16 * r1[1] - r1[0]) / 15


def _functor(fun, par=None):
    if getattr(fun, '_lrvb_device_functor', False):
        return fun
    if par is not None and callable(fun):
        return _NumericFunctor(par, fun)
    raise NotImplementedError(_NO_DERIV)


def safe_matmul(x, y):
    """x y for any mix of dense arrays and scipy sparse matrices (the reference's helper of the same name,
    LRVB/SparseObjectives.py:21-25)."""
    sparse_operand = sparse.issparse(x) or sparse.issparse(y)
    return x * y if sparse_operand else np.matmul(x, y)


def compress(x):
    """A dense, squeezed ndarray from a dense or sparse result (LRVB/SparseObjectives.py:27-31)."""
    return np.squeeze(np.asarray(x.todense() if sparse.issparse(x) else x))


class Timer(object):
    """Named wall-clock intervals: tic() starts the clock, toc(name) stores the time since in `time_dict[name]`
    (and prints it unless verbose=False).  Same surface as the reference's utility (LRVB/SparseObjectives.py:35-45)."""

    def __init__(self):
        self.time_dict = {}
        self.tic_time = None

    def tic(self):
        self.tic_time = time.perf_counter()

    def toc(self, time_name, verbose=True):
        if self.tic_time is None:
            raise RuntimeError('toc() without a preceding tic()')
        elapsed = time.perf_counter() - self.tic_time
        self.time_dict[time_name] = elapsed
        if verbose:
            print('{}: {} seconds'.format(time_name, elapsed))
        return elapsed

    def __str__(self):
        return str(self.time_dict)


class Logger(object):
    """Iteration log of an optimisation: `fun_free(..., verbose=True)` hands every evaluation to `log(value, x)`.
    Attribute names are the reference's (LRVB/SparseObjectives.py:48-87: iter, x, value, last_x, last_value, x_array,
    val_array, print_every, callback), because optimiser callbacks read them; the history lists are the state and the
    `last_*` / current fields are views of their tails."""

    def __init__(self, print_every=1):
        self.print_every = print_every
        self.print_x_diff = True
        self.callback = None
        self.initialize()

    def initialize(self):
        # plain attributes, assigned here and in log(), as in the reference (LRVB/SparseObjectives.py:53-60): subclasses
        # that override initialize() the reference's way, and callbacks that reset last_x, keep working
        self.iter = 0
        self.last_x = None
        self.x = None
        self.value = None
        self.last_value = None
        self.x_array = []
        self.val_array = []

    @property
    def x_diff(self):
        """Largest coordinate change between the two most recent points (inf before there are two)."""
        if len(self.x_array) < 2:
            return float('inf')
        return float(np.max(np.abs(np.asarray(self.x_array[-1]) - np.asarray(self.x_array[-2]))))

    def print_message(self):
        print('Iter ', self.iter, ' value: ', self.value)

    def log(self, value, x):
        self.value, self.x = value, x
        self.x_array.append(x)
        self.val_array.append(value)
        self.last_x, self.last_value = x, value          # before the callback runs, as in the reference (:77-78)
        # (the reference's `iter % print_every` raises ZeroDivisionError for print_every = 0; so does this)
        if self.iter % self.print_every == 0:
            (self.callback or type(self).print_message)(self)
        self.iter += 1


class Objective(object):
    def __init__(self, par, fun):
        self.par = par
        self.fun = fun
        self.preconditioner = None
        self.logger = Logger()

    # ---- values: plain plumbing, any callable -------------------------------------------
    def fun_free(self, free_val, *argv, verbose=False, **argk):
        self.par.set_free(free_val)
        if getattr(self.fun, '_lrvb_device_functor', False) and hasattr(self.fun, 'value'):
            # a declared objective is evaluated AT free_val (not at get_free() of the state just set: that
            # round trip through the bounds-checked unconstraining map would turn a nan probe of an
            # optimiser into a ValueError, where the reference's closure simply returns nan)
            val = self.fun.value(np.asarray(free_val, dtype=np.float64), True, *argv, **argk)
        else:
            val = self.fun(*argv, **argk)
        if verbose:
            self.logger.log(val, free_val)
        return val

    def fun_vector(self, vec_val, *argv, **argk):
        self.par.set_vector(vec_val)
        if getattr(self.fun, '_lrvb_device_functor', False) and hasattr(self.fun, 'value'):
            try:
                return self.fun.value(np.asarray(vec_val, dtype=np.float64), False, *argv, **argk)
            except NotImplementedError:
                pass                               # objectives that are evaluated in free coordinates only
        return self.fun(*argv, **argk)

    # ---- derivatives: one device call each, then restore `par` (reference :142-150) -------
    def _eval(self, method, val, is_free, *argv, **argk):
        result = getattr(_functor(self.fun, self.par), method)(np.asarray(val, dtype=np.float64), is_free, *argv, **argk)
        if is_free:
            self.par.set_free(val)
        else:
            self.par.set_vector(val)
        return result

    def cache_free_and_eval(self, autograd_fun, free_val, *argv, **argk):
        """Evaluate `autograd_fun(free_val, ...)` and leave `par` at free_val: the side-effect contract every derivative
        method of the reference goes through (LRVB/SparseObjectives.py:142-145)."""
        result = autograd_fun(free_val, *argv, **argk)
        self.par.set_free(free_val)
        return result

    def cache_vector_and_eval(self, autograd_fun, vec_val, *argv, **argk):
        """LRVB/SparseObjectives.py:147-150."""
        result = autograd_fun(vec_val, *argv, **argk)
        self.par.set_vector(vec_val)
        return result

    def fun_free_grad(self, free_val, *argv, **argk):
        return self._eval('grad', free_val, True, *argv, **argk)

    def fun_free_hessian(self, free_val, *argv, **argk):
        return self._eval('hessian', free_val, True, *argv, **argk)

    def fun_free_jacobian(self, free_val, *argv, **argk):
        return self._eval('jacobian', free_val, True, *argv, **argk)

    def fun_vector_grad(self, vec_val, *argv, **argk):
        return self._eval('grad', vec_val, False, *argv, **argk)

    def fun_vector_hessian(self, vec_val, *argv, **argk):
        return self._eval('hessian', vec_val, False, *argv, **argk)

    def fun_vector_jacobian(self, vec_val, *argv, **argk):
        return self._eval('jacobian', vec_val, False, *argv, **argk)

    # Argument order (theta, *extra, vec, **kw), as autograd's hessian_vector_product imposes on
    # the reference (comment at LRVB/SparseObjectives.py:176-182).
    def fun_free_hvp(self, *argv, **argk):
        args, vec = argv[:-1], argv[-1]
        result = _functor(self.fun, self.par).hvp(np.asarray(args[0], dtype=np.float64), vec, True, *args[1:], **argk)
        self.par.set_free(args[0])
        return result

    def fun_vector_hvp(self, *argv, **argk):
        args, vec = argv[:-1], argv[-1]
        result = _functor(self.fun, self.par).hvp(np.asarray(args[0], dtype=np.float64), vec, False, *args[1:], **argk)
        self.par.set_vector(args[0])
        return result

    # ---- preconditioned family: x = A y  (LRVB/SparseObjectives.py:202-240) ----------------
    def get_conditioned_x(self, free_val):
        return safe_matmul(self.preconditioner, free_val)

    def fun_free_cond(self, free_val, *argv, verbose=False, **argk):
        assert self.preconditioner is not None
        return self.fun_free(self.get_conditioned_x(free_val), *argv, verbose=verbose, **argk)

    def fun_free_grad_cond(self, free_val, *argv, **argk):
        assert self.preconditioner is not None
        grad = self.fun_free_grad(self.get_conditioned_x(free_val), *argv, **argk)
        return safe_matmul(self.preconditioner.T, grad)

    def fun_free_hessian_cond(self, free_val, *argv, **argk):
        assert self.preconditioner is not None
        hess = self.fun_free_hessian(self.get_conditioned_x(free_val), *argv, **argk)
        return safe_matmul(self.preconditioner.T, safe_matmul(hess, self.preconditioner))

    def fun_free_hvp_cond(self, *argv, **argk):
        assert self.preconditioner is not None
        args, vec = argv[1:-1], argv[-1]
        y = self.get_conditioned_x(argv[0])
        return safe_matmul(self.preconditioner.T,
                           self.fun_free_hvp(y, *args, safe_matmul(self.preconditioner, vec), **argk))

    def uncondition_x(self, cond_x):
        return safe_matmul(self.preconditioner, cond_x)


class ParameterConverter(object):
    """Jacobians of a map from `par_in` to `par_out` (LRVB/SparseObjectives.py:245-308).

    `converter` is a zero-argument callable that sets `par_out` from the current `par_in` (as in
    the reference) AND carries its own derivative: an attribute `vec_jacobian(vec_in)` returning
    d vec_out / d vec_in^T.  `LinearConverter` and `ElementwiseConverter` below are declared
    converters; the free-coordinate variants chain through the packing Jacobians."""

    def __init__(self, par_in, par_out, converter):
        self.par_in = par_in
        self.par_out = par_out
        self.converter = converter

    def _convert(self, val_in, in_is_free, out_is_free):
        set_par(self.par_in, val_in, in_is_free)
        self.converter()
        return self.par_out.get_free() if out_is_free else self.par_out.get_vector()

    def cache_free_and_eval(self, autograd_fun, free_val_in):
        """Evaluate `autograd_fun(free_val_in)`, then leave par_in at free_val_in and par_out where it was
        (LRVB/SparseObjectives.py:280-285)."""
        vec_val_out = self.par_out.get_vector()
        result = autograd_fun(free_val_in)
        self.par_in.set_free(free_val_in)
        self.par_out.set_vector(vec_val_out)
        return result

    def cache_vector_and_eval(self, autograd_fun, vec_val_in):
        """LRVB/SparseObjectives.py:287-292."""
        vec_val_out = self.par_out.get_vector()
        result = autograd_fun(vec_val_in)
        self.par_in.set_vector(vec_val_in)
        self.par_out.set_vector(vec_val_out)
        return result

    def converter_free_to_vec(self, free_par_in):
        return self._convert(free_par_in, True, False)

    def converter_free_to_free(self, free_par_in):
        return self._convert(free_par_in, True, True)

    def converter_vec_to_vec(self, vec_par_in):
        return self._convert(vec_par_in, False, False)

    def converter_vec_to_free(self, vec_par_in):
        return self._convert(vec_par_in, False, True)

    def _vec_jac(self, vec_in):
        """d vec_out / d vec_in^T: the converter's own `vec_jacobian` if it declares one; otherwise -- an opaque
        Python closure, which the reference hands to autograd -- Richardson-extrapolated central differences of the
        closure itself (converters are small N-independent maps on the host; error ~1e-10 relative)."""
        vec_in = np.asarray(vec_in, dtype=np.float64)
        if hasattr(self.converter, 'vec_jacobian'):
            return np.asarray(self.converter.vec_jacobian(vec_in))
        return numeric_jacobian(self.converter_vec_to_vec, vec_in)

    def _out_free_from_vec(self):
        # d free_out / d vec_out = (d vec_out / d free_out)^-1 at the current par_out
        Jout = np.asarray(self.par_out.free_to_vector_jac(self.par_out.get_free()).todense())
        return np.linalg.pinv(Jout) if Jout.shape[0] != Jout.shape[1] else np.linalg.inv(Jout)

    def _restore(self, fn, val_in, set_in):
        vec_out = self.par_out.get_vector()
        result = fn()
        set_in(val_in)
        self.par_out.set_vector(vec_out)
        return result

    def free_to_vec_jacobian(self, free_par_in):
        def fn():
            Jin = np.asarray(self.par_in.free_to_vector_jac(free_par_in).todense())
            self.par_in.set_free(free_par_in)
            return self._vec_jac(self.par_in.get_vector()) @ Jin
        return self._restore(fn, free_par_in, self.par_in.set_free)

    def free_to_free_jacobian(self, free_par_in):
        def fn():
            Jin = np.asarray(self.par_in.free_to_vector_jac(free_par_in).todense())
            self.converter_free_to_vec(free_par_in)
            return self._out_free_from_vec() @ self._vec_jac(self.par_in.get_vector()) @ Jin
        return self._restore(fn, free_par_in, self.par_in.set_free)

    def vec_to_vec_jacobian(self, vec_par_in):
        return self._restore(lambda: self._vec_jac(vec_par_in), vec_par_in, self.par_in.set_vector)

    def vec_to_free_jacobian(self, vec_par_in):
        def fn():
            self.converter_vec_to_vec(vec_par_in)
            return self._out_free_from_vec() @ self._vec_jac(vec_par_in)
        return self._restore(fn, vec_par_in, self.par_in.set_vector)


class LinearConverter(object):
    """par_out.vector = B par_in.vector."""

    def __init__(self, par_in, par_out, B):
        self.par_in, self.par_out, self.B = par_in, par_out, np.asarray(B, dtype=np.float64)

    def __call__(self):
        self.par_out.set_vector(self.B @ self.par_in.get_vector())

    def vec_jacobian(self, vec_in):
        return self.B


class ElementwiseConverter(object):
    """par_out.vector = g(par_in.vector) elementwise, with derivative dg (e.g. x -> x**2, the
    converter of LRVB/test_objectives.py:80-88)."""

    def __init__(self, par_in, par_out, g, dg):
        self.par_in, self.par_out, self.g, self.dg = par_in, par_out, g, dg

    def __call__(self):
        self.par_out.set_vector(self.g(self.par_in.get_vector()))

    def vec_jacobian(self, vec_in):
        return np.diag(self.dg(vec_in))


def numeric_jacobian(f, x, rel_step=1e-3):
    """Jacobian of a smooth vector function by central differences with two Richardson extrapolations (steps h, h/2,
    h/4: error O(h^6)); used for opaque host-side converters only."""
    x = np.asarray(x, dtype=np.float64)
    cols = []
    for k in range(x.size):
        h = rel_step * max(1.0, abs(x[k]))
        d = []
        for step in (h, h / 2, h / 4):
            e = np.zeros_like(x); e[k] = step
            d.append(np.atleast_1d(np.asarray(f(x + e), dtype=np.float64) - np.asarray(f(x - e), dtype=np.float64)) / (2 * step))
        r1 = [(4 * d[1] - d[0]) / 3, (4 * d[2] - d[1]) / 3]
        cols.append((16 * r1[1] - r1[0]) / 15)
    return np.stack(cols, axis=1)


def set_par(par, val, is_free):
    if is_free:
        par.set_free(val)
    else:
        par.set_vector(val)


class TwoParameterObjective(object):
    """Cross Hessians d2 f / d par1 d par2^T (LRVB/SparseObjectives.py:321-449).  With a declared objective `par2` is one
    of the hyper-parameters it declares (`hyper_pars`: observation weights, tilt, prior mean / information / scale,
    likelihood information, the priors of the model families) and the cross Hessian is a closed form on the device; with a
    plain closure of up to NUMERIC_FALLBACK_MAX_D entries per parameter it comes from Richardson-extrapolated mixed
    differences on the host (`_NumericTwoParFunctor`; plumbing for the reference's small test models)."""

    def __init__(self, par1, par2, fun):
        self.par1 = par1
        self.par2 = par2
        self.fun = fun

    def cache_and_eval(self, autograd_fun, val1, val2, val1_is_free, val2_is_free, *argv, **argk):
        """Evaluate `autograd_fun(val1, val2, val1_is_free, val2_is_free, ...)` and leave both parameters at the
        evaluation point (LRVB/SparseObjectives.py:341-351)."""
        result = autograd_fun(val1, val2, val1_is_free, val2_is_free, *argv, **argk)
        set_par(self.par1, val1, val1_is_free)
        set_par(self.par2, val2, val2_is_free)
        return result

    def eval_fun(self, val1, val2, val1_is_free, val2_is_free, *argv, **argk):
        set_par(self.par1, val1, val1_is_free)
        set_par(self.par2, val2, val2_is_free)
        return self.fun(*argv, **argk)

    def fun_free(self, free_val1, free_val2, *argv, **argk):
        return self.eval_fun(free_val1, free_val2, True, True, *argv, **argk)

    def fun_vector(self, vec_val1, vec_val2, *argv, **argk):
        return self.eval_fun(vec_val1, vec_val2, False, False, *argv, **argk)

    def _jac2(self, val2, val2_is_free):
        # d vec2 / d val2: identity in vector coordinates, the packing Jacobian in free ones
        if not val2_is_free:
            return None
        return self.par2.free_to_vector_jac(np.asarray(val2, dtype=np.float64)).tocsr()

    def _numeric(self):
        """The host fallback for a plain closure (None for a declared objective)."""
        if getattr(self.fun, '_lrvb_device_functor', False):
            return None
        if not callable(self.fun):
            raise NotImplementedError(_NO_DERIV)
        return _NumericTwoParFunctor(self.par1, self.par2, self.fun)

    def _restore(self, result, val1, val2, val1_is_free, val2_is_free):
        set_par(self.par1, val1, val1_is_free)
        set_par(self.par2, val2, val2_is_free)
        return result

    def _cross12(self, val1, val2, val1_is_free, val2_is_free, *argv, **argk):
        num = self._numeric()
        if num is not None:
            return self._restore(num.cross12(val1, val2, val1_is_free, val2_is_free, *argv, **argk),
                                 val1, val2, val1_is_free, val2_is_free)
        f = self.fun
        set_par(self.par2, val2, val2_is_free)
        C = f.cross_hessian(self.par2, np.asarray(val1, dtype=np.float64), val1_is_free, *argv, **argk)
        J2 = self._jac2(val2, val2_is_free)
        if J2 is not None:
            C = np.asarray(C @ J2)
        set_par(self.par1, val1, val1_is_free)
        set_par(self.par2, val2, val2_is_free)
        return C

    def fun_grad1(self, val1, val2, val1_is_free, val2_is_free, *argv, **argk):
        num = self._numeric()
        if num is not None:
            return self._restore(num.grad1(val1, val2, val1_is_free, val2_is_free, *argv, **argk),
                                 val1, val2, val1_is_free, val2_is_free)
        f = self.fun
        set_par(self.par2, val2, val2_is_free)
        g = f.grad(np.asarray(val1, dtype=np.float64), val1_is_free, *argv, **argk)
        set_par(self.par1, val1, val1_is_free)
        return g

    def fun_grad2(self, val1, val2, val1_is_free, val2_is_free, *argv, **argk):
        """d f / d par2 (LRVB/SparseObjectives.py:381-387): per-observation losses for the weights, s * eta for the
        tilt, chained through par2's packing Jacobian when val2 is free."""
        num = self._numeric()
        if num is not None:
            return self._restore(num.grad2(val1, val2, val1_is_free, val2_is_free, *argv, **argk),
                                 val1, val2, val1_is_free, val2_is_free)
        f = self.fun
        set_par(self.par2, val2, val2_is_free)
        g = np.asarray(f.hyper_grad(self.par2, np.asarray(val1, dtype=np.float64), val1_is_free, *argv, **argk))
        J2 = self._jac2(val2, val2_is_free)
        if J2 is not None:
            g = np.asarray(J2.T @ g).ravel()
        set_par(self.par1, val1, val1_is_free)
        set_par(self.par2, val2, val2_is_free)
        return g

    def fun_free_hessian12(self, free_val1, free_val2, *argv, **argk):
        return self._cross12(free_val1, free_val2, True, True, *argv, **argk)

    def fun_free_hessian21(self, free_val1, free_val2, *argv, **argk):
        return self._cross12(free_val1, free_val2, True, True, *argv, **argk).T

    def fun_vector_hessian12(self, vec_val1, vec_val2, *argv, **argk):
        return self._cross12(vec_val1, vec_val2, False, False, *argv, **argk)

    def fun_vector_hessian21(self, vec_val1, vec_val2, *argv, **argk):
        return self._cross12(vec_val1, vec_val2, False, False, *argv, **argk).T

    def fun_hessian_free1_vector2(self, free_val1, vec_val2, *argv, **argk):
        return self._cross12(free_val1, vec_val2, True, False, *argv, **argk)

    def fun_hessian_vector1_free2(self, vec_val1, free_val2, *argv, **argk):
        return self._cross12(vec_val1, free_val2, False, True, *argv, **argk)


class ParametricSensitivity(object):
    """All-in-one linear sensitivity with an output map (LRVB/SparseObjectives.py:487-573);
    deprecated in the reference in favour of ParametricSensitivityLinearApproximation."""

    def __init__(self, objective_fun, input_par, output_par, hyper_par, input_to_output_converter,
                 optimal_input_par=None, objective_hessian=None, hyper_par_objective_fun=None):
        warnings.warn('ParametricSensitivity is deprecated.  Please use '
                      'ParametricSensitivityTaylorExpansion or ParametricSensitivityLinearApproximation.',
                      DeprecationWarning)
        self.input_par = input_par
        self.output_par = output_par
        self.hyper_par = hyper_par
        self.input_to_output_converter = input_to_output_converter
        self.objective_fun = objective_fun
        self.hyper_par_objective_fun = objective_fun if hyper_par_objective_fun is None else hyper_par_objective_fun
        self.parameter_converter = ParameterConverter(input_par, output_par, self.input_to_output_converter)
        self.objective = Objective(self.input_par, self.objective_fun)
        self.sensitivity_objective = TwoParameterObjective(self.input_par, self.hyper_par, self.hyper_par_objective_fun)
        self.set_optimal_input_par(optimal_input_par, objective_hessian)

    def set_optimal_input_par(self, optimal_input_par=None, objective_hessian=None):
        """Base point of the approximation: Hessian (given or built on the device), output Jacobian, cross Hessian with
        the hyper-parameter, one Cholesky solve -- the sequence of LRVB/SparseObjectives.py:524-552."""
        from .sensitivity import _factor_and_solve
        theta = self.input_par.get_free() if optimal_input_par is None else deepcopy(optimal_input_par)
        self.optimal_input_par = theta
        self.objective_hessian = (self.objective.fun_free_hessian(theta) if objective_hessian is None
                                  else objective_hessian)
        self.dout_din = self.parameter_converter.free_to_vec_jacobian(theta)
        self.optimal_hyper_par = self.hyper_par.get_vector()
        self.optimal_output_par = self.output_par.get_vector()
        cross = self.sensitivity_objective.fun_hessian_free1_vector2(theta, self.optimal_hyper_par)
        self.hyper_par_cross_hessian = cross
        self.hessian_chol, h_inv_cross = _factor_and_solve(self.objective_fun, self.objective_hessian, cross)
        self.hyper_par_sensitivity = -h_inv_cross

    def get_dinput_dhyper(self):
        return self.hyper_par_sensitivity

    def get_doutput_dhyper(self):
        return self.dout_din @ self.hyper_par_sensitivity

    def _input_shift(self, new_hyper_par):
        return self.hyper_par_sensitivity @ (np.asarray(new_hyper_par) - self.optimal_hyper_par)

    def predict_input_par_from_hyperparameters(self, new_hyper_par):
        return self.optimal_input_par + self._input_shift(new_hyper_par)

    def predict_output_par_from_hyperparameters(self, new_hyper_par, linear):
        """linear=True: first-order in the output too; False: the converter applied to the predicted input."""
        if linear:
            return self.optimal_output_par + self.dout_din @ self._input_shift(new_hyper_par)
        return self.parameter_converter.converter_free_to_vec(self.predict_input_par_from_hyperparameters(new_hyper_par))


# ---- index / sparse helpers (LRVB/SparseObjectives.py:581-657) --------------------------------
def make_index_param(param):
    index_param = deepcopy(param)
    index_param.set_vector(np.arange(0, index_param.vector_size()))
    return index_param


def get_sparse_sub_matrix(sub_matrix, row_indices, col_indices, row_dim, col_dim):
    sub_matrix = np.asarray(sub_matrix)
    r, c = np.nonzero(sub_matrix)
    rows = np.asarray(row_indices).astype(int)[r]
    cols = np.asarray(col_indices).astype(int)[c]
    return sp.sparse.csr_matrix((sub_matrix[r, c], (rows, cols)), (row_dim, col_dim))


def get_sparse_sub_hessian(sub_hessian, full_indices, full_hess_dim):
    return get_sparse_sub_matrix(sub_hessian, full_indices, full_indices, full_hess_dim, full_hess_dim)


def pack_csr_matrix(sp_mat):
    sp_mat = sp.sparse.csr_matrix(sp_mat)
    return {'data': sp_mat.data, 'indices': sp_mat.indices, 'indptr': sp_mat.indptr, 'shape': sp_mat.shape}


def unpack_csr_matrix(sp_mat_dict):
    return sp.sparse.csr_matrix((sp_mat_dict['data'], sp_mat_dict['indices'], sp_mat_dict['indptr']),
                                shape=sp_mat_dict['shape'])


def json_pack_csr_matrix(sp_mat):
    """JSON-serialisable form of a CSR matrix (LRVB/SparseObjectives.py:631-639; the reference encodes
    the three arrays with json_tricks, absent here -- plain lists carry the same content)."""
    assert sp.sparse.isspmatrix_csr(sp_mat)
    sp_mat = sp.sparse.csr_matrix(sp_mat)
    return {'data': sp_mat.data.tolist(), 'indices': sp_mat.indices.tolist(), 'indptr': sp_mat.indptr.tolist(),
            'shape': [int(t) for t in sp_mat.shape], 'type': 'csr_matrix'}


def json_unpack_csr_matrix(sp_mat_dict):
    assert sp_mat_dict['type'] == 'csr_matrix'
    return sp.sparse.csr_matrix((np.asarray(sp_mat_dict['data'], dtype=np.float64),
                                 np.asarray(sp_mat_dict['indices'], dtype=np.int64),
                                 np.asarray(sp_mat_dict['indptr'], dtype=np.int64)),
                                shape=tuple(sp_mat_dict['shape']))


def get_sym_matrix_inv_sqrt(block_hessian, ev_min=None, ev_max=None):
    """Deprecated in the reference as well (LRVB/SparseObjectives.py:662-664): it raises and points at
    `OptimizationUtils.get_sym_matrix_inv_sqrt`."""
    raise DeprecationWarning('Deprecated.  Use OptimizationUtils.get_sym_matrix_inv_sqrt instead.')
