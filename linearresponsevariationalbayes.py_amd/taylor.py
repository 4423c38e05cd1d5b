"""Higher-order sensitivity of an optimum to a hyper-parameter: d^k eta_hat / d eps^k and the Taylor series of
eta_hat(eps) -- `ParametricSensitivityTaylorExpansion` of LRVB/ModelSensitivity.py:382-515 (with the term algebra of
:83-316) for declared objectives.

The reference differentiates the gradient closure g(eta, eps) with nested autograd JVPs.  Here g is declared, so its
mixed directional derivatives have closed forms that run on the device:

* in vector coordinates the linear predictor is linear in eta, and D^j g_eta [u_1 .. u_j] is one fused pass with a
  per-observation coefficient (`lrvb_dk_grad_vec`);
* in free coordinates with element-wise packing maps eta = c(phi) (the four box kinds of LRVB/Parameters.py:31-61)
  the gradient is G(phi) = c'(phi) o g_eta(c(phi)), and D^i G [v_1 .. v_i] follows from the product rule over subsets
  of the directions and Faa di Bruno's formula over set partitions, each leaf being one `lrvb_dk_grad_vec` call;
* the objective is LINEAR in the two hyper-parameters the device path declares (observation weights, linear tilt),
  so derivatives of order >= 2 in eps vanish and the first one is g evaluated with the direction as weights / tilt.

PSD and simplex blocks are not element-wise: with them the class raises NotImplementedError (use vector coordinates).
The recursion for d^k eta_hat / d eps^k is the implicit-function theorem applied k times to g(eta_hat(eps), eps) = 0.
"""
import math
from copy import deepcopy

import numpy as np
from numpy.polynomial import polynomial as _poly

from . import _hip
from .objectives import Objective, set_par


# ---- the terms of d^k/dt^k g(eta(t), eps0 + t d eps) -------------------------------------------------------------
class DerivativeTerm(object):
    """prefactor * D_eta^(sum m) D_eps^(eps_order) g [eta^(1) x m_1, eta^(2) x m_2, ..., d eps x eps_order] with
    eta_orders = [m_1, m_2, ...] (LRVB/ModelSensitivity.py:83-204)."""

    def __init__(self, eps_order, eta_orders, prefactor):
        self.eps_order = int(eps_order)
        self.eta_orders = [int(m) for m in eta_orders]
        self.prefactor = float(prefactor)
        self.order = self.eps_order + sum((i + 1) * m for i, m in enumerate(self.eta_orders))
        assert self.eps_order >= 0 and all(m >= 0 for m in self.eta_orders)
        assert len(self.eta_orders) == self.order

    def __str__(self):
        return 'Order: {}\t{} * eta{} * eps[{}]'.format(self.order, self.prefactor, self.eta_orders, self.eps_order)

    def key(self):
        return (self.eps_order, tuple(self.eta_orders))

    def differentiate(self):
        """The terms of d/dt of this one: through the explicit eps argument, through g's eta argument (a new factor
        eta^(1)), and through each factor eta^(i) -> eta^(i+1) (m_i ways)."""
        grown = self.eta_orders + [0]
        out = [DerivativeTerm(self.eps_order + 1, grown, self.prefactor)]
        first = list(grown)
        first[0] += 1
        out.append(DerivativeTerm(self.eps_order, first, self.prefactor))
        for i, m in enumerate(self.eta_orders):
            if m > 0:
                moved = list(grown)
                moved[i] -= 1
                moved[i + 1] += 1
                out.append(DerivativeTerm(self.eps_order, moved, self.prefactor * m))
        return out


def get_taylor_base_terms():
    """d/dt g = D_eps g [d eps] + D_eta g [eta^(1)]  (LRVB/ModelSensitivity.py:282-298)."""
    return [DerivativeTerm(1, [0], 1.0), DerivativeTerm(0, [1], 1.0)]


def consolidate_terms(terms):
    """Sum the prefactors of terms with equal orders, first-seen order kept (:255-266)."""
    merged = {}
    for t in terms:
        k = t.key()
        if k in merged:
            merged[k] = DerivativeTerm(t.eps_order, t.eta_orders, merged[k].prefactor + t.prefactor)
        else:
            merged[k] = t
    return list(merged.values())


def differentiate_terms(terms):
    out = []
    for t in terms:
        out += t.differentiate()
    return consolidate_terms(out)


def _set_partitions(items):
    """All partitions of a list into non-empty blocks."""
    if not items:
        yield []
        return
    head, rest = items[0], items[1:]
    for part in _set_partitions(rest):
        yield [[head]] + part
        for i in range(len(part)):
            yield part[:i] + [[head] + part[i]] + part[i + 1:]


def _subsets(items):
    for mask in range(1 << len(items)):
        yield [x for b, x in enumerate(items) if mask >> b & 1], [x for b, x in enumerate(items) if not mask >> b & 1]


def _sigmoid_derivatives(s, max_order):
    """[sigma^(1)(z), ..., sigma^(max_order)(z)] from s = sigma(z): P_1 = s - s^2, P_(m+1) = P_m' (s - s^2)."""
    p = np.array([0.0, 1.0, -1.0])
    out = []
    for _ in range(max_order):
        out.append(_poly.polyval(s, p))
        p = _poly.polymul(_poly.polyder(p), np.array([0.0, 1.0, -1.0]))
    return out


def box_map_derivatives(free_val, blocks, max_order):
    """c^(m)(phi) element-wise for m = 0 .. max_order, as an array (max_order + 1, D), for a layout made of box
    blocks only (identity / exp + lb / ub - exp(-.) / scaled logistic; LRVB/Parameters.py:47-61)."""
    free_val = np.asarray(free_val, dtype=np.float64)
    out = np.zeros((max_order + 1, free_val.size))
    off = 0
    for b in blocks:
        if b['kind'] != _hip.BLOCK_BOX:
            raise NotImplementedError('higher-order sensitivity in free coordinates needs element-wise (box) packing maps; '
                                      'PSD and simplex blocks are not: use input_is_free=False')
        n = b['free_size']
        f = free_val[off:off + n]
        lb, ub = b['lb'], b['ub']
        if lb == -np.inf and ub == np.inf:
            out[0, off:off + n] = f
            if max_order >= 1:
                out[1, off:off + n] = 1.0
        elif ub == np.inf:
            e = np.exp(f)
            out[0, off:off + n] = e + lb
            out[1:, off:off + n] = e
        elif lb == -np.inf:
            e = np.exp(-f)
            out[0, off:off + n] = ub - e
            for m in range(1, max_order + 1):
                out[m, off:off + n] = e if m % 2 == 1 else -e
        else:
            s = 1.0 / (1.0 + np.exp(-f))
            out[0, off:off + n] = (ub - lb) * s + lb
            for m, d in enumerate(_sigmoid_derivatives(s, max_order), start=1):
                out[m, off:off + n] = (ub - lb) * d
        off += n
    return out


class ParametricSensitivityTaylorExpansion(object):
    """Same constructor and methods as the reference class (LRVB/ModelSensitivity.py:382-515).  `objective_functor`
    must be a device functor and `hyper_par` its `weights_par` or `tilt_par`, in vector coordinates."""

    def __init__(self, objective_functor, input_par, hyper_par, input_val0, hyper_val0, order,
                 input_is_free=True, hyper_is_free=False, hess0=None, hyper_par_objective_functor=None):
        if hyper_is_free:
            raise NotImplementedError('the declared hyper-parameters (weights, tilt) live in vector coordinates')
        fun = objective_functor if hyper_par_objective_functor is None else hyper_par_objective_functor
        if not hasattr(objective_functor, 'ctx') or not hasattr(objective_functor.ctx, 'dk_grad_vec') \
                or not hasattr(fun, 'hyper_kind'):
            raise NotImplementedError('higher-order sensitivity needs a device functor (DeviceObjective); an opaque '
                                      'closure would have to be traced')
        if getattr(objective_functor, 'scale_fun', None) is not None:
            raise NotImplementedError('objectives with extra arguments (scale_fun) are not supported here')
        self.objective_functor = objective_functor
        self.hyper_par_objective_functor = fun
        self.input_par = input_par
        self.hyper_par = hyper_par
        self.input_is_free = input_is_free
        self.hyper_is_free = hyper_is_free
        self.hyper_kind = fun.hyper_kind(hyper_par)
        self.ctx = objective_functor.ctx
        self.objective = Objective(self.input_par, self.objective_functor)
        self.set_base_values(input_val0, hyper_val0, hess0=hess0)
        self.set_order(order)

    # ---- base point ------------------------------------------------------------------------------------------
    def set_par_to_base_values(self):
        set_par(self.input_par, self.input_val0, self.input_is_free)
        set_par(self.hyper_par, self.hyper_val0, self.hyper_is_free)

    def set_base_values(self, input_val0, hyper_val0, hess0=None):
        self.input_val0 = np.array(deepcopy(input_val0), dtype=np.float64)
        self.hyper_val0 = np.array(deepcopy(hyper_val0), dtype=np.float64)
        self.set_par_to_base_values()
        if hess0 is None:
            hess = self.objective.fun_free_hessian if self.input_is_free else self.objective.fun_vector_hessian
            self.hess0 = hess(self.input_val0)
        else:
            self.hess0 = np.asarray(hess0, dtype=np.float64)
        self.ctx.chol_factor(self.hess0)                          # raises LinAlgError if not positive definite
        self._chol_token = getattr(self.ctx, 'chol_token', None)
        self._cache = {}

    def _solve(self, rhs):
        if getattr(self.ctx, 'chol_token', None) != self._chol_token:
            self.ctx.chol_factor(self.hess0)
            self._chol_token = getattr(self.ctx, 'chol_token', None)
        return self.ctx.chol_solve(np.asarray(rhs, dtype=np.float64).reshape(-1, 1)).ravel()

    # ---- mixed directional derivatives of the gradient at the base point ----------------------------------------
    def _box(self, max_order):
        if not self.input_is_free:
            return None
        have = self._cache.get('box')
        if have is None or have.shape[0] <= max_order:
            have = box_map_derivatives(self.input_val0, self.input_par.layout_blocks(), max_order)
            self._cache['box'] = have
        return have

    def _eta0(self):
        return self._box(1)[0] if self.input_is_free else self.input_val0

    def _h(self, dirs_eta, eps_dir):
        """D_eta^r h [dirs] in vector coordinates, h = g_eta (eps_dir None) or d g_eta / d eps [eps_dir]."""
        r = len(dirs_eta)
        # the derivative is symmetric in its directions and the same leaves recur across subsets and partitions:
        # memoised per evaluation (the key is the multiset of direction vectors)
        key = (eps_dir is None, tuple(sorted(np.ascontiguousarray(u).tobytes() for u in dirs_eta)))
        memo = self._cache.setdefault('h', {})
        if key in memo:
            return memo[key]
        self.objective_functor._push_state()
        U = np.array(dirs_eta) if r else None
        if eps_dir is None:
            memo[key] = self.ctx.dk_grad_vec(self._eta0(), U, None, True)
            return memo[key]
        if self.hyper_kind == 'weights':
            memo[key] = self.ctx.dk_grad_vec(self._eta0(), U, eps_dir, False)
            return memo[key]
        # tilt: the objective holds eps^T eta, so d g_eta / d eps [d eps] = d eps, constant in eta
        if r > 0:
            return np.zeros(self.input_val0.size if not self.input_is_free else self._eta0().size)
        return np.asarray(eps_dir, dtype=np.float64).copy()

    def _dg(self, dirs, eps_dir):
        """D_input^i [gradient in input coordinates] [dirs], the gradient optionally differentiated once along eps."""
        i = len(dirs)
        if not self.input_is_free:
            return self._h(list(dirs), eps_dir)
        c = self._box(i + 1)
        idx = list(range(i))
        total = np.zeros(self.input_val0.size)
        for S, T in _subsets(idx):
            lead = c[len(S) + 1].copy()
            for k in S:
                lead *= dirs[k]
            if not T:
                total += lead * self._h([], eps_dir)
                continue
            inner = np.zeros_like(total)
            for part in _set_partitions(T):
                vecs = []
                for block in part:
                    u = c[len(block)].copy()
                    for k in block:
                        u *= dirs[k]
                    vecs.append(u)
                inner += self._h(vecs, eps_dir)
            total += lead * inner
        return total

    def _evaluate_term(self, term, eta_derivs, dhyper):
        if term.eps_order >= 2:
            return 0.0                                           # the objective is linear in the declared hyper-parameters
        dirs = []
        for i, m in enumerate(term.eta_orders):
            if m > 0:
                dirs += [eta_derivs[i]] * m
        return term.prefactor * self._dg(dirs, dhyper if term.eps_order == 1 else None)

    # ---- the recursion -----------------------------------------------------------------------------------------
    def set_order(self, order):
        if order < 1:
            raise ValueError('order must be at least one.')
        self.order = int(order)
        self.taylor_terms_list = [get_taylor_base_terms()]
        for _ in range(self.order - 1):
            self.taylor_terms_list.append(differentiate_terms(self.taylor_terms_list[-1]))

    def _eta_derivatives(self, dhyper, k):
        """[eta^(1), ..., eta^(k)] along dhyper: eta^(j) = -H^-1 (terms of order j that do not contain eta^(j))."""
        dhyper = np.asarray(dhyper, dtype=np.float64).ravel()
        if dhyper.size != self.hyper_val0.size:
            raise ValueError('dhyper is the wrong size')
        self._cache['h'] = {}                                    # leaves are memoised per direction dhyper
        derivs = []
        for j in range(1, k + 1):
            rhs = np.zeros(self.input_val0.size)
            for term in self.taylor_terms_list[j - 1]:
                if term.eta_orders[-1] == 0:                     # every term but H eta^(j)
                    rhs = rhs + self._evaluate_term(term, derivs, dhyper)
            derivs.append(-self._solve(rhs))
        return derivs

    def evaluate_dkinput_dhyperk(self, dhyper, k):
        if k <= 0:
            raise ValueError('k must be at least one.')
        if k > self.order:
            raise ValueError('k must be no greater than the declared order={}'.format(self.order))
        out = self._eta_derivatives(dhyper, k)[k - 1]
        self.set_par_to_base_values()
        return out

    def evaluate_taylor_series(self, dhyper, add_offset=True, max_order=None):
        if max_order is None:
            max_order = self.order
        if max_order <= 0:
            raise ValueError('max_order must be greater than zero.')
        if max_order > self.order:
            raise ValueError('max_order must be no greater than the declared order={}'.format(self.order))
        derivs = self._eta_derivatives(dhyper, max_order)
        dinput = sum(d / float(math.factorial(k)) for k, d in enumerate(derivs, start=1))
        self.set_par_to_base_values()
        return dinput + self.input_val0 if add_offset else dinput

    def print_terms(self, k=None):
        if k is not None and k > self.order:
            raise ValueError('k must be no greater than order={}'.format(self.order))
        for order in range(self.order):
            if k is None or order == (k - 1):
                print('\nTerms for order {}:'.format(order + 1))
                for term in self.taylor_terms_list[order]:
                    print(term)
