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
* with PSD (log-Cholesky) or simplex (softmax) blocks the packing map is not element-wise.  The same two rules then
  read, for G(phi) = J(phi)^T g_eta(c(phi)):
      D^i G [v_1 .. v_i] = sum over subsets S of the directions  ( D^(|S|+1) c [v_S, .] )^T
                           sum over set partitions pi of the rest  D^|pi| g_eta [ D^|B| c [v_B] : B in pi ]
  (multivariate Faa di Bruno).  The leaves D^r g_eta [...] are still `lrvb_dk_grad_vec` calls on the device -- every
  O(N) contraction -- while the derivatives of the small N-independent map c come from forward-mode AD of c on the
  host (torch.func.jvp, the counterpart of the autograd JVPs the reference applies to the whole objective,
  LRVB/ModelSensitivity.py:38-62), see `PackingJet`.
* the GRADIENT is linear in every hyper-parameter the device path declares, taken in its vector coordinates
  (observation weights, linear tilt, prior mean m, prior information A, quadratic scale s, Gaussian likelihood precision):
  derivatives of order >= 2 in eps vanish there and the first one is a closed form (`hyper_direction_vec` of the
  functor; its O(N) leaves are `lrvb_dk_grad_vec` calls).  A hyper-parameter given in FREE coordinates eps = c_h^-1(.)
  enters through its own packing map: D_eps^j g [d eps^j] = (d g / d eps_vec) [ D^j c_h [d eps^j] ].

The recursion for d^k eta_hat / d eps^k is the implicit-function theorem applied k times to g(eta_hat(eps), eps) = 0.
`append_jvp` and `generate_two_term_derivative_array` (LRVB/ModelSensitivity.py:38-62, 221-234) are provided for
closures written with torch operations (torch.func in place of autograd), and `DerivativeTerm` carries the optional
evaluator lists of the reference class so that such closures can be expanded with the same term algebra.
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

    def __init__(self, eps_order, eta_orders, prefactor, eval_eta_derivs=None, eval_g_derivs=None):
        self.eps_order = int(eps_order)
        self.eta_orders = [int(m) for m in eta_orders]
        self.prefactor = float(prefactor)
        self.order = self.eps_order + sum((i + 1) * m for i, m in enumerate(self.eta_orders))
        assert self.eps_order >= 0 and all(m >= 0 for m in self.eta_orders)
        assert len(self.eta_orders) == self.order
        # optional evaluators, as in the reference class: eval_eta_derivs[i](eta0, eps0, deps) = d^(i+1) eta / d eps^(i+1)
        # along deps; eval_g_derivs[i][j](eta0, eps0, v_1 .. v_i, w_1 .. w_j) = D_eta^i D_eps^j g [v.., w..]
        self.eval_eta_derivs = eval_eta_derivs
        self.eval_g_derivs = eval_g_derivs
        if eval_g_derivs is not None:
            assert len(eval_g_derivs) > sum(self.eta_orders) and len(eval_g_derivs[sum(self.eta_orders)]) > self.eps_order
            self.eval_g_deriv = eval_g_derivs[sum(self.eta_orders)][self.eps_order]

    def evaluate(self, eta0, eps0, deps):
        """prefactor * D g [eta^(1) x m_1, ..., deps x eps_order] through the evaluator lists (LRVB/ModelSensitivity.py:139-153)."""
        if self.eval_g_derivs is None or self.eval_eta_derivs is None:
            raise ValueError('this term carries no evaluators (it was built for the device path)')
        vec_args = []
        for i, m in enumerate(self.eta_orders):
            if m > 0:
                vec = self.eval_eta_derivs[i](eta0, eps0, deps)
                vec_args += [vec] * m
        vec_args += [deps] * self.eps_order
        return self.prefactor * self.eval_g_deriv(eta0, eps0, *vec_args)

    def check_similarity(self, term):
        return self.key() == term.key()

    def combine_with(self, term):
        assert self.check_similarity(term)
        return DerivativeTerm(self.eps_order, self.eta_orders, self.prefactor + term.prefactor,
                              self.eval_eta_derivs, self.eval_g_derivs)

    def __str__(self):
        return 'Order: {}\t{} * eta{} * eps[{}]'.format(self.order, self.prefactor, self.eta_orders, self.eps_order)

    def key(self):
        return (self.eps_order, tuple(self.eta_orders))

    def differentiate(self, eval_next_eta_deriv=None):
        """The terms of d/dt of this one: through the explicit eps argument, through g's eta argument (a new factor
        eta^(1)), and through each factor eta^(i) -> eta^(i+1) (m_i ways).  `eval_next_eta_deriv` extends the
        evaluator list when the term carries one (the reference's signature)."""
        etas = self.eval_eta_derivs
        if etas is not None and eval_next_eta_deriv is not None:
            etas = list(etas) + [eval_next_eta_deriv]
        gd = self.eval_g_derivs
        grown = self.eta_orders + [0]
        out = [DerivativeTerm(self.eps_order + 1, grown, self.prefactor, etas, gd)]
        first = list(grown)
        first[0] += 1
        out.append(DerivativeTerm(self.eps_order, first, self.prefactor, etas, gd))
        for i, m in enumerate(self.eta_orders):
            if m > 0:
                moved = list(grown)
                moved[i] -= 1
                moved[i + 1] += 1
                out.append(DerivativeTerm(self.eps_order, moved, self.prefactor * m, etas, gd))
        return out


def get_taylor_base_terms(eval_g_derivs=None):
    """d/dt g = D_eps g [d eps] + D_eta g [eta^(1)]  (LRVB/ModelSensitivity.py:282-298)."""
    etas = None if eval_g_derivs is None else []
    return [DerivativeTerm(1, [0], 1.0, etas, eval_g_derivs), DerivativeTerm(0, [1], 1.0, etas, eval_g_derivs)]


def append_jvp(fun, num_base_args=1, argnum=0):
    """fun(x_1 .. x_B, v_1 .. v_r) -> fun_jvp(x_1 .. x_B, v_1 .. v_r, v) = d fun / d x_argnum [v] at the same base
    point and earlier directions: the building block of nested Jacobian-vector products (LRVB/ModelSensitivity.py:38-62),
    for closures written with torch operations (torch.func.jvp in place of autograd.make_jvp)."""
    import torch
    assert argnum < num_base_args

    def fun_jvp(*argv):
        base, vecs = list(argv[:num_base_args]), argv[num_base_args:]
        earlier, new = vecs[:-1], vecs[-1]

        def of_x(x):
            args = base[:argnum] + [x] + base[argnum + 1:]
            return fun(*args, *earlier)
        return torch.func.jvp(of_x, (base[argnum],), (new,))[1]
    return fun_jvp


def generate_two_term_derivative_array(fun, order):
    """eval_fun_derivs[i][j](x1, x2, v_1 .. v_i, w_1 .. w_j) = D_x1^i D_x2^j fun [v.., w..], i, j <= order
    (LRVB/ModelSensitivity.py:221-234)."""
    derivs = [[fun]]
    for i in range(order):
        if i > 0:
            derivs.append([append_jvp(derivs[i - 1][0], num_base_args=2, argnum=0)])
        for j in range(order):
            derivs[i].append(_append_x2(derivs[i][j], i))
    return derivs


def _append_x2(f, n_x1_dirs):
    """One more x2 direction, appended AFTER the existing x2 directions (argument order x1, x2, v.., w..)."""
    import torch

    def g(*argv):
        x1, x2, rest = argv[0], argv[1], argv[2:-1]
        return torch.func.jvp(lambda z: f(x1, z, *rest), (x2,), (argv[-1],))[1]
    return g


def consolidate_terms(dterms):
    """Sum the prefactors of terms with equal orders, first-seen order kept (:255-266)."""
    merged = {}
    for t in dterms:
        k = t.key()
        merged[k] = merged[k].combine_with(t) if k in merged else t
    return list(merged.values())


def _differentiate_terms(dterms, eval_next_eta_deriv=None):
    out = []
    for t in dterms:
        out += t.differentiate(eval_next_eta_deriv)
    return consolidate_terms(out)


def differentiate_terms(hess0, dterms):
    """Derivatives of the terms with respect to the hyper-parameter, the next eta derivative being
    -hess0^-1 (terms without it): the module-level function of LRVB/ModelSensitivity.py:327-335 (the terms must carry
    evaluators; `hess0 = None` differentiates the orders only)."""
    if hess0 is None:
        return _differentiate_terms(dterms)

    def eval_next_eta_deriv(eta, eps, deps):
        return evaluate_dketa_depsk(hess0, dterms, eta, eps, deps)
    return _differentiate_terms(dterms, eval_next_eta_deriv)


def evaluate_terms(dterms, eta0, eps0, deps, include_highest_eta_order=True):
    """Sum of `term.evaluate(eta0, eps0, deps)` over the terms (those without the highest eta derivative if asked):
    LRVB/ModelSensitivity.py:274-282.  The terms must carry callables (`eval_eta_derivs`, `eval_g_derivs`), as the
    reference's do; the terms this package's Taylor class builds for DECLARED objectives carry orders only and are
    evaluated by the class on the device (`evaluate_dkinput_dhyperk`)."""
    vec = None
    for term in dterms:
        if include_highest_eta_order or (term.eta_orders[-1] == 0):
            val = term.evaluate(eta0, eps0, deps)
            vec = val if vec is None else vec + val
    return vec


def evaluate_dketa_depsk(hess0, dterms, eta0, eps0, deps):
    """d^k eta / d eps^k [deps] = -H^-1 (terms that do not contain it): LRVB/ModelSensitivity.py:312-316."""
    vec = evaluate_terms(dterms, eta0, eps0, deps, include_highest_eta_order=False)
    assert vec is not None
    return -1 * np.linalg.solve(hess0, vec)


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


class PackingJet(object):
    """Directional derivatives of the packing map c: free -> vector at one point phi, for ANY layout (box, log-Cholesky
    PSD, softmax simplex blocks; LRVB/Parameters.py:47-61, MatrixParameters.py:101-112, SimplexParams.py:11-18):

        vec(dirs)  = D^m c [w_1 .. w_m]        (a V-vector; m = 0 is c(phi) itself)
        mat(dirs)  = D^(m+1) c [w_1 .. w_m, .]  (V x D: one slot left open)

    by nested forward-mode AD (torch.func.jvp / jacfwd) of a torch restatement of the map.  The map is small and
    N-independent; this is the host-side counterpart of the autograd JVPs the reference nests (it differentiates
    the whole objective that way, :38-62).  Results are memoised per set of directions."""

    def __init__(self, free_val, blocks):
        import torch
        self._torch = torch
        self._phi = torch.tensor(np.asarray(free_val, dtype=np.float64))
        self._blocks = [dict(b) for b in blocks]
        self._memo = {}

    def _map(self, phi):
        torch = self._torch
        out, off = [], 0
        for b in self._blocks:
            n = int(b['free_size'])
            f = phi[off:off + n]
            off += n
            if b['kind'] == _hip.BLOCK_BOX:
                lb, ub = b['lb'], b['ub']
                if lb == -np.inf and ub == np.inf:
                    out.append(f)
                elif ub == np.inf:
                    out.append(torch.exp(f) + lb)
                elif lb == -np.inf:
                    out.append(ub - torch.exp(-f))
                else:
                    out.append((ub - lb) * torch.sigmoid(f) + lb)
            elif b['kind'] == _hip.BLOCK_PSD:
                k = int(b['dim0'])
                r, c = np.tril_indices(k)                      # row-major lower triangle: index c + r (r + 1) / 2
                L = torch.zeros((k, k), dtype=phi.dtype).index_put((torch.tensor(r), torch.tensor(c)), f)
                d = torch.diagonal(L)
                L = L - torch.diag(d) + torch.diag(torch.exp(d))
                A = L @ L.T + float(b['lb']) * torch.eye(k, dtype=phi.dtype)
                out.append(A[torch.tensor(r), torch.tensor(c)])
            else:                                              # simplex rows: softmax([0, f])
                rows, K = int(b['dim0']), int(b['dim1'])
                z = torch.cat([torch.zeros((rows, 1), dtype=phi.dtype), f.reshape(rows, K - 1)], dim=1)
                out.append(torch.softmax(z, dim=1).reshape(-1))
        return torch.cat(out)

    def _directional(self, dirs):
        torch = self._torch
        f = self._map
        for w in dirs:
            f = (lambda g, wt: (lambda p: torch.func.jvp(g, (p,), (wt,))[1]))(f, torch.tensor(np.asarray(w, dtype=np.float64)))
        return f

    @staticmethod
    def _key(dirs):
        return tuple(sorted(np.ascontiguousarray(w, dtype=np.float64).tobytes() for w in dirs))

    def vec(self, dirs):
        key = ('v', self._key(dirs))
        if key not in self._memo:
            self._memo[key] = self._directional(dirs)(self._phi).numpy()
        return self._memo[key]

    def mat(self, dirs):
        key = ('m', self._key(dirs))
        if key not in self._memo:
            self._memo[key] = self._torch.func.jacfwd(self._directional(dirs))(self._phi).numpy()
        return self._memo[key]


class ParametricSensitivityTaylorExpansion(object):
    """Same constructor and methods as the reference class (LRVB/ModelSensitivity.py:382-515).  `objective_functor`
    must be a device functor and `hyper_par` one of its declared hyper-parameters (`hyper_pars`), in vector or free
    coordinates."""

    def __init__(self, objective_functor, input_par, hyper_par, input_val0, hyper_val0, order,
                 input_is_free=True, hyper_is_free=False, hess0=None, hyper_par_objective_functor=None):
        fun = objective_functor if hyper_par_objective_functor is None else hyper_par_objective_functor
        if not hasattr(objective_functor, 'ctx') or not hasattr(objective_functor.ctx, 'dk_grad_vec') \
                or not hasattr(fun, 'hyper_kind') or not hasattr(fun, 'hyper_direction_vec'):
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

    # ---- reference helper methods (LRVB/ModelSensitivity.py:412-450) ---------------------------------------------
    def cache_and_eval(self, diff_fun, *argv, **argk):
        """Evaluate and put the parameters back to the base values (:412-415)."""
        result = diff_fun(*argv, **argk)
        self.set_par_to_base_values()
        return result

    def objective_gradient(self, input_val, hyper_val, *argv, **argk):
        """Gradient of the objective in the input parameter at (input_val, hyper_val): the function whose Taylor
        expansion the class forms (:430-433), evaluated on the device."""
        from .objectives import TwoParameterObjective
        fun = self.hyper_par_objective_functor
        two = TwoParameterObjective(self.input_par, self.hyper_par, fun)
        out = two.fun_grad1(input_val, hyper_val, self.input_is_free, self.hyper_is_free, *argv, **argk)
        self.set_par_to_base_values()
        return out

    def get_dkinput_dhyperk_from_terms(self, dterms):
        """A function (input_val, hyper_val, dhyper, tolerance) -> next derivative from the terms, checked to be asked
        for at the base point (:436-444); the evaluation is this class's device recursion for the order of `dterms`."""
        k = max(sum(t.eta_orders[i] * (i + 1) for i in range(len(t.eta_orders))) + t.eps_order for t in dterms)

        def dkinput_dhyperk(input_val, hyper_val, dhyper, tolerance=1e-8):
            if tolerance is not None:
                assert np.max(np.abs(np.asarray(input_val) - self.input_val0)) <= tolerance
                assert np.max(np.abs(np.asarray(hyper_val) - self.hyper_val0)) <= tolerance
            return self.evaluate_dkinput_dhyperk(dhyper, k)
        return dkinput_dhyperk

    def differentiate_terms(self, dterms, eval_next_eta_deriv=None):
        """:446-450."""
        return _differentiate_terms(dterms, eval_next_eta_deriv)

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
    def _all_box(self):
        return all(b['kind'] == _hip.BLOCK_BOX for b in self.input_par.layout_blocks())

    def _jet(self):
        jet = self._cache.get('jet')
        if jet is None:
            jet = self._cache['jet'] = PackingJet(self.input_val0, self.input_par.layout_blocks())
        return jet

    def _box(self, max_order):
        if not self.input_is_free:
            return None
        have = self._cache.get('box')
        if have is None or have.shape[0] <= max_order:
            have = box_map_derivatives(self.input_val0, self.input_par.layout_blocks(), max_order)
            self._cache['box'] = have
        return have

    def _eta0(self):
        if not self.input_is_free:
            return self.input_val0
        return self._box(1)[0] if self._all_box() else self._jet().vec([])

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
        # d g_eta / d eps_vec [eps_dir], differentiated r times in eta: a closed form per hyper-parameter kind
        key = key + (np.ascontiguousarray(eps_dir, dtype=np.float64).tobytes(),)
        if key not in memo:
            memo[key] = self.hyper_par_objective_functor.hyper_direction_vec(self.hyper_par, self._eta0(), U, eps_dir)
        return memo[key]

    def _eps_direction(self, dhyper, j):
        """The direction in the hyper-parameter's VECTOR coordinates that carries the j-th eps derivative along dhyper:
        dhyper itself for j = 1 in vector coordinates (None for j >= 2: the gradient is linear there), D^j c_h [dhyper^j]
        for a free hyper-parameter with packing map c_h."""
        if not self.hyper_is_free:
            return np.asarray(dhyper, dtype=np.float64) if j == 1 else None
        blocks = self.hyper_par.layout_blocks()
        if all(b['kind'] == _hip.BLOCK_BOX for b in blocks):
            return box_map_derivatives(self.hyper_val0, blocks, j)[j] * np.asarray(dhyper, dtype=np.float64) ** j
        jet = self._cache.get('hjet')
        if jet is None:
            jet = self._cache['hjet'] = PackingJet(self.hyper_val0, blocks)
        return jet.vec([dhyper] * j)

    def _dg(self, dirs, eps_dir):
        """D_input^i [gradient in input coordinates] [dirs], the gradient optionally differentiated once along eps."""
        i = len(dirs)
        if not self.input_is_free:
            return self._h(list(dirs), eps_dir)
        if not self._all_box():
            return self._dg_general(dirs, eps_dir)
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

    def _dg_general(self, dirs, eps_dir):
        """The same sum for a packing map that is not element-wise (PSD / simplex blocks): the leading factor is the
        transposed open-slot derivative of the map, the inner directions its directional derivatives."""
        jet = self._jet()
        idx = list(range(len(dirs)))
        total = np.zeros(self.input_val0.size)
        for S, T in _subsets(idx):
            lead = jet.mat([dirs[k] for k in S])                 # V x D
            if not T:
                total += lead.T @ self._h([], eps_dir)
                continue
            inner = np.zeros(lead.shape[0])
            for part in _set_partitions(T):
                inner += self._h([jet.vec([dirs[k] for k in block]) for block in part], eps_dir)
            total += lead.T @ inner
        return total

    def _evaluate_term(self, term, eta_derivs, dhyper):
        eps_dir = None
        if term.eps_order >= 1:
            eps_dir = self._eps_direction(dhyper, term.eps_order)
            if eps_dir is None:
                return 0.0                                       # the gradient is linear in the hyper-parameter's vector form
        dirs = []
        for i, m in enumerate(term.eta_orders):
            if m > 0:
                dirs += [eta_derivs[i]] * m
        return term.prefactor * self._dg(dirs, eps_dir)

    # ---- the recursion -----------------------------------------------------------------------------------------
    def set_order(self, order):
        if order < 1:
            raise ValueError('order must be at least one.')
        self.order = int(order)
        self.taylor_terms_list = [get_taylor_base_terms()]
        for _ in range(self.order - 1):
            self.taylor_terms_list.append(_differentiate_terms(self.taylor_terms_list[-1]))

    def _eta_derivatives(self, dhyper, k):
        """[eta^(1), ..., eta^(k)] along dhyper: eta^(j) = -H^-1 (terms of order j that do not contain eta^(j))."""
        dhyper = np.asarray(dhyper, dtype=np.float64).ravel()
        if dhyper.size != self.hyper_val0.size:
            raise ValueError('dhyper is the wrong size')
        # the derivatives are those AT THE BASE VALUES, whatever the parameter objects hold at the moment (the reference
        # evaluates its closures at (input_val0, hyper_val0) explicitly, LRVB/ModelSensitivity.py:452-470)
        self.set_par_to_base_values()
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
