"""H^-1 v by conjugate gradients on Hessian-vector products, many masked right-hand sides.

Drop-in for LRVB/ConjugateGradient.py: `ConjugateGradientSolver` (:63-105), `get_masks` (:46-57),
`split_vector` (:19-30), `recursive_split` (:36-43).  When `eval_hessian_vector_product` is the
`fun_free_hvp` of an Objective over a declared device objective, the whole CG loop runs on the
device (`lrvb_cg_solve`: one fused observation pass per iteration, scipy's stopping rule
||r|| < tol ||b|| with tol = 1e-8 as at :70, :82-84); any other callable is driven through
scipy's `cg` on the host exactly as the reference does, with `tol` spelled `rtol` for
scipy >= 1.14.
"""
import time

import numpy as np
import scipy.sparse.linalg
from scipy.sparse.linalg import LinearOperator


def split_vector(vec):
    vec = np.asarray(vec, dtype=bool)
    true_inds = np.flatnonzero(vec)
    half = len(true_inds) // 2
    first = np.full(len(vec), False)
    second = np.full(len(vec), False)
    first[true_inds[:half]] = True
    second[true_inds[half:]] = True
    return first, second


def recursive_split(mask, results=None, terminate_len=10):
    """Appends to `results` masks with at most terminate_len True values each.  (The reference's
    mutable default `results=[]`, :36, is not replicated: pass a list or use the return value.)"""
    if results is None:
        results = []
    if np.sum(mask) > terminate_len:
        m1, m2 = split_vector(mask)
        recursive_split(m1, results=results, terminate_len=terminate_len)
        recursive_split(m2, results=results, terminate_len=terminate_len)
    else:
        results.append(mask)
    return results


def get_masks(full_len, min_mask_len):
    assert min_mask_len > 0
    assert min_mask_len < full_len
    masks = []
    for ind in range(0, full_len, min_mask_len):
        mask = np.full(full_len, False)
        mask[ind:min(ind + min_mask_len, full_len)] = True
        masks.append(mask)
    return masks


def _device_objective_of(hvp_callable):
    """The DeviceObjective behind a bound `Objective.fun_free_hvp`, or None."""
    owner = getattr(hvp_callable, '__self__', None)
    if owner is None or getattr(hvp_callable, '__name__', '') != 'fun_free_hvp':
        return None
    fun = getattr(owner, 'fun', None)
    if getattr(fun, '_lrvb_device_functor', False) and hasattr(fun, 'ctx'):
        return owner, fun
    return None


class ConjugateGradientSolver(object):
    def __init__(self, eval_hessian_vector_product, x0):
        self.dim = len(x0)
        self.ObjHessVecProdLO = LinearOperator(
            (self.dim, self.dim), lambda vec: eval_hessian_vector_product(x0, vec))
        self.x0 = x0
        self.preconditioner = None
        self.tol = 1e-8
        self._device = _device_objective_of(eval_hessian_vector_product)
        self.initialize()

    def initialize(self):
        self.vecs = []
        self.hinv_vecs = []
        self.masks = []
        self.times = []
        self.cg_infos = []

    def get_hinv_vec(self, vec, x0=None):
        if self._device is not None:
            objective, fun = self._device
            fun._push_state()
            minv = None if self.preconditioner is None else np.asarray(self.preconditioner, dtype=np.float64)
            solve = fun.cg_solve if hasattr(fun, 'cg_solve') else fun.ctx.cg_solve
            hinv_vec, cg_info, _ = solve(self.x0, vec, x0=x0, Minv=minv, tol=self.tol)
            objective.par.set_free(self.x0)
            return hinv_vec, cg_info
        return scipy.sparse.linalg.cg(self.ObjHessVecProdLO, vec, x0=x0, rtol=self.tol, atol=0.0,
                                      M=self.preconditioner)

    def get_hinv_vec_subsets(self, vec, masks, verbose=False, print_every=10):
        num_masks = len(masks)
        fun = self._device[1] if self._device is not None else None
        if fun is not None and not hasattr(fun, 'cg_solve') and hasattr(fun.ctx, 'cg_solve_multi') and num_masks > 1:
            # all masked right-hand sides advance together on the device: every CG iteration makes ONE
            # pair of passes over the observations for the whole batch (`lrvb_cg_solve_multi`); each
            # system still follows its own recurrence and stopping rule, so the results are those of the loop
            objective = self._device[0]
            fun._push_state()
            minv = None if self.preconditioner is None else np.asarray(self.preconditioner, dtype=np.float64)
            vec = np.asarray(vec, dtype=np.float64)
            B = np.zeros((num_masks, len(vec)))
            for i, mask in enumerate(masks):
                B[i, mask] = vec[mask]
            cg_time = time.time()
            X, infos, _ = fun.ctx.cg_solve_multi(self.x0, B, Minv=minv, tol=self.tol)
            elapsed = (time.time() - cg_time) / num_masks
            objective.par.set_free(self.x0)
            for i, mask in enumerate(masks):
                self.times.append(elapsed)
                self.vecs.append(B[i].copy())
                self.masks.append(mask)
                self.hinv_vecs.append(X[i].copy())
                self.cg_infos.append(int(infos[i]))
            return
        for ind, mask in enumerate(masks, start=1):
            if verbose and ind % print_every == 0:
                print('{} of {}\n'.format(ind, num_masks))
            vec_masked = np.zeros(len(vec))
            vec_masked[mask] = vec[mask]
            cg_time = time.time()
            hinv_vec, cg_info = self.get_hinv_vec(vec_masked)
            self.times.append(time.time() - cg_time)
            self.vecs.append(vec_masked)
            self.masks.append(mask)
            self.hinv_vecs.append(hinv_vec)
            self.cg_infos.append(cg_info)
