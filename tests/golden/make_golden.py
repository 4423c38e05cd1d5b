#!/usr/bin/env python3
"""Generates tests/golden/reference_vectors.npz by RUNNING the reference's own code, in the build
container only (the reference does not exist on the GPU box; only the .npz travels).

Only the two reference modules that import without the absent third-party `autograd` package are
used -- LinearResponseVariationalBayes/ConjugateGradient.py and OptimizationUtils.py -- loaded by
file path so that the package __init__ (which imports autograd) is not executed.  No stand-in for
autograd is installed.  The optimiser wrappers of OptimizationUtils.py take any object with the
`Objective` method names; they are run on `DuckObjective` (below), whose arithmetic is the repo's
numpy oracle for a small seeded declared model.  One compatibility shim: the reference calls scipy's cg(..., tol=) which
scipy >= 1.14 spells rtol= (legacy `tol` was relative to ||b||, i.e. rtol=tol, atol=0).

Usage: python tests/golden/make_golden.py   (writes next to itself)
"""
import importlib.util
import os
import sys

import numpy as np
import scipy.sparse.linalg

REF = '/root/reference/LinearResponseVariationalBayes'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _QuietLogger(object):
    """What the reference's optimiser wrappers touch on `objective.logger` (OptimizationUtils.py:49-52)."""
    print_every = 1

    def initialize(self):
        pass


class DuckObjective(object):
    """The slice of the reference's `Objective` that its OptimizationUtils functions call, for a small declared model
    whose arithmetic is the repo's numpy oracle: `fun_free`, `fun_free_grad`, `fun_free_hvp`, `fun_free_hessian` and
    the preconditioned family exactly as LRVB/SparseObjectives.py:202-240 defines it (y = A x; A^T grad; A^T H A v;
    uncondition = A x).  The reference's OWN wrapper code (`set_objective_preconditioner`,
    `minimize_objective_trust_ncg`, `minimize_objective_bfgs`, `repeatedly_optimize`) then runs on it unchanged, so
    the fixtures pin that wrapper logic -- start point mapping, which callbacks scipy is handed, the convergence
    rule of the restarts -- for the same model on the device."""

    def __init__(self, model):
        self.model = model
        self.preconditioner = None
        self.preconditioning = False
        self.logger = _QuietLogger()

    def fun_free(self, x, verbose=False):
        return self.model.value(x)

    def fun_free_grad(self, x):
        return self.model.grad(x)

    def fun_free_hessian(self, x):
        return self.model.hessian(x)

    def fun_free_hvp(self, x, v):
        return self.model.hvp(x, v)

    def get_conditioned_x(self, x):
        return self.preconditioner @ x

    def fun_free_cond(self, x, verbose=False):
        assert self.preconditioner is not None
        return self.fun_free(self.get_conditioned_x(x), verbose=verbose)

    def fun_free_grad_cond(self, x):
        assert self.preconditioner is not None
        return self.preconditioner.T @ self.fun_free_grad(self.get_conditioned_x(x))

    def fun_free_hvp_cond(self, x, v):
        assert self.preconditioner is not None
        return self.preconditioner.T @ self.fun_free_hvp(self.get_conditioned_x(x), self.preconditioner @ v)

    def uncondition_x(self, x):
        return self.preconditioner @ x


def optimiser_model(np_rng):
    """Seeded logistic regression with all four kinds of box constraint and a Gaussian prior; returned as plain
    arrays (they travel in the fixture) plus the oracle model."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import packing as opk, models as om
    N = 600
    sizes = (5, 4, 3, 4)                                     # free | lb = 0 | ub = 2 | [-1, 3]
    bounds = ((-np.inf, np.inf), (0.0, np.inf), (-np.inf, 2.0), (-1.0, 3.0))
    P = sum(sizes)
    x = np_rng.normal(size=(N, P)) / np.sqrt(P)
    beta = np.concatenate([np_rng.normal(size=5), np_rng.uniform(0.2, 1.5, size=4), np_rng.uniform(-1.0, 1.5, size=3),
                           np_rng.uniform(-0.5, 2.5, size=4)])
    y = (np_rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-x @ beta))).astype(np.float64)
    w = np_rng.uniform(0.5, 1.5, size=N)
    prior = 0.8
    layout = opk.Layout([opk.box_block(n, lb=lb, ub=ub) for n, (lb, ub) in zip(sizes, bounds)])
    model = om.DeclaredModel(layout, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, prior))
    arrays = dict(opt_x=x, opt_y=y, opt_w=w, opt_sizes=np.array(sizes), opt_bounds=np.array(bounds), opt_prior=np.array(prior))
    return model, arrays


def optimiser_fixtures(opt, out):
    """Runs the reference's OptimizationUtils.py:25-162 on the duck-typed objective and stores what it returned."""
    rng = np.random.default_rng(20241)
    model, arrays = optimiser_model(rng)
    out.update(arrays)
    D = model.layout.D
    x0 = rng.normal(size=D) * 0.3
    out['opt_x0'] = x0
    obj = DuckObjective(model)

    # set_objective_preconditioner (:25-41), eigenvalues clamped from below
    hess, inv_sqrt, corrected = opt.set_objective_preconditioner(obj, free_par=x0, ev_min=0.5)
    out['opt_precond_hessian'], out['opt_precond_inv_sqrt'], out['opt_precond_corrected'] = hess, inv_sqrt, corrected

    # minimize_objective_trust_ncg (:44-75), plain and preconditioned
    x_plain, res = opt.minimize_objective_trust_ncg(obj, x0, False, maxiter=100, gtol=1e-7, disp=False)
    out['opt_tncg_x'], out['opt_tncg_nit'], out['opt_tncg_fun'] = x_plain, np.array(res.nit), np.array(res.fun)
    assert res.success
    x_cond, res = opt.minimize_objective_trust_ncg(obj, x0, True, maxiter=100, gtol=1e-7, disp=False)
    out['opt_tncg_cond_x'], out['opt_tncg_cond_nit'], out['opt_tncg_cond_fun'] = x_cond, np.array(res.nit), np.array(res.fun)
    out['opt_tncg_cond_y'] = res.x                             # the optimum in the optimiser's own coordinates
    assert res.success

    # minimize_objective_bfgs (:78-108), plain and preconditioned
    x_b, res = opt.minimize_objective_bfgs(obj, x0, precondition=False, maxiter=500, disp=False)
    out['opt_bfgs_x'], out['opt_bfgs_fun'] = x_b, np.array(res.fun)
    x_bc, res = opt.minimize_objective_bfgs(obj, x0, precondition=True, maxiter=500, disp=False)
    out['opt_bfgs_cond_x'], out['opt_bfgs_cond_fun'] = x_bc, np.array(res.fun)

    # repeatedly_optimize (:114-162): three-iteration trust-ncg restarts after a short BFGS start
    ret = opt.repeatedly_optimize(
        obj, lambda x: opt.minimize_objective_trust_ncg(obj, x, False, maxiter=3, gtol=1e-8, disp=False), x0,
        initial_optimization_fun=lambda x: opt.minimize_objective_bfgs(obj, x, precondition=False, maxiter=5, disp=False),
        max_iter=50, gtol=1e-7, ftol=1e-12, xtol=1e-10, keep_intermediate_optimizations=True)
    new_x, converged, x_conv, f_conv, grad_conv, last, results = ret
    out['opt_repeat_x'] = new_x
    out['opt_repeat_flags'] = np.array([converged, x_conv, f_conv, grad_conv])
    out['opt_repeat_n_results'] = np.array(len(results))
    out['opt_repeat_nits'] = np.array([r.nit for r in results])


def main():
    cg = load('ConjugateGradient')
    opt = load('OptimizationUtils')
    out = {}

    # masks and splits (ConjugateGradient.py:19-57)
    masks = cg.get_masks(20, 3)
    out['masks_20_3'] = np.array(masks)
    vec = np.zeros(23, dtype=bool)
    vec[[1, 2, 5, 8, 13, 14, 20]] = True
    s1, s2 = cg.split_vector(vec)
    out['split_in'], out['split_1'], out['split_2'] = vec, s1, s2
    res = []
    big = np.zeros(64, dtype=bool)
    big[::2] = True
    cg.recursive_split(big, results=res, terminate_len=5)
    out['rsplit_in'], out['rsplit_out'] = big, np.array(res)

    # get_sym_matrix_inv_sqrt with eigenvalue clamping (OptimizationUtils.py:6-20)
    rng = np.random.default_rng(20240)
    a = rng.normal(size=(6, 6))
    h = a @ a.T + 0.1 * np.eye(6) + 0.01 * rng.normal(size=(6, 6))        # slightly asymmetric on purpose
    out['invsqrt_in'] = h
    for tag, kw in (('plain', {}), ('min', {'ev_min': 1.0}), ('max', {'ev_max': 5.0}), ('both', {'ev_min': 1.0, 'ev_max': 5.0})):
        isq, corr = opt.get_sym_matrix_inv_sqrt(h, **kw)
        out['invsqrt_' + tag], out['invsqrt_corr_' + tag] = isq, corr

    # ConjugateGradientSolver on the test_cg problem (test_objectives.py:524-554), closed-form HVP 2 mat v
    orig_cg = scipy.sparse.linalg.cg

    def cg_compat(A, b, x0=None, tol=1e-5, M=None, **kw):
        return orig_cg(A, b, x0=x0, rtol=tol, atol=0.0, M=M, **kw)
    cg.sp.sparse.linalg.cg = cg_compat
    try:
        K = 50
        mat = rng.random((K, K))
        mat = 0.5 * (mat + mat.T) + 10 * np.eye(K)
        loc = np.array([k / 7. for k in range(K)])
        x = loc + 0.1 * rng.random(K)
        solver = cg.ConjugateGradientSolver(lambda x0, v: 2.0 * (mat @ v), loc)
        cg_masks = cg.get_masks(K, 10)
        solver.get_hinv_vec_subsets(x, cg_masks)
        out['cg_mat'], out['cg_loc'], out['cg_x'] = mat, loc, x
        out['cg_masks'] = np.array(cg_masks)
        out['cg_vecs'] = np.array(solver.vecs)
        out['cg_hinv_vecs'] = np.array(solver.hinv_vecs)
        out['cg_infos'] = np.array(solver.cg_infos)
    finally:
        cg.sp.sparse.linalg.cg = orig_cg

    optimiser_fixtures(opt, out)

    path = os.path.join(HERE, 'reference_vectors.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: np.shape(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
