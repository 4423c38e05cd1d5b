"""Rebuilds, from the arrays stored in tests/golden/reference_vectors.npz, the problems that tests/golden/make_golden.py
ran the reference's own code on -- for the oracle (numpy) and for the product's parameter classes."""
import os

import numpy as np

from oracle import packing as opk, models as om

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors.npz'))


def optimiser_problem(vb):
    """The seeded logistic regression of make_golden.optimiser_model: (par, oracle layout, oracle model, arrays)."""
    sizes, bounds = G['opt_sizes'], G['opt_bounds']
    par = vb.ModelParamsDict('par')
    blocks = []
    for i, (n, (lb, ub)) in enumerate(zip(sizes, bounds)):
        par.push_param(vb.VectorParam('b{}'.format(i), int(n), lb=float(lb), ub=float(ub)))
        blocks.append(opk.box_block(int(n), lb=float(lb), ub=float(ub)))
    lay = opk.Layout(blocks)
    P = int(sizes.sum())
    prior = float(G['opt_prior'])
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=G['opt_x'], y=G['opt_y'], w=G['opt_w'], quad_A=np.full(P, prior))
    return par, lay, model, dict(x=G['opt_x'], y=G['opt_y'], w=G['opt_w'], prior=prior, P=P)


def check_optimiser_wrappers(vb, objective, lay, on_device_too):
    """Runs the package's OptimizationUtils on `objective` exactly as make_golden.py ran the reference's, and compares
    with what the reference's functions returned (LRVB/OptimizationUtils.py:25-162)."""
    opt = vb.OptimizationUtils
    x0 = G['opt_x0']

    def same_point(x, want, tol):
        assert np.max(np.abs(lay.constrain(x) - lay.constrain(want))) < tol

    hess, inv_sqrt, corrected = opt.set_objective_preconditioner(objective, free_par=x0, ev_min=0.5)
    assert np.max(np.abs(hess - G['opt_precond_hessian'])) < 1e-11 * np.max(np.abs(hess))
    assert np.max(np.abs(inv_sqrt - G['opt_precond_inv_sqrt'])) < 1e-10 * np.max(np.abs(inv_sqrt))
    assert np.max(np.abs(corrected - G['opt_precond_corrected'])) < 1e-10 * np.max(np.abs(corrected))
    assert np.array_equal(objective.preconditioner, inv_sqrt)

    routes = [False, True] if on_device_too else [False]
    for on_device in routes:
        kw = {'on_device': True} if on_device else {}
        x, res = opt.minimize_objective_trust_ncg(objective, x0, False, maxiter=100, gtol=1e-7, disp=False, **kw)
        assert res.success and res.nit == int(G['opt_tncg_nit'])
        assert abs(res.fun - float(G['opt_tncg_fun'])) < 1e-11 * abs(res.fun)
        same_point(x, G['opt_tncg_x'], 1e-8)
        x, res = opt.minimize_objective_trust_ncg(objective, x0, True, maxiter=100, gtol=1e-7, disp=False, **kw)
        assert res.success and res.nit == int(G['opt_tncg_cond_nit'])
        assert abs(res.fun - float(G['opt_tncg_cond_fun'])) < 1e-11 * abs(res.fun)
        same_point(x, G['opt_tncg_cond_x'], 1e-8)
        # res.x lives in the optimiser's coordinates; mapped back it is the same point (uncondition_x)
        same_point(objective.uncondition_x(res.x), G['opt_tncg_cond_x'], 1e-8)

    # BFGS paths amplify rounding differences between gradient implementations: the optimum, not the path
    x, res = opt.minimize_objective_bfgs(objective, x0, precondition=False, maxiter=500, disp=False)
    assert abs(res.fun - float(G['opt_bfgs_fun'])) < 1e-8 * abs(res.fun)
    same_point(x, G['opt_bfgs_x'], 1e-4)
    x, res = opt.minimize_objective_bfgs(objective, x0, precondition=True, maxiter=500, disp=False)
    assert abs(res.fun - float(G['opt_bfgs_cond_fun'])) < 1e-8 * abs(res.fun)
    same_point(x, G['opt_bfgs_cond_x'], 1e-4)

    ret = opt.repeatedly_optimize(
        objective, lambda z: opt.minimize_objective_trust_ncg(objective, z, False, maxiter=3, gtol=1e-8, disp=False), x0,
        initial_optimization_fun=lambda z: opt.minimize_objective_bfgs(objective, z, precondition=False, maxiter=5, disp=False),
        max_iter=50, gtol=1e-7, ftol=1e-12, xtol=1e-10, keep_intermediate_optimizations=True)
    new_x, converged, x_conv, f_conv, grad_conv, last, results = ret
    assert converged == bool(G['opt_repeat_flags'][0])
    same_point(new_x, G['opt_repeat_x'], 1e-7)
    # the restarts stop by the same rule: as many optimiser runs as the reference made, give or take the last one
    # (its step is at the rounding level of the x / f tolerances)
    assert abs(len(results) - int(G['opt_repeat_n_results'])) <= 1
    assert [r.nit for r in results][:4] == list(G['opt_repeat_nits'][:4])
