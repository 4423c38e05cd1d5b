"""Callers of the hot path: preconditioner construction and scipy optimiser wrappers.

Same entry points, arguments and return values as LRVB/OptimizationUtils.py
(get_sym_matrix_inv_sqrt :6-20, set_objective_preconditioner :25-41, minimize_objective_trust_ncg :44-75,
minimize_objective_bfgs :78-108, repeatedly_optimize :114-162).  scipy drives the iterations; what it is
handed are the device-backed Objective methods, so every function / gradient / Hessian-vector evaluation
inside the optimiser is one pass of the HIP path.
"""
import numpy as np
import scipy.optimize


def get_sym_matrix_inv_sqrt(hessian, ev_min=None, ev_max=None):
    """(H^-1/2, H clipped) from the eigendecomposition of the symmetrised matrix, eigenvalues clamped to
    [ev_min, ev_max] where given."""
    lam, Q = np.linalg.eigh((hessian + hessian.T) / 2.0)
    if ev_min is not None or ev_max is not None:
        lam = np.clip(lam, ev_min, ev_max)
    return np.array((Q / np.sqrt(lam)) @ Q.T), np.array((Q * lam) @ Q.T)


def set_objective_preconditioner(objective, free_par=None, hessian=None, ev_min=None, ev_max=None):
    """Installs H^-1/2 as objective.preconditioner; H is given or built at free_par (one device Hessian)."""
    if hessian is None:
        if free_par is None:
            raise ValueError('You must specify either a Hessian or the free_par at which '
                             'the objective\'s Hessian is to be evaluated.')
        hessian = objective.fun_free_hessian(free_par)
    inv_sqrt, clipped = get_sym_matrix_inv_sqrt(hessian, ev_min=ev_min, ev_max=ev_max)
    objective.preconditioner = inv_sqrt
    return hessian, inv_sqrt, clipped


def _minimize(objective, init_x, precondition, method, options, with_hessp, print_every, init_logger):
    """One scipy.optimize.minimize run in plain or preconditioned coordinates; returns (x, scipy result)
    with x mapped back to the objective's own free coordinates."""
    log = objective.logger
    if init_logger:
        log.initialize()
    if print_every is not None:
        log.print_every = print_every
    objective.preconditioning = precondition
    verbose = options['disp']
    if precondition:
        assert objective.preconditioner is not None
        names = ('fun_free_cond', 'fun_free_grad_cond', 'fun_free_hvp_cond')
        start = np.linalg.solve(objective.preconditioner, init_x)
        back = objective.uncondition_x
    else:
        names = ('fun_free', 'fun_free_grad', 'fun_free_hvp')
        start = init_x
        back = lambda z: z                                   # noqa: E731
    value = getattr(objective, names[0])
    extra = {'hessp': getattr(objective, names[2])} if with_hessp else {}
    res = scipy.optimize.minimize(lambda z: value(z, verbose=verbose), x0=start, jac=getattr(objective, names[1]),
                                  method=method, options=options, **extra)
    return back(res.x), res


_TRUST_NCG_MESSAGES = ('Optimization terminated successfully.',
                       'Maximum number of iterations has been exceeded.',
                       'A bad approximation caused failure to predict improvement.')


def _trust_ncg_on_device(objective, init_x, precondition, maxiter, gtol, disp):
    """The same optimiser as one library call (lrvb_minimize_trust_ncg): outer trust-region loop and Steihaug CG on
    the device, one curvature pass per accepted point.  Needs a device functor without extra arguments."""
    fun = objective.fun
    ctx = getattr(fun, 'ctx', None)
    if ctx is None or not hasattr(ctx, 'minimize_trust_ncg') or not hasattr(fun, '_push_state'):
        raise NotImplementedError('on_device=True needs an objective built on a device functor (DeviceObjective, '
                                  'GLMObjective, QuadraticObjective); this one is evaluated through host callbacks')
    if getattr(fun, 'scale_fun', None) is not None:
        raise NotImplementedError('on_device=True takes no extra objective arguments (this functor declares scale_fun)')
    A = None
    start = np.asarray(init_x, dtype=np.float64)
    if precondition:
        assert objective.preconditioner is not None
        A = np.asarray(objective.preconditioner, dtype=np.float64)
        start = np.linalg.solve(A, start)
    fun._push_state()                                        # weights / data the functor keeps on the host side
    y, x, info = ctx.minimize_trust_ncg(start, precond=A, gtol=gtol, maxiter=maxiter)
    objective.par.set_free(x)                                # side effect of every evaluation: par sits at the last point
    res = scipy.optimize.OptimizeResult(
        x=y, fun=info['fun'], status=info['status'], success=info['status'] == 0,
        message=_TRUST_NCG_MESSAGES[info['status']], nit=info['nit'], nfev=info['nfev'], njev=info['njev'],
        nhev=info['nhev'], nbuild=info.get('nbuild', 0), jac_mag=info['jac_mag'], trust_radius=info['trust_radius'])
    if disp:
        print('{}\n         Current function value: {:f}\n         Iterations: {:d}\n         Function evaluations: {:d}'
              '\n         Gradient evaluations: {:d}\n         Hessian evaluations: {:d}'.format(
                  res.message, res.fun, res.nit, res.nfev, res.njev, res.nhev))
    return x, res


def minimize_objective_trust_ncg(objective, init_x, precondition, maxiter=50, gtol=1e-6, disp=True,
                                 print_every=None, init_logger=True, on_device=False):
    """on_device=False: scipy drives, every callback is a device evaluation (the reference's structure).
    on_device=True: the whole optimisation is one call into the HIP library."""
    if on_device:
        if init_logger:
            objective.logger.initialize()
        objective.preconditioning = precondition
        return _trust_ncg_on_device(objective, init_x, precondition, maxiter, gtol, disp)
    return _minimize(objective, init_x, precondition, 'trust-ncg', {'maxiter': maxiter, 'gtol': gtol, 'disp': disp},
                     True, print_every, init_logger)


def minimize_objective_bfgs(objective, init_x, precondition=False, maxiter=500, disp=True,
                            print_every=None, init_logger=True):
    return _minimize(objective, init_x, precondition, 'BFGS', {'maxiter': maxiter, 'disp': disp},
                     False, print_every, init_logger)


def repeatedly_optimize(objective, optimization_fun, init_x, initial_optimization_fun=None,
                        max_iter=100, gtol=1e-8, ftol=1e-8, xtol=1e-8, disp=False,
                        keep_intermediate_optimizations=False):
    """Restarts optimization_fun from its own answer until the step, the decrease or the gradient is small (L1
    norms, absolute).  Returns (x, converged, x_conv, f_conv, grad_conv, last scipy result, kept results)."""
    kept = []

    def remember(res):
        if keep_intermediate_optimizations:
            kept.append(res)

    x = init_x
    if initial_optimization_fun is not None:
        if disp:
            print('Running intitial optimization.')
        x, res0 = initial_optimization_fun(x)
        remember(res0)
    f_here = objective.fun_free(x)
    flags = (False, False, False)
    last = None
    for it in range(1, max_iter + 1):
        if disp:
            print('\n---------------------------------\nRepeated optimization iteration ', it - 1)
        x_next, last = optimization_fun(x)
        remember(last)
        f_next = objective.fun_free(x_next)
        step = np.sum(np.abs(x_next - x))
        drop = np.abs(f_next - f_here)
        slope = np.sum(np.abs(objective.fun_free_grad(x_next)))
        flags = (step < xtol, drop < ftol, slope < gtol)
        x, f_here = x_next, f_next
        if disp:
            print('Iter {}: x_diff = {}, f_diff = {}, grad_l1 = {}'.format(it, step, drop, slope))
        if any(flags):
            break
    return (x, any(flags)) + flags + (last, kept)
