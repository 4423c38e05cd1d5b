"""Callers of the hot path: preconditioner construction and scipy optimiser wrappers.

Drop-in for LRVB/OptimizationUtils.py: get_sym_matrix_inv_sqrt :6-20,
set_objective_preconditioner :25-41, minimize_objective_trust_ncg :44-75,
minimize_objective_bfgs :78-108, repeatedly_optimize :114-162.  scipy drives the iterations; the
value / gradient / Hessian-vector callables it is handed are the device-backed Objective methods.
"""
import numpy as np
import scipy.optimize


def get_sym_matrix_inv_sqrt(hessian, ev_min=None, ev_max=None):
    hessian_sym = 0.5 * (hessian + hessian.T)
    eig_val, eig_vec = np.linalg.eigh(hessian_sym)
    if ev_min is not None:
        eig_val = np.where(eig_val <= ev_min, ev_min, eig_val)
    if ev_max is not None:
        eig_val = np.where(eig_val >= ev_max, ev_max, eig_val)
    hess_corrected = (eig_vec * eig_val) @ eig_vec.T
    hess_inv_sqrt = (eig_vec / np.sqrt(eig_val)) @ eig_vec.T
    return np.array(hess_inv_sqrt), np.array(hess_corrected)


def set_objective_preconditioner(objective, free_par=None, hessian=None, ev_min=None, ev_max=None):
    if free_par is None and hessian is None:
        raise ValueError('You must specify either a Hessian or the free_par at which '
                         'the objective\'s Hessian is to be evaluated.')
    if hessian is None:
        hessian = objective.fun_free_hessian(free_par)
    inv_hess_sqrt, hessian_corrected = get_sym_matrix_inv_sqrt(hessian, ev_min=ev_min, ev_max=ev_max)
    objective.preconditioner = inv_hess_sqrt
    return hessian, inv_hess_sqrt, hessian_corrected


def _prepare(objective, precondition, print_every, init_logger):
    if init_logger:
        objective.logger.initialize()
    if print_every is not None:
        objective.logger.print_every = print_every
    objective.preconditioning = precondition
    if precondition:
        assert objective.preconditioner is not None


def minimize_objective_trust_ncg(objective, init_x, precondition, maxiter=50, gtol=1e-6, disp=True,
                                 print_every=None, init_logger=True):
    _prepare(objective, precondition, print_every, init_logger)
    options = {'maxiter': maxiter, 'gtol': gtol, 'disp': disp}
    if precondition:
        obj_opt = scipy.optimize.minimize(
            lambda par: objective.fun_free_cond(par, verbose=disp),
            x0=np.linalg.solve(objective.preconditioner, init_x),
            jac=objective.fun_free_grad_cond, hessp=objective.fun_free_hvp_cond,
            method='trust-ncg', options=options)
        return objective.uncondition_x(obj_opt.x), obj_opt
    obj_opt = scipy.optimize.minimize(
        lambda par: objective.fun_free(par, verbose=disp), x0=init_x,
        jac=objective.fun_free_grad, hessp=objective.fun_free_hvp,
        method='trust-ncg', options=options)
    return obj_opt.x, obj_opt


def minimize_objective_bfgs(objective, init_x, precondition=False, maxiter=500, disp=True,
                            print_every=None, init_logger=True):
    _prepare(objective, precondition, print_every, init_logger)
    options = {'maxiter': maxiter, 'disp': disp}
    if precondition:
        obj_opt = scipy.optimize.minimize(
            lambda par: objective.fun_free_cond(par, verbose=disp),
            x0=np.linalg.solve(objective.preconditioner, init_x),
            jac=objective.fun_free_grad_cond, method='BFGS', options=options)
        return objective.uncondition_x(obj_opt.x), obj_opt
    obj_opt = scipy.optimize.minimize(
        lambda par: objective.fun_free(par, verbose=disp), x0=init_x,
        jac=objective.fun_free_grad, method='BFGS', options=options)
    return obj_opt.x, obj_opt


def repeatedly_optimize(objective, optimization_fun, init_x, initial_optimization_fun=None,
                        max_iter=100, gtol=1e-8, ftol=1e-8, xtol=1e-8, disp=False,
                        keep_intermediate_optimizations=False):
    opt_results = []
    if initial_optimization_fun is not None:
        if disp:
            print('Running intitial optimization.')
        init_x, init_opt = initial_optimization_fun(init_x)
        if keep_intermediate_optimizations:
            opt_results.append(init_opt)
    converged = x_conv = f_conv = grad_conv = False
    f_val = objective.fun_free(init_x)
    x = new_x = init_x
    obj_opt = None
    i = 0
    while i < max_iter and not converged:
        if disp:
            print('\n---------------------------------\nRepeated optimization iteration ', i)
        i += 1
        new_x, obj_opt = optimization_fun(x)
        if keep_intermediate_optimizations:
            opt_results.append(obj_opt)
        new_f_val = objective.fun_free(new_x)
        grad_val = objective.fun_free_grad(new_x)
        x_diff = np.sum(np.abs(new_x - x))
        f_diff = np.abs(new_f_val - f_val)
        grad_l1 = np.sum(np.abs(grad_val))
        x_conv, f_conv, grad_conv = x_diff < xtol, f_diff < ftol, grad_l1 < gtol
        x, f_val = new_x, new_f_val
        converged = x_conv or f_conv or grad_conv
        if disp:
            print('Iter {}: x_diff = {}, f_diff = {}, grad_l1 = {}'.format(i, x_diff, f_diff, grad_l1))
    return new_x, converged, x_conv, f_conv, grad_conv, obj_opt, opt_results
