"""The walk-through of the reference's Example.ipynb (cells 3-18) on the MI355X path: a multivariate normal
regression y_n ~ N(x_n beta, Lambda^-1), fitted by trust-region Newton-CG, followed by the linear-response
covariance of beta and the sensitivity of beta to every observation's weight -- which predicts what happens when
an observation is left out.  Runs on one GPU:

    python -c "import __graft_entry__ as g; g.build()"
    python examples/normal_model.py [d] [N]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrvb_amd as vb                                               # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(42)

# ---- data (Example.ipynb cell 3) -------------------------------------------------------------------------------
true_beta = np.exp(rng.random((d, d)))
true_lambda = np.eye(d) + np.full((d, d), 0.5)
x = rng.random((N, d))
y = x @ true_beta + rng.multivariate_normal(np.zeros(d), np.linalg.inv(true_lambda), size=N)

# ---- parameters: same classes, same free-vector layout as the reference (cells 5-8) -----------------------------
par = vb.ModelParamsDict('params')
par.push_param(vb.ArrayParam(name='beta', shape=(d, d), lb=0.))
par.push_param(vb.PosDefMatrixParam('lambda', size=d))
print('free parameters:', par.free_size())

# ---- objective: the notebook's closure, declared instead of traced (cell 9) -----------------------------------
fun = vb.NormalRegressionObjective(par, x, y)              # data go to the GPU once; weights are a parameter
objective = vb.Objective(par, fun)                         # the reference's Objective: fun_free, _grad, _hvp, _hessian

# ---- fit (cell 12) -------------------------------------------------------------------------------------------
t0 = time.perf_counter()
opt_free, result = vb.OptimizationUtils.minimize_objective_trust_ncg(
    objective, par.get_free(), precondition=False, gtol=1e-8, maxiter=100, disp=False)
par.set_free(opt_free)
print('fit: %d iterations, %.1f ms;  beta =\n%s' % (result.nit, 1e3 * (time.perf_counter() - t0), par['beta'].get()))
print('truth:\n%s' % true_beta)

# ---- linear response (cells 15-16) ---------------------------------------------------------------------------
summary = vb.LinearMoments(par, select='beta')             # the quantity of interest and its free Jacobian
summary_jac = vb.Objective(par, summary).fun_free_jacobian(opt_free)
linresp = vb.ParametricSensitivityLinearApproximation(fun, par, fun.weights_par, opt_free, np.ones(N))
cov = linresp.get_lrvb_cov(summary_jac)                    # summary_jac H^-1 summary_jac^T: Hessian + Cholesky on the GPU
print('LRVB standard deviations of beta:', np.sqrt(np.diag(cov)))
weight_sens = (summary_jac @ linresp.get_dinput_dhyper()).T    # N x |beta|: d beta / d w_n for every observation

# ---- leave one observation out: prediction against an actual refit (cells 17-18) -------------------------------
row = 25
w = np.ones(N)
w[row] = 0.0
fun.weights_par.set_vector(w)
refit_free, _ = vb.OptimizationUtils.minimize_objective_trust_ncg(objective, opt_free, False, gtol=1e-10, disp=False)


def beta_at(free):
    par.set_free(free)
    return par['beta'].get_vector().copy()


actual = beta_at(opt_free) - beta_at(refit_free)
print('leaving out observation %d changes beta by\n  actual    %s\n  predicted %s' % (row, actual, weight_sens[row]))
