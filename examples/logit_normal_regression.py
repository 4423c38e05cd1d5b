"""Logistic regression with a Gaussian variational posterior -- the model the reference's `Modeling.get_e_logistic_term_guass_hermite`
is written for -- on the MI355X path: fit, linear-response covariance (which corrects the mean-field variances), and the
sensitivity of the posterior means to every observation's weight.  Runs on one GPU:

    python -c "import __graft_entry__ as g; g.build()"
    python examples/logit_normal_regression.py [P] [N]
"""
import os
import sys
import time

import numpy as np
import scipy.optimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrvb_amd as vb                                               # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(7)
x = rng.normal(size=(N, P)) / np.sqrt(P)
true_beta = rng.normal(size=P) * 2.0
y = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-x @ true_beta))).astype(np.float64)

# q(beta_j) = N(mean_j, 1 / info_j): the reference's UVNParamVector, same free-vector layout
par = vb.ModelParamsDict('params')
par.push_param(vb.UVNParamVector('beta', length=P))
fun = vb.LogitNormalRegressionObjective(par, x, y, prior_info=0.1, gh_deg=20)      # X, y go to the GPU once
objective = vb.Objective(par, fun)

t0 = time.perf_counter()
opt = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp,
                              x0=np.zeros(2 * P), method='trust-ncg', options={'gtol': 1e-6})
t1 = time.perf_counter()
theta = opt.x
par.set_free(theta)
print('fit: %d iterations, %.0f ms, |grad| = %.1e' % (opt.nit, (t1 - t0) * 1e3, np.max(np.abs(objective.fun_free_grad(theta)))))
print('posterior means      :', np.round(par['beta']['mean'].get(), 3))
print('true coefficients    :', np.round(true_beta, 3))

# linear-response covariance of the means: M H^-1 M^T (device Cholesky), against the mean-field variances 1 / info
H = objective.fun_free_hessian(theta)
fun.ctx.chol_factor(H)
M = np.eye(2 * P)[:P]
cov = fun.ctx.lrvb_cov(M)
print('mean-field sd        :', np.round(1.0 / np.sqrt(par['beta']['info'].get()), 4))
print('linear-response sd   :', np.round(np.sqrt(np.diag(cov)), 4))

# weight sensitivity d mean / d w_n for every observation; predicts the effect of leaving one out
sens = vb.ParametricSensitivityLinearApproximation(fun, par, fun.weights_par, theta, np.ones(N))
dmean_dw = sens.get_dinput_dhyper()[:P]                     # P x N (the mean block is unconstrained: free = value)
n_out = int(np.argmax(np.abs(dmean_dw).sum(axis=0)))        # the most influential observation
w = np.ones(N); w[n_out] = 0.0
fun.weights_par.set_vector(w)
opt2 = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp, x0=theta,
                               method='trust-ncg', options={'gtol': 1e-7})
print('leave out observation %d:' % n_out)
print('  predicted change of the means:', np.round(-dmean_dw[:, n_out], 6))
print('  refit                        :', np.round(opt2.x[:P] - theta[:P], 6))
