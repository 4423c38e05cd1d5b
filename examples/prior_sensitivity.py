"""Prior sensitivity by linear response -- the defining use of `HyperparameterSensitivityLinearApproximation`
(`ParametricSensitivityLinearApproximation` in this reference checkout, LRVB/ModelSensitivity.py:555-612) -- on the MI355X path:
fit a conjugate-normal regression (BASELINE.json configuration 2: q(beta) = MVNParam, q(tau) = GammaParam), then ask how the
fitted variational parameters move when the PRIOR moves, without refitting:

    d theta_hat / d prior^T = -H^-1  d2 KL / d theta d prior^T

The Hessian build, the cross Hessian with the prior (a closed form, J^T applied on the device), the Cholesky and the solve run on
the GPU; the prediction is compared with an actual refit.  Runs on one GPU:

    python -c "import __graft_entry__ as g; g.build()"
    python examples/prior_sensitivity.py [k] [N]
"""
import os
import sys
import time

import numpy as np
import scipy.optimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrvb_amd as vb                                               # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100000
rng = np.random.default_rng(3)
x = rng.normal(size=(N, k))
y = x @ rng.normal(size=k) + rng.normal(size=N) / np.sqrt(2.0)

par = vb.ModelParamsDict('params')
par.push_param(vb.MVNParam('beta', dim=k))
par.push_param(vb.GammaParam('tau'))
prior_mean, prior_info = np.zeros(k), 50.0 * np.eye(k)            # an informative prior, so that it matters
fun = vb.MVNRegressionObjective(par, x, y, prior_mean=prior_mean, prior_info=prior_info, prior_shape=2.0, prior_rate=1.5)
objective = vb.Objective(par, fun)
print('free parameters D = %d, observations N = %d; declared hyper-parameters: %s' % (par.free_size(), N, ', '.join(fun.hyper_pars)))


def fit(start):
    res = scipy.optimize.minimize(objective.fun_free, start, jac=objective.fun_free_grad, hess=objective.fun_free_hessian,
                                  method='trust-exact', options={'gtol': 1e-8})
    theta = res.x
    for _ in range(2):                                               # Newton polish: stationary to rounding
        theta = theta - np.linalg.solve(objective.fun_free_hessian(theta), objective.fun_free_grad(theta))
    return theta


mean, info = vb.regression_utils.get_posterior_regression_coefficients(y, x, 2.0, prior_mean, prior_info)
par['beta']['mean'].set(mean)
par['beta']['info'].set(0.5 * (info + info.T))
par['tau']['shape'].set(np.array(2.0 + 0.5 * N))
par['tau']['rate'].set(np.array(1.5 + 0.25 * N))
t0 = time.perf_counter()
theta0 = fit(par.get_free())
print('fit: %.0f ms, |grad| = %.1e' % ((time.perf_counter() - t0) * 1e3, np.max(np.abs(objective.fun_free_grad(theta0)))))

t0 = time.perf_counter()
sens = vb.HyperparameterSensitivityLinearApproximation(
    objective_functor=fun, input_par=par, hyper_par=fun.prior_mean_par,
    input_val0=theta0, hyper_val0=fun.prior_mean_par.get_vector())
print('sensitivity to the prior mean (%d x %d): %.1f ms (Hessian build + cross Hessian + Cholesky + solve)'
      % (sens.get_dinput_dhyper().shape + ((time.perf_counter() - t0) * 1e3,)))

# move the prior mean, predict, refit
new_prior_mean = prior_mean + 0.5 * rng.normal(size=k)
predicted = sens.predict_input_par_from_hyperparameters(new_prior_mean)
fun.prior_mean_par.set_vector(new_prior_mean)
refit = fit(theta0)
par.set_free(theta0)
m0 = par['beta']['mean'].get().copy()
par.set_free(predicted)
m_pred = par['beta']['mean'].get().copy()
par.set_free(refit)
m_refit = par['beta']['mean'].get().copy()
print('posterior mean of beta[0..3] at the base prior :', np.round(m0[:4], 5))
print('  predicted under the new prior (no refit)      :', np.round(m_pred[:4], 5))
print('  refitted under the new prior                  :', np.round(m_refit[:4], 5))
print('relative error of the prediction: %.2e of the move' % (np.linalg.norm(predicted - refit) / np.linalg.norm(refit - theta0)))
