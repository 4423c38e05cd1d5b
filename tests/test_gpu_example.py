"""BASELINE.json config 1 on the GPU: the reference's Example.ipynb replayed end to end through
the drop-in surface -- objective, trust-ncg fit, summary sensitivity operator, LRVB covariance,
weight cross Hessian, and the leave-one-out prediction the notebook verifies (cells 9-18).
Both sizes SURVEY.md section 8(d) names: d = 1, N = 1000 (D = 2) and d = 2, N = 10000 (D = 7).
Oracle: exact AD (torch.func, fp64) of the restated closure (oracle/example_model.py)."""
import numpy as np
import pytest
import scipy.optimize
import torch

import torch_ref as tr
from oracle import example_model as oex
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _data(rng, d, N):
    """Example.ipynb:44-53."""
    true_beta = np.exp(rng.random((d, d)))
    true_lambda = np.eye(d) + np.full((d, d), 0.5)
    x = rng.random((N, d))
    y = x @ true_beta + rng.multivariate_normal(np.zeros(d), np.linalg.inv(true_lambda), size=N)
    return x, y


@pytest.mark.parametrize('d,N', [(1, 1000), (2, 10000)])
def test_example_notebook_flow(vb, d, N):
    rng = np.random.default_rng(20240 + d)
    x, y = _data(rng, d, N)
    beta = vb.ArrayParam(name='beta', shape=(d, d), lb=0.)
    lamb = vb.PosDefMatrixParam('lambda', size=d)
    par = vb.ModelParamsDict('params')
    par.push_param(beta); par.push_param(lamb)
    D = par.free_size()
    assert D == d * d + d * (d + 1) // 2
    fun = vb.NormalRegressionObjective(par, x, y)
    objective = vb.Objective(par, fun)
    lay = oex.layout(d, d)
    ft = tr.example_objective(lay, x, y)
    w1 = torch.ones(N, dtype=torch.float64)

    # derivatives at a generic point, weights != 1
    theta = rng.normal(size=D) * 0.3
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)
    tt, tw = torch.tensor(theta), torch.tensor(w)
    H_ad = torch.func.hessian(ft)(tt, tw).numpy()
    assert abs(objective.fun_free(theta) - ft(tt, tw).item()) < 1e-11 * abs(ft(tt, tw).item())
    assert rel_err(objective.fun_free_grad(theta), torch.func.grad(ft)(tt, tw).numpy()) < 1e-10
    assert rel_err(objective.fun_free_hessian(theta), H_ad) < 1e-10
    v = rng.normal(size=D)
    assert rel_err(objective.fun_free_hvp(theta, v), H_ad @ v) < 1e-10
    cross_ad = torch.func.jacrev(torch.func.grad(ft, argnums=0), argnums=1)(tt, tw).numpy()     # D x N
    two = vb.TwoParameterObjective(par, fun.weights_par, fun)
    assert rel_err(two.fun_hessian_free1_vector2(theta, w), cross_ad) < 1e-10

    # fit (cell 12): Newton trust region on the device-backed callables
    fun.weights_par.set_vector(np.ones(N))
    init = par.get_free() * 0.0 + np.asarray(vb.ModelParamsDict.get_free(par))
    opt = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp,
                                  x0=init, method='trust-ncg', options={'gtol': 1e-8})
    opt_free = opt.x
    assert np.max(np.abs(objective.fun_free_grad(opt_free))) < 1e-5 * N

    # summary = beta, its Jacobian, sensitivity operator and LRVB covariance (cell 15)
    summary = vb.LinearMoments(par, select='beta')
    summary_jac = vb.Objective(par, summary).fun_free_jacobian(opt_free)
    H = objective.fun_free_hessian(opt_free)
    H_ad = torch.func.hessian(ft)(torch.tensor(opt_free), w1).numpy()
    assert rel_err(H, H_ad) < 1e-10
    sens_op = -np.linalg.solve(H_ad, summary_jac.T)
    linresp = vb.ParametricSensitivityLinearApproximation(fun, par, fun.weights_par, opt_free, np.ones(N), hess0=H)
    assert rel_err(linresp.get_lrvb_cov(summary_jac), summary_jac @ np.linalg.solve(H_ad, summary_jac.T)) < 1e-8

    # weight sensitivity (cell 16): weight_sens = par_weight_hess^T @ summary_sens_operator  (N x |beta|)
    par_weight_hess = two.fun_hessian_free1_vector2(opt_free, np.ones(N))                  # D x N
    weight_sens = par_weight_hess.T @ sens_op
    assert rel_err(summary_jac @ linresp.get_dinput_dhyper(), weight_sens.T) < 1e-8

    # leave one observation out and refit (cells 17-18): prediction matches the actual change
    row = 25
    wl = np.ones(N); wl[row] = 0.0
    fun.weights_par.set_vector(wl)
    refit = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp,
                                    x0=opt_free, method='trust-ncg', options={'gtol': 1e-10})
    def beta_of(free):
        par.set_free(free)
        return par['beta'].get_vector().copy()
    actual = beta_of(opt_free) - beta_of(refit.x)
    predicted = weight_sens[row]
    assert np.linalg.norm(actual - predicted) < 0.02 * np.linalg.norm(actual) + 1e-9
