"""Hyper-parameters beyond weights and tilt on the GPU: the reference's
`TwoParameterObjective` and `ParametricSensitivityLinearApproximation` take ANY hyper_par (LRVB/SparseObjectives.py:321-449,
LRVB/ModelSensitivity.py:555-612) and the defining use is prior sensitivity.  Here: every hyper-parameter a declared
objective lists in `hyper_pars` -- prior mean / information / scale and likelihood precision of `DeviceObjective`, the priors
of the MVN-regression, Wishart, hierarchical and mixture models -- against the oracle / exact AD, refits predicted to second
order in the step at D >= 254, the Taylor class on the same hyper-parameters, and the reference's own plain-closure test of
the linear approximation.  Tolerances: 1e-11 relative for closed forms without solves, 1e-9 through solves."""
import numpy as np
import pytest
import torch

import torch_ref as tr
from oracle import packing as opk
from oracle import models as om
from helpers import make_par, glm_data, rel_err
from oracle_functor import OracleFunctor

pytestmark = pytest.mark.gpu

KINDS = ('tilt', 'prior_mean', 'prior_info', 'quad_scale', 'lik_info')


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _declared(vb, rng, spec, N, dense, glm_name, P):
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.GAUSSIAN, scale=0.5)
    V = lay.V
    a = rng.normal(size=(V, V))
    A = a @ a.T / V + np.eye(V)
    qA = A if dense else np.diag(A).copy()
    qm, qb = lay.constrain(rng.normal(size=lay.D) * 0.3), rng.normal(size=V) * 0.05      # a feasible centre: the optimum is interior
    fun = vb.DeviceObjective(par, x=x, y=y, loss='gaussian', glm_param=glm_name, lik_info=1.7, quad_A=qA, quad_m=qm, quad_b=qb,
                             weights=w)
    fun.quad_scale_par.set_vector(np.array([0.6]))
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, w=w, glm_off=par.vector_indices_dict[glm_name].start, lik_info=1.7,
                             quad_A=qA, quad_m=qm, quad_b=qb, quad_scale=0.6)
    return par, lay, fun, model


@pytest.mark.parametrize('dense', [False, True])
@pytest.mark.parametrize('layout', ['box', 'mixed'])
def test_declared_hyper_cross_hessians_and_gradients_match_oracle(vb, layout, dense):
    rng = np.random.default_rng(3 + dense)
    if layout == 'box':
        spec = [('box', 'free', 7, -np.inf, np.inf), ('box', 'beta', 9, 0.0, np.inf), ('box', 'ub', 4, -np.inf, 2.0), ('box', 'bb', 5, -1.0, 3.0)]
        glm_name, P = 'beta', 9
    else:
        spec = [('box', 'beta', 6, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3), ('box', 'u', 2, -np.inf, np.inf)]
        glm_name, P = 'beta', 6
    par, lay, fun, model = _declared(vb, rng, spec, 300, dense, glm_name, P)
    assert set(KINDS) | {'weights'} == set(fun.hyper_pars)
    theta = rng.normal(size=lay.D) * 0.4
    eta = lay.constrain(theta)
    for kind in KINDS:
        hp = fun.hyper_pars[kind]
        h0 = hp.get_vector().copy()
        np.testing.assert_allclose(h0, model.hyper_value(kind), rtol=1e-15)
        two = vb.TwoParameterObjective(par, hp, fun)
        C = model.cross_hessian_hyper(kind, theta)
        Cv = model.cross_hessian_hyper_vec(kind, eta)
        assert rel_err(two.fun_hessian_free1_vector2(theta, h0), C) < 1e-11, kind
        assert rel_err(two.fun_vector_hessian12(eta, h0), Cv) < 1e-11, kind
        assert rel_err(two.fun_vector_hessian21(eta, h0), Cv.T) < 1e-11, kind
        assert rel_err(two.fun_grad2(theta, h0, True, False), model.hyper_grad(kind, theta)) < 1e-11, kind
        assert rel_err(two.fun_grad2(eta, h0, False, False), model.hyper_grad_vec(kind, eta)) < 1e-11, kind
        assert rel_err(two.fun_grad1(theta, h0, True, False), model.grad(theta)) < 1e-11, kind
        np.testing.assert_allclose(hp.get_vector(), h0, rtol=0, atol=0)         # left at the evaluation point
        np.testing.assert_allclose(par.get_free(), theta, rtol=1e-12, atol=1e-13)


def test_hyper_values_reach_the_device_and_free_hypers_chain(vb):
    """A new value of a hyper-parameter object is used by the next evaluation (value, gradient, Hessian equal the oracle with the
    new value); a lower-bounded hyper-parameter in FREE coordinates chains through its own packing Jacobian."""
    rng = np.random.default_rng(11)
    spec = [('box', 'a', 5, -np.inf, np.inf), ('box', 'beta', 6, 0.0, np.inf)]
    par, lay, fun, model = _declared(vb, rng, spec, 200, False, 'beta', 6)
    objective = vb.Objective(par, fun)
    theta = rng.normal(size=lay.D) * 0.3
    new = {'prior_mean': rng.normal(size=lay.V), 'prior_info': rng.uniform(0.5, 2.0, lay.V), 'quad_scale': np.array([1.3]),
           'lik_info': np.array([0.45]), 'tilt': rng.normal(size=lay.V)}
    for kind, val in new.items():
        fun.hyper_pars[kind].set_vector(val)
        model.set_hyper(kind, val)
        assert abs(objective.fun_free(theta) - model.value(theta)) < 1e-12 * abs(model.value(theta)), kind
        assert rel_err(objective.fun_free_grad(theta), model.grad(theta)) < 1e-11, kind
        assert rel_err(objective.fun_free_hessian(theta), model.hessian(theta)) < 1e-11, kind
        v = rng.normal(size=lay.D)
        assert rel_err(objective.fun_free_hvp(theta, v), model.hessian(theta) @ v) < 1e-11, kind
    # free coordinates of a lower-bounded hyper-parameter: replace the objective's own parameter object
    pinfo = vb.VectorParam('prior_info', lay.V, lb=0.0, val=new['prior_info'].copy())
    fun.prior_info_par = pinfo
    assert fun.hyper_kind(pinfo) == 'prior_info'
    two = vb.TwoParameterObjective(par, pinfo, fun)
    f_info = pinfo.get_free().copy()
    C = model.cross_hessian_hyper('prior_info', theta)
    assert rel_err(two.fun_free_hessian12(theta, f_info), C * new['prior_info'][None, :]) < 1e-11      # d a / d log a = a
    assert rel_err(two.fun_grad2(theta, f_info, True, True), model.hyper_grad('prior_info', theta) * new['prior_info']) < 1e-11
    with pytest.raises(NotImplementedError):
        fun.hyper_kind(vb.VectorParam('stranger', 3))
    with pytest.raises(ValueError):
        fun.lik_info_par.set_vector(np.array([-1.0])); objective.fun_free(theta)
    fun.lik_info_par.set_vector(np.array([0.45]))


def _newton(objective, theta, iters=8):
    """Trust-region fit (scipy drives the device callbacks), then Newton steps to stationarity at rounding level."""
    import scipy.optimize
    fit = scipy.optimize.minimize(objective.fun_free, theta, jac=objective.fun_free_grad, hess=objective.fun_free_hessian,
                                  method='trust-exact', options={'gtol': 1e-9, 'maxiter': 200})
    theta = fit.x
    for _ in range(3):
        theta = theta - np.linalg.solve(objective.fun_free_hessian(theta), objective.fun_free_grad(theta))
    return theta


def test_prior_sensitivity_predicts_refits_to_second_order_at_d256(vb):
    """`HyperparameterSensitivityLinearApproximation` (the name BASELINE.json's north star uses) at D = 256: the fitted
    coefficients of a Gaussian GLM under a perturbation of the prior MEAN and of the prior INFORMATION; the prediction error of
    the linear approximation falls by four when the step halves (second order), and the sensitivity equals the oracle's."""
    rng = np.random.default_rng(256)
    N, P = 3000, 256
    spec = [('box', 'free', 192, -np.inf, np.inf), ('box', 'pos', 64, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x = rng.normal(size=(N, P)) / np.sqrt(P)
    beta_true = np.concatenate([rng.normal(size=192), rng.uniform(0.5, 1.5, 64)])
    y = x @ beta_true + 0.3 * rng.normal(size=N)
    m0 = np.concatenate([np.zeros(192), np.ones(64)])
    a0 = rng.uniform(0.5, 1.5, P)
    fun = vb.GLMObjective(par, x, y, loss='gaussian', lik_info=4.0, prior_info=a0, prior_mean=m0)
    objective = vb.Objective(par, fun)
    theta0 = _newton(objective, np.zeros(P))
    assert np.linalg.norm(objective.fun_free_grad(theta0)) < 1e-8
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, lik_info=4.0, quad_A=a0, quad_m=m0)
    Hw = model.hessian(theta0)
    for kind, hp, direction in (('prior_mean', fun.prior_mean_par, rng.normal(size=P)),
                                ('prior_info', fun.prior_info_par, rng.uniform(0.2, 1.0, P))):
        h0 = hp.get_vector().copy()
        sens = vb.HyperparameterSensitivityLinearApproximation(fun, par, hp, theta0, h0)
        S = sens.get_dinput_dhyper()
        assert rel_err(S, -np.linalg.solve(Hw, model.cross_hessian_hyper(kind, theta0))) < 1e-8, kind
        errs = []
        for step in (0.2, 0.1):
            hp.set_vector(h0 + step * direction)
            refit = _newton(objective, theta0)
            assert np.linalg.norm(objective.fun_free_grad(refit)) < 1e-8
            errs.append(np.linalg.norm(sens.predict_input_par_from_hyperparameters(h0 + step * direction) - refit))
            moved = np.linalg.norm(refit - theta0)
        hp.set_vector(h0)
        assert errs[1] < 0.05 * moved, (kind, errs, moved)                  # first order is most of the move
        assert 3.0 < errs[0] / errs[1] < 5.5, (kind, errs)                  # and the remainder is second order


def test_mvn_regression_prior_sensitivity_at_d254(vb):
    """BASELINE.json's configuration 2 (k = 21, D = 254): cross Hessians with respect to the four priors against exact AD, and
    the refit under a prior-mean / prior-information perturbation predicted to second order."""
    rng = np.random.default_rng(254)
    N, k = 20000, 21
    x = rng.normal(size=(N, k)); y = x @ rng.normal(size=k) + rng.normal(size=N) / np.sqrt(2.0)
    par = vb.ModelParamsDict('params')
    par.push_param(vb.MVNParam('beta', dim=k)); par.push_param(vb.GammaParam('tau'))
    mu0 = rng.normal(size=k) * 0.3
    a = rng.normal(size=(k, k)); lam0 = 5.0 * (a @ a.T / k + np.eye(k))
    fun = vb.MVNRegressionObjective(par, x, y, prior_mean=mu0, prior_info=lam0, prior_shape=2.0, prior_rate=1.5)
    assert par.free_size() == 254
    lay = opk.Layout([opk.box_block(k), opk.psd_block(k), opk.box_block(1, lb=0.0), opk.box_block(1, lb=0.0)])
    objective = vb.Objective(par, fun)
    mean, info = vb.regression_utils.get_posterior_regression_coefficients(y, x, 2.0, mu0, lam0)
    par['beta']['mean'].set(mean); par['beta']['info'].set(0.5 * (info + info.T))
    par['tau']['shape'].set(np.array(2.0 + 0.5 * N)); par['tau']['rate'].set(np.array(1.5 + 0.25 * N))
    theta0 = _newton(objective, par.get_free(), iters=12)
    assert np.linalg.norm(objective.fun_free_grad(theta0)) < 1e-6 * abs(objective.fun_free(theta0))
    tt, tw = torch.tensor(theta0), torch.ones(N, dtype=torch.float64)
    tri = torch.tril_indices(k, k)

    def sym(v):
        L = torch.zeros(k, k, dtype=v.dtype).index_put((tri[0], tri[1]), v)
        return L + L.T - torch.diag(torch.diagonal(L))
    builders = {'prior_mean': lambda e: tr.mvn_regression_objective(x, y, k, e, lam0, 2.0, 1.5, layout=lay),
                'prior_info': lambda e: tr.mvn_regression_objective(x, y, k, mu0, sym(e), 2.0, 1.5, layout=lay),
                'prior_shape': lambda e: tr.mvn_regression_objective(x, y, k, mu0, lam0, e[0], 1.5, layout=lay),
                'prior_rate': lambda e: tr.mvn_regression_objective(x, y, k, mu0, lam0, 2.0, e[0], layout=lay)}
    for kind, build in builders.items():
        hp = fun.hyper_pars[kind]
        e0 = torch.tensor(hp.get_vector().copy())
        F = lambda th, e: build(e)(th, tw)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(tt, e0).numpy()
        two = vb.TwoParameterObjective(par, hp, fun)
        assert rel_err(two.fun_hessian_free1_vector2(theta0, hp.get_vector()), C) < 1e-9, kind
        assert rel_err(two.fun_grad2(theta0, hp.get_vector(), True, False), torch.func.grad(F, argnums=1)(tt, e0).numpy()) < 1e-9, kind
    H = objective.fun_free_hessian(theta0)
    for kind, direction in (('prior_mean', rng.normal(size=k)), ('prior_info', vb.models.sym_to_vech(np.eye(k) + 0.1 * (a + a.T)))):
        hp = fun.hyper_pars[kind]
        h0 = hp.get_vector().copy()
        sens = vb.HyperparameterSensitivityLinearApproximation(fun, par, hp, theta0, h0, hess0=H)
        errs = []
        for step in (0.5, 0.25):
            hp.set_vector(h0 + step * direction)
            refit = _newton(objective, theta0, iters=10)
            errs.append(np.linalg.norm(sens.predict_input_par_from_hyperparameters(h0 + step * direction) - refit))
            moved = np.linalg.norm(refit - theta0)
        hp.set_vector(h0)
        assert moved > 1e-4 and errs[1] < 0.1 * moved, (kind, errs, moved)
        assert 3.0 < errs[0] / errs[1] < 5.5, (kind, errs)


def test_wishart_priors_in_free_coordinates(vb):
    rng = np.random.default_rng(7)
    N, d = 500, 3
    y = rng.normal(size=(N, d)) @ np.array([[1.0, 0.2, 0.0], [0.0, 0.8, 0.3], [0.0, 0.0, 1.2]]) + 0.4
    par = vb.ModelParamsDict('params')
    par.push_param(vb.MVNParam('mu', dim=d)); par.push_param(vb.WishartParam('lambda', size=d))
    mu0 = rng.normal(size=d) * 0.2
    a = rng.normal(size=(d, d)); lam0 = a @ a.T / d + np.eye(d)
    b = rng.normal(size=(d, d)); w0 = b @ b.T / d + np.eye(d)
    fun = vb.WishartMVNObjective(par, y, prior_mean=mu0, prior_info=lam0, prior_df=d + 2.5, prior_inv_scale=w0)
    lay = opk.Layout([opk.box_block(d), opk.psd_block(d), opk.box_block(1, lb=d - 1.0), opk.psd_block(d)])
    theta = rng.normal(size=lay.D) * 0.3
    tt, tw = torch.tensor(theta), torch.ones(N, dtype=torch.float64)
    tri = torch.tril_indices(d, d)

    def sym(v):
        L = torch.zeros(d, d, dtype=v.dtype).index_put((tri[0], tri[1]), v)
        return L + L.T - torch.diag(torch.diagonal(L))
    builders = {'prior_mean': lambda e: tr.wishart_mvn_objective(y, d, e, lam0, d + 2.5, w0, layout=lay),
                'prior_info': lambda e: tr.wishart_mvn_objective(y, d, mu0, sym(e), d + 2.5, w0, layout=lay),
                'prior_df': lambda e: tr.wishart_mvn_objective(y, d, mu0, lam0, e[0], w0, layout=lay),
                'prior_inv_scale': lambda e: tr.wishart_mvn_objective(y, d, mu0, lam0, d + 2.5, sym(e), layout=lay)}
    assert set(builders) | {'weights'} == set(fun.hyper_pars)
    for kind, build in builders.items():
        hp = fun.hyper_pars[kind]
        e0 = torch.tensor(hp.get_vector().copy())
        F = lambda th, e: build(e)(th, tw)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(tt, e0).numpy()
        two = vb.TwoParameterObjective(par, hp, fun)
        assert rel_err(two.fun_hessian_free1_vector2(theta, hp.get_vector()), C) < 1e-9, kind
    # a new prior value changes the objective the way the restatement says
    fun.prior_inv_scale_par.set_vector(vb.models.sym_to_vech(2.0 * w0))
    f2 = tr.wishart_mvn_objective(y, d, mu0, lam0, d + 2.5, 2.0 * w0, layout=lay)
    objective = vb.Objective(par, fun)
    assert abs(objective.fun_free(theta) - f2(tt, tw).item()) < 1e-10 * abs(f2(tt, tw).item())
    assert rel_err(objective.fun_free_hessian(theta), torch.func.hessian(f2)(tt, tw).numpy()) < 1e-9


def test_lmm_prior_sensitivity_through_the_schur_complement(vb):
    from test_gpu_lmm import _layout, _cavi_optimum
    from test_lmm_host_math import make_par as lmm_make_par
    rng = np.random.default_rng(12)
    N, p, G = 600, 3, 8
    x = rng.normal(size=(N, p))
    gid = rng.integers(0, G, size=N).astype(np.int32); gid[:G] = np.arange(G)
    y = x @ rng.normal(size=p) + (rng.normal(size=G) * 0.7)[gid] + rng.normal(size=N) * 0.5
    par = lmm_make_par(p, G)
    a = rng.normal(size=(p, p))
    pri = dict(beta_prior_mean=rng.normal(size=p) * 0.2, beta_prior_info=a @ a.T / p + 0.5 * np.eye(p), mu_prior_mean=0.1, mu_prior_info=0.3,
               tau_y_prior=(2.0, 1.0), tau_mu_prior=(1.5, 0.5))
    fun = vb.LMMObjective(par, x, y, gid, G, **pri)
    lay = _layout(p, G)
    # the coordinate-ascent optimum (closed-form updates, stationary to rounding): the arrow Hessian is positive definite there
    theta = lay.unconstrain(_cavi_optimum(x, y, gid, G, pri['beta_prior_mean'], pri['beta_prior_info'], 0.1, 0.3, (2.0, 1.0), (1.5, 0.5)))
    theta = theta + 1e-3 * rng.normal(size=theta.size)
    tt, tw = torch.tensor(theta), torch.ones(N, dtype=torch.float64)
    tri = torch.tril_indices(p, p)

    def sym(v):
        L = torch.zeros(p, p, dtype=v.dtype).index_put((tri[0], tri[1]), v)
        return L + L.T - torch.diag(torch.diagonal(L))
    b0, l0 = pri['beta_prior_mean'], pri['beta_prior_info']
    builders = {
        'beta_prior_mean': lambda e: tr.lmm_objective(x, y, gid, G, e, l0, 0.1, 0.3, (2.0, 1.0), (1.5, 0.5), layout=lay),
        'beta_prior_info': lambda e: tr.lmm_objective(x, y, gid, G, b0, sym(e), 0.1, 0.3, (2.0, 1.0), (1.5, 0.5), layout=lay),
        'mu_prior': lambda e: tr.lmm_objective(x, y, gid, G, b0, l0, e[0], e[1], (2.0, 1.0), (1.5, 0.5), layout=lay),
        'tau_y_prior': lambda e: tr.lmm_objective(x, y, gid, G, b0, l0, 0.1, 0.3, (e[0], e[1]), (1.5, 0.5), layout=lay),
        'tau_mu_prior': lambda e: tr.lmm_objective(x, y, gid, G, b0, l0, 0.1, 0.3, (2.0, 1.0), (e[0], e[1]), layout=lay)}
    assert set(builders) | {'weights'} == set(fun.hyper_pars)
    H_ad = torch.func.hessian(lambda th: builders['mu_prior'](torch.tensor([0.1, 0.3]))(th, tw))(tt).numpy()
    assert np.min(np.linalg.eigvalsh(H_ad)) > 0
    ng = fun.n_global
    for kind, build in builders.items():
        hp = fun.hyper_pars[kind]
        e0 = torch.tensor(hp.get_vector().copy())
        F = lambda th, e: build(e)(th, tw)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(tt, e0).numpy()
        assert rel_err(fun.global_cross_hessian(hp, theta), C[:ng]) < 1e-9, kind
        two = vb.TwoParameterObjective(par, hp, fun)
        assert rel_err(two.fun_hessian_free1_vector2(theta, hp.get_vector()), C) < 1e-9, kind
        # linear response of the global parameters: the global rows of -H^-1 C of the FULL arrow matrix
        want = -np.linalg.solve(H_ad, C)[:ng]
        assert rel_err(fun.global_sensitivity(hp, theta), want) < 1e-7, kind


def test_mixture_prior_sensitivity_through_the_schur_complement(vb):
    from test_mixture_host_math import make_par as mixture_par, near_optimum_problem
    N, V, K = 60, 4, 3
    x, w, theta = near_optimum_problem(N, V, K, seed=21)            # the seed of tests/test_gpu_mixture.py: positive definite there
    par = mixture_par(N, V, K)
    a0 = np.array([1.5, 1.4, 1.6]); b0 = 0.8 + np.random.default_rng(1).uniform(-0.05, 0.05, (V, K))
    fun = vb.MixtureObjective(par, x, pi_prior=a0, phi_prior=b0, weights=w)
    tt, tw = torch.tensor(theta), torch.tensor(w)
    builders = {'pi_prior': (a0, lambda e: tr.mixture_objective(x, K, e, b0)),
                'phi_prior': (b0.ravel(), lambda e: tr.mixture_objective(x, K, a0, e.reshape(V, K)))}
    H_ad = torch.func.hessian(lambda th: tr.mixture_objective(x, K, a0, b0)(th, tw))(tt).numpy()
    assert np.min(np.linalg.eigvalsh(H_ad)) > 0
    ng = fun.n_global
    for kind, (e0, build) in builders.items():
        hp = fun.hyper_pars[kind]
        np.testing.assert_allclose(hp.get_vector(), e0)
        F = lambda th, e: build(e)(th, tw)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(tt, torch.tensor(e0)).numpy()
        assert rel_err(fun.global_cross_hessian(hp, theta), C[:ng]) < 1e-9, kind
        assert rel_err(fun.hyper_grad(hp, theta), torch.func.grad(F, argnums=1)(tt, torch.tensor(e0)).numpy()) < 1e-9, kind
        want = -np.linalg.solve(H_ad, C)[:ng]
        assert rel_err(fun.global_sensitivity(hp, theta), want) < 1e-6, kind
    # a new prior is used by the next evaluation
    fun.pi_prior_par.set_vector(2.0 * a0)
    f2 = tr.mixture_objective(x, K, 2.0 * a0, b0)
    assert abs(fun.value(theta) - f2(tt, tw).item()) < 1e-10 * abs(f2(tt, tw).item())


@pytest.mark.parametrize('kind', ['prior_mean', 'prior_info', 'lik_info', 'quad_scale'])
def test_taylor_expansion_on_the_new_hyper_parameters(vb, kind):
    """`ParametricSensitivityTaylorExpansion` with a prior / likelihood hyper-parameter, in vector and (where the parameter is
    bounded) free coordinates: the device functor gives the derivatives the same class gives on the oracle's arithmetic (which
    tests/test_taylor_host_math.py pins by exact nested AD), and the series predicts refits with the error falling like t^(K+1)."""
    rng = np.random.default_rng(70)
    spec = [('box', 'beta', 8, -1.0, np.inf), ('psd', 'm', 2, 0.2), ('box', 'u', 3, -np.inf, np.inf)]
    par, lay, fun, model = _declared(vb, rng, spec, 400, kind == 'prior_info', 'beta', 8)
    objective = vb.Objective(par, fun)
    phi0 = _newton(objective, np.zeros(lay.D), iters=12)
    assert np.linalg.norm(model.grad(phi0)) < 1e-8
    K = 3
    for hyper_is_free in ((False, True) if kind in ('lik_info',) else (False,)):
        hp = fun.hyper_pars[kind]
        h0 = (hp.get_free() if hyper_is_free else hp.get_vector()).copy()
        opar = vb.VectorParam(kind, hp.size(), lb=hp._lb, ub=hp._ub, val=hp.get_vector().copy())
        ofun = OracleFunctor(par, model, **{kind + '_par': opar})
        tay = vb.ParametricSensitivityTaylorExpansion(fun, par, hp, phi0, h0, K, hyper_is_free=hyper_is_free)
        otay = vb.ParametricSensitivityTaylorExpansion(ofun, par, opar, phi0, h0, K, hyper_is_free=hyper_is_free)
        de = rng.normal(size=h0.size) * 0.3
        for k in range(1, K + 1):
            got, want = tay.evaluate_dkinput_dhyperk(de, k), otay.evaluate_dkinput_dhyperk(de, k)
            assert rel_err(got, want) < 1e-7, (kind, hyper_is_free, k)
        errs = []
        for t in (0.3, 0.15):
            (hp.set_free if hyper_is_free else hp.set_vector)(h0 + t * de)
            refit = _newton(objective, phi0, iters=12)
            errs.append(np.linalg.norm(tay.evaluate_taylor_series(t * de) - refit))
        (hp.set_free if hyper_is_free else hp.set_vector)(h0)
        assert errs[1] < errs[0] / 10.0 or errs[0] < 1e-9, (kind, hyper_is_free, errs)      # t^(K+1) = 1/16 per halving


def test_reference_quadratic_model_with_plain_closures(vb):
    """LRVB/test_model_sensitivity.py:367-424 as the reference writes it: `QuadraticModel` (:36-88) with plain closures for the
    objective and for the hyper-parameter part, BFGS fit, the Jacobian of the closed-form optimum (autograd.jacobian there,
    its analytic form here), and the variant with `hyper_par_objective_functor`."""
    import scipy.optimize
    from copy import deepcopy
    sens_lib, obj_lib = vb.ModelSensitivity, vb.SparseObjectives

    class QuadraticModel(object):
        def __init__(self, dim):
            self.dim = dim
            self.param = vb.VectorParam('theta', size=dim, lb=-10.0)
            self.param_copy = deepcopy(self.param)
            self.hyper_param = vb.VectorParam('lambda', size=dim, lb=-2.0)
            self.hyper_param.set_vector(np.linspace(0.5, 10.0, num=dim))
            vec = np.linspace(0.1, 0.3, num=dim)
            self.matrix = np.outer(vec, vec) + np.eye(dim)
            self.objective = obj_lib.Objective(self.param, self.get_objective)

        def get_hyper_par_objective(self):
            theta = self.param.get()
            return self.hyper_param.get() @ theta

        def get_objective(self):
            theta = self.param.get()
            return 0.5 * theta.T @ self.matrix @ theta + self.get_hyper_par_objective()

        def get_true_optimum_theta(self, hyper_param_val):
            return -1 * np.linalg.solve(self.matrix, hyper_param_val)

        def get_true_optimum(self, hyper_param_val):
            self.param_copy.set_vector(self.get_true_optimum_theta(hyper_param_val))
            return self.param_copy.get_free()

    model = QuadraticModel(3)
    opt_output = scipy.optimize.minimize(fun=model.objective.fun_free, jac=model.objective.fun_free_grad, x0=np.zeros(model.dim), method='BFGS')
    hyper_param_val = model.hyper_param.get_vector()
    theta0 = model.get_true_optimum(hyper_param_val)
    np.testing.assert_array_almost_equal(theta0, opt_output.x)
    model.param.set_free(theta0)
    parametric_sens = sens_lib.ParametricSensitivityLinearApproximation(
        objective_functor=model.get_objective, input_par=model.param, hyper_par=model.hyper_param,
        input_val0=theta0, hyper_val0=hyper_param_val)
    epsilon = 0.01
    new_hyper_param_val = hyper_param_val + epsilon
    pred_diff = parametric_sens.predict_input_par_from_hyperparameters(new_hyper_param_val) - theta0
    true_diff = model.get_true_optimum(new_hyper_param_val) - theta0
    assert np.linalg.norm(true_diff - pred_diff) <= epsilon * np.linalg.norm(true_diff)
    # d log(theta_hat + 10) / d eps with theta_hat = -A^-1 eps
    jac = np.diag(1.0 / (model.get_true_optimum_theta(hyper_param_val) + 10.0)) @ (-np.linalg.inv(model.matrix))
    np.testing.assert_array_almost_equal(jac, parametric_sens.get_dinput_dhyper())
    model.param.set_free(theta0)
    model.hyper_param.set_vector(hyper_param_val)
    parametric_sens2 = sens_lib.ParametricSensitivityLinearApproximation(
        objective_functor=model.get_objective, input_par=model.param, hyper_par=model.hyper_param,
        input_val0=theta0, hyper_val0=hyper_param_val, hyper_par_objective_functor=model.get_hyper_par_objective)
    np.testing.assert_array_almost_equal(jac, parametric_sens2.get_dinput_dhyper())
    # and in free coordinates of the hyper-parameter (lb = -2): chain rule d eps / d free = eps + 2
    sens3 = sens_lib.ParametricSensitivityLinearApproximation(
        objective_functor=model.get_objective, input_par=model.param, hyper_par=model.hyper_param,
        input_val0=theta0, hyper_val0=model.hyper_param.get_free(), hyper_is_free=True)
    np.testing.assert_array_almost_equal(jac * (hyper_param_val + 2.0)[None, :], sens3.get_dinput_dhyper())
