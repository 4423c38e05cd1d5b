"""The non-conjugate logistic term (LRVB/Modeling.py:16-52) and the regression model built on it, on the GPU
(SURVEY.md section 8(f) item 4).  Oracle: oracle/logitnormal.py (pinned by the reference's own Monte-Carlo check and by exact
AD, tests/test_logitnormal_host_math.py).  Tolerances: quadrature sums 1e-13 absolute on O(1) values; value / gradient /
Hessian of the model 1e-11 / 1e-10 / 1e-9 relative (sums over N observations in a different order than numpy's);
LRVB covariance rtol 1e-6 (BASELINE.json)."""
import math

import numpy as np
import pytest
import scipy.optimize
import torch

from oracle import logitnormal as ol
from helpers import rel_err
from test_logitnormal_host_math import problem, torch_kl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _model(vb, x, y, w, tau=0.7, deg=20):
    P = x.shape[1]
    par = vb.ModelParamsDict('params')
    par.push_param(vb.UVNParamVector('beta', length=P))
    fun = vb.LogitNormalRegressionObjective(par, x, y, prior_info=tau, gh_deg=deg, weights=w)
    return par, fun


@pytest.mark.parametrize('shape,K', [((1,), 1), ((7,), 5), ((300,), 20), ((33, 4), 32), ((1000,), 128)])
def test_quadrature_term_and_derivatives_match_oracle(vb, shape, K):
    rng = np.random.default_rng(K + len(shape))
    gx, gw = np.polynomial.hermite.hermgauss(K)
    m = rng.normal(size=shape) * 3.0
    s = rng.uniform(0.0, 3.0, size=shape)
    s.flat[0] = 0.0                                            # sd = 0: the plug-in value
    m.flat[-1] = 750.0                                         # past the overflow point of log1p(exp(.))
    ctx = vb.scratch_context()
    val, d1, d2 = ctx.gh_logistic(m, s, gx, gw, order=2)
    o_val, o_d1, o_d2 = ol.gh_logistic_derivs(m, s, gx, gw)
    assert val.shape == shape and d1.shape == shape + (2,) and d2.shape == shape + (3,)
    assert np.max(np.abs(val - o_val) / np.maximum(1.0, np.abs(o_val))) < 1e-13
    assert np.max(np.abs(d1 - o_d1)) < 1e-13 * (1 + np.max(np.abs(o_d1)))
    assert np.max(np.abs(d2 - o_d2)) < 1e-13 * (1 + np.max(np.abs(o_d2)))
    assert np.all(np.isfinite(val))
    # the reference-named wrappers with a context: aggregate and per-element forms, and the draws form
    md = vb.Modeling
    assert abs(md.get_e_logistic_term_guass_hermite(m, s, gx, gw, ctx=ctx) - np.sum(o_val)) < 1e-12 * max(1.0, abs(np.sum(o_val)))
    np.testing.assert_allclose(md.get_e_logistic_term_guass_hermite(m, s, gx, gw, aggregate_all=False, ctx=ctx), o_val, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(md.get_e_logistic_term_guass_hermite(m, s, gx, gw, ctx=ctx), md.get_e_logistic_term_guass_hermite(m, s, gx, gw), rtol=1e-12)
    y = rng.random(size=shape)
    draws = md.get_standard_draws(25)
    assert abs(md.get_e_logistic_term(y, m, s, draws, ctx=ctx) - ol.draws_logistic(y, m, s, draws)) < 1e-11 * (1 + abs(ol.draws_logistic(y, m, s, draws)))
    with pytest.raises(Exception):
        ctx.gh_logistic(m, s, np.zeros(129), np.zeros(129))    # at most 128 nodes


@pytest.mark.parametrize('N,P', [(1, 2), (37, 3), (500, 8), (1999, 17), (4096, 64), (20011, 130)])
def test_model_matches_oracle_in_vector_and_free_coordinates(vb, N, P):
    """P even -> the LDS-DMA MFMA kernel forms the three Hessian products, P odd -> the generic tile GEMM."""
    x, y, w, eta = problem(N, P, seed=N + P)
    gx, gw = np.polynomial.hermite.hermgauss(20)
    par, fun = _model(vb, x, y, w)
    o_val, o_g, o_H = ol.kl_terms(eta, x, y, w, 0.7, gx, gw)
    assert abs(fun.value(eta, False) - o_val) < 1e-11 * max(1.0, abs(o_val))
    assert rel_err(fun.grad(eta, False), o_g) < 1e-10
    H = fun.hessian(eta, False)
    assert rel_err(H, o_H) < 1e-9 and np.max(np.abs(H - H.T)) < 1e-11 * np.max(np.abs(H))
    # free coordinates: mean unconstrained, info = exp(f) (lower bound 0: LRVB/Parameters.py:47-61)
    theta = np.concatenate([eta[:P], np.log(eta[P:])])
    objective = vb.Objective(par, fun)
    J = np.concatenate([np.ones(P), eta[P:]])
    assert abs(objective.fun_free(theta) - o_val) < 1e-11 * max(1.0, abs(o_val))
    assert rel_err(objective.fun_free_grad(theta), J * o_g) < 1e-10
    H_free = J[:, None] * o_H * J[None, :] + np.diag(np.concatenate([np.zeros(P), o_g[P:] * eta[P:]]))
    assert rel_err(objective.fun_free_hessian(theta), H_free) < 1e-9
    v = np.random.default_rng(1).normal(size=2 * P)
    assert rel_err(objective.fun_free_hvp(theta, v), H_free @ v) < 1e-9
    assert np.allclose(par['beta']['mean'].get(), eta[:P])     # the side-effect contract: par holds the evaluation point


def test_fit_lrvb_covariance_and_weight_sensitivity(vb):
    N, P = 3000, 6
    x, y, w, eta = problem(N, P, seed=77)
    w = np.ones(N)
    par, fun = _model(vb, x, y, w, tau=0.5, deg=30)
    objective = vb.Objective(par, fun)
    theta0 = np.concatenate([np.zeros(P), np.zeros(P)])
    opt = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp, x0=theta0,
                                  method='trust-ncg', options={'gtol': 1e-7, 'maxiter': 100})
    # (scipy may stop with status 2 once the predicted decrease falls below the rounding of f ~ 1.6e3: judge by the gradient)
    assert np.max(np.abs(objective.fun_free_grad(opt.x))) < 1e-6
    theta_hat = opt.x
    gx, gw = np.polynomial.hermite.hermgauss(30)
    args = (torch.tensor(x), torch.tensor(y), torch.tensor(w), 0.5, torch.tensor(gx), torch.tensor(gw))

    def f_free(th, ww):
        eta_t = torch.cat([th[:P], torch.exp(th[P:])])
        return torch_kl(eta_t, args[0], args[1], ww, 0.5, args[4], args[5])
    tt = torch.tensor(theta_hat)
    H_ad = torch.func.hessian(f_free)(tt, args[2]).numpy()
    H = objective.fun_free_hessian(theta_hat)
    assert rel_err(H, H_ad) < 1e-9
    assert np.min(np.linalg.eigvalsh(0.5 * (H + H.T))) > 0
    # linear-response covariance of the coefficient means (the reference's use of the Hessian), device Cholesky path
    M = np.eye(2 * P)[:P]
    fun.ctx.chol_factor(H)
    cov = fun.ctx.lrvb_cov(M)
    assert rel_err(cov, M @ np.linalg.solve(H_ad, M.T)) < 1e-6
    # the LRVB covariance of the means exceeds the mean-field variances 1 / info (the point of linear response)
    par.set_free(theta_hat)
    assert np.all(np.diag(cov) >= 1.0 / par['beta']['info'].get() * (1 - 1e-9))
    # weight sensitivity by linear response against exact AD, and against a refit with one observation left out
    cross = torch.func.jacrev(torch.func.grad(f_free, argnums=0), argnums=1)(tt, args[2]).numpy()
    two = vb.TwoParameterObjective(par, fun.weights_par, fun)
    assert rel_err(two.fun_hessian_free1_vector2(theta_hat, w), cross) < 1e-9
    sens = -np.linalg.solve(H, two.fun_hessian_free1_vector2(theta_hat, w))          # d theta / d w^T
    w2 = w.copy(); w2[11] = 0.0
    fun.weights_par.set_vector(w2)
    opt2 = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp, x0=theta_hat,
                                   method='trust-ncg', options={'gtol': 1e-7, 'maxiter': 100})
    pred = theta_hat + sens @ (w2 - w)
    assert np.max(np.abs(pred - opt2.x)) < 0.05 * np.max(np.abs(opt2.x - theta_hat)) + 1e-8


def test_bad_arguments_fail_loudly(vb):
    x, y, w, eta = problem(30, 4, seed=2)
    par, fun = _model(vb, x, y, w)
    gx, gw = np.polynomial.hermite.hermgauss(10)
    with pytest.raises(ValueError):
        fun.ctx.logitnormal_terms(np.zeros(4), np.array([1.0, 0.0, 1.0, 1.0]), gx, gw)      # a variance that is not positive
    with pytest.raises(ValueError):
        fun.ctx.logitnormal_terms(np.zeros(3), np.ones(3), gx, gw)                           # wrong length
    par2 = vb.ModelParamsDict('params')
    par2.push_param(vb.UVNParamVector('beta', length=5))
    with pytest.raises(ValueError):
        vb.LogitNormalRegressionObjective(par2, x, y)                                         # parameter does not match the design


def test_statistics_are_additive_over_shards(vb):
    """The multi-GPU reduction of this model: [value | gradient | Hessian blocks] of row shards add up to the full data term,
    and a shard object with the summed statistics installed returns the full-data value, gradient and Hessian."""
    N, P = 5001, 12
    x, y, w, eta = problem(N, P, seed=9)
    gx, gw = np.polynomial.hermite.hermgauss(20)
    _, full = _model(vb, x, y, w)
    n1 = 1777
    _, f1 = _model(vb, x[:n1], y[:n1], w[:n1])
    _, f2 = _model(vb, x[n1:], y[n1:], w[n1:])
    s_sum = f1.local_stats(eta) + f2.local_stats(eta)
    assert rel_err(s_sum, full.local_stats(eta)) < 1e-11
    o_val, o_g, o_H = ol.kl_terms(eta, x, y, w, 0.7, gx, gw)
    f1.set_reduced_stats(s_sum, eta)
    assert abs(f1.value(eta, False) - o_val) < 1e-11 * abs(o_val)
    assert rel_err(f1.grad(eta, False), o_g) < 1e-10 and rel_err(f1.hessian(eta, False), o_H) < 1e-9
    with pytest.raises(ValueError):
        f1.value(eta * 1.01, False)                             # statistics of another point
    f1.set_reduced_stats(None)
    assert abs(f1.value(eta, False) - o_val) > 1e-3 * abs(o_val)      # back to its own rows


@pytest.mark.parametrize('P', [62, 66, 126, 128, 190, 192, 194, 254, 256, 258, 320, 382])
def test_hessian_blocks_over_tile_edge_shapes(vb, P):
    """The three Hessian products run on the two-operand LDS-DMA MFMA kernel for every even P: widths just below, at and
    above the 128-column tile and its 64-column half (an edge tile whose real columns end inside it, a full tile next to a
    sliver, ...), and row counts that end inside a 16-row stage and inside a split."""
    N = 1000 + 37 * (P % 13)
    x, y, w, eta = problem(N, P, seed=P)
    gx, gw = np.polynomial.hermite.hermgauss(12)
    par, fun = _model(vb, x, y, w, deg=12)
    o_val, o_g, o_H = ol.kl_terms(eta, x, y, w, 0.7, gx, gw)
    assert rel_err(fun.grad(eta, False), o_g) < 1e-10
    assert rel_err(fun.hessian(eta, False), o_H) < 1e-9
