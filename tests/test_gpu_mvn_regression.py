"""BASELINE.json config 2 on the GPU: MVNParam regression, N = 1e5, k = 21 -> D = 21 + 231 + 2 = 254
free parameters, dense Hessian on one MI355X.  Oracle: exact AD (torch.func fp64 on the host) of
the restated -ELBO at the full size, plus small-size cases; tolerance 1e-9 relative (the Hessian
contains Lambda^-1 products: its condition enters the rounding)."""
import numpy as np
import pytest
import scipy.optimize
import torch

import torch_ref as tr
from oracle import packing as opk
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _build(vb, rng, N, k):
    x = rng.normal(size=(N, k))
    beta_true = rng.normal(size=k)
    y = x @ beta_true + rng.normal(size=N) / np.sqrt(2.0)          # tau* = 2
    par = vb.ModelParamsDict('params')
    par.push_param(vb.MVNParam('beta', dim=k))
    par.push_param(vb.GammaParam('tau'))
    mu0 = np.zeros(k); lam0 = 0.1 * np.eye(k)
    fun = vb.MVNRegressionObjective(par, x, y, prior_mean=mu0, prior_info=lam0, prior_shape=2.0, prior_rate=1.5)
    lay = opk.Layout([opk.box_block(k), opk.psd_block(k), opk.box_block(1, lb=0.0), opk.box_block(1, lb=0.0)])
    ft = tr.mvn_regression_objective(x, y, k, mu0, lam0, 2.0, 1.5, layout=lay)
    return x, y, par, fun, lay, ft


@pytest.mark.parametrize('N,k', [(200, 2), (5000, 6)])
def test_small_cases_all_derivatives(vb, N, k):
    rng = np.random.default_rng(N + k)
    x, y, par, fun, lay, ft = _build(vb, rng, N, k)
    assert par.free_size() == lay.D == k + k * (k + 1) // 2 + 2
    objective = vb.Objective(par, fun)
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)
    theta = rng.normal(size=lay.D) * 0.3
    tt, tw = torch.tensor(theta), torch.tensor(w)
    H_ad = torch.func.hessian(ft)(tt, tw).numpy()
    assert abs(objective.fun_free(theta) - ft(tt, tw).item()) < 1e-10 * abs(ft(tt, tw).item())
    assert rel_err(objective.fun_free_grad(theta), torch.func.grad(ft)(tt, tw).numpy()) < 1e-9
    assert rel_err(objective.fun_free_hessian(theta), H_ad) < 1e-9
    v = rng.normal(size=lay.D)
    assert rel_err(objective.fun_free_hvp(theta, v), H_ad @ v) < 1e-9
    cross = torch.func.jacrev(torch.func.grad(ft, argnums=0), argnums=1)(tt, tw).numpy()
    two = vb.TwoParameterObjective(par, fun.weights_par, fun)
    assert rel_err(two.fun_hessian_free1_vector2(theta, w), cross) < 1e-9


def test_config2_full_size(vb):
    N, k = 100000, 21
    rng = np.random.default_rng(20242)
    x, y, par, fun, lay, ft = _build(vb, rng, N, k)
    D = par.free_size()
    assert D == 254
    objective = vb.Objective(par, fun)
    # start from the closed-form conditional posterior of beta (regression_utils) at tau = 2
    mean, info = vb.regression_utils.get_posterior_regression_coefficients(y, x, 2.0, np.zeros(k), 0.1 * np.eye(k))
    par['beta']['mean'].set(mean)
    par['beta']['info'].set(0.5 * (info + info.T))
    par['tau']['shape'].set(np.array(2.0 + 0.5 * N)); par['tau']['rate'].set(np.array(1.5 + 0.25 * N))
    theta0 = par.get_free()
    opt = scipy.optimize.minimize(objective.fun_free, jac=objective.fun_free_grad, hessp=objective.fun_free_hvp,
                                  x0=theta0, method='trust-ncg', options={'gtol': 1e-6, 'maxiter': 50})
    theta_hat = opt.x + 1e-3 * rng.normal(size=D)          # perturbed optimum: third-order terms non-zero
    w1 = torch.ones(N, dtype=torch.float64)
    tt = torch.tensor(theta_hat)
    H = objective.fun_free_hessian(theta_hat)
    H_ad = torch.func.hessian(ft)(tt, w1).numpy()
    assert rel_err(H, H_ad) < 1e-9
    assert np.allclose(H, H.T, rtol=0, atol=1e-9 * np.max(np.abs(H)))
    assert rel_err(objective.fun_free_grad(theta_hat), torch.func.grad(ft)(tt, w1).numpy()) < 1e-8
    assert np.min(np.linalg.eigvalsh(0.5 * (H + H.T))) > 0
    # LRVB covariance of the coefficient means: M H^-1 M^T with M = d mean / d theta, vs the oracle
    M = vb.Objective(par, vb.LinearMoments(par, B=np.eye(par.vector_size())[:k])).fun_free_jacobian(theta_hat)
    fun.ctx.chol_factor(H)
    cov = fun.ctx.lrvb_cov(M)
    assert rel_err(cov, M @ np.linalg.solve(H_ad, M.T)) < 1e-6        # BASELINE.json's rtol 1e-6
    # the optimum's posterior covariance of beta is close to the classical (X^T tau X + Lambda0)^-1
    par.set_free(opt.x)
    e_tau = par['tau'].e()
    np.testing.assert_allclose(np.linalg.inv(par['beta']['info'].get()),
                               np.linalg.inv(e_tau * x.T @ x + 0.1 * np.eye(k)), rtol=1e-4, atol=1e-10)


def test_statistics_are_not_reduced_twice_under_a_hook(vb):
    """Advisor finding, round 3: with a sum-over-ranks hook on the context the statistics calls return GLOBAL sums; the host flow
    `set_reduced_stats(allreduce_stats(local_stats()))` would multiply them by the world size.  It is refused, and the host
    caches of the statistics follow the hook (installed / removed) instead of going stale."""
    rng = np.random.default_rng(5)
    x, y, par, fun, lay, ft = _build(vb, rng, 500, 3)
    local = fun.local_stats().copy()
    calls = []

    def doubling_hook(ptr, n, stream):                      # stands for "sum over two identical ranks"
        import torch
        from lrvb_amd.distributed import _DevicePointer
        t = torch.as_tensor(_DevicePointer(ptr, n), device='cuda:0')
        t.mul_(2.0)
        calls.append(n)
    import torch
    fun.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    fun.ctx.set_reduce_hook(doubling_hook)
    try:
        under_hook = fun.local_stats()
        assert calls and np.allclose(under_hook, 2.0 * local, rtol=1e-14)      # not the cached local statistics
        with pytest.raises(RuntimeError):
            fun.set_reduced_stats(under_hook)
    finally:
        fun.ctx.set_reduce_hook(None)
        fun.ctx.set_stream(None)
    np.testing.assert_allclose(fun.local_stats(), local, rtol=1e-14)          # hook removed: local again, not the doubled cache
    fun.set_reduced_stats(local)                                               # the host-side exchange is fine without a hook
    fun.set_reduced_stats(None)


@pytest.mark.parametrize('N,k', [(300, 1), (5000, 6), (20000, 21), (4000, 31), (4000, 32), (3000, 63)])
def test_one_call_hessian_matches_the_stepwise_route_and_ad(vb, N, k):
    """`MVNRegressionObjective.device_hessian` (lrvb_mvnreg_hessian: statistics, closed forms in (m, Lambda, a, b) evaluated on
    the device where the statistics lie, Kronecker block, free conversion -- one call, no copy back inside) against round 3's
    stepwise route (statistics to the host, numpy closed forms, blocks sent up) and against exact AD; the value comes from the
    same kernel; want_host=False leaves the matrix where the Cholesky finds it.  k = 31 and 63 (q = 32, 64: no spare column in
    the narrow Gram kernel for the sum of the weights), k = 32 (an odd row width on the two-pair instantiation) and k = 63 (the
    largest log-Cholesky block the structured Jacobian product takes) are compared with the stepwise route only."""
    rng = np.random.default_rng(N + k)
    x, y, par, fun, lay, ft = _build(vb, rng, N, k)
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)
    theta = rng.normal(size=lay.D) * (0.3 if k <= 21 else 0.05)      # (the larger information matrices stay well conditioned: the two routes invert them differently)
    H1, val = fun.device_hessian(theta, want_value=True)
    fun.stepwise = True
    H0 = vb.Objective(par, fun).fun_free_hessian(theta)
    fun.stepwise = False
    assert rel_err(H1, H0) < (1e-12 if k <= 21 else 1e-11)
    assert np.max(np.abs(H1 - H1.T)) < 1e-12 * np.max(np.abs(H1))
    tt, tw = torch.tensor(theta), torch.tensor(w)
    assert abs(val - ft(tt, tw).item()) < 1e-11 * abs(ft(tt, tw).item())
    if k > 21:
        return
    assert rel_err(H1, torch.func.hessian(ft)(tt, tw).numpy()) < 1e-9
    assert rel_err(vb.Objective(par, fun).fun_free_hessian(theta), H1) == 0.0          # the Objective route IS the one call
    # resident result: factor it where it lies (shifted copy on the host for the comparison)
    assert fun.device_hessian(theta, want_host=False)[0] is None
    ev = np.linalg.eigvalsh(H1).min()
    if ev > 1e-8 * np.abs(H1).max():
        fun.ctx.chol_factor_last()
        M = np.eye(lay.D)[:2]
        assert rel_err(fun.ctx.lrvb_cov(M), M @ np.linalg.solve(H1, M.T)) < 1e-7
    # new weights and a new prior reach the next call
    fun.weights_par.set_vector(w * 1.5)
    fun.prior_mean_par.set_vector(np.full(k, 0.3))
    ft2 = tr.mvn_regression_objective(x, y, k, np.full(k, 0.3), 0.1 * np.eye(k), 2.0, 1.5, layout=lay)
    assert rel_err(fun.device_hessian(theta)[0], torch.func.hessian(ft2)(tt, 1.5 * tw).numpy()) < 1e-9


def test_replayed_launch_chain_follows_every_change(vb):
    """From the third call of one shape on, `lrvb_mvnreg_hessian` replays its launch chain as a captured graph.  The replay must
    see everything a plain call sees: a new point and new prior values (they travel in the per-call upload), new weights (same
    buffer: contents; a larger problem: new buffers, the graph is rebuilt), the profile marks (plain launches under them) and a
    change of stream; results compared with exact AD each time."""
    rng = np.random.default_rng(909)
    N, k = 3000, 5
    x, y, par, fun, lay, ft = _build(vb, rng, N, k)
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)

    def check(theta, weights, ftorch=ft):
        H = fun.device_hessian(theta)[0]
        want = torch.func.hessian(ftorch)(torch.tensor(theta), torch.tensor(weights)).numpy()
        assert rel_err(H, want) < 1e-9
        return H
    th = rng.normal(size=lay.D) * 0.3
    H1 = check(th, w)                                          # call 1: plain (warms the buffers)
    H2 = check(th, w)                                          # call 2: captured
    H3 = check(th, w)                                          # call 3: replayed
    assert np.array_equal(H1, H2) and np.array_equal(H2, H3)
    th2 = rng.normal(size=lay.D) * 0.3
    check(th2, w)                                              # a new point through the replay
    w2 = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w2)
    check(th2, w2)                                             # new weights
    fun.prior_mean_par.set_vector(np.full(k, 0.3))
    ft2 = tr.mvn_regression_objective(x, y, k, np.full(k, 0.3), 0.1 * np.eye(k), 2.0, 1.5, layout=lay)
    check(th2, w2, ft2)                                        # a new prior
    fun.ctx.profile_enable(True)
    Hp = check(th, w2, ft2)                                    # under the profile marks: plain launches
    fun.ctx.profile_enable(False)
    assert np.array_equal(Hp, check(th, w2, ft2))              # and the replay again: the same bits
    s = torch.cuda.Stream()
    fun.ctx.set_stream(s.cuda_stream)
    for _ in range(3):
        Hs = check(th, w2, ft2)                                # another stream: warmed, captured and replayed there
    fun.ctx.set_stream(None)
    assert np.array_equal(Hs, Hp)
