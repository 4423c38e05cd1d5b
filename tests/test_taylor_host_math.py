"""Higher-order sensitivity, host half: the term algebra, the element-wise packing-map
derivatives and the oracle's closed form of D^j g [u_1 .. u_j], pinned by exact nested forward-mode AD (torch.func.jvp)
of the torch restatement of the gradient -- what the reference computes with nested autograd JVPs
(LRVB/ModelSensitivity.py:38-62, 221-234)."""
import math

import numpy as np
import pytest
import torch

import lrvb_amd as vb
from lrvb_amd import taylor
import torch_ref as tr
from oracle import models as om, packing as opk
from helpers import glm_data


def test_term_algebra_matches_the_known_expansions():
    """d^k/dt^k g(eta(t), eps0 + t d) written out by hand for k = 1, 2, 3 (the k = 2, 3 lines are the expressions in
    the comments of LRVB/ModelSensitivity.py:320-345)."""
    t1 = taylor.get_taylor_base_terms()
    assert sorted((t.key(), t.prefactor) for t in t1) == [((0, (1,)), 1.0), ((1, (0,)), 1.0)]
    t2 = taylor.differentiate_terms(None, t1)
    assert dict((t.key(), t.prefactor) for t in t2) == {(2, (0, 0)): 1.0, (1, (1, 0)): 2.0, (0, (2, 0)): 1.0, (0, (0, 1)): 1.0}
    t3 = taylor.differentiate_terms(None, t2)
    assert dict((t.key(), t.prefactor) for t in t3) == {
        (3, (0, 0, 0)): 1.0, (2, (1, 0, 0)): 3.0, (1, (2, 0, 0)): 3.0, (1, (0, 1, 0)): 3.0,
        (0, (3, 0, 0)): 1.0, (0, (1, 1, 0)): 3.0, (0, (0, 0, 1)): 1.0}
    for k, terms in enumerate([t1, t2, t3, taylor.differentiate_terms(None, t3)], start=1):
        assert all(t.order == k for t in terms)
        assert sum(1 for t in terms if t.eta_orders[-1] == 1) == 1          # exactly one term carries eta^(k): H eta^(k)
    # the pure-eta terms of order k count the set partitions of k elements (Faa di Bruno): Bell numbers 1, 2, 5, 15
    t4 = taylor.differentiate_terms(None, t3)
    for terms, bell in ((t1, 1), (t2, 2), (t3, 5), (t4, 15)):
        assert sum(t.prefactor for t in terms if t.eps_order == 0) == bell
    with pytest.raises(AssertionError):
        taylor.DerivativeTerm(1, [1], 1.0)                                    # orders must add up to len(eta_orders)
    assert len(list(taylor._set_partitions([0, 1, 2, 3]))) == 15


@pytest.mark.parametrize('lb,ub', [(-np.inf, np.inf), (0.5, np.inf), (-np.inf, 2.0), (-1.0, 3.0)])
def test_box_map_derivatives_against_nested_ad(lb, ub):
    D = 5
    rng = np.random.default_rng(3)
    phi = rng.normal(size=D)
    blocks = [dict(kind=0, free_size=D, vec_size=D, dim0=D, dim1=0, lb=lb, ub=ub)]
    got = taylor.box_map_derivatives(phi, blocks, 6)
    block = opk.box_block(D, lb=lb, ub=ub)
    f = lambda t: tr.constrain_block(t, block)
    x = torch.tensor(phi)
    ones = torch.ones(D, dtype=torch.float64)
    np.testing.assert_allclose(got[0], f(x).numpy(), rtol=1e-14)
    g = f
    for m in range(1, 7):                                                     # element-wise map: m nested JVPs along 1
        g = (lambda h: (lambda t: torch.func.jvp(h, (t,), (ones,))[1]))(g)
        np.testing.assert_allclose(got[m], g(x).numpy(), rtol=1e-11, atol=1e-13)
    with pytest.raises(NotImplementedError):
        taylor.box_map_derivatives(phi[:3], [dict(kind=1, free_size=3, vec_size=3, dim0=2, dim1=0, lb=0.0, ub=np.inf)], 2)


@pytest.mark.parametrize('loss', [om.GAUSSIAN, om.LOGISTIC, om.POISSON])
def test_oracle_directional_derivatives_against_nested_jvps(loss):
    rng = np.random.default_rng(40 + loss)
    N, P, V = 60, 5, 8
    lay = opk.Layout([opk.box_block(2), opk.box_block(P), opk.box_block(1)])
    x, y, w = glm_data(rng, N, P, loss)
    A = rng.normal(size=(V, V)); A = A @ A.T / V + np.eye(V)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, glm_off=2, lik_info=1.3, quad_A=A,
                             quad_m=rng.normal(size=V), quad_b=rng.normal(size=V))
    f = tr.make_objective(model)
    grad = torch.func.grad(f)
    eta = rng.normal(size=V) * 0.3
    te = torch.tensor(eta)
    U = rng.normal(size=(5, V))
    g = grad
    np.testing.assert_allclose(model.dk_grad_vec(eta), g(te).numpy(), rtol=1e-12, atol=1e-12)
    for j in range(1, 6):
        u = torch.tensor(U[j - 1])
        g = (lambda h, d: (lambda t: torch.func.jvp(h, (t,), (d,))[1]))(g, u)
        want = g(te).numpy()
        got = model.dk_grad_vec(eta, U[:j])
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-11 * max(1.0, np.max(np.abs(want))))
    # a direction in weight space: the objective is linear in the weights
    dw = rng.normal(size=N)
    m2 = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=dw, glm_off=2, lik_info=1.3)
    np.testing.assert_allclose(model.dk_grad_vec(eta, U[:2], w_override=dw, include_quad=False), m2.dk_grad_vec(eta, U[:2]),
                               rtol=1e-13, atol=1e-13)
    # loss derivatives against the sigmoid recurrence and the closed forms
    z = rng.normal(size=7)
    for m in range(1, 8):
        if loss == om.POISSON:
            want = np.exp(z) - (y[:7] if m == 1 else 0.0)
            np.testing.assert_allclose(om.loss_derivative(loss, m, y[:7], z), want, rtol=1e-14)


# ---- the class itself on the CPU: oracle arithmetic behind the device-functor protocol (tests/oracle_functor.py) ----
def _oracle_setup(loss, spec_blocks, N, seed):
    from oracle_functor import OracleFunctor
    rng = np.random.default_rng(seed)
    par = vb.ModelParamsDict('p')
    blocks = []
    for name, n, lb, ub in spec_blocks:
        par.push_param(vb.VectorParam(name, n, lb=lb, ub=ub))
        blocks.append(opk.box_block(n, lb=lb, ub=ub))
    lay = opk.Layout(blocks)
    P = lay.V
    x, y, w = glm_data(rng, N, P, loss)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w.copy(), quad_A=np.full(P, 0.8), quad_b=np.zeros(P))
    wpar = vb.VectorParam('weights', N, val=w.copy())
    tpar = vb.VectorParam('tilt', P, val=np.zeros(P))
    fun = OracleFunctor(par, model, weights_par=wpar, tilt_par=tpar)
    return rng, par, lay, model, fun, w


@pytest.mark.parametrize('loss', [om.LOGISTIC, om.POISSON])
def test_taylor_class_against_refits(loss):
    """d^k phi_hat / d w^k along dw, k = 1..3, against 8th-order central differences of the refitted optimum."""
    spec = [('u', 2, -np.inf, np.inf), ('lo', 2, -1.0, np.inf), ('hi', 1, -np.inf, 2.0), ('both', 2, -2.0, 3.0)]
    rng, par, lay, model, fun, w0 = _oracle_setup(loss, spec, 120, seed=10 + loss)

    def optimum(w, start):
        model.w = w
        phi = start.copy()
        for _ in range(200):                                   # Newton with step halving on the value
            step = np.linalg.solve(model.hessian(phi), model.grad(phi))
            t, f0 = 1.0, model.value(phi)
            while not model.value(phi - t * step) <= f0 and t > 1e-8:
                t *= 0.5
            phi = phi - t * step
            if np.max(np.abs(step)) < 1e-14:
                break
        model.w = w0
        return phi

    phi0 = optimum(w0, np.zeros(lay.D))
    assert np.linalg.norm(model.grad(phi0)) < 1e-10
    tay = vb.ParametricSensitivityTaylorExpansion(fun, par, fun.weights_par, phi0, w0, 3)
    dw = rng.normal(size=w0.size) * 0.5
    h = 2e-2
    path = {m: optimum(w0 + m * h * dw, phi0) for m in range(-4, 5)}
    # central differences of 8th / 8th / 6th order for the first three derivatives
    d1 = (672 * (path[1] - path[-1]) - 168 * (path[2] - path[-2]) + 32 * (path[3] - path[-3]) - 3 * (path[4] - path[-4])) / (840 * h)
    d2 = (-14350 * path[0] + 8064 * (path[1] + path[-1]) - 1008 * (path[2] + path[-2]) + 128 * (path[3] + path[-3])
          - 9 * (path[4] + path[-4])) / (5040 * h * h)
    d3 = (-488 * (path[1] - path[-1]) + 338 * (path[2] - path[-2]) - 72 * (path[3] - path[-3]) + 7 * (path[4] - path[-4])) / (240 * h ** 3)
    for k, want, tol in ((1, d1, 1e-8), (2, d2, 1e-6), (3, d3, 1e-4)):
        got = tay.evaluate_dkinput_dhyperk(dw, k)
        assert np.max(np.abs(got - want)) < tol * max(1.0, np.max(np.abs(want))), (k, got, want)
    # the series at a finite step beats the linear approximation by orders of magnitude
    t = 0.1
    truth = optimum(w0 + t * dw, phi0)
    e1 = np.max(np.abs(tay.evaluate_taylor_series(t * dw, max_order=1) - truth))
    e3 = np.max(np.abs(tay.evaluate_taylor_series(t * dw, max_order=3) - truth))
    assert e3 < 1e-2 * e1
    np.testing.assert_allclose(par.get_free(), phi0, rtol=1e-12, atol=1e-14)    # base values restored (free -> vector -> free)


def test_taylor_class_tilt_and_vector_coordinates():
    spec = [('a', 3, -10.0, np.inf)]
    rng, par, lay, model, fun, w0 = _oracle_setup(om.GAUSSIAN, spec, 40, seed=3)
    H = model.hessian_vec(np.zeros(3))                                   # Gaussian loss: constant in eta
    lam0 = rng.normal(size=3)
    fun.tilt_par.set_vector(lam0)
    model.quad_b = lam0.copy()
    eta0 = -np.linalg.solve(H, model.grad_vec(np.zeros(3)))               # quadratic objective: one Newton step from 0
    par.set_vector(eta0)
    phi0 = par.get_free()
    d = rng.normal(size=3)
    b = -np.linalg.solve(H, d)                                            # eta_hat is linear in the tilt
    tay = vb.ParametricSensitivityTaylorExpansion(fun, par, fun.tilt_par, phi0, lam0, 4)
    ratio = b / (eta0 + 10.0)
    for k in range(1, 5):
        want = (-1.0) ** (k - 1) * math.factorial(k - 1) * ratio ** k
        np.testing.assert_allclose(tay.evaluate_dkinput_dhyperk(d, k), want, rtol=1e-8, atol=1e-12)
    tv = vb.ParametricSensitivityTaylorExpansion(fun, par, fun.tilt_par, eta0, lam0, 3, input_is_free=False)
    np.testing.assert_allclose(tv.evaluate_dkinput_dhyperk(d, 1), b, rtol=1e-10)
    assert np.max(np.abs(tv.evaluate_dkinput_dhyperk(d, 2))) < 1e-10
    with pytest.raises(ValueError):
        tay.evaluate_dkinput_dhyperk(d, 5)
    with pytest.raises(ValueError):
        tay.evaluate_taylor_series(d[:2])


# ---- layouts with PSD and simplex blocks: multivariate Faa di Bruno through the packing-map jets ------------------
def _mixed_setup(seed):
    """Logistic GLM whose coefficient slice is a box block, next to a log-Cholesky PSD block and a simplex block that
    enter through a dense quadratic term (so every block's higher map derivatives carry weight)."""
    from oracle_functor import OracleFunctor
    from helpers import make_par
    rng = np.random.default_rng(seed)
    spec = [('box', 'beta', 4, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3), ('box', 'u', 2, -np.inf, np.inf)]
    par, lay = make_par(vb, spec)
    N, P = 150, 4
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    a = rng.normal(size=(lay.V, lay.V)); A = a @ a.T / lay.V + 2.0 * np.eye(lay.V)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w.copy(), glm_off=0, quad_A=A,
                             quad_m=lay.constrain(rng.normal(size=lay.D) * 0.3), quad_b=np.zeros(lay.V))
    wpar = vb.VectorParam('weights', N, val=w.copy())
    tpar = vb.VectorParam('tilt', lay.V, val=np.zeros(lay.V))
    return rng, par, lay, model, OracleFunctor(par, model, weights_par=wpar, tilt_par=tpar), w


def _optimum(model, start):
    import scipy.optimize
    fit = scipy.optimize.minimize(model.value, start, jac=model.grad, hess=model.hessian, method='trust-exact', options={'gtol': 1e-13})
    phi = fit.x
    for _ in range(3):
        phi = phi - np.linalg.solve(model.hessian(phi), model.grad(phi))
    return phi


def test_packing_jet_against_nested_ad():
    import torch_ref as tr
    rng, par, lay, model, fun, w0 = _mixed_setup(21)
    from lrvb_amd.taylor import PackingJet
    phi = rng.normal(size=lay.D) * 0.4
    jet = PackingJet(phi, par.layout_blocks())
    np.testing.assert_allclose(jet.vec([]), lay.constrain(phi), rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(jet.mat([]), lay.jac(phi), rtol=1e-13, atol=1e-14)
    dirs = [rng.normal(size=lay.D) for _ in range(3)]
    f = lambda p: tr.constrain(p, lay)
    for m in (1, 2, 3):
        g = f
        for w in dirs[:m]:
            g = (lambda h, wt: (lambda p: torch.func.jvp(h, (p,), (wt,))[1]))(g, torch.tensor(w))
        np.testing.assert_allclose(jet.vec(dirs[:m]), g(torch.tensor(phi)).numpy(), rtol=1e-12, atol=1e-13)
    # the open slot closes onto the next directional derivative
    np.testing.assert_allclose(jet.mat(dirs[:2]) @ dirs[2], jet.vec(dirs[:3]), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('hyper', ['weights', 'tilt'])
def test_taylor_class_with_psd_and_simplex_blocks(hyper):
    """d^k/dt^k of the optimality condition along the Taylor polynomial vanishes for k <= K (exact nested AD of an
    independent torch restatement), in FREE coordinates of a layout that is not element-wise."""
    import torch_ref as tr
    rng, par, lay, model, fun, w0 = _mixed_setup(22)
    phi0 = _optimum(model, np.zeros(lay.D))
    assert np.linalg.norm(model.grad(phi0)) < 1e-10
    K = 3
    hyper_par = fun.weights_par if hyper == 'weights' else fun.tilt_par
    hyper0 = w0 if hyper == 'weights' else np.zeros(lay.V)
    tay = vb.ParametricSensitivityTaylorExpansion(fun, par, hyper_par, phi0, hyper0, K)
    de = rng.normal(size=hyper0.size) * 0.5
    derivs = [tay.evaluate_dkinput_dhyperk(de, k) for k in range(1, K + 1)]
    # first order = the linear-response formula with the oracle's dense matrices
    cross = model.obs_grad(phi0).T if hyper == 'weights' else model.cross_hessian_tilt(phi0)
    np.testing.assert_allclose(derivs[0], -np.linalg.solve(model.hessian(phi0), cross @ de), rtol=1e-8, atol=1e-10)
    f_vec = tr.make_objective(model)
    tx, ty = torch.tensor(model.x), torch.tensor(model.y)
    A, mq = torch.tensor(model.quad_A), torch.tensor(model.quad_m)
    coefs = [torch.tensor(phi0)] + [torch.tensor(dk / math.factorial(k)) for k, dk in enumerate(derivs, start=1)]
    tde, th0 = torch.tensor(de), torch.tensor(hyper0)

    def f_free(phi, eps):
        eta = tr.constrain(phi, lay)
        z = tx @ eta[:model.P]
        l = torch.nn.functional.softplus(z) - ty * z
        d = eta - mq
        wts = eps if hyper == 'weights' else torch.tensor(w0)
        b = eps if hyper == 'tilt' else torch.zeros(lay.V, dtype=torch.float64)
        return torch.sum(wts * l) + 0.5 * torch.dot(d, A @ d) + torch.dot(b, eta)

    def residual(t):
        phi = sum(c * t ** k for k, c in enumerate(coefs))
        return torch.func.grad(f_free)(phi, th0 + t * tde)

    t0, one = torch.zeros((), dtype=torch.float64), torch.ones((), dtype=torch.float64)
    scale = np.linalg.norm(model.hessian(phi0) @ derivs[0])
    h = residual
    assert torch.linalg.norm(h(t0)).item() < 1e-9 * scale
    for k in range(1, K + 1):
        h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
        assert torch.linalg.norm(h(t0)).item() < 1e-7 * scale * math.factorial(k), k
    h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
    assert torch.linalg.norm(h(t0)).item() > 1e-6 * scale          # one order further it does not vanish
    np.testing.assert_allclose(par.get_free(), phi0, rtol=1e-10, atol=1e-12)


def test_append_jvp_and_derivative_array_counterparts():
    """LRVB/ModelSensitivity.py:38-62, 221-234 for closures written with torch operations: the (i, j) entry of the array
    is D_x1^i D_x2^j fun contracted with its trailing direction arguments -- checked on a polynomial with known
    derivatives, then used through DerivativeTerm.evaluate as the reference class uses it."""
    from lrvb_amd.taylor import append_jvp, generate_two_term_derivative_array, DerivativeTerm, get_taylor_base_terms
    a = torch.tensor([0.7, -1.2, 0.4], dtype=torch.float64)

    def fun(x1, x2):                                  # vector valued: x1^3 * (a . x2)  +  x1 * x2^2
        return x1 ** 3 * torch.dot(a, x2) + x1 * x2 ** 2
    x1 = torch.tensor([0.5, -0.3, 1.1], dtype=torch.float64); x2 = torch.tensor([0.2, 0.9, -0.6], dtype=torch.float64)
    v = torch.tensor([1.0, 0.5, -0.2], dtype=torch.float64); w = torch.tensor([-0.4, 0.3, 0.8], dtype=torch.float64)
    arr = generate_two_term_derivative_array(fun, 3)
    assert len(arr) == 3 and all(len(row) == 4 for row in arr)
    np.testing.assert_allclose(arr[0][0](x1, x2).numpy(), fun(x1, x2).numpy())
    np.testing.assert_allclose(arr[1][0](x1, x2, v).numpy(), (3 * x1 ** 2 * v * torch.dot(a, x2) + v * x2 ** 2).numpy(), rtol=1e-13)
    np.testing.assert_allclose(arr[0][1](x1, x2, w).numpy(), (x1 ** 3 * torch.dot(a, w) + 2 * x1 * x2 * w).numpy(), rtol=1e-13)
    np.testing.assert_allclose(arr[1][1](x1, x2, v, w).numpy(), (3 * x1 ** 2 * v * torch.dot(a, w) + 2 * v * x2 * w).numpy(), rtol=1e-13)
    np.testing.assert_allclose(arr[2][0](x1, x2, v, v).numpy(), (6 * x1 * v * v * torch.dot(a, x2)).numpy(), rtol=1e-13)
    np.testing.assert_allclose(arr[0][2](x1, x2, w, w).numpy(), (2 * x1 * w * w).numpy(), rtol=1e-13)
    jv = append_jvp(fun, num_base_args=2, argnum=0)
    np.testing.assert_allclose(jv(x1, x2, v).numpy(), arr[1][0](x1, x2, v).numpy(), rtol=1e-14)
    # the base terms evaluate through the array: d/dt g = D_eps g [d eps] + D_eta g [eta^(1)]
    terms = get_taylor_base_terms(eval_g_derivs=arr)
    eta1 = lambda e0, p0, dp: v
    terms = [DerivativeTerm(t.eps_order, t.eta_orders, t.prefactor, [eta1], arr) for t in terms]
    total = sum(t.evaluate(x1, x2, w) for t in terms)
    np.testing.assert_allclose(total.numpy(), (arr[0][1](x1, x2, w) + arr[1][0](x1, x2, v)).numpy(), rtol=1e-13)
    second = []
    for t in terms:
        second += t.differentiate(eval_next_eta_deriv=lambda e0, p0, dp: w)
    assert all(len(t.eval_eta_derivs) == 2 for t in second) and sorted(t.order for t in second) == [2] * len(second)


@pytest.mark.parametrize('hyper_is_free', [False, True])
@pytest.mark.parametrize('kind', ['prior_mean', 'prior_info', 'quad_scale', 'lik_info', 'tilt'])
def test_taylor_class_with_prior_hyper_parameters(kind, hyper_is_free):
    """The Taylor class for the OTHER declared hyper-parameters (LRVB/ModelSensitivity.py:382-515 takes any hyper_par), in
    vector and in free coordinates of the hyper-parameter: d^k/dt^k of the optimality condition along the Taylor polynomial
    vanishes for k <= K, by exact nested AD of an independent torch restatement with the hyper-parameter as a variable."""
    import torch_ref as tr
    from oracle_functor import OracleFunctor
    from helpers import make_par
    rng = np.random.default_rng(90)
    spec = [('box', 'beta', 4, -1.0, np.inf), ('psd', 'm', 2, 0.2), ('box', 'u', 2, -np.inf, np.inf)]
    par, lay = make_par(vb, spec)
    N, P = 80, 4
    x = rng.normal(size=(N, P)) * 0.6
    y = rng.normal(size=N)
    a = rng.normal(size=(lay.V, lay.V)); A = a @ a.T / lay.V + 2.0 * np.eye(lay.V)
    dense = kind == 'prior_info' and not hyper_is_free
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, w=rng.uniform(0.5, 1.5, N), lik_info=1.4,
                             quad_A=A if dense else np.diag(A).copy(), quad_m=lay.constrain(rng.normal(size=lay.D) * 0.3),
                             quad_b=rng.normal(size=lay.V) * 0.1, quad_scale=0.7)
    h_vec0 = model.hyper_value(kind)
    lb, ub = {'prior_mean': (-6.0, 6.0), 'tilt': (-np.inf, 2.0)}.get(kind, (0.0, np.inf))
    if not hyper_is_free:
        lb, ub = -np.inf, np.inf
    hp = vb.VectorParam(kind, h_vec0.size, lb=lb, ub=ub, val=h_vec0.copy())
    fun = OracleFunctor(par, model, **{kind + '_par': hp})
    phi0 = _optimum(model, np.zeros(lay.D))
    assert np.linalg.norm(model.grad(phi0)) < 1e-9
    K = 3
    hyper0 = hp.get_free().copy() if hyper_is_free else h_vec0.copy()
    tay = vb.ParametricSensitivityTaylorExpansion(fun, par, hp, phi0, hyper0, K, hyper_is_free=hyper_is_free)
    de = rng.normal(size=hyper0.size) * (0.3 if kind != 'prior_info' else 0.2)
    derivs = [tay.evaluate_dkinput_dhyperk(de, k) for k in range(1, K + 1)]
    np.testing.assert_allclose(hp.get_vector(), h_vec0, rtol=1e-12)              # hyper-parameter left at the base value
    f = tr.make_hyper_objective(model, kind)
    coefs = [torch.tensor(phi0)] + [torch.tensor(dk / math.factorial(k)) for k, dk in enumerate(derivs, start=1)]
    tde, th0 = torch.tensor(de), torch.tensor(hyper0)

    def c_h(e):                                              # the hyper-parameter's own packing map
        if not hyper_is_free:
            return e
        if np.isinf(ub):
            return torch.exp(e) + lb
        if np.isinf(lb):
            return ub - torch.exp(-e)
        return (ub - lb) * torch.sigmoid(e) + lb

    def residual(t):
        phi = sum(c * t ** k for k, c in enumerate(coefs))
        return torch.func.grad(f, argnums=0)(phi, c_h(th0 + t * tde))

    t0, one = torch.zeros((), dtype=torch.float64), torch.ones((), dtype=torch.float64)
    scale = np.linalg.norm(model.hessian(phi0) @ derivs[0])
    assert scale > 1e-3
    h = residual
    assert torch.linalg.norm(h(t0)).item() < 1e-8 * max(scale, 1.0)
    for k in range(1, K + 1):
        h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
        assert torch.linalg.norm(h(t0)).item() < 1e-7 * scale * math.factorial(k), k
    h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
    assert torch.linalg.norm(h(t0)).item() > 1e-7 * scale          # one order further it does not vanish
