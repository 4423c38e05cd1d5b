"""SURVEY.md section 8(f) item 4 on the GPU: `lrvb_dk_grad_vec` against the oracle (itself pinned by nested forward-mode
AD in tests/test_taylor_host_math.py) and `ParametricSensitivityTaylorExpansion` against (i) the closed-form optimum of
the reference's own test model, a quadratic with a linear tilt and lower-bounded parameters
(LRVB/test_model_sensitivity.py:35-90, TestTaylorExpansion :91-364), and (ii) the defining identity
d^k/dt^k g(eta_K(t), eps0 + t d eps) = 0, k <= K, checked by exact nested AD of a torch restatement."""
import math

import numpy as np
import pytest
import scipy.optimize
import torch

import torch_ref as tr
from oracle import models as om
from helpers import make_par, glm_data, rel_err, LOSS_NAME

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


@pytest.mark.parametrize('loss,N,P', [(om.GAUSSIAN, 211, 6), (om.LOGISTIC, 300, 7), (om.POISSON, 257, 8), (om.LOGISTIC, 1000, 130),
                                      (om.POISSON, 300, 1030), (om.LOGISTIC, 5, 2)])
def test_directional_derivatives_of_the_gradient(vb, loss, N, P):
    """Odd P takes the generic product for z and X u_k, even P the fused multi-vector pass; P = 1030 the wide-design
    passes; N = 5 a single partial chunk."""
    rng = np.random.default_rng(N + P)
    spec = [('box', 'pre', 2, -np.inf, np.inf), ('box', 'beta', P, -np.inf, np.inf), ('box', 'post', 1, -np.inf, np.inf)]
    par, lay = make_par(vb, spec)
    V = lay.V
    x, y, w = glm_data(rng, N, P, loss)
    A = rng.normal(size=(V, V)); A = A @ A.T / V + np.eye(V)
    qm, qb = rng.normal(size=V), rng.normal(size=V)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], glm_param='beta', lik_info=1.3, quad_A=A, quad_m=qm,
                             quad_b=qb, weights=w)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, glm_off=2, lik_info=1.3, quad_A=A, quad_m=qm, quad_b=qb)
    eta = rng.normal(size=V) * 0.3
    U = rng.normal(size=(6, V)) * 0.7
    dw = rng.normal(size=N)
    fun._push_state()
    for j in range(0, 7):
        want = model.dk_grad_vec(eta, U[:j])
        assert rel_err(fun.ctx.dk_grad_vec(eta, U[:j]), want) < 1e-11 or np.max(np.abs(want)) == 0.0
        want_w = model.dk_grad_vec(eta, U[:j], w_override=dw, include_quad=False)
        got_w = fun.ctx.dk_grad_vec(eta, U[:j], w_override=dw, include_quad=False)
        assert rel_err(got_w, want_w) < 1e-11 or np.max(np.abs(want_w)) == 0.0
    # the context's own weights are untouched by an override, and the entry point checks its arguments
    assert rel_err(fun.ctx.grad(eta, is_free=False), model.grad_vec(eta)) < 1e-12
    with pytest.raises(ValueError):
        fun.ctx.dk_grad_vec(eta[:-1])
    with pytest.raises(ValueError):
        fun.ctx.dk_grad_vec(eta, np.zeros((7, V)))
    with pytest.raises(ValueError):
        fun.ctx.dk_grad_vec(eta, U[:1], w_override=dw[:-1])


def test_taylor_expansion_on_the_reference_test_model(vb):
    """QuadraticModel of LRVB/test_model_sensitivity.py:35-90: theta lower-bounded at -10, objective
    1/2 theta^T M theta + lambda^T theta, optimum theta*(lambda) = -M^-1 lambda, free optimum log(theta* + 10).
    Along lambda0 + t d the vector optimum is a + t b, so d^k phi / dt^k = (-1)^(k-1) (k-1)! (b / (a + 10))^k exactly."""
    dim = 3
    vec = np.linspace(0.1, 0.3, num=dim)
    M = np.outer(vec, vec) + np.eye(dim)
    param = vb.VectorParam('theta', size=dim, lb=-10.0)
    lam0 = np.linspace(0.5, 10.0, num=dim)
    fun = vb.QuadraticObjective(param, A=M, b=lam0)
    theta0 = -np.linalg.solve(M, lam0)
    param.set_vector(theta0)
    phi0 = param.get_free()
    order = 4
    tay = vb.ParametricSensitivityTaylorExpansion(fun, param, fun.tilt_par, phi0, lam0, order)
    d = np.array([0.3, -0.2, 0.5])
    b = -np.linalg.solve(M, d)
    ratio = b / (theta0 + 10.0)
    for k in range(1, order + 1):
        want = (-1.0) ** (k - 1) * math.factorial(k - 1) * ratio ** k
        assert rel_err(tay.evaluate_dkinput_dhyperk(d, k), want) < 1e-9
    # the series: truncation error falls with the order, as in the reference's test (:300-364)
    lam1 = lam0 + 0.1
    param.set_vector(-np.linalg.solve(M, lam1))
    phi1 = param.get_free()
    errs = [np.max(np.abs(tay.evaluate_taylor_series(lam1 - lam0, max_order=k) - phi1)) for k in range(1, order + 1)]
    assert all(e2 < e1 for e1, e2 in zip(errs, errs[1:])) and errs[-1] < 1e-3 * errs[0]      # 4th order: remainder ~ t^5 / 5
    np.testing.assert_allclose(tay.evaluate_taylor_series(lam1 - lam0, add_offset=False) + phi0,
                               tay.evaluate_taylor_series(lam1 - lam0), rtol=1e-15)
    # side effect contract: the parameters sit at the base values afterwards
    np.testing.assert_allclose(param.get_free(), phi0, rtol=1e-15)
    np.testing.assert_allclose(fun.tilt_par.get_vector(), lam0, rtol=1e-15)
    # argument errors of the reference class (:476-481, 489-497)
    for bad in (0, order + 1):
        with pytest.raises(ValueError):
            tay.evaluate_dkinput_dhyperk(d, bad)
        with pytest.raises(ValueError):
            tay.evaluate_taylor_series(d, max_order=bad)
    tay.print_terms(2)
    # the reference class's helper methods (LRVB/ModelSensitivity.py:412-450) on the device path: the gradient being expanded
    # vanishes at the base point and equals d f / d phi elsewhere; the derivative function made from a term list is the
    # class's own k-th derivative, checked to be asked for at the base point; both leave the parameters at the base values
    assert np.max(np.abs(tay.objective_gradient(phi0, lam0))) < 1e-10
    g_shift = tay.objective_gradient(phi0 + 0.1, lam0)
    th = np.exp(phi0 + 0.1) - 10.0
    np.testing.assert_allclose(g_shift, (M @ th + lam0) * np.exp(phi0 + 0.1), rtol=1e-12)
    for k in (1, 2, 3):
        fk = tay.get_dkinput_dhyperk_from_terms(tay.taylor_terms_list[k - 1])
        np.testing.assert_allclose(fk(phi0, lam0, d), tay.evaluate_dkinput_dhyperk(d, k), rtol=1e-13)
    with pytest.raises(AssertionError):
        tay.get_dkinput_dhyperk_from_terms(tay.taylor_terms_list[0])(phi0 + 1e-3, lam0, d)
    assert tay.cache_and_eval(lambda a: a + 1, 2) == 3
    assert len(tay.differentiate_terms(tay.taylor_terms_list[0])) == len(tay.taylor_terms_list[1])
    np.testing.assert_allclose(param.get_free(), phi0, rtol=1e-15)
    # vector coordinates: the optimum is linear in lambda, every higher derivative vanishes
    tv = vb.ParametricSensitivityTaylorExpansion(fun, param, fun.tilt_par, theta0, lam0, 3, input_is_free=False)
    assert rel_err(tv.evaluate_dkinput_dhyperk(d, 1), b) < 1e-12
    assert np.max(np.abs(tv.evaluate_dkinput_dhyperk(d, 2))) < 1e-12 and np.max(np.abs(tv.evaluate_dkinput_dhyperk(d, 3))) < 1e-12


@pytest.mark.parametrize('loss', [om.LOGISTIC, om.POISSON])
def test_weight_sensitivity_to_fourth_order(vb, loss):
    """GLM with mixed box constraints, hyper-parameter = observation weights.  The Taylor polynomial phi_K(t) of the
    optimum must satisfy d^k/dt^k grad f(phi_K(t); w0 + t dw) = 0 at t = 0 for k = 0 .. K: checked with exact nested
    forward-mode AD of a torch restatement.  Also checked against re-optimisation at w0 + t dw."""
    rng = np.random.default_rng(77 + loss)
    spec = [('box', 'u', 3, -np.inf, np.inf), ('box', 'lo', 2, -1.0, np.inf), ('box', 'hi', 2, -np.inf, 2.0), ('box', 'both', 3, -2.0, 3.0)]
    par, lay = make_par(vb, spec)
    N, P = 400, lay.V
    x, y, w0 = glm_data(rng, N, P, loss)
    prior = np.full(P, 0.8)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], quad_A=prior, weights=w0)
    obj = vb.Objective(par, fun)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w0, quad_A=prior)
    fit = scipy.optimize.minimize(model.value, np.zeros(lay.D), jac=model.grad, hess=model.hessian, method='trust-exact',
                                  options={'gtol': 1e-13})
    phi0 = fit.x
    for _ in range(3):                                            # polish: Newton steps to machine precision
        phi0 = phi0 - np.linalg.solve(model.hessian(phi0), model.grad(phi0))
    assert np.linalg.norm(model.grad(phi0)) < 1e-10
    K = 4
    tay = vb.ParametricSensitivityTaylorExpansion(fun, par, fun.weights_par, phi0, w0, K)
    dw = rng.normal(size=N) * 0.5
    derivs = [tay.evaluate_dkinput_dhyperk(dw, k) for k in range(1, K + 1)]
    # first order against the linear-response class
    lin = vb.ParametricSensitivityLinearApproximation(fun, par, fun.weights_par, phi0, w0)
    assert rel_err(derivs[0], lin.get_dinput_dhyper() @ dw) < 1e-9

    # the optimality condition along the Taylor polynomial, by exact AD
    tx, ty = torch.tensor(x), torch.tensor(y)
    tw0, tdw, tprior = torch.tensor(w0), torch.tensor(dw), torch.tensor(prior)
    coefs = [torch.tensor(phi0)] + [torch.tensor(dk / math.factorial(k)) for k, dk in enumerate(derivs, start=1)]

    def f_free(phi, w):
        eta = tr.constrain(phi, lay)
        z = tx @ eta
        l = torch.nn.functional.softplus(z) - ty * z if loss == om.LOGISTIC else torch.exp(z) - ty * z
        return torch.sum(w * l) + 0.5 * torch.sum(tprior * eta * eta)

    def residual(t):
        phi = sum(c * t ** k for k, c in enumerate(coefs))
        return torch.func.grad(f_free)(phi, tw0 + t * tdw)

    t0, one = torch.zeros((), dtype=torch.float64), torch.ones((), dtype=torch.float64)
    h = residual
    scale = np.linalg.norm(obj.fun_free_hessian(phi0) @ derivs[0])
    assert torch.linalg.norm(h(t0)).item() < 1e-9 * scale
    for k in range(1, K + 1):
        h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
        # d^k/dt^k of the residual at 0 vanishes when the first k Taylor coefficients are right
        assert torch.linalg.norm(h(t0)).item() < 1e-7 * scale * math.factorial(k), k
    # and one more derivative does NOT vanish (the check has teeth)
    h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
    assert torch.linalg.norm(h(t0)).item() > 1e-6 * scale

    # against an actual refit at perturbed weights: the error of the K-th order series falls like t^(K+1)
    t = 0.2
    model.w = w0 + t * dw
    refit = phi0.copy()
    for _ in range(30):
        refit = refit - np.linalg.solve(model.hessian(refit), model.grad(refit))
    errs = [np.max(np.abs(tay.evaluate_taylor_series(t * dw, max_order=k) - refit)) for k in range(1, K + 1)]
    assert all(e2 < e1 for e1, e2 in zip(errs, errs[1:]))
    assert errs[-1] < 1e-3 * errs[0]


def test_free_coordinates_with_psd_and_simplex_blocks(vb):
    """Layouts that are not element-wise: the device supplies every O(N) leaf (lrvb_dk_grad_vec), the host the
    derivatives of the packing map; d^k phi_hat / d w^k for k <= 3 satisfies the optimality condition along the Taylor
    polynomial (exact nested AD), and the series beats the linear approximation against a refit."""
    rng = np.random.default_rng(31)
    spec = [('box', 'beta', 6, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3)]
    par, lay = make_par(vb, spec)
    N, P = 800, 6
    x, y, w0 = glm_data(rng, N, P, om.LOGISTIC)
    a = rng.normal(size=(lay.V, lay.V)); A = a @ a.T / lay.V + 2.0 * np.eye(lay.V)
    qm = lay.constrain(rng.normal(size=lay.D) * 0.3)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', glm_param='beta', quad_A=A, quad_m=qm, weights=w0)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w0.copy(), glm_off=0, quad_A=A, quad_m=qm)
    fit = scipy.optimize.minimize(model.value, np.zeros(lay.D), jac=model.grad, hess=model.hessian, method='trust-exact',
                                  options={'gtol': 1e-13})
    phi0 = fit.x
    for _ in range(3):
        phi0 = phi0 - np.linalg.solve(model.hessian(phi0), model.grad(phi0))
    assert np.linalg.norm(model.grad(phi0)) < 1e-10
    K = 3
    tay = vb.ParametricSensitivityTaylorExpansion(fun, par, fun.weights_par, phi0, w0, K)
    dw = rng.normal(size=N) * 0.5
    derivs = [tay.evaluate_dkinput_dhyperk(dw, k) for k in range(1, K + 1)]
    assert rel_err(derivs[0], -np.linalg.solve(model.hessian(phi0), model.obs_grad(phi0).T @ dw)) < 1e-8
    tx, ty, tA, tm = torch.tensor(x), torch.tensor(y), torch.tensor(A), torch.tensor(qm)
    coefs = [torch.tensor(phi0)] + [torch.tensor(dk / math.factorial(k)) for k, dk in enumerate(derivs, start=1)]
    tw0, tdw = torch.tensor(w0), torch.tensor(dw)

    def f_free(phi, w):
        eta = tr.constrain(phi, lay)
        z = tx @ eta[:P]
        d = eta - tm
        return torch.sum(w * (torch.nn.functional.softplus(z) - ty * z)) + 0.5 * torch.dot(d, tA @ d)

    def residual(t):
        phi = sum(c * t ** k for k, c in enumerate(coefs))
        return torch.func.grad(f_free)(phi, tw0 + t * tdw)

    t0, one = torch.zeros((), dtype=torch.float64), torch.ones((), dtype=torch.float64)
    scale = np.linalg.norm(model.hessian(phi0) @ derivs[0])
    h = residual
    assert torch.linalg.norm(h(t0)).item() < 1e-9 * scale
    for k in range(1, K + 1):
        h = (lambda g: (lambda t: torch.func.jvp(g, (t,), (one,))[1]))(h)
        assert torch.linalg.norm(h(t0)).item() < 1e-7 * scale * math.factorial(k), k
    t = 0.2
    model.w = w0 + t * dw
    refit = phi0.copy()
    for _ in range(30):
        refit = refit - np.linalg.solve(model.hessian(refit), model.grad(refit))
    errs = [np.max(np.abs(tay.evaluate_taylor_series(t * dw, max_order=k) - refit)) for k in range(1, K + 1)]
    assert errs[2] < errs[1] < errs[0] and errs[2] < 1e-2 * errs[0]


def test_refusals(vb):
    p2 = vb.VectorParam('x', 3)
    host = lambda: float(np.sum(p2.get() ** 2))
    with pytest.raises(NotImplementedError):
        vb.ParametricSensitivityTaylorExpansion(host, p2, p2, np.zeros(3), np.zeros(3), 2)
    f2 = vb.QuadraticObjective(p2, A=np.eye(3))
    with pytest.raises(NotImplementedError):          # a hyper-parameter the objective does not declare
        vb.ParametricSensitivityTaylorExpansion(f2, p2, vb.VectorParam('stranger', 3), np.zeros(3), np.zeros(3), 2)
    # free coordinates of the hyper-parameter are accepted since round 4 (an unbounded tilt: free = vector, same derivatives)
    tf = vb.ParametricSensitivityTaylorExpansion(f2, p2, f2.tilt_par, np.zeros(3), np.zeros(3), 2, hyper_is_free=True)
    tv = vb.ParametricSensitivityTaylorExpansion(f2, p2, f2.tilt_par, np.zeros(3), np.zeros(3), 2)
    de = np.array([0.3, -0.2, 0.1])
    np.testing.assert_allclose(tf.evaluate_dkinput_dhyperk(de, 1), tv.evaluate_dkinput_dhyperk(de, 1), rtol=1e-13)
    np.testing.assert_allclose(tv.evaluate_dkinput_dhyperk(de, 1), -de, rtol=1e-12)
    with pytest.raises(ValueError):
        vb.ParametricSensitivityTaylorExpansion(f2, p2, f2.tilt_par, np.zeros(3), np.zeros(3), 0)
