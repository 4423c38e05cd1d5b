"""`TorchObjective`: an objective written as a closure in torch operations of the vector-coordinate tensor -- the route for
opaque objectives above the 64 parameters of the Richardson fallback (round-3 verdict, "missing" item 6; the reference
differentiates any closure with autograd, LRVB/SparseObjectives.py:95-116).  torch.func forms the vector-coordinate
derivatives, the library converts to free coordinates.  Checked against exact AD of the composed map free -> vector ->
value (tests/torch_ref.py restates the packing maps), D = 318 with bounded boxes and two log-Cholesky blocks."""
import numpy as np
import pytest
import torch

import torch_ref as tr
from helpers import make_par, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _problem(vb, seed=0):
    rng = np.random.default_rng(seed)
    spec = [('box', 'a', 40, 0.0, np.inf), ('psd', 'm1', 20, 0.0), ('box', 'b', 30, -1.0, 2.0), ('psd', 'm2', 6, 0.2), ('box', 'c', 17, -np.inf, np.inf)]
    par, lay = make_par(vb, spec)
    V = lay.V
    A = rng.normal(size=(V, V)); A = A @ A.T / V + np.eye(V)
    B = rng.normal(size=(25, V)) / np.sqrt(V)
    At = {'cpu': torch.tensor(A), 'B': torch.tensor(B)}

    def closure_on(dev):
        Ad, Bd = At['cpu'].to(dev), At['B'].to(dev)
        return lambda eta, scale=1.0: scale * (0.5 * eta @ (Ad @ eta) + torch.logsumexp(Bd @ eta, 0) + torch.sum(torch.cos(eta)))
    return rng, par, lay, closure_on


def test_derivatives_match_ad_of_the_composed_map(vb):
    rng, par, lay, closure_on = _problem(vb)
    assert lay.D == 40 + 210 + 30 + 21 + 17
    fun = vb.TorchObjective(par, closure_on(torch.device('cuda', 0)))
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=lay.D) * 0.3
    f_cpu = closure_on(torch.device('cpu'))
    comp = lambda t, scale=1.0: f_cpu(tr.constrain(t, lay), scale)
    tt = torch.tensor(theta)
    assert abs(obj.fun_free(theta) - comp(tt).item()) < 1e-12 * abs(comp(tt).item())
    assert rel_err(obj.fun_free_grad(theta), torch.func.grad(comp)(tt).numpy()) < 1e-12
    Hw = torch.func.hessian(comp)(tt).numpy()
    assert rel_err(obj.fun_free_hessian(theta), Hw) < 1e-12
    v = rng.normal(size=lay.D)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < 1e-12
    # keyword arguments travel to the closure, as the reference passes them on
    assert rel_err(obj.fun_free_grad(theta, scale=3.0), 3.0 * torch.func.grad(comp)(tt).numpy()) < 1e-12
    # vector coordinates
    eta = lay.constrain(theta)
    et = torch.tensor(eta)
    assert rel_err(obj.fun_vector_grad(eta), torch.func.grad(f_cpu)(et).numpy()) < 1e-12
    assert rel_err(obj.fun_vector_hessian(eta), torch.func.hessian(f_cpu)(et).numpy()) < 1e-12
    assert rel_err(obj.fun_vector_hvp(eta, v), torch.func.hessian(f_cpu)(et).numpy() @ v) < 1e-12
    # the side-effect contract: par holds the evaluation point
    assert np.max(np.abs(par.get_vector() - eta)) < 1e-12


def test_jacobian_of_a_tensor_valued_closure_and_the_lrvb_covariance(vb):
    rng, par, lay, closure_on = _problem(vb, seed=3)
    dev = torch.device('cuda', 0)
    W = torch.tensor(rng.normal(size=(3, 2, lay.V)), device=dev)
    fun = vb.TorchObjective(par, lambda eta: torch.tanh(W @ eta))
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=lay.D) * 0.3
    Wc = W.cpu()
    want = torch.func.jacrev(lambda t: torch.tanh(Wc @ tr.constrain(t, lay)))(torch.tensor(theta)).numpy()
    got = obj.fun_free_jacobian(theta)
    assert got.shape == (3, 2, lay.D)
    assert rel_err(got, want) < 1e-12
    # the linear-response covariance of a scalar objective through the device factorisation (north_star's get_lrvb_cov)
    fun2 = vb.TorchObjective(par, closure_on(dev))
    obj2 = vb.Objective(par, fun2)
    H = obj2.fun_free_hessian(theta)
    Hs = H + (1.0 - min(0.0, np.linalg.eigvalsh(H).min())) * np.eye(lay.D)        # not at an optimum: shifted for the factorisation
    M = rng.normal(size=(4, lay.D))
    cov = vb.ModelSensitivity.get_lrvb_cov(obj2, theta, M, kl_hessian=Hs)
    assert rel_err(cov, M @ np.linalg.solve(Hs, M.T)) < 1e-9


def test_hyper_parameter_of_a_torch_closure_and_the_linear_response(vb):
    """A closure of (eta, eps): the cross Hessian d2 f / d theta d eps^T and d f / d eps against exact AD of the composed map, in
    free and vector coordinates of both; `ParametricSensitivityLinearApproximation` over it predicts the refit under a moved
    hyper-parameter to second order (the error falls by ~4 when the step halves) -- the reference's TwoParameterObjective /
    sensitivity classes on an arbitrary objective (LRVB/SparseObjectives.py:321-449, ModelSensitivity.py:555-612) at D = 97."""
    import scipy.optimize
    rng = np.random.default_rng(11)
    spec = [('box', 'a', 30, 0.0, np.inf), ('psd', 'm1', 10, 0.0), ('box', 'c', 12, -np.inf, np.inf)]
    par, lay = make_par(vb, spec)
    V = lay.V
    hyper = vb.VectorParam('eps', 7, lb=0.0)
    hyper.set_vector(np.linspace(0.5, 1.5, 7))
    A = rng.normal(size=(V, V)); A = A @ A.T / V + np.eye(V)
    Wm = rng.normal(size=(V, 7)) / np.sqrt(V)
    B = rng.normal(size=(25, V)) / np.sqrt(V)
    dev = torch.device('cuda', 0)

    def closure_on(d):
        Ad, Wd, Bd = torch.tensor(A, device=d), torch.tensor(Wm, device=d), torch.tensor(B, device=d)
        return lambda eta, eps: 0.5 * eta @ (Ad @ eta) + torch.logsumexp(Bd @ eta, 0) - eta @ (Wd @ torch.log(eps)) + 0.1 * torch.sum(eps * eps)
    fun = vb.TorchObjective(par, closure_on(dev), hyper_par=hyper)
    two = vb.TwoParameterObjective(par, hyper, fun)
    f_cpu = closure_on(torch.device('cpu'))
    hlay = [('box', 'eps', 7, 0.0, np.inf)]
    _, hl = make_par(vb, hlay)
    theta = rng.normal(size=lay.D) * 0.2
    ef = hyper.get_free()
    comp = lambda t, e: f_cpu(tr.constrain(t, lay), tr.constrain(e, hl))
    tt, te = torch.tensor(theta), torch.tensor(ef)
    want12 = torch.func.jacfwd(torch.func.grad(comp, argnums=0), argnums=1)(tt, te).numpy()
    assert rel_err(two.fun_free_hessian12(theta, ef), want12) < 1e-11
    ev = hyper.get_vector()
    want12v = torch.func.jacfwd(torch.func.grad(lambda t, e: f_cpu(tr.constrain(t, lay), e), argnums=0), argnums=1)(tt, torch.tensor(ev)).numpy()
    assert rel_err(two.fun_hessian_free1_vector2(theta, ev), want12v) < 1e-11
    assert rel_err(two.fun_grad2(theta, ef, True, True), torch.func.grad(comp, argnums=1)(tt, te).numpy()) < 1e-11
    # linear response on a second model (boxes only, strongly convex in eta): fit, move eps, predict, refit
    par2, lay2 = make_par(vb, [('box', 'u', 40, -np.inf, np.inf), ('box', 'pos', 20, 0.0, np.inf)])
    V2 = lay2.V
    A2 = rng.normal(size=(V2, V2)); A2 = A2 @ A2.T / V2 + 2.0 * np.eye(V2)
    W2 = rng.normal(size=(V2, 7)) / np.sqrt(7.0)
    B2 = rng.normal(size=(15, V2)) / np.sqrt(V2)
    A2d, W2d, B2d = torch.tensor(A2, device=dev), torch.tensor(W2, device=dev), torch.tensor(B2, device=dev)
    shift = torch.tensor(np.concatenate([np.zeros(40), np.full(20, 1.5)]), device=dev)

    def model(eta, eps):
        r = eta - shift - W2d @ torch.log(eps)
        return 0.5 * r @ (A2d @ r) + torch.logsumexp(B2d @ eta, 0)
    fun2 = vb.TorchObjective(par2, model, hyper_par=hyper)
    obj = vb.Objective(par2, fun2)

    def fit(start):
        res = scipy.optimize.minimize(obj.fun_free, start, jac=obj.fun_free_grad, hess=obj.fun_free_hessian, method='trust-exact', options={'gtol': 1e-9})
        x = res.x
        for _ in range(2):
            x = x - np.linalg.solve(obj.fun_free_hessian(x), obj.fun_free_grad(x))
        return x
    theta0 = fit(np.zeros(lay2.D))
    assert np.max(np.abs(obj.fun_free_grad(theta0))) < 1e-8
    sens = vb.ParametricSensitivityLinearApproximation(fun2, par2, hyper, theta0, ev.copy(), hyper_is_free=False)
    direction = rng.normal(size=7) * 0.03
    errs = []
    for step in (1.0, 0.5):
        new = ev + step * direction
        pred = sens.predict_input_par_from_hyperparameters(new)
        hyper.set_vector(new)
        refit = fit(theta0)
        errs.append(np.linalg.norm(pred - refit))
        move = np.linalg.norm(refit - theta0)
        hyper.set_vector(ev)
    assert errs[1] < 0.05 * move and 2.5 < errs[0] / errs[1] < 6.0
