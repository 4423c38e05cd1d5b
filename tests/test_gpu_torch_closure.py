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
