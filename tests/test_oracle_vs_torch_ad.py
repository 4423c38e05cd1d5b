"""The oracle's ANALYTIC derivatives against exact AD (torch.func, fp64) of an independent
restatement (tests/torch_ref.py).  Exact derivatives are unique, so this pins the oracle to what
the reference's autograd calls would return for the same function.  Tolerance 1e-12 relative."""
import numpy as np
import pytest
import torch

import torch_ref as tr
from oracle import packing as opk
from oracle import models as om


def _layout_mixed():
    return opk.Layout([opk.box_block(3), opk.box_block(2, lb=0.5), opk.box_block(2, ub=3.0),
                       opk.box_block(3, lb=-2, ub=5), opk.psd_block(3, diag_lb=0.3), opk.simplex_block(2, 4)])


def test_packing_forward_jacobian_third_order():
    rng = np.random.default_rng(0)
    lay = _layout_mixed()
    theta = rng.normal(size=lay.D) * 0.7
    t = torch.tensor(theta)
    np.testing.assert_allclose(lay.constrain(theta), tr.constrain(t, lay).numpy(), rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(lay.unconstrain(lay.constrain(theta)), theta, rtol=1e-12, atol=1e-13)
    J = torch.func.jacrev(lambda th: tr.constrain(th, lay))(t).numpy()
    np.testing.assert_allclose(lay.jac(theta), J, rtol=1e-13, atol=1e-15)
    g = rng.normal(size=lay.V)
    T = torch.func.hessian(lambda th: torch.dot(torch.tensor(g), tr.constrain(th, lay)))(t).numpy()
    np.testing.assert_allclose(lay.third_order(theta, g), T, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('loss', [om.GAUSSIAN, om.LOGISTIC, om.POISSON])
def test_declared_model_value_grad_hessian_hvp(loss):
    rng = np.random.default_rng(10 + loss)
    lay = opk.Layout([opk.box_block(2), opk.box_block(3, lb=-1.0), opk.psd_block(2), opk.simplex_block(1, 3)])
    N, P = 60, 5
    x = rng.normal(size=(N, P)) * 0.3
    w = rng.uniform(0.5, 1.5, N)
    y = {om.GAUSSIAN: rng.normal(size=N), om.LOGISTIC: rng.integers(0, 2, N).astype(float),
         om.POISSON: rng.poisson(1.0, N).astype(float)}[loss]
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T + np.eye(lay.V)
    m = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, glm_off=0, lik_info=1.7, quad_A=A,
                         quad_m=rng.normal(size=lay.V), quad_b=rng.normal(size=lay.V), quad_scale=0.6)
    theta = rng.normal(size=lay.D) * 0.5
    f = tr.make_free_objective(m)
    t = torch.tensor(theta)
    H = torch.func.hessian(f)(t).numpy()
    scale = np.max(np.abs(H))
    assert abs(f(t).item() - m.value(theta)) < 1e-12 * max(1.0, abs(m.value(theta)))
    np.testing.assert_allclose(m.grad(theta), torch.func.grad(f)(t).numpy(), rtol=1e-11, atol=1e-12 * scale)
    np.testing.assert_allclose(m.hessian(theta), H, rtol=0, atol=1e-12 * scale)
    v = rng.normal(size=lay.D)
    np.testing.assert_allclose(m.hvp(theta, v), H @ v, rtol=0, atol=1e-11 * scale)
    np.testing.assert_allclose(m.hessian_by_hvps(theta), H, rtol=0, atol=1e-12 * scale)
    # cross Hessian w.r.t. the weights = Jacobian of the gradient in w (Example.ipynb:425-441)
    def f_w(wt, th):
        mm = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=np.ones(N), lik_info=1.7)
        eta = tr.constrain(th, lay)
        z = torch.tensor(x) @ eta[:P]
        yt = torch.tensor(y)
        l = {1: 0.5 * 1.7 * (yt - z) ** 2, 2: torch.nn.functional.softplus(z) - yt * z, 3: torch.exp(z) - yt * z}[loss]
        return torch.sum(wt * l)
    cross = torch.func.jacrev(torch.func.grad(f_w, argnums=1), argnums=0)(torch.tensor(w), t).numpy()   # D x N
    np.testing.assert_allclose(m.obs_grad(theta).T, cross, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(cross))))


@pytest.mark.parametrize('dense', [False, True])
def test_hyper_parameter_cross_hessians_and_gradients(dense):
    """The closed forms of d2 f / d theta d eps^T and d f / d eps for every declared hyper-parameter (tilt, prior mean,
    prior information -- diagonal and symmetric-matrix vector form --, quadratic scale, likelihood precision) against exact
    AD of an independent restatement: what `jacobian(grad_1, argnum=hyper)` of LRVB/SparseObjectives.py:333-339 returns."""
    rng = np.random.default_rng(31 + dense)
    lay = opk.Layout([opk.box_block(2), opk.box_block(3, lb=-1.0), opk.psd_block(2), opk.simplex_block(1, 3)])
    N, P = 40, 5
    x = rng.normal(size=(N, P)) * 0.3
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T + np.eye(lay.V)
    m = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=rng.normal(size=N), w=rng.uniform(0.5, 1.5, N), lik_info=1.7,
                         quad_A=A if dense else np.diag(A).copy(), quad_m=rng.normal(size=lay.V),
                         quad_b=rng.normal(size=lay.V), quad_scale=0.6)
    m_id = om.DeclaredModel(opk.Layout([opk.box_block(lay.V)]), loss=om.GAUSSIAN, x=m.x, y=m.y, w=m.w, lik_info=1.7, quad_A=m.quad_A,
                            quad_m=m.quad_m, quad_b=m.quad_b, quad_scale=0.6)
    theta = rng.normal(size=lay.D) * 0.5
    t = torch.tensor(theta)
    for kind in ('tilt', 'prior_mean', 'prior_info', 'quad_scale', 'lik_info'):
        eps = torch.tensor(m.hyper_value(kind))
        f = tr.make_hyper_objective(m, kind)
        assert abs(f(t, eps).item() - m.value(theta)) < 1e-12 * max(1.0, abs(m.value(theta)))
        C = torch.func.jacrev(torch.func.grad(f, argnums=0), argnums=1)(t, eps).numpy()
        g = torch.func.grad(f, argnums=1)(t, eps).numpy()
        scale = max(np.max(np.abs(C)), 1.0)
        np.testing.assert_allclose(m.cross_hessian_hyper(kind, theta), C, rtol=0, atol=1e-12 * scale, err_msg=kind)
        np.testing.assert_allclose(m.hyper_grad(kind, theta), g, rtol=1e-12, atol=1e-12 * max(np.max(np.abs(g)), 1.0), err_msg=kind)
        # vector coordinates of the input: AD of the same function under the identity packing map
        eta = lay.constrain(theta)
        Cv = torch.func.jacrev(torch.func.grad(tr.make_hyper_objective(m_id, kind), argnums=0), argnums=1)(torch.tensor(eta), eps).numpy()
        np.testing.assert_allclose(m.cross_hessian_hyper_vec(kind, eta), Cv, rtol=0, atol=1e-12 * scale, err_msg=kind)
        # round trip of the setter
        m.set_hyper(kind, m.hyper_value(kind))
        assert abs(m.value(theta) - f(t, eps).item()) < 1e-12 * max(1.0, abs(m.value(theta)))
