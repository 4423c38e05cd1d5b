"""BASELINE.json config 5 on the GPU: Wishart + MVN full-covariance model, D = (d+1)^2.
Small sizes: every derivative against exact AD (torch.func fp64).  d = 63 (D = 4096): the dense
Hessian against AD Hessian-vector products, the Kronecker-row MFMA Gram matrix G^T G against a
numpy evaluation of the same G, and the conjugate-gradient LRVB solve against a direct solve."""
import numpy as np
import pytest
import torch

import torch_ref as tr
from oracle import packing as opk
from helpers import rel_err
from test_wishart_mvn_host_math import random_point

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _build(vb, rng, N, d):
    a = rng.normal(size=(d, d)); cov = a @ a.T / d + np.eye(d)
    y = rng.multivariate_normal(rng.normal(size=d), cov, size=N)
    par = vb.ModelParamsDict('params')
    par.push_param(vb.MVNParam('mu', dim=d))
    par.push_param(vb.WishartParam('lambda', size=d))
    mu0 = np.zeros(d); lam0 = 0.5 * np.eye(d); nu0 = d + 2.0; w0 = np.eye(d)
    fun = vb.WishartMVNObjective(par, y, prior_mean=mu0, prior_info=lam0, prior_df=nu0, prior_inv_scale=w0)
    lay = opk.Layout([opk.box_block(d), opk.psd_block(d), opk.box_block(1, lb=d - 1.0), opk.psd_block(d)])
    ft = tr.wishart_mvn_objective(y, d, mu0, lam0, nu0, w0, layout=lay)
    return y, par, fun, lay, ft


@pytest.mark.parametrize('d,N', [(1, 300), (3, 1000)])
def test_small_all_derivatives(vb, d, N):
    rng = np.random.default_rng(7 * d)
    y, par, fun, lay, ft = _build(vb, rng, N, d)
    D = par.free_size()
    assert D == lay.D == (d + 1) ** 2
    objective = vb.Objective(par, fun)
    theta = lay.unconstrain(random_point(rng, d))
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)
    tt, tw = torch.tensor(theta), torch.tensor(w)
    H_ad = torch.func.hessian(ft)(tt, tw).numpy()
    assert abs(objective.fun_free(theta) - ft(tt, tw).item()) < 1e-10 * abs(ft(tt, tw).item())
    assert rel_err(objective.fun_free_grad(theta), torch.func.grad(ft)(tt, tw).numpy()) < 1e-9
    assert rel_err(objective.fun_free_hessian(theta), H_ad) < 1e-9
    cross = torch.func.jacrev(torch.func.grad(ft, argnums=0), argnums=1)(tt, tw).numpy()      # D x N
    two = vb.TwoParameterObjective(par, fun.weights_par, fun)
    assert rel_err(two.fun_hessian_free1_vector2(theta, w), cross) < 1e-9
    # Gram matrix of the per-observation gradients: Kronecker-row kernel vs G^T G of the AD cross Hessian
    assert rel_err(fun.gram(theta), cross @ cross.T) < 1e-9
    # conjugate gradients through the reference's solver class, on the device-resident dense Hessian
    Hs = 0.5 * (H_ad + H_ad.T) + 10.0 * np.abs(H_ad).max() * np.eye(D) * 0     # SPD at this point? use H^T H shift if not
    if np.min(np.linalg.eigvalsh(Hs)) > 0:
        solver = vb.ConjugateGradientSolver(objective.fun_free_hvp, theta)
        b = rng.normal(size=D)
        x, info = solver.get_hinv_vec(b)
        assert info == 0 and np.max(np.abs(x - np.linalg.solve(H_ad, b))) < 1e-6 * np.max(np.abs(np.linalg.solve(H_ad, b)))


def test_config5_d63(vb):
    d, N = 63, 4096
    rng = np.random.default_rng(20245)
    y, par, fun, lay, ft = _build(vb, rng, N, d)
    D = par.free_size()
    assert D == 4096
    objective = vb.Objective(par, fun)
    eta0 = random_point(rng, d)
    eta0[d + d * (d + 1) // 2] = d + 10.0              # df well inside its domain
    theta = lay.unconstrain(eta0)
    tt = torch.tensor(theta)
    w1 = torch.ones(N, dtype=torch.float64)
    H = objective.fun_free_hessian(theta)
    assert np.allclose(H, H.T, rtol=0, atol=1e-9 * np.abs(H).max())
    grad_fn = torch.func.grad(ft)
    for seed in range(3):
        v = np.random.default_rng(seed).normal(size=D)
        hv_ad = torch.func.jvp(lambda th: grad_fn(th, w1), (tt,), (torch.tensor(v),))[1].numpy()
        assert rel_err(H @ v, hv_ad) < 1e-8
    assert rel_err(objective.fun_free_grad(theta), grad_fn(tt, w1).numpy()) < 1e-9
    # G^T G: numpy evaluation of the same per-observation gradients
    eta = lay.constrain(theta)
    M, c = fun._obs_terms(eta)
    z = np.hstack([y, np.ones((N, 1))])
    zz = (z[:, :, None] * z[:, None, :]).reshape(N, -1)
    G = 0.5 * zz @ M.reshape(M.shape[0], -1).T + c[None, :]
    J = lay.jac(theta)
    want = J.T @ (G.T @ G) @ J
    got = fun.gram(theta)
    assert rel_err(got, want) < 1e-9
    assert np.allclose(got, got.T, rtol=0, atol=1e-10 * np.abs(got).max())
    # LRVB solve by conjugate gradients (tol 1e-8) at D = 4096, against the device Cholesky
    lam_min = np.min(np.linalg.eigvalsh(0.5 * (H + H.T)))
    if lam_min > 0:
        b = rng.normal(size=D)
        x, info, iters = fun.cg_solve(theta, b, tol=1e-10, maxiter=20000)
        fun.ctx.chol_factor(H)
        xc = fun.ctx.chol_solve(b)
        assert info == 0
        assert np.max(np.abs(x - xc)) < 1e-6 * np.max(np.abs(xc))
