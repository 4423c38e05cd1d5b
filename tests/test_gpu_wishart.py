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


def cavi_optimum(y, w, mu0, lam0, nu0, w0, sweeps=200):
    """The minimiser of the -ELBO in vector coordinates, from the stationarity conditions of the model's closed form
    (standard conjugate updates): nu = sum w + nu0, then iterate V^-1 = W0 + sum w (y-m)(y-m)^T + (sum w) Lambda_mu^-1,
    Lambda_mu = (sum w) nu V + Lambda0, m = Lambda_mu^-1 (nu V sum w y + Lambda0 mu0).  At (and near) it the Hessian
    is positive definite, which the CG checks need -- chosen by construction instead of testing a random point."""
    d = y.shape[1]
    W = float(np.sum(w)); sy = y.T @ w; Syy = y.T @ (w[:, None] * y)
    nu = W + nu0
    m = sy / W; P = np.eye(d) / W
    for _ in range(sweeps):
        A = Syy - np.outer(sy, m) - np.outer(m, sy) + W * np.outer(m, m)
        v = np.linalg.inv(A + w0 + W * P)
        C = W * nu * v + lam0
        P = np.linalg.inv(C)
        m_new = P @ (nu * v @ sy + lam0 @ mu0)
        done = np.max(np.abs(m_new - m)) < 1e-14
        m = m_new
        if done:
            break
    return np.concatenate([m, C[np.tril_indices(d)], [nu], v[np.tril_indices(d)]])


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
    # conjugate gradients through the reference's solver class, at a point where the Hessian is positive definite by
    # construction: next to the optimum (slightly off it, so that the packing second-order terms are not zero)
    mu0 = np.zeros(d); lam0 = 0.5 * np.eye(d); nu0 = d + 2.0; w0 = np.eye(d)
    theta_opt = lay.unconstrain(cavi_optimum(y, w, mu0, lam0, nu0, w0))
    assert np.linalg.norm(objective.fun_free_grad(theta_opt)) < 1e-6 * max(1.0, abs(objective.fun_free(theta_opt)))
    theta_cg = theta_opt + 1e-3 * rng.normal(size=D)
    H_cg = torch.func.hessian(ft)(torch.tensor(theta_cg), tw).numpy()
    assert np.min(np.linalg.eigvalsh(0.5 * (H_cg + H_cg.T))) > 0
    solver = vb.ConjugateGradientSolver(objective.fun_free_hvp, theta_cg)
    b = rng.normal(size=D)
    x, info = solver.get_hinv_vec(b)
    want = np.linalg.solve(H_cg, b)
    assert info == 0 and np.max(np.abs(x - want)) < 1e-6 * np.max(np.abs(want))


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
    # G^T G.  Row n of G is d/dtheta of observation n's own term, so (i) on a row subsample the AD cross Hessian
    # d2 f / d theta d w^T (forward mode over the weights of an objective built on those rows) gives G itself, and
    # (ii) the Gram matrix is additive over any split of the rows -- which carries (i) to all N rows.
    ns = 48
    par_s = vb.ModelParamsDict('params')
    par_s.push_param(vb.MVNParam('mu', dim=d)); par_s.push_param(vb.WishartParam('lambda', size=d))
    mu0 = np.zeros(d); lam0 = 0.5 * np.eye(d); nu0 = d + 2.0; w0 = np.eye(d)
    fun_s = vb.WishartMVNObjective(par_s, y[:ns], prior_mean=mu0, prior_info=lam0, prior_df=nu0, prior_inv_scale=w0)
    ft_s = tr.wishart_mvn_objective(y[:ns], d, mu0, lam0, nu0, w0, layout=lay)
    cross = torch.func.jacfwd(torch.func.grad(ft_s, argnums=0), argnums=1)(tt, torch.ones(ns, dtype=torch.float64)).numpy()
    assert cross.shape == (D, ns)
    assert rel_err(fun_s.gram(theta), cross @ cross.T) < 1e-9
    got = fun.gram(theta)
    assert np.allclose(got, got.T, rtol=0, atol=1e-10 * np.abs(got).max())
    parts = []
    for rows in (slice(0, ns), slice(ns, 1500), slice(1500, N)):
        par_r = vb.ModelParamsDict('params')
        par_r.push_param(vb.MVNParam('mu', dim=d)); par_r.push_param(vb.WishartParam('lambda', size=d))
        parts.append(vb.WishartMVNObjective(par_r, y[rows], prior_mean=mu0, prior_info=lam0, prior_df=nu0,
                                            prior_inv_scale=w0).gram(theta))
    assert rel_err(parts[0], cross @ cross.T) < 1e-9
    assert rel_err(got, parts[0] + parts[1] + parts[2]) < 1e-11
    # LRVB solve by conjugate gradients (tol 1e-10) at D = 4096 against the device Cholesky, next to the optimum,
    # where the Hessian is positive definite
    theta_cg = lay.unconstrain(cavi_optimum(y, np.ones(N), mu0, lam0, nu0, w0)) + 1e-4 * rng.normal(size=D)
    H_cg = objective.fun_free_hessian(theta_cg)
    assert np.min(np.linalg.eigvalsh(0.5 * (H_cg + H_cg.T))) > 0
    b = rng.normal(size=D)
    x, info, iters = fun.cg_solve(theta_cg, b, tol=1e-10, maxiter=20000)
    fun.ctx.chol_factor(H_cg)
    xc = fun.ctx.chol_solve(b)
    assert info == 0
    assert np.max(np.abs(x - xc)) < 1e-6 * np.max(np.abs(xc))


def test_config5_full_size_gram_properties(vb):
    """G^T G at the full configuration (N = 1e6, d = 63, D = 4096: the Kronecker-row fp64-MFMA kernel on 1.7e13
    flops).  No oracle finishes at this size; the size-independent properties do: the matrix is symmetric positive
    semi-definite, additive over a split of the rows, and the first 48 rows' share -- a split part of its own --
    equals the AD cross Hessian d2 f / d theta d w^T of those rows."""
    d, N, ns = 63, 1_000_000, 48
    rng = np.random.default_rng(20246)
    a = rng.normal(size=(d, d)); cov = a @ a.T / d + np.eye(d)
    y = rng.normal(size=(N, d)) @ np.linalg.cholesky(cov).T + rng.normal(size=d)
    mu0 = np.zeros(d); lam0 = 0.5 * np.eye(d); nu0 = d + 2.0; w0 = np.eye(d)
    lay = opk.Layout([opk.box_block(d), opk.psd_block(d), opk.box_block(1, lb=d - 1.0), opk.psd_block(d)])
    eta0 = random_point(rng, d)
    eta0[d + d * (d + 1) // 2] = d + 10.0
    theta = lay.unconstrain(eta0)

    def gram_of(rows):
        par = vb.ModelParamsDict('params')
        par.push_param(vb.MVNParam('mu', dim=d)); par.push_param(vb.WishartParam('lambda', size=d))
        return vb.WishartMVNObjective(par, y[rows], prior_mean=mu0, prior_info=lam0, prior_df=nu0, prior_inv_scale=w0).gram(theta)

    full = gram_of(slice(0, N))
    scale = np.abs(full).max()
    assert np.max(np.abs(full - full.T)) < 1e-10 * scale
    parts = [gram_of(slice(0, ns)), gram_of(slice(ns, 400_000)), gram_of(slice(400_000, N))]
    assert rel_err(parts[0] + parts[1] + parts[2], full) < 1e-11
    ft_s = tr.wishart_mvn_objective(y[:ns], d, mu0, lam0, nu0, w0, layout=lay)
    cross = torch.func.jacfwd(torch.func.grad(ft_s, argnums=0), argnums=1)(torch.tensor(theta),
                                                                          torch.ones(ns, dtype=torch.float64)).numpy()
    assert rel_err(parts[0], cross @ cross.T) < 1e-9
    lam = np.linalg.eigvalsh(full)
    assert lam[0] > -1e-9 * lam[-1]


@pytest.mark.parametrize('d,N', [(1, 50), (4, 700), (17, 3000)])
def test_gram_with_device_generated_matrices(vb, d, N):
    """`WishartMVNObjective.gram` writes the V = (d + 1)^2 per-coordinate matrices of the per-observation gradient on the device
    (lrvb_wishart_gram) instead of building V q^2 doubles on the host: the same G^T G as the generic entry point fed with the
    host-built matrices (`_obs_terms`, pinned against exact AD in tests/test_wishart_mvn_host_math.py), and with
    want_host=False the result stays in HBM, where the Cholesky factors it."""
    rng = np.random.default_rng(100 + d)
    y, par, fun, lay, ft = _build(vb, rng, N, d)
    theta = lay.unconstrain(random_point(rng, d))
    got = fun.gram(theta)
    eta = lay.constrain(theta)
    M, c = fun._obs_terms(eta)
    want = fun.ctx.quadform_gram(M, c, theta)
    assert rel_err(got, want) < 1e-13
    c2, m, nu, v = fun._obs_constants(eta)
    np.testing.assert_allclose(c2, c, rtol=1e-14, atol=1e-15)
    assert fun.gram(theta, want_host=False) is None
    shift = 1e-3 * np.abs(got).max()
    # the resident matrix is the one that came back: factor a shifted copy both ways and compare the covariances
    fun.ctx.chol_factor(got + shift * np.eye(lay.D))
    cov_host = fun.ctx.lrvb_cov(np.eye(lay.D)[:3])
    assert rel_err(cov_host, np.linalg.inv(got + shift * np.eye(lay.D))[:3, :3]) < 1e-7
    with pytest.raises(ValueError):
        fun.ctx.wishart_gram(d, [0, d, d + d * (d + 1) // 2, d + d * (d + 1) // 2 + 1], nu, m[:-1] if d > 1 else np.zeros(2), v, c2, theta)


@pytest.mark.parametrize('q,N', [(2, 40), (7, 900), (18, 2500), (64, 700)])
def test_quadform_gram_folds_general_matrices_onto_the_triangle(vb, q, N):
    """`lrvb_quadform_gram` for matrices M_k that are NOT symmetric (the Kronecker SYRK forms the packed lower triangle of
    z z^T only -- q (q + 1) / 2 virtual columns -- and `mtilde_kernel` folds M_k[a][b] + M_k[b][a] onto it): G^T G against
    numpy with the rows g_n[k] = 1/2 z_n^T M_k z_n + c_k written out, a box layout so that the free conversion is a scaling;
    q = 64 is the largest row the kernel takes (configuration 5's shape: 2080 virtual columns, 17 tile rows)."""
    rng = np.random.default_rng(500 + q)
    V = 9
    z = rng.normal(size=(N, q)) / np.sqrt(q)
    M = rng.normal(size=(V, q, q))                                   # general, not symmetric
    c = rng.normal(size=V)
    blocks = [dict(kind=0, free_size=4, vec_size=4, dim0=4, dim1=0, lb=0.0, ub=np.inf),
              dict(kind=0, free_size=5, vec_size=5, dim0=5, dim1=0, lb=-np.inf, ub=np.inf)]
    ctx = vb.DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=q, device=0)
    ctx.set_data(vb._hip.SLOT_X, z)
    theta = rng.normal(size=V) * 0.3
    got = ctx.quadform_gram(M, c, theta)
    G = 0.5 * np.einsum('na,kab,nb->nk', z, M, z) + c[None, :]
    j = np.concatenate([np.exp(theta[:4]), np.ones(5)])
    want = (G.T @ G) * j[:, None] * j[None, :]
    assert rel_err(got, want) < 1e-12
