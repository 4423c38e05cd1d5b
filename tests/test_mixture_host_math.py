"""Config 3: the host half of MixtureObjective (Dirichlet digamma chains, Schur assembly) and the
oracle's per-row restatement, against exact AD of a torch restatement of the same -ELBO."""
import numpy as np
import pytest
import torch

import lrvb_amd as vb
import torch_ref as tr
from oracle import mixture as om
from synthetic import clustered_problem  # noqa: F401  (tools/synthetic.py; re-exported for the GPU tests)


def make_par(N, V, K, lb=0.0):
    par = vb.ModelParamsDict('params')
    par.push_param(vb.DirichletParamArray('pi', shape=(K,), min_alpha=lb))
    par.push_param(vb.DirichletParamArray('phi', shape=(V, K), min_alpha=lb))
    par.push_param(vb.SimplexParam('z', shape=(N, K)))
    return par


class OracleSchurCtx(object):
    """Stands in for DeviceContext.mixture_schur (the CPU suite has no device): the oracle's restatement."""

    def mixture_schur(self, K, q, R, jlam, hgg, scale=None, diag_add=None):
        assert R is not None, 'the shell always installs reduced statistics'
        return om.mixture_schur(K, q, R, jlam, hgg, scale, diag_add)


def shell(par, x, K, a0, b0):
    """A MixtureObjective without a device context: statistics and the Schur assembly come from the oracle."""
    f = vb.MixtureObjective.__new__(vb.MixtureObjective)
    f.ctx = OracleSchurCtx()
    N, V = x.shape
    f.par, f.n_obs, f.V, f.K, f.n_global = par, N, V, K, K + V * K
    vi = par.vector_indices_dict
    f._ipi = np.arange(vi['pi'].start, vi['pi'].stop)
    f._iphi = np.arange(vi['phi'].start, vi['phi'].stop).reshape(V, K)
    f._lb = np.full(f.n_global, par['pi']['alpha']._lb)
    f._declare_priors(a0, b0)
    f._external_stats = None
    return f


def problem(N, V, K, seed, lb=0.0):
    rng = np.random.default_rng(seed)
    x = rng.poisson(3.0, size=(N, V)).astype(np.float64)
    w = rng.uniform(0.5, 1.5, N)
    fg = np.concatenate([rng.normal(size=K) * 0.3 + 0.5, rng.normal(size=V * K) * 0.3 + 0.5])
    # responsibilities near their optimum for these globals (z ~ softmax(s)), so that the local
    # blocks are positive definite and well conditioned, plus a perturbation
    alpha, beta = lb + np.exp(fg[:K]), lb + np.exp(fg[K:]).reshape(V, K)
    from scipy import special
    s = (special.digamma(alpha) - special.digamma(alpha.sum()))[None, :] + x @ (
        special.digamma(beta) - special.digamma(beta.sum(0, keepdims=True)))
    fz = (s[:, 1:] - s[:, :1]) + 0.2 * rng.normal(size=(N, K - 1))
    return x, w, np.concatenate([fg, fz.ravel()])


def near_optimum_problem(N, V, K, seed, a0=1.5, b0=0.8, sweeps=200, jitter=0.01, trials=4):
    """A few coordinate-ascent sweeps (z | globals, globals | z) from a random start, then a small
    perturbation: the full Hessian is positive definite there, as it is where LRVB is applied."""
    from scipy import special
    rng = np.random.default_rng(seed)
    centers = rng.dirichlet(np.ones(V) * 2.0, size=K)
    lab = rng.integers(0, K, size=N)
    x = np.stack([rng.multinomial(trials, centers[c]) for c in lab]).astype(np.float64)
    w = rng.uniform(0.5, 1.5, N)
    alpha = np.ones(K) + rng.uniform(0, 1, K)
    beta = np.ones((V, K)) + rng.uniform(0, 1, (V, K))
    for _ in range(sweeps):
        s = (special.digamma(alpha) - special.digamma(alpha.sum()))[None, :] + x @ (
            special.digamma(beta) - special.digamma(beta.sum(0, keepdims=True)))
        z = np.exp(s - s.max(1, keepdims=True)); z /= z.sum(1, keepdims=True)
        alpha = a0 + (w[:, None] * z).sum(0)
        beta = b0 + x.T @ (w[:, None] * z)
    fz = np.log(z[:, 1:]) - np.log(z[:, :1]) + jitter * rng.normal(size=(N, K - 1))
    fg = np.concatenate([np.log(alpha), np.log(beta).ravel()]) + jitter * rng.normal(size=K + V * K)
    return x, w, np.concatenate([fg, fz.ravel()])




def oracle_stats(f, x, w, theta):
    fg, fz = theta[:f.n_global], theta[f.n_global:]
    _, _, lam = f._lam(f._lb + np.exp(fg))
    val2, gfree, Hloc, S64, R = om.mixture_rows(fz, x, w, lam)
    return np.concatenate([val2, S64.ravel(), R.ravel()]), gfree, Hloc


@pytest.mark.parametrize('N,V,K,lb', [(12, 3, 2, 0.0), (25, 4, 3, 0.0), (30, 5, 4, 0.1)])
def test_value_and_schur_complement_match_ad(N, V, K, lb):
    x, w, theta = problem(N, V, K, seed=N + K, lb=lb)
    par = make_par(N, V, K, lb)
    f = shell(par, x, K, 1.5, 0.8)
    stats, gfree, Hloc = oracle_stats(f, x, w, theta)
    f.set_reduced_stats(stats)
    ft = tr.mixture_objective(x, K, 1.5, 0.8, lb=lb)
    tt, tw = torch.tensor(theta), torch.tensor(w)
    v_ad = ft(tt, tw).item()
    assert abs(f.value(theta) - v_ad) < 1e-11 * max(1.0, abs(v_ad))
    g_ad = torch.func.grad(ft)(tt, tw).numpy()
    H_ad = torch.func.hessian(ft)(tt, tw).numpy()
    ng = f.n_global
    # oracle rows: local gradient and block-diagonal local Hessian
    np.testing.assert_allclose(gfree.ravel(), g_ad[ng:], rtol=1e-10, atol=1e-11)
    Hzz = H_ad[ng:, ng:]
    for n in range(N):
        s = slice(n * (K - 1), (n + 1) * (K - 1))
        np.testing.assert_allclose(Hloc[n], Hzz[s, s], rtol=1e-9, atol=1e-11)
    off = Hzz.copy()
    for n in range(N):
        s = slice(n * (K - 1), (n + 1) * (K - 1))
        off[s, s] = 0.0
    assert np.max(np.abs(off)) < 1e-12
    # host assembly: Schur complement onto the Dirichlet block
    HS, Hgg_free, schur = f.global_hessian(theta, return_parts=True)
    np.testing.assert_allclose(Hgg_free, H_ad[:ng, :ng], rtol=1e-9, atol=1e-10)
    HS_ad = H_ad[:ng, :ng] - H_ad[:ng, ng:] @ np.linalg.solve(Hzz, H_ad[ng:, :ng])
    np.testing.assert_allclose(HS, HS_ad, rtol=1e-7, atol=1e-8 * np.max(np.abs(HS_ad)))   # difference of two large terms
    # the LRVB covariance of the global block is the corresponding block of the full inverse
    if np.all(np.linalg.eigvalsh(H_ad) > 0):
        cov = np.linalg.inv(H_ad)[:ng, :ng]
        np.testing.assert_allclose(np.linalg.inv(HS), cov, rtol=1e-6, atol=1e-7 * np.max(np.abs(cov)))


def test_dirichlet_block_closed_forms():
    from lrvb_amd.mixture import _dirichlet_terms
    rng = np.random.default_rng(3)
    alpha = rng.uniform(0.3, 4.0, 6)
    d = rng.uniform(0.0, 5.0, 6)

    def t(al):
        a0 = al.sum()
        elog = torch.digamma(al) - torch.digamma(a0)
        ent = (torch.lgamma(al).sum() - torch.lgamma(a0) + (a0 - al.numel()) * torch.digamma(a0)
               - ((al - 1.0) * torch.digamma(al)).sum())
        return -(torch.tensor(d) * elog).sum() - ent
    ta = torch.tensor(alpha)
    val, g, H = _dirichlet_terms(alpha, d)
    assert abs(val - t(ta).item()) < 1e-12 * max(1.0, abs(val))
    # torch's fp64 trigamma is accurate to ~1e-9 only (scipy's agrees with mpmath to 1e-15)
    np.testing.assert_allclose(g, torch.func.grad(t)(ta).numpy(), rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(H, torch.func.hessian(t)(ta).numpy(), rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize('lb', [0.0, 0.5])
def test_prior_hyper_parameter_closed_forms_match_ad(lb):
    """Cross Hessians and gradients of the mixture with respect to the two Dirichlet priors against exact AD of the torch
    restatement with the prior as a variable; the N (K - 1) simplex rows of the cross Hessian are zero."""
    N, V, K = 12, 3, 3
    x, w, theta = problem(N, V, K, seed=5, lb=lb)
    par = make_par(N, V, K, lb)
    a0 = np.array([1.5, 0.9, 2.0]); b0 = np.random.default_rng(0).uniform(0.5, 2.0, (V, K))
    f = shell(par, x, K, a0, b0)
    tt, tw = torch.tensor(theta), torch.tensor(w)
    ng = f.n_global
    fg = theta[:ng]
    eta_g = lb + np.exp(fg)
    alpha, beta, _ = f._lam(eta_g)
    jg = eta_g - lb
    builders = {
        'pi_prior': (a0, lambda e: tr.mixture_objective(x, K, e, b0, lb=lb)),
        'phi_prior': (b0.ravel(), lambda e: tr.mixture_objective(x, K, a0, e.reshape(V, K), lb=lb)),
    }
    for kind, (e0, build) in builders.items():
        F = lambda th, e: build(e)(th, tw)
        te0 = torch.tensor(e0)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(tt, te0).numpy()
        g = torch.func.grad(F, argnums=1)(tt, te0).numpy()
        assert np.max(np.abs(C[ng:])) == 0.0
        np.testing.assert_allclose(jg[:, None] * f._prior_hyper(kind, alpha, beta, 'cross'), C[:ng], rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(C))), err_msg=kind)
        np.testing.assert_allclose(f._prior_hyper(kind, alpha, beta, 'grad'), g, rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(g))), err_msg=kind)
